// include/sventt/status.hpp -- C-ABI status codes back to the exception types the
// reference throws (std::invalid_argument for bad shapes / fields, modulus.hpp:118-120
// and transposition/sve/in-register.hpp:121-128; std::logic_error, wrapper.hpp:55;
// std::bad_alloc, vector.hpp:131); anything HIP-specific is a std::runtime_error.
#ifndef SVENTT_GPU_STATUS_HPP_INCLUDED
#define SVENTT_GPU_STATUS_HPP_INCLUDED

#include <new>
#include <stdexcept>
#include <string>

#include "sventt_hip.h"

namespace sventt {

namespace detail {

inline void throw_on_error(const int status) {
  if (status == SVENTT_OK) {
    return;
  }
  const std::string message{sventt_last_error()};
  switch (status) {
  case SVENTT_ERR_INVALID_ARGUMENT:
    throw std::invalid_argument{message};
  case SVENTT_ERR_ALLOC:
    throw std::bad_alloc{};
  case SVENTT_ERR_LOGIC:
    throw std::logic_error{message};
  default:
    throw std::runtime_error{"sventt-hip: " + message};
  }
}

} // namespace detail

} // namespace sventt

#endif /* SVENTT_GPU_STATUS_HPP_INCLUDED */
