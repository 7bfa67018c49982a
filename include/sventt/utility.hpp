// include/sventt/utility.hpp -- sventt::bitreverse, same contract as the
// reference's helper (include/sventt/utility.hpp:12-23 there): reverses all 64
// bits.  Callers shift the result down to reverse a shorter index, e.g.
// bitreverse(j) >> (64 - log2(n)) is where element j of a forward transform's
// output belongs in natural order.
#ifndef SVENTT_GPU_UTILITY_HPP_INCLUDED
#define SVENTT_GPU_UTILITY_HPP_INCLUDED

#include <cstdint>

namespace sventt {

static inline constexpr std::uint64_t bitreverse(std::uint64_t x) {
  std::uint64_t reversed{0};
  for (int bit{0}; bit < 64; ++bit) {
    reversed = (reversed << 1) | (x & 1);
    x >>= 1;
  }
  return reversed;
}

} // namespace sventt

#endif /* SVENTT_GPU_UTILITY_HPP_INCLUDED */
