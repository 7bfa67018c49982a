// include/sventt/wrapper.hpp -- sventt::NTT<kernel_type> on the GPU engine.
//
// Same public surface as the reference's wrapper (include/sventt/wrapper.hpp:
// 13-83 there): NTT(enable_forward, enable_inverse, allocate_huge_pages),
// get_m(), modulus_type, compute_forward/compute_inverse(dst[, src]) const.
// Construction builds the plan through the C ABI (include/sventt_hip.h) instead
// of filling an SVE twiddle blob; the transforms run in the HIP kernels.
// Pointers may be host memory (PageMemory, std::vector: staged through the
// device, the call returns when dst is complete) or device memory (asynchronous
// on the default stream).  Errors come back as the exception types the reference
// throws: std::invalid_argument, std::logic_error, std::bad_alloc; anything
// HIP-specific is a std::runtime_error.
#ifndef SVENTT_GPU_WRAPPER_HPP_INCLUDED
#define SVENTT_GPU_WRAPPER_HPP_INCLUDED

#include <bit>
#include <cstdint>
#include <new>
#include <stdexcept>
#include <string>

#include "sventt/plan_handle.hpp"
#include "sventt/status.hpp"
#include "sventt_hip.h"

namespace sventt {

template <class kernel_type_> class NTT {
public:
  using kernel_type = kernel_type_;
  using modulus_type = typename kernel_type::modulus_type;

private:
  sventt_plan *plan{};

public:
  NTT(const bool enable_forward = true, const bool enable_inverse = true,
      [[maybe_unused]] const bool allocate_huge_pages = true) {
    const std::uint32_t flags{(enable_forward ? std::uint32_t{SVENTT_FORWARD} : 0u) |
                              (enable_inverse ? std::uint32_t{SVENTT_INVERSE} : 0u)};
    if (flags == 0) {
      return;  // the reference allows an NTT with neither table; it then cannot transform
    }
    plan = detail::create_plan<kernel_type>(flags);
  }

  NTT(const NTT &) = delete;
  NTT &operator=(const NTT &) = delete;
  NTT(NTT &&that) noexcept : plan{that.plan} { that.plan = nullptr; }
  NTT &operator=(NTT &&that) noexcept {
    if (this != &that) {
      sventt_plan_destroy(plan);
      plan = that.plan;
      that.plan = nullptr;
    }
    return *this;
  }
  ~NTT(void) { sventt_plan_destroy(plan); }

  static constexpr std::uint64_t get_m(void) { return kernel_type::get_m(); }

  // what the engine made of kernel_type, e.g. "col 2^8 x T16 (stride 512) | row 2^9 (tile 2^12)"
  std::string describe(void) const { return plan ? sventt_plan_describe(plan) : ""; }

  void compute_forward(std::uint64_t *const dst, const std::uint64_t *const src) const {
    if (plan == nullptr) {
      throw std::logic_error{"NTT was constructed with forward and inverse disabled"};
    }
    detail::throw_on_error(sventt_forward(plan, dst, src, nullptr));
  }

  void compute_forward(std::uint64_t *const dst) const { compute_forward(dst, dst); }

  void compute_inverse(std::uint64_t *const dst, const std::uint64_t *const src) const {
    if (plan == nullptr) {
      throw std::logic_error{"NTT was constructed with forward and inverse disabled"};
    }
    detail::throw_on_error(sventt_inverse(plan, dst, src, nullptr));
  }

  void compute_inverse(std::uint64_t *const dst) const { compute_inverse(dst, dst); }

  // ---- extensions for data that stays on the device (not in the reference's wrapper) ----
  // What its caller does around the transforms (examples/magic-series/
  // gaussian-polynomial.hpp:176-212): convert a spectrum to Montgomery form once, then
  // multiply every forward transform by it.  All pointers are device pointers.

  // dst[i] = src[i] * 2^64 mod N over get_m() elements (PAdic64SVE::to_montgomery)
  void to_montgomery(std::uint64_t *const dst, const std::uint64_t *const src) const {
    detail::throw_on_error(sventt_to_montgomery(plan, dst, src, get_m(), nullptr));
  }

  // dst = forward(src) (.) operand_montgomery, the product fused into the last pass
  void compute_forward_multiply(std::uint64_t *const dst, const std::uint64_t *const src,
                                const std::uint64_t *const operand_montgomery) const {
    detail::throw_on_error(sventt_forward_multiply(plan, dst, src, operand_montgomery, nullptr));
  }

  // dst[i] = a[i] * b[i] mod N (plain residues) over get_m() elements
  void pointwise_multiply(std::uint64_t *const dst, const std::uint64_t *const a,
                          const std::uint64_t *const b) const {
    detail::throw_on_error(sventt_pointwise_multiply(plan, dst, a, b, get_m(), nullptr));
  }
};

} // namespace sventt

#endif /* SVENTT_GPU_WRAPPER_HPP_INCLUDED */
