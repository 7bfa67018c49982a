// include/sventt/plan_handle.hpp -- from a kernel_type to an engine plan.
//
// Shared by NTT<kernel_type> (wrapper.hpp) and by the kernel concept's own entry points
// (prepare_forward / compute_forward(dst, src, aux) ... in plan_types.hpp): what the
// reference does in kernel_type::prepare_* -- fill a twiddle blob (kernel/iterative.hpp:
// 80-83, kernel/recursive.hpp:35-46) -- is here "create the device plan".
#ifndef SVENTT_GPU_PLAN_HANDLE_HPP_INCLUDED
#define SVENTT_GPU_PLAN_HANDLE_HPP_INCLUDED

#include <bit>
#include <cstdint>

#include "sventt/status.hpp"
#include "sventt_hip.h"

namespace sventt {

namespace detail {

// log2 of R when kernel_type spells out a six-step n = R x C, else 0 (engine's choice)
template <class kernel_type> constexpr std::uint32_t requested_rows_log2(void) {
  constexpr std::uint64_t rows{kernel_type::get_six_step_rows()};
  constexpr std::uint64_t m{kernel_type::get_m()};
  if (rows < 2 || rows >= m || !std::has_single_bit(rows)) {
    return 0;
  }
  return static_cast<std::uint32_t>(std::countr_zero(rows));
}

// The plan for kernel_type with the given SVENTT_FORWARD/SVENTT_INVERSE flags.  A split the
// engine's tiles do not cover (too few columns for the column pass, rows longer than a row
// tile ...) is not an error of the caller's kernel_type: the engine then plans on its own.
template <class kernel_type> sventt_plan *create_plan(std::uint32_t flags) {
  using modulus_type = typename kernel_type::modulus_type;
  if constexpr (kernel_type::uses_fixed_point()) {
    flags |= SVENTT_FIXED_POINT;  // FixedPoint64SVE / FixedPoint64Scalar layers: the Shoup kernels
  }
  sventt_plan *plan{};
  constexpr std::uint32_t rows_log2{requested_rows_log2<kernel_type>()};
  // the inverse divides by the product of the layers' inverse_factor arguments: 1 (no
  // scaling) for README-shaped kernels, m for kernels whose last layer ends in `..., m>`
  constexpr std::uint64_t inverse_divisor{kernel_type::get_inverse_factor()};
  static_assert(inverse_divisor != 0, "an inverse_factor is a multiple of the modulus");
  int status{sventt_plan_create_ex(modulus_type::get_modulus(), modulus_type::get_generator(),
                                   kernel_type::get_m(), rows_log2, 1, flags, inverse_divisor, &plan)};
  if (status == SVENTT_ERR_INVALID_ARGUMENT && rows_log2 != 0) {
    status = sventt_plan_create_ex(modulus_type::get_modulus(), modulus_type::get_generator(),
                                   kernel_type::get_m(), 0, 1, flags, inverse_divisor, &plan);
  }
  throw_on_error(status);
  return plan;
}

// One process-wide plan per (kernel_type, direction) for the static kernel-concept entry
// points, created on first use and released at exit.
template <class kernel_type, std::uint32_t flags> const sventt_plan *shared_plan(void) {
  struct holder {
    sventt_plan *plan;
    holder(void) : plan{create_plan<kernel_type>(flags)} {}
    ~holder(void) { sventt_plan_destroy(plan); }
  };
  static const holder h;
  return h.plan;
}

// What prepare_* leaves in the auxiliary vector in place of a twiddle blob.
struct plan_record {
  const sventt_plan *plan;
};

} // namespace detail

} // namespace sventt

#endif /* SVENTT_GPU_PLAN_HANDLE_HPP_INCLUDED */
