// include/sventt/plan_types.hpp -- the reference's template vocabulary as a
// compile-time PLAN DESCRIPTION.
//
// In the reference these class templates contain the SVE inner loops.  Here
// they keep their names and template parameter lists so that existing
// kernel_type definitions (README.md:13-82, tests/ntt-tests/*-sve-*.hpp of the
// reference) compile unchanged, but they compute nothing: each one only knows
// its transform length, its radix and which field it works in, and
// sventt::NTT<kernel_type> (wrapper.hpp) lowers the outermost type to the
// (modulus, generator, n, n0) the C ABI takes.  Knobs that tune the SVE code
// (vector width, shuffle stages, paddings, unroll counts, NUMA blocks) are
// accepted and ignored: the GPU engine chooses its own tiles.
//
//   reference file (include/sventt/...)          names provided here
//   modmul/{sve,scalar}/p-adic-64.hpp            PAdic64SVE, PAdic64Scalar
//   modmul/{sve,scalar}/fixed-point-64.hpp       FixedPoint64SVE, FixedPoint64Scalar
//   layer/sve/radix-{two,four,eight}.hpp         Radix{Two,Four,Eight}SVELayer
//   layer/scalar/radix-{two,four,eight}.hpp      Radix{Two,Four,Eight}ScalarLayer
//   layer/sve/generic.hpp, blocked-generic.hpp   GenericSVELayer, BlockedGenericSVELayer
//   layer/scalar/generic.hpp                     GenericScalarLayer
//   transposition/sve/*.hpp                      Transpose[Parallel]SVE... (tags)
//   kernel/iterative.hpp, kernel/recursive.hpp   IterativeNTT, RecursiveNTT
#ifndef SVENTT_GPU_PLAN_TYPES_HPP_INCLUDED
#define SVENTT_GPU_PLAN_TYPES_HPP_INCLUDED

#include <bit>
#include <cstdint>
#include <type_traits>

#include "sventt/modulus.hpp"
#include "sventt/plan_handle.hpp"
#include "sventt/transposition.hpp"
#include "sventt/vector.hpp"

namespace sventt {

// ---- modular multiplication back ends ---------------------------------------
// PAdic64* select the engine's Montgomery kernels (sve_ntt_amd/csrc/field64.h: ARITH_MONT; the
// Goldilocks prime gets its own reduction automatically), FixedPoint64* its Shoup kernels
// (ARITH_SHOUP: c = a*w - hi64(a*w')*N with w' stored beside each twiddle, the arithmetic of
// modmul/sve/fixed-point-64.hpp:60-68 of the reference; modulus < 2^63).  Every back end returns
// canonical residues, so the results do not depend on the choice.  The host-side conversions the
// reference's callers use on data are kept.
namespace detail {

template <class modulus_type_, bool fixed_point_ = false> struct modmul_tag {
  using modulus_type = modulus_type_;
  static constexpr bool is_fixed_point{fixed_point_};

  // b * 2^64 mod N  (modmul/scalar/p-adic-64.hpp:16-19 of the reference)
  static constexpr std::uint64_t to_montgomery(const std::uint64_t b) {
    constexpr std::uint64_t N{modulus_type::get_modulus()};
    return static_cast<std::uint64_t>((static_cast<unsigned __int128>(b % N) << 64) % N);
  }

  // b * 2^-64 mod N  (:21-24)
  static constexpr std::uint64_t from_montgomery(const std::uint64_t b) {
    constexpr std::uint64_t N{modulus_type::get_modulus()};
    constexpr std::uint64_t r{static_cast<std::uint64_t>((static_cast<unsigned __int128>(1) << 64) % N)};
    return modulus_type::multiply(b % N, modulus_type::invert(r));
  }

  // b * N^{-1} mod 2^64  (:26-29)
  static constexpr std::uint64_t precompute(const std::uint64_t b) {
    return b * modulus_type::get_montgomery_inverse();
  }
};

constexpr bool is_power_of_two(const std::uint64_t x) { return std::has_single_bit(x); }

} // namespace detail

template <class modulus_type> class PAdic64SVE : public detail::modmul_tag<modulus_type> {};
template <class modulus_type> class PAdic64Scalar : public detail::modmul_tag<modulus_type> {};
// FixedPoint64 keeps data and twiddles in the plain domain: its to/from_montgomery are the
// identity (modmul/scalar/fixed-point-64.hpp:16-22 of the reference)
template <class modulus_type> class FixedPoint64SVE : public detail::modmul_tag<modulus_type, true> {
public:
  static constexpr std::uint64_t to_montgomery(const std::uint64_t b) { return b; }
  static constexpr std::uint64_t from_montgomery(const std::uint64_t b) { return b; }
};
template <class modulus_type> class FixedPoint64Scalar : public detail::modmul_tag<modulus_type, true> {
public:
  static constexpr std::uint64_t to_montgomery(const std::uint64_t b) { return b; }
  static constexpr std::uint64_t from_montgomery(const std::uint64_t b) { return b; }
};

// ---- butterfly layers -------------------------------------------------------------
// <modmul, m, n, inverse_factor = 1, store_precomputation = true>: `radix` fused
// stages of the length-n sub-transforms of a length-m vector
// (layer/sve/radix-two.hpp:18-20 of the reference and its siblings).
namespace detail {

template <std::uint64_t radix, class modmul_type_, std::uint64_t m, std::uint64_t n,
          std::uint64_t inverse_factor>
struct radix_layer {
  using modmul_type = modmul_type_;
  using modulus_type = typename modmul_type::modulus_type;
  static_assert(n >= radix, "layer span smaller than its radix");
  static_assert(is_power_of_two(m) && is_power_of_two(n) && m % n == 0);
  static constexpr std::uint64_t get_radix(void) { return radix; }
  static constexpr std::uint64_t get_m(void) { return m; }
  static constexpr std::uint64_t get_n(void) { return n; }
  static constexpr std::uint64_t get_inverse_factor(void) { return inverse_factor; }
  static constexpr bool uses_fixed_point(void) { return modmul_type::is_fixed_point; }
  static constexpr bool is_six_step_layer{false};
};

} // namespace detail

#define SVENTT_GPU_DEFINE_RADIX_LAYER(NAME, RADIX)                                             \
  template <class modmul_type, std::uint64_t m, std::uint64_t n, std::uint64_t inverse_factor = 1, \
            bool store_precomputation = true>                                                   \
  class NAME : public detail::radix_layer<RADIX, modmul_type, m, n, inverse_factor> {};

SVENTT_GPU_DEFINE_RADIX_LAYER(RadixTwoSVELayer, 2)
SVENTT_GPU_DEFINE_RADIX_LAYER(RadixFourSVELayer, 4)
SVENTT_GPU_DEFINE_RADIX_LAYER(RadixEightSVELayer, 8)
SVENTT_GPU_DEFINE_RADIX_LAYER(RadixTwoScalarLayer, 2)
SVENTT_GPU_DEFINE_RADIX_LAYER(RadixFourScalarLayer, 4)
SVENTT_GPU_DEFINE_RADIX_LAYER(RadixEightScalarLayer, 8)
#undef SVENTT_GPU_DEFINE_RADIX_LAYER

// ---- six-step column layers ---------------------------------------------------------
// m = R x C with R = inner_kernel_type::get_m(): C column transforms of length
// R, then the row twiddle (layer/sve/generic.hpp:26-40, blocked-generic.hpp:27-46).
namespace detail {

template <class modmul_type_, std::uint64_t m, class inner_kernel_type_> struct six_step_layer {
  using modmul_type = modmul_type_;
  using modulus_type = typename modmul_type::modulus_type;
  using inner_kernel_type = inner_kernel_type_;
  static_assert(std::is_same_v<modulus_type, typename inner_kernel_type::modulus_type>);
  static_assert(m % inner_kernel_type::get_m() == 0);
  static constexpr std::uint64_t get_m(void) { return m; }
  static constexpr std::uint64_t get_radix(void) { return inner_kernel_type::get_m(); }
  // a six-step layer has no inverse_factor of its own; its inner kernel's layers may
  static constexpr std::uint64_t get_inverse_factor(void) { return inner_kernel_type::get_inverse_factor(); }
  static constexpr bool uses_fixed_point(void) {
    return modmul_type::is_fixed_point || inner_kernel_type::uses_fixed_point();
  }
  static constexpr bool is_six_step_layer{true};
  class buffer_type {};  // scratch lived here in the reference; the GPU passes need none
};

} // namespace detail

template <class modmul_type, std::uint64_t m, class inner_kernel_type,
          std::uint64_t buffer_padding_elements, std::uint64_t twiddle_unroll_count,
          class transposition_type, bool transpose_in_place = false>
class GenericSVELayer : public detail::six_step_layer<modmul_type, m, inner_kernel_type> {
  static_assert(!transpose_in_place ||
                m / inner_kernel_type::get_m() == inner_kernel_type::get_m());
};

template <class modmul_type, std::uint64_t m, class inner_kernel_type,
          std::uint64_t block_padding_elements, std::uint64_t twiddle_unroll_count,
          std::uint64_t block_rows, class transposition_type>
class BlockedGenericSVELayer : public detail::six_step_layer<modmul_type, m, inner_kernel_type> {};

template <class modmul_type, std::uint64_t m, class inner_kernel_type>
class GenericScalarLayer : public detail::six_step_layer<modmul_type, m, inner_kernel_type> {};

// ---- the kernel concept ------------------------------------------------------------------
// What NTT<kernel_type> of the reference calls on its kernel (kernel/iterative.hpp:78-106,
// kernel/recursive.hpp:33-145): prepare_* appends the kernel's auxiliary data to a byte
// vector, compute_* consumes it through a cursor.  Here the auxiliary data of a whole kernel
// is one plan record; the plan itself is shared per kernel_type and direction.
namespace detail {

template <class kernel_type> struct kernel_concept {
  template <class vector_type> static void prepare_forward(vector_type &aux) {
    if constexpr (std::is_same_v<vector_type, FakeByteVector>) {
      aux.push_back(plan_record{});  // sizing only: needs no device
    } else {
      aux.push_back(plan_record{shared_plan<kernel_type, SVENTT_FORWARD>()});
    }
  }

  template <class vector_type> static void prepare_inverse(vector_type &aux) {
    if constexpr (std::is_same_v<vector_type, FakeByteVector>) {
      aux.push_back(plan_record{});
    } else {
      aux.push_back(plan_record{shared_plan<kernel_type, SVENTT_INVERSE>()});
    }
  }

  static void compute_forward(std::uint64_t *const dst, const std::uint64_t *const src,
                              const std::byte *&aux) {
    const plan_record &record{pointer_utility::get_and_advance<plan_record>(aux)};
    throw_on_error(sventt_forward(record.plan, dst, src, nullptr));
  }

  static void compute_forward(std::uint64_t *const dst, const std::byte *&aux) {
    compute_forward(dst, dst, aux);
  }

  static void compute_inverse(std::uint64_t *const dst, const std::uint64_t *const src,
                              const std::byte *&aux) {
    const plan_record &record{pointer_utility::get_and_advance<plan_record>(aux)};
    throw_on_error(sventt_inverse(record.plan, dst, src, nullptr));
  }

  static void compute_inverse(std::uint64_t *const dst, const std::byte *&aux) {
    compute_inverse(dst, dst, aux);
  }
};

} // namespace detail

// ---- kernels -------------------------------------------------------------------------
// IterativeNTT<modulus, m, layers...>: the product of the layer radices must be m
// (kernel/iterative.hpp:24-27 of the reference).
template <class modulus_type_, std::uint64_t m, class... layer_types>
class IterativeNTT : public detail::kernel_concept<IterativeNTT<modulus_type_, m, layer_types...>> {
public:
  using modulus_type = modulus_type_;

private:
  static_assert((std::is_same_v<modulus_type, typename layer_types::modulus_type> && ...));
  static_assert(((layer_types::get_m() == m) && ...));
  // The reference's header demands radix product == m (kernel/iterative.hpp:27).  Its own
  // README.md:46-56 spells the inner kernel of a RecursiveNTT with the PARENT's m and only
  // the lower layers (product m/8 there); both spellings are accepted here.
  static constexpr std::uint64_t radix_product{(layer_types::get_radix() * ...)};
  static_assert(radix_product == m || (radix_product < m && m % radix_product == 0),
                "the layer radices do not multiply to (a divisor of) the transform length");

public:
  static constexpr std::uint64_t get_m(void) { return m; }
  static constexpr std::uint64_t get_span(void) { return radix_product; }
  // no preferred split: the engine plans the decomposition itself
  static constexpr std::uint64_t get_six_step_rows(void) { return 0; }
  // What the inverse divides by: every layer with inverse_factor != 1 multiplies by its
  // modular inverse (layer/sve/radix-two.hpp:208-235 of the reference), so the kernel as a
  // whole scales by the inverse of the product (mod p).  1 = unscaled (README.md:36-68).
  static constexpr std::uint64_t get_inverse_factor(void) {
    std::uint64_t f{1};
    ((f = modulus_type::multiply(f, layer_types::get_inverse_factor() % modulus_type::get_modulus())), ...);
    return f;
  }
  // one plan runs one arithmetic: the FixedPoint64 kernels as soon as any layer names them (the
  // reference's tests mix both per layer, tests/ntt-tests/iterative-scalar-radix8-two12.hpp:11-18)
  static constexpr bool uses_fixed_point(void) { return (layer_types::uses_fixed_point() || ...); }
};

// RecursiveNTT<modulus, m, layer, inner_kernel, separate_twiddle>: one outer layer
// of radix r, then r inner transforms of length m/r; with separate_twiddle the
// outer layer is a six-step column layer (kernel/recursive.hpp:15-31, :61-75).
template <class modulus_type_, std::uint64_t m, class layer_type_, class inner_kernel_type_,
          bool separate_twiddle>
class RecursiveNTT
    : public detail::kernel_concept<
          RecursiveNTT<modulus_type_, m, layer_type_, inner_kernel_type_, separate_twiddle>> {
public:
  using modulus_type = modulus_type_;
  using layer_type = layer_type_;
  using inner_kernel_type = inner_kernel_type_;

private:
  static_assert(std::is_same_v<modulus_type, typename layer_type::modulus_type>);
  static_assert(std::is_same_v<modulus_type, typename inner_kernel_type::modulus_type>);
  static_assert(layer_type::get_m() == m);
  // kernel/recursive.hpp:30 of the reference, plus the README.md:46-56 spelling in which
  // the inner kernel carries the parent's m and spans m / radix.
  static_assert(inner_kernel_type::get_m() * layer_type::get_radix() == m ||
                (inner_kernel_type::get_m() == m &&
                 inner_kernel_type::get_span() * layer_type::get_radix() == m));
  static_assert(!separate_twiddle || layer_type::is_six_step_layer,
                "separate_twiddle needs a Generic/BlockedGeneric layer");

public:
  static constexpr std::uint64_t get_m(void) { return m; }
  static constexpr std::uint64_t get_span(void) { return m; }
  // R of the six-step split n = R x C when the user spelled one out
  static constexpr std::uint64_t get_six_step_rows(void) {
    return (separate_twiddle || layer_type::is_six_step_layer) ? layer_type::get_radix() : 0;
  }
  // product of the inverse factors of the outer layer (and, for a six-step layer, of its
  // column kernel) and of the inner kernel: see IterativeNTT::get_inverse_factor
  static constexpr std::uint64_t get_inverse_factor(void) {
    return modulus_type::multiply(layer_type::get_inverse_factor() % modulus_type::get_modulus(),
                                  inner_kernel_type::get_inverse_factor());
  }
  static constexpr bool uses_fixed_point(void) {
    return layer_type::uses_fixed_point() || inner_kernel_type::uses_fixed_point();
  }
};

} // namespace sventt

#endif /* SVENTT_GPU_PLAN_TYPES_HPP_INCLUDED */
