// include/sventt/modulus.hpp -- prime field description used at compile time.
//
// API-compatible with the reference's sventt::Modulus<modulus, generator>
// (include/sventt/modulus.hpp:14-133 there): same static member names, same
// results, same std::invalid_argument when an order does not divide p-1.  On the
// GPU build this class is host-side only: the NTT facade reads get_modulus() /
// get_generator() from it and the device tables are built by the engine.
#ifndef SVENTT_GPU_MODULUS_HPP_INCLUDED
#define SVENTT_GPU_MODULUS_HPP_INCLUDED

#include <bit>
#include <cstdint>
#include <stdexcept>

namespace sventt {

template <std::uint64_t modulus, std::uint64_t generator = 0> class Modulus {
  using wide = unsigned __int128;

  static constexpr bool divides_group_order(const std::uint64_t order) {
    return order != 0 && (modulus - 1) % order == 0;
  }

public:
  struct shoup_inverse_type {
    std::uint64_t modulus_inverse_lo, modulus_inverse_hi;
  };

  static constexpr std::uint64_t get_modulus(void) { return modulus; }
  static constexpr std::uint64_t get_generator(void) { return generator; }

  // floor((2^128 - 1) / modulus), or 2^128 / modulus exactly for powers of two.
  static constexpr shoup_inverse_type get_shoup_inverse(void) {
    wide q{};
    if (std::has_single_bit(modulus)) {
      q = wide{1} << (128 - std::countr_zero(modulus));
    } else {
      q = ~wide{0} / modulus;
    }
    return {static_cast<std::uint64_t>(q), static_cast<std::uint64_t>(q >> 64)};
  }

  // modulus^{-1} mod 2^64 by Newton's iteration x <- x(2 - modulus*x); an odd
  // number is its own inverse modulo 8, each round doubles the valid bits.
  static constexpr std::uint64_t get_montgomery_inverse(void) {
    std::uint64_t x{modulus};
    for (int valid_bits{3}; valid_bits < 64; valid_bits *= 2) {
      x *= 2 - modulus * x;
    }
    return x;
  }

  static constexpr std::uint64_t reduce(const std::uint64_t a) { return a % modulus; }

  static constexpr std::uint64_t negate(const std::uint64_t a) { return subtract(0, a); }

  static constexpr std::uint64_t add(const std::uint64_t a, const std::uint64_t b) {
    return static_cast<std::uint64_t>((wide{reduce(a)} + reduce(b)) % modulus);
  }

  static constexpr std::uint64_t subtract(const std::uint64_t a, const std::uint64_t b) {
    return static_cast<std::uint64_t>((wide{reduce(a)} + modulus - reduce(b)) % modulus);
  }

  static constexpr std::uint64_t multiply(const std::uint64_t a, const std::uint64_t b) {
    return static_cast<std::uint64_t>(wide{a} * b % modulus);
  }

  static constexpr std::uint64_t power(std::uint64_t base, std::uint64_t exponent) {
    std::uint64_t result{1};
    while (exponent != 0) {
      if (exponent % 2 != 0) {
        result = multiply(result, base);
      }
      base = multiply(base, base);
      exponent /= 2;
    }
    return result;
  }

  // Fermat: the modulus is assumed prime, as in the reference.
  static constexpr std::uint64_t invert(const std::uint64_t a) { return power(a, modulus - 2); }

  static constexpr std::uint64_t divide(const std::uint64_t a, const std::uint64_t b) {
    return multiply(a, invert(b));
  }

  static constexpr std::uint64_t get_root_forward(const std::uint64_t order)
    requires(generator != 0)
  {
    if (!divides_group_order(order)) {
      throw std::invalid_argument{"the field has no such root"};
    }
    return power(generator, (modulus - 1) / order);
  }

  static constexpr std::uint64_t get_root_inverse(const std::uint64_t order)
    requires(generator != 0)
  {
    if (!divides_group_order(order)) {
      throw std::invalid_argument{"the field has no such root"};
    }
    // generator^{-(p-1)/order} = generator^{(p-1) - (p-1)/order}
    const std::uint64_t step{(modulus - 1) / order};
    return power(generator, (modulus - 1 - step) % (modulus - 1));
  }
};

} // namespace sventt

#endif /* SVENTT_GPU_MODULUS_HPP_INCLUDED */
