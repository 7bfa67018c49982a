// oracle/ref_shim.cpp -- C entry points around the REAL reference headers.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
// the upstream headers where they lie (REFERENCE_ROOT, default /root/reference)
// and forwards to them, so that oracle/_ref/libntt_ref.so *is* the reference's
// scalar path (tests/ntt-reference.hpp: class NTTReference) and its constexpr
// field (include/sventt/modulus.hpp: class Modulus).  Built by oracle/Makefile
// into oracle/_ref/ only; /root/reference does not exist on the GPU box, the
// prebuilt .so travels there instead.
#include <cstdint>
#include <stdexcept>

#include "tests/ntt-reference.hpp"
#include "include/sventt/modulus.hpp"
#include "include/sventt/utility.hpp"
#include "include/sventt/modmul/scalar/p-adic-64.hpp"

namespace {

// The reference's Modulus is a compile-time template; expose the primes the
// reference itself uses (README.md:19, tests/ntt-tests/*.hpp:4-5,
// tests/test-modulus.cpp:13, tests/test-ntt-reference.cpp:17-23).
template <class M> struct field_ops {
  static std::uint64_t root_forward(std::uint64_t order, int *ok) {
    try {
      *ok = 1;
      return M::get_root_forward(order);
    } catch (const std::invalid_argument &) {
      *ok = 0;
      return 0;
    }
  }
  static std::uint64_t root_inverse(std::uint64_t order, int *ok) {
    try {
      *ok = 1;
      return M::get_root_inverse(order);
    } catch (const std::invalid_argument &) {
      *ok = 0;
      return 0;
    }
  }
};

using M_baseline = sventt::Modulus<UINT64_C(0xfffffc6e80000001), 3>;
using M_test62 = sventt::Modulus<UINT64_C(0x3a00000000000001), 3>;
using M_goldilocks = sventt::Modulus<UINT64_C(0xffffffff00000001), 7>;

template <class F>
std::uint64_t dispatch(std::uint64_t N, int *ok, F &&f) {
  switch (N) {
  case UINT64_C(0xfffffc6e80000001):
    return f(M_baseline{});
  case UINT64_C(0x3a00000000000001):
    return f(M_test62{});
  case UINT64_C(0xffffffff00000001):
    return f(M_goldilocks{});
  default:
    *ok = -1; /* prime not instantiated in this shim */
    return 0;
  }
}

} // namespace

extern "C" {

int ref_ntt_forward(std::uint64_t *dst, const std::uint64_t *src,
                    std::uint64_t m, std::uint64_t N, std::uint64_t g) {
  try {
    const NTTReference ntt{m, N, g};
    ntt.compute_forward(dst, src);
    return 0;
  } catch (const std::invalid_argument &) {
    return -1;
  }
}

int ref_ntt_inverse(std::uint64_t *dst, const std::uint64_t *src,
                    std::uint64_t m, std::uint64_t N, std::uint64_t g) {
  try {
    const NTTReference ntt{m, N, g};
    ntt.compute_inverse(dst, src);
    return 0;
  } catch (const std::invalid_argument &) {
    return -1;
  }
}

std::uint64_t ref_root_forward(std::uint64_t N, std::uint64_t order, int *ok) {
  return dispatch(N, ok, [&](auto m) {
    return field_ops<decltype(m)>::root_forward(order, ok);
  });
}

std::uint64_t ref_root_inverse(std::uint64_t N, std::uint64_t order, int *ok) {
  return dispatch(N, ok, [&](auto m) {
    return field_ops<decltype(m)>::root_inverse(order, ok);
  });
}

std::uint64_t ref_montgomery_inverse(std::uint64_t N, int *ok) {
  *ok = 1;
  return dispatch(
      N, ok, [&](auto m) { return decltype(m)::get_montgomery_inverse(); });
}

std::uint64_t ref_generator(std::uint64_t N, int *ok) {
  *ok = 1;
  return dispatch(N, ok, [&](auto m) { return decltype(m)::get_generator(); });
}

std::uint64_t ref_to_montgomery(std::uint64_t N, std::uint64_t b, int *ok) {
  *ok = 1;
  return dispatch(N, ok, [&](auto m) {
    return sventt::PAdic64Scalar<decltype(m)>::to_montgomery(b);
  });
}

std::uint64_t ref_from_montgomery(std::uint64_t N, std::uint64_t b, int *ok) {
  *ok = 1;
  return dispatch(N, ok, [&](auto m) {
    return sventt::PAdic64Scalar<decltype(m)>::from_montgomery(b);
  });
}

std::uint64_t ref_padic_precompute(std::uint64_t N, std::uint64_t b, int *ok) {
  *ok = 1;
  return dispatch(N, ok, [&](auto m) {
    return sventt::PAdic64Scalar<decltype(m)>::precompute(b);
  });
}

// PAdic64Scalar::multiply returns a value in (0, 2N) that is only meaningful
// when 2N < 2^64 (include/sventt/modmul/scalar/p-adic-64.hpp:35-45); exposed
// for the 62-bit test prime.
std::uint64_t ref_padic_multiply_lazy(std::uint64_t N, std::uint64_t a,
                                      std::uint64_t b, std::uint64_t bp,
                                      int *ok) {
  *ok = 1;
  return dispatch(N, ok, [&](auto m) {
    return sventt::PAdic64Scalar<decltype(m)>::multiply(a, b, bp);
  });
}

std::uint64_t ref_bitreverse(std::uint64_t x) { return sventt::bitreverse(x); }

} // extern "C"
