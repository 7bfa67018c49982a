// sve_ntt_amd/csrc/field64.h -- 64-bit prime-field arithmetic for gfx950 (and host).
//
// Device counterpart of the reference's PAdic64 modular multiplication
// (include/sventt/modmul/sve/p-adic-64.hpp:80-115, the bit_width(N)==64
// branches: every result canonical in [0,N) by compare-and-add) and of its
// add/subtract (:40-62) and butterflies (:117-246).  The reference streams a
// second operand w' = w*N^{-1} mod 2^64 beside every twiddle; on gfx950 that
// would double twiddle traffic while saving nothing (a 64x64->128 product costs
// the same four v_mad_u64_u32 whether or not its low half is kept), so the
// quotient is taken from the low half of a*w instead:
//
//     t = a*w                   (128 bit, four 32x32 v_mad_u64_u32)
//     q = lo64(t) * N^{-1}      (mod 2^64)
//     c = hi64(t) - hi64(q*N)   (+N on borrow)        = a*w*2^-64 mod N
//
// which is the same number PAdic64*::multiply_normalize returns.  Twiddles are
// kept in Montgomery form (w*2^64 mod N), data stays in the plain domain.
//
// gfx950 has no 64-bit integer multiplier: everything is built from
// v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 (measured at ~half the v_add_u32
// rate, tools/ubench_valu.hip), which is what bounds this path.
#pragma once

#include <stdint.h>

#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define F64_HD __host__ __device__ __forceinline__
#else
#define F64_HD inline
#endif

namespace sventt_hip {

typedef uint64_t u64;
typedef uint32_t u32;

struct Field {
  u64 N;     // modulus (odd prime, up to 64 bits)
  u64 Ninv;  // N^{-1} mod 2^64   (Modulus::get_montgomery_inverse, modulus.hpp:36-68)
  u64 negN;  // 2^64 - N
};

F64_HD u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }

// 64-bit add / subtract returning the carry / borrow.  On the device these are
// written with the multiprecision builtins so that the carry comes out of the
// v_add_co/v_addc_co (v_sub_co/v_subb_co) pair itself instead of a separate
// 64-bit compare.
F64_HD bool add64_carry(u64 a, u64 b, u64 &d) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 c0, c1;
  const u32 lo = __builtin_addc((u32)a, (u32)b, 0u, &c0);
  const u32 hi = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
  d = ((u64)hi << 32) | lo;
  return c1 != 0;
#else
  d = a + b;
  return d < a;
#endif
}

F64_HD bool sub64_borrow(u64 a, u64 b, u64 &d) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 c0, c1;
  const u32 lo = __builtin_subc((u32)a, (u32)b, 0u, &c0);
  const u32 hi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
  d = ((u64)hi << 32) | lo;
  return c1 != 0;
#else
  d = a - b;
  return a < b;
#endif
}

// hi:lo = a*b through 32-bit limbs (maps onto four v_mad_u64_u32).
F64_HD void mul64x64(u64 a, u64 b, u64 &hi, u64 &lo) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 m0 = mad32(a0, b0, 0);
  const u64 m1 = mad32(a0, b1, m0 >> 32);
  const u64 m2 = mad32(a1, b0, (u32)m1);
  hi = mad32(a1, b1, m1 >> 32) + (m2 >> 32);
  lo = (m2 << 32) | (u32)m0;
}

F64_HD u64 mulhi64(u64 a, u64 b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 m0h = ((u64)a0 * b0) >> 32;
  const u64 m1 = mad32(a0, b1, m0h);
  const u64 m2 = mad32(a1, b0, (u32)m1);
  return mad32(a1, b1, m1 >> 32) + (m2 >> 32);
}

// a * w * 2^-64 mod N, canonical.  Needs w < N; a may be any 64-bit value.
F64_HD u64 montmul(u64 a, u64 w, const Field &f) {
  u64 thi, tlo;
  mul64x64(a, w, thi, tlo);
  const u32 t0 = (u32)tlo, t1 = (u32)(tlo >> 32);
  const u32 ni0 = (u32)f.Ninv, ni1 = (u32)(f.Ninv >> 32);
  const u64 r0 = mad32(t0, ni0, 0);
  const u32 q0 = (u32)r0;
  const u32 q1 = (u32)(r0 >> 32) + t0 * ni1 + t1 * ni0;
  const u64 g = mulhi64(((u64)q1 << 32) | q0, f.N);
  u64 c;
  const bool borrow = sub64_borrow(thi, g, c);
  return c + (borrow ? f.N : 0);
}

// PAdic64SVE::add, 64-bit-N branch (p-adic-64.hpp:40-50): canonical sum.
// a + b >= N  <=>  a + (b - N mod 2^64) carries out of 64 bits.
F64_HD u64 addmod(u64 a, u64 b, const Field &f) {
  u64 e;
  const bool wrapped = add64_carry(a, b + f.negN, e);
  return e + (wrapped ? 0 : f.N);
}

// PAdic64SVE::subtract (p-adic-64.hpp:52-62): canonical difference.
F64_HD u64 submod(u64 a, u64 b, const Field &f) {
  u64 d;
  const bool borrow = sub64_borrow(a, b, d);
  return d + (borrow ? f.N : 0);
}

// Gentleman-Sande butterfly, PAdic64SVE::butterfly_forward (p-adic-64.hpp:142-178):
//   (x0, x1) -> (x0 + x1, (x0 - x1) * w)
F64_HD void butterfly_fwd(u64 &x0, u64 &x1, u64 w, const Field &f) {
  const u64 s = addmod(x0, x1, f);
  const u64 d = submod(x0, x1, f);
  x0 = s;
  x1 = montmul(d, w, f);
}

// twiddle-less form (p-adic-64.hpp:117-140)
F64_HD void butterfly_fwd(u64 &x0, u64 &x1, const Field &f) {
  const u64 s = addmod(x0, x1, f);
  x1 = submod(x0, x1, f);
  x0 = s;
}

// Cooley-Tukey butterfly, PAdic64SVE::butterfly_inverse (p-adic-64.hpp:225-246):
//   (x0, x1) -> (x0 + x1*w, x0 - x1*w)
F64_HD void butterfly_inv(u64 &x0, u64 &x1, u64 w, const Field &f) {
  const u64 t = montmul(x1, w, f);
  const u64 s = addmod(x0, t, f);
  x1 = submod(x0, t, f);
  x0 = s;
}

F64_HD void butterfly_inv(u64 &x0, u64 &x1, const Field &f) { butterfly_fwd(x0, x1, f); }

// ---- arithmetic back ends -----------------------------------------------------------
// The tile kernels are instantiated per back end (TileNTT's ARITH parameter); every one of them
// takes canonical residues in and gives canonical residues out, so passes of different back
// ends can follow each other and every result equals the oracle's bit for bit.
//   ARITH_MONT   any odd prime < 2^64: Montgomery product above; twiddles as w * 2^64 mod N.
//                (PAdic64SVE / PAdic64Scalar of the reference, modmul/sve/p-adic-64.hpp)
//   ARITH_GOLD   N = 2^64 - 2^32 + 1 only: 2^64 = 2^32 - 1 and 2^96 = -1 (mod N), so a 128-bit
//                product folds with one subtraction and one 32 x 32 multiply-add; twiddles plain.
//                (the "Goldilocks reduction" fast path of SURVEY.md 8f; the reference has none)
//   ARITH_SHOUP  N < 2^63: c = a*w - floor(a*w'/2^64)*N with the precomputed w' = floor(w*2^64/N)
//                stored beside every twiddle (FixedPoint64SVE / FixedPoint64Scalar of the
//                reference, modmul/sve/fixed-point-64.hpp:60-68); twiddles plain, two words each.
enum : int { ARITH_MONT = 0, ARITH_GOLD = 1, ARITH_SHOUP = 2, ARITH_COUNT = 3 };

constexpr u64 GOLDILOCKS_N = 0xffffffff00000001ull;
constexpr u64 GOLDILOCKS_EPS = 0xffffffffull;  // 2^64 mod N = 2^32 - 1

// a * w mod N for N = 2^64 - 2^32 + 1, canonical; w < N, a any 64-bit value.
F64_HD u64 gold_mul(u64 a, u64 w, const Field &f) {
  u64 hi, lo;
  mul64x64(a, w, hi, lo);
  const u32 t2 = (u32)hi, t3 = (u32)(hi >> 32);
  u64 r;
  const bool borrow = sub64_borrow(lo, (u64)t3, r);       // t3 * 2^96 = -t3
  r += borrow ? f.N : 0;                                   // the wrap added 2^64 = eps: take it back (-eps = +N mod 2^64)
  u64 r2;
  const bool carry = add64_carry(r, mad32(t2, (u32)GOLDILOCKS_EPS, 0), r2);  // t2 * 2^64 = t2 * eps
  r2 += carry ? GOLDILOCKS_EPS : 0;                        // cannot wrap again: t2*eps <= 2^64 - 2^33 + 1
  return r2 + (r2 >= f.N ? GOLDILOCKS_EPS : 0);            // - N
}

// FixedPoint64*::multiply (modmul/scalar/fixed-point-64.hpp:47-55) followed by the one
// conditional subtraction that makes the result canonical; w < N < 2^63, a any 64-bit value.
F64_HD u64 shoup_mul(u64 a, u64 w, u64 wp, const Field &f) {
  const u64 q = mulhi64(a, wp);
  const u64 c = a * w - q * f.N;  // in [0, 2N)
  return c - (c >= f.N ? f.N : 0);
}

// One twiddle of a stage / twist table is TW_WORDS consecutive 64-bit words.
template <int ARITH> struct Arith;
template <> struct Arith<ARITH_MONT> {
  static constexpr int TW_WORDS = 1;
  F64_HD static u64 mul(u64 a, const u64 *tw, const Field &f) { return montmul(a, tw[0], f); }
};
template <> struct Arith<ARITH_GOLD> {
  static constexpr int TW_WORDS = 1;
  F64_HD static u64 mul(u64 a, const u64 *tw, const Field &f) { return gold_mul(a, tw[0], f); }
};
template <> struct Arith<ARITH_SHOUP> {
  static constexpr int TW_WORDS = 2;
  F64_HD static u64 mul(u64 a, const u64 *tw, const Field &f) { return shoup_mul(a, tw[0], tw[1], f); }
};

// butterflies with the twiddle taken from a table entry
template <int ARITH> F64_HD void butterfly_fwd_tw(u64 &x0, u64 &x1, const u64 *tw, const Field &f) {
  const u64 s = addmod(x0, x1, f);
  const u64 d = submod(x0, x1, f);
  x0 = s;
  x1 = Arith<ARITH>::mul(d, tw, f);
}
template <int ARITH> F64_HD void butterfly_inv_tw(u64 &x0, u64 &x1, const u64 *tw, const Field &f) {
  const u64 t = Arith<ARITH>::mul(x1, tw, f);
  const u64 s = addmod(x0, t, f);
  x1 = submod(x0, t, f);
  x0 = s;
}

// ---- host-side field helpers (plan construction only) ----------------------
typedef unsigned __int128 u128;
inline u64 h_mulmod(u64 a, u64 b, u64 N) { return (u64)(((u128)a * b) % N); }
inline u64 h_powmod(u64 a, u64 e, u64 N) {
  u64 r = 1 % N;
  a %= N;
  for (; e; e >>= 1) {
    if (e & 1) r = h_mulmod(r, a, N);
    a = h_mulmod(a, a, N);
  }
  return r;
}
inline u64 h_invmod(u64 a, u64 N) { return h_powmod(a, N - 2, N); }
// N^{-1} mod 2^64 by Newton iteration (same value as modulus.hpp:36-68).
inline u64 h_montgomery_inverse(u64 N) {
  u64 x = N;  // correct to 3 bits for odd N
  for (int i = 0; i < 6; ++i) x *= 2 - N * x;
  return x;
}
inline u64 h_to_montgomery(u64 a, u64 N) { return (u64)((((u128)a) << 64) % N); }
// floor(w * 2^64 / N): FixedPoint64*::precompute with its correction (fixed-point-64.hpp:25-44)
inline u64 h_shoup_precompute(u64 w, u64 N) { return (u64)((((u128)w) << 64) / N); }
// appends the table form of the plain residue `a` for a back end
inline void h_push_twiddle(std::vector<u64> &t, int arith, u64 a, u64 N);

inline void h_push_twiddle(std::vector<u64> &t, int arith, u64 a, u64 N) {
  if (arith == ARITH_MONT) {
    t.push_back(h_to_montgomery(a, N));
  } else {
    t.push_back(a);
    if (arith == ARITH_SHOUP) t.push_back(h_shoup_precompute(a, N));
  }
}

}  // namespace sventt_hip
