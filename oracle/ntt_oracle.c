/*
 * oracle/ntt_oracle.c -- CPU restatement of the reference's scalar NTT path.
 *
 * TEST INFRASTRUCTURE ONLY (see ntt_oracle.h).  Parity pinned against the
 * reference header compiled in place (oracle/_ref) and tests/golden/.
 */
#include "ntt_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* tests/ntt-reference.hpp:16-18 */
uint64_t oracle_modmul(uint64_t x, uint64_t y, uint64_t N) {
  return (uint64_t)(((u128)x * y) % N);
}

/* tests/ntt-reference.hpp:20-29 */
uint64_t oracle_modpow(uint64_t x, uint64_t e, uint64_t N) {
  uint64_t acc = 1;
  while (e != 0) {
    if (e & 1u)
      acc = oracle_modmul(acc, x, N);
    x = oracle_modmul(x, x, N);
    e >>= 1;
  }
  return acc;
}

/* include/sventt/modulus.hpp:76-80 (operands reduced first, as there) */
uint64_t oracle_modadd(uint64_t a, uint64_t b, uint64_t N) {
  a %= N;
  b %= N;
  return (a < N - b) ? a + b : a + b - N;
}

/* include/sventt/modulus.hpp:82-86 */
uint64_t oracle_modsub(uint64_t a, uint64_t b, uint64_t N) {
  a %= N;
  b %= N;
  return (a >= b) ? a - b : a - b + N;
}

/* include/sventt/modulus.hpp:45-67: seed correct to 5 bits, then four
 * Goldschmidt refinements doubling the number of correct low bits. */
uint64_t oracle_montgomery_inverse(uint64_t N) {
  uint64_t num = (N * 3u) ^ 2u;
  uint64_t den = num * N;
  for (int round = 0; round < 4; ++round) {
    const uint64_t t = 2u - den;
    num *= t;
    den *= t;
  }
  return num;
}

/* include/sventt/modulus.hpp:115-122 */
uint64_t oracle_root_forward(uint64_t N, uint64_t g, uint64_t order, int *ok) {
  if (order == 0 || (N - 1) % order != 0) {
    if (ok)
      *ok = 0;
    return 0;
  }
  if (ok)
    *ok = 1;
  return oracle_modpow(g, (N - 1) / order, N);
}

/* include/sventt/modulus.hpp:124-132: exponent (N-1)/order * (N-2) taken
 * modulo N-1, i.e. the negated exponent. */
uint64_t oracle_root_inverse(uint64_t N, uint64_t g, uint64_t order, int *ok) {
  if (order == 0 || (N - 1) % order != 0) {
    if (ok)
      *ok = 0;
    return 0;
  }
  if (ok)
    *ok = 1;
  const uint64_t e = oracle_modmul((N - 1) / order, N - 2, N - 1);
  return oracle_modpow(g, e, N);
}

/* include/sventt/modmul/scalar/p-adic-64.hpp:16-19: b * (2^64 mod N) mod N;
 * the reference writes 2^64 mod N as the unsigned negation of N. */
uint64_t oracle_to_montgomery(uint64_t b, uint64_t N) {
  return oracle_modmul(b, (uint64_t)(0u - N), N);
}

/* include/sventt/modmul/scalar/p-adic-64.hpp:21-24 */
uint64_t oracle_from_montgomery(uint64_t b, uint64_t N) {
  const uint64_t r = (uint64_t)(0u - N) % N;
  return oracle_modmul(b, oracle_modpow(r, N - 2, N), N);
}

/* include/sventt/modmul/scalar/p-adic-64.hpp:26-29 */
uint64_t oracle_padic_precompute(uint64_t b, uint64_t N) {
  return b * oracle_montgomery_inverse(N);
}

/* include/sventt/modmul/sve/p-adic-64.hpp:101-115, else-branch: the high
 * halves are subtracted and N is added back when the subtraction borrowed. */
uint64_t oracle_padic_multiply_normalize(uint64_t a, uint64_t b, uint64_t bp,
                                         uint64_t N) {
  const uint64_t q = a * bp;
  const uint64_t ab_hi = (uint64_t)(((u128)a * b) >> 64);
  const uint64_t qn_hi = (uint64_t)(((u128)q * N) >> 64);
  uint64_t c = ab_hi - qn_hi;
  if (ab_hi < qn_hi)
    c += N;
  return c;
}

/* include/sventt/utility.hpp:12-23 */
uint64_t oracle_bitreverse64(uint64_t x) {
  uint64_t r = 0;
  for (int i = 0; i < 64; ++i) {
    r = (r << 1) | (x & 1u);
    x >>= 1;
  }
  return r;
}

static int is_pow2(uint64_t m) { return m != 0 && (m & (m - 1)) == 0; }

static unsigned ilog2(uint64_t m) {
  unsigned l = 0;
  while ((m >> l) > 1)
    ++l;
  return l;
}

/* tests/ntt-reference.hpp:43-61.  Stage order, butterfly form and the running
 * twiddle omega_2l_j are kept exactly; the first stage reads src, later ones
 * read dst. */
int oracle_ntt_forward(uint64_t *dst, const uint64_t *src, uint64_t m,
                       uint64_t N, uint64_t g) {
  if (!is_pow2(m))
    return -1;
  const unsigned log2m = ilog2(m);
  uint64_t w_stage = oracle_modpow(g, (N - 1) >> log2m, N); /* omega_m */
  if (log2m == 0) {
    /* the reference's loop body never runs for m == 1: dst is left as is */
    return 0;
  }
  const uint64_t *in = src;
  for (unsigned s = log2m; s-- > 0;) {
    const uint64_t half = (uint64_t)1 << s;
    uint64_t w = 1;
    for (uint64_t j = 0; j < half; ++j) {
      for (uint64_t k = j; k < m; k += 2 * half) {
        const uint64_t x0 = in[k], x1 = in[k + half];
        dst[k] = (x0 < N - x1) ? x0 + x1 : x0 + x1 - N;
        dst[k + half] =
            oracle_modmul((x0 >= x1) ? x0 - x1 : x0 - x1 + N, w, N);
      }
      w = oracle_modmul(w, w_stage, N);
    }
    w_stage = oracle_modmul(w_stage, w_stage, N);
    in = dst;
  }
  return 0;
}

/* tests/ntt-reference.hpp:63-83 */
int oracle_ntt_inverse(uint64_t *dst, const uint64_t *src, uint64_t m,
                       uint64_t N, uint64_t g) {
  if (!is_pow2(m))
    return -1;
  const unsigned log2m = ilog2(m);
  const uint64_t omega_m = oracle_modpow(g, (N - 1) >> log2m, N);
  const uint64_t omegainv_m = oracle_modpow(omega_m, N - 2, N);
  const uint64_t minv = oracle_modpow(m, N - 2, N);
  for (uint64_t i = 0; i < m; ++i)
    dst[i] = oracle_modmul(src[i], minv, N);
  for (unsigned s = 0; s < log2m; ++s) {
    const uint64_t half = (uint64_t)1 << s;
    const uint64_t w_stage =
        oracle_modpow(omegainv_m, (uint64_t)1 << (log2m - s - 1), N);
    uint64_t w = 1;
    for (uint64_t j = 0; j < half; ++j) {
      for (uint64_t k = j; k < m; k += 2 * half) {
        const uint64_t x0 = dst[k];
        const uint64_t x1 = oracle_modmul(dst[k + half], w, N);
        dst[k] = (x0 < N - x1) ? x0 + x1 : x0 + x1 - N;
        dst[k + half] = (x0 >= x1) ? x0 - x1 : x0 - x1 + N;
      }
      w = oracle_modmul(w, w_stage, N);
    }
  }
  return 0;
}

/* Strided gather/scatter helpers for the six-step restatement. */
static void gather_col(uint64_t *col, const uint64_t *a, uint64_t R, uint64_t C,
                       uint64_t c) {
  for (uint64_t r = 0; r < R; ++r)
    col[r] = a[r * C + c];
}
static void scatter_col(uint64_t *a, const uint64_t *col, uint64_t R,
                        uint64_t C, uint64_t c) {
  for (uint64_t r = 0; r < R; ++r)
    a[r * C + c] = col[r];
}

/* include/sventt/kernel/recursive.hpp:61-75 (separate_twiddle driver) over
 * include/sventt/layer/sve/generic.hpp:112-161 (column phase; the transposes
 * there only change where the column NTTs run) and :95-105 / :169-188 (row j
 * is scaled by omega_j^i with omega_j = omega_m^(bitreverse(j) >> (65 -
 * bit_width(R)))). */
int oracle_ntt_forward_sixstep(uint64_t *dst, const uint64_t *src, uint64_t m,
                               uint64_t R, uint64_t N, uint64_t g) {
  if (!is_pow2(m) || !is_pow2(R) || R > m)
    return -1;
  const uint64_t C = m / R;
  const unsigned log2R = ilog2(R);
  uint64_t *col = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (R > C ? R : C));
  if (!col)
    return -2;
  uint64_t *tmp = col + (R > C ? R : C);
  for (uint64_t c = 0; c < C; ++c) {
    gather_col(col, src, R, C, c);
    if (R > 1) {
      oracle_ntt_forward(tmp, col, R, N, g);
      scatter_col(dst, tmp, R, C, c);
    } else {
      scatter_col(dst, col, R, C, c);
    }
  }
  const uint64_t omega_m = oracle_modpow(g, (N - 1) / m, N);
  for (uint64_t j = 0; j < R; ++j) {
    const uint64_t jr = log2R ? (oracle_bitreverse64(j) >> (64 - log2R)) : 0;
    const uint64_t omega_j = oracle_modpow(omega_m, jr, N);
    uint64_t w = 1;
    uint64_t *row = dst + j * C;
    for (uint64_t i = 0; i < C; ++i) {
      row[i] = oracle_modmul(row[i], w, N);
      w = oracle_modmul(w, omega_j, N);
    }
    if (C > 1) {
      memcpy(col, row, sizeof(uint64_t) * C);
      oracle_ntt_forward(row, col, C, N, g);
    }
  }
  free(col);
  return 0;
}

/* include/sventt/kernel/recursive.hpp:116-130: rows inverse + inverse twiddle,
 * then the column phase.  Each inner inverse scales by its own length^{-1}
 * (as the oracle's inverse does), so the product is m^{-1}. */
int oracle_ntt_inverse_sixstep(uint64_t *dst, const uint64_t *src, uint64_t m,
                               uint64_t R, uint64_t N, uint64_t g) {
  if (!is_pow2(m) || !is_pow2(R) || R > m)
    return -1;
  const uint64_t C = m / R;
  const unsigned log2R = ilog2(R);
  uint64_t *col = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (R > C ? R : C));
  if (!col)
    return -2;
  uint64_t *tmp = col + (R > C ? R : C);
  int ok = 1;
  const uint64_t omegainv_m = oracle_root_inverse(N, g, m, &ok);
  for (uint64_t j = 0; j < R; ++j) {
    const uint64_t jr = log2R ? (oracle_bitreverse64(j) >> (64 - log2R)) : 0;
    const uint64_t omega_j = oracle_modpow(omegainv_m, jr, N);
    uint64_t *row = dst + j * C;
    if (C > 1)
      oracle_ntt_inverse(row, src + j * C, C, N, g);
    else
      row[0] = src[j * C];
    uint64_t w = 1;
    for (uint64_t i = 0; i < C; ++i) {
      row[i] = oracle_modmul(row[i], w, N);
      w = oracle_modmul(w, omega_j, N);
    }
  }
  for (uint64_t c = 0; c < C; ++c) {
    gather_col(col, dst, R, C, c);
    if (R > 1) {
      oracle_ntt_inverse(tmp, col, R, N, g);
      scatter_col(dst, tmp, R, C, c);
    }
  }
  free(col);
  return ok ? 0 : -1;
}

/* tests/bench-ntt.cpp:31-33 with tests/utility.hpp:140-154 (iota) */
void oracle_fill_iota(uint64_t *dst, uint64_t m, uint64_t start) {
  for (uint64_t i = 0; i < m; ++i)
    dst[i] = start + i;
}

static uint64_t splitmix64_next(uint64_t *state) {
  uint64_t z = (*state += UINT64_C(0x9e3779b97f4a7c15));
  z = (z ^ (z >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
  z = (z ^ (z >> 27)) * UINT64_C(0x94d049bb133111eb);
  return z ^ (z >> 31);
}

void oracle_fill_splitmix(uint64_t *dst, uint64_t m, uint64_t seed,
                          uint64_t N) {
  uint64_t state = seed;
  for (uint64_t i = 0; i < m; ++i) {
    uint64_t v;
    do {
      v = splitmix64_next(&state);
    } while (v >= N);
    dst[i] = v;
  }
}

void oracle_digest(const uint64_t *v, uint64_t m, uint64_t out[3]) {
  uint64_t fnv = UINT64_C(0xcbf29ce484222325), x = 0, s = 0;
  for (uint64_t i = 0; i < m; ++i) {
    uint64_t w = v[i];
    x ^= w;
    s += w;
    for (int b = 0; b < 8; ++b) {
      fnv ^= (w & 0xffu);
      fnv *= UINT64_C(0x100000001b3);
      w >>= 8;
    }
  }
  out[0] = fnv;
  out[1] = x;
  out[2] = s;
}
