/*
 * oracle/ntt_oracle.h -- CPU restatement of the reference's scalar NTT path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sve_ntt_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Parity is PINNED: this restatement is checked (tests/test_oracle.py) against
 *   - the real reference header compiled in place (oracle/_ref, see Makefile),
 *   - the golden vectors under tests/golden/ that the real reference generated,
 *   - the closed-form known answers of the reference's own
 *     tests/test-ntt-reference.cpp:45-85 and tests/test-modulus.cpp:17-46.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the upstream repository root).
 */
#ifndef NTT_ORACLE_H_INCLUDED
#define NTT_ORACLE_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tests/ntt-reference.hpp:16-18 -- x*y mod N through a 128-bit product. */
uint64_t oracle_modmul(uint64_t x, uint64_t y, uint64_t N);
/* tests/ntt-reference.hpp:20-29 -- square-and-multiply. */
uint64_t oracle_modpow(uint64_t x, uint64_t e, uint64_t N);

/* include/sventt/modulus.hpp:76-88 -- canonical add / subtract. */
uint64_t oracle_modadd(uint64_t a, uint64_t b, uint64_t N);
uint64_t oracle_modsub(uint64_t a, uint64_t b, uint64_t N);

/* include/sventt/modulus.hpp:36-68 -- N^{-1} mod 2^64 (Newton/Goldschmidt). */
uint64_t oracle_montgomery_inverse(uint64_t N);
/* include/sventt/modulus.hpp:115-122 -- g^((N-1)/order); returns 0 and sets
 * *ok=0 when order does not divide N-1 (the reference throws). */
uint64_t oracle_root_forward(uint64_t N, uint64_t g, uint64_t order, int *ok);
/* include/sventt/modulus.hpp:124-132 -- inverse root of the same order. */
uint64_t oracle_root_inverse(uint64_t N, uint64_t g, uint64_t order, int *ok);

/* include/sventt/modmul/scalar/p-adic-64.hpp:16-29 -- Montgomery domain. */
uint64_t oracle_to_montgomery(uint64_t b, uint64_t N);
uint64_t oracle_from_montgomery(uint64_t b, uint64_t N);
uint64_t oracle_padic_precompute(uint64_t b, uint64_t N);
/* include/sventt/modmul/sve/p-adic-64.hpp:98-115 (multiply_normalize, the
 * bit_width(N)==64 branch: compare-and-add, canonical result). */
uint64_t oracle_padic_multiply_normalize(uint64_t a, uint64_t b, uint64_t bp,
                                         uint64_t N);

/* include/sventt/utility.hpp:12-23 -- 64-bit bit reversal. */
uint64_t oracle_bitreverse64(uint64_t x);

/* tests/ntt-reference.hpp:43-61 -- radix-2 DIF, natural in, bit-reversed out.
 * Returns 0 on success, -1 if m is not a power of two (reference throws). */
int oracle_ntt_forward(uint64_t *dst, const uint64_t *src, uint64_t m,
                       uint64_t N, uint64_t g);
/* tests/ntt-reference.hpp:63-83 -- scale by m^{-1}, radix-2 DIT,
 * bit-reversed in, natural out. */
int oracle_ntt_inverse(uint64_t *dst, const uint64_t *src, uint64_t m,
                       uint64_t N, uint64_t g);

/* Six-step restatement (include/sventt/kernel/recursive.hpp:61-75 with
 * include/sventt/layer/sve/generic.hpp:95-161): view src as R x C row-major,
 * C column NTTs of length R, row j times omega_m^(bitrev_R(j)*i), R row NTTs
 * of length C.  Used to pin the decomposition the GPU plan uses; must equal
 * oracle_ntt_forward bit for bit. */
int oracle_ntt_forward_sixstep(uint64_t *dst, const uint64_t *src, uint64_t m,
                               uint64_t R, uint64_t N, uint64_t g);
int oracle_ntt_inverse_sixstep(uint64_t *dst, const uint64_t *src, uint64_t m,
                               uint64_t R, uint64_t N, uint64_t g);

/* tests/bench-ntt.cpp:31-33 -- src[i] = start + i. */
void oracle_fill_iota(uint64_t *dst, uint64_t m, uint64_t start);
/* SURVEY.md 8(d) input I2: splitmix64 stream, values >= N rejected. */
void oracle_fill_splitmix(uint64_t *dst, uint64_t m, uint64_t seed, uint64_t N);

/* Order-sensitive digest of a vector (FNV-1a over the little-endian bytes),
 * plus XOR and wrapping sum, for fixtures too large to commit in full. */
void oracle_digest(const uint64_t *v, uint64_t m, uint64_t out[3]);

#ifdef __cplusplus
}
#endif

#endif /* NTT_ORACLE_H_INCLUDED */
