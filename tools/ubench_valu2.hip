// tools/ubench_valu2.hip -- per-instruction VALU issue cost on gfx950, second pass:
// 8 independent chains per wave, explicit SGPR-pair carry/mask registers so that
// no VCC hazard padding is needed.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu2.hip -o tools/ubench_valu2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITER = 2048;

// Each test body is a string of 8 independent instructions over registers
// a0..a7 (32-bit), b0..b7 (32-bit), c0..c7 (64-bit pairs).
#define DEF(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                               \
  __global__ void NAME(uint32_t *out, uint32_t seed) {                                          \
    uint32_t a0 = seed * (threadIdx.x + 1), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, \
             a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                             \
    uint32_t b0 = a0 ^ 0x9e3779b9u, b1 = a1 ^ 0x9e3779b9u, b2 = a2 ^ 0x9e3779b9u,               \
             b3 = a3 ^ 0x9e3779b9u, b4 = a4 ^ 0x9e3779b9u, b5 = a5 ^ 0x9e3779b9u,               \
             b6 = a6 ^ 0x9e3779b9u, b7 = a7 ^ 0x9e3779b9u;                                      \
    uint64_t c0 = ((uint64_t)a0 << 32) | b0, c1 = ((uint64_t)a1 << 32) | b1,                    \
             c2 = ((uint64_t)a2 << 32) | b2, c3 = ((uint64_t)a3 << 32) | b3,                    \
             c4 = ((uint64_t)a4 << 32) | b4, c5 = ((uint64_t)a5 << 32) | b5,                    \
             c6 = ((uint64_t)a6 << 32) | b6, c7 = ((uint64_t)a7 << 32) | b7;                    \
    uint64_t m0 = seed, m1 = seed + 1;                                                          \
    for (int it = 0; it < ITER; ++it) {                                                         \
      asm volatile(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7     \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),      \
                     "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5),      \
                     "+v"(b6), "+v"(b7), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4),      \
                     "+v"(c5), "+v"(c6), "+v"(c7), "+s"(m0), "+s"(m1)                           \
                   : "s"(seed)                                                                  \
                   : "vcc");                                                                    \
    }                                                                                           \
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;  \
    uint64_t rc = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ m0 ^ m1;                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ (uint32_t)rc ^ (uint32_t)(rc >> 32);       \
  }

// operand numbering: a0..a7 = %0..%7, b0..b7 = %8..%15, c0..c7 = %16..%23, m0 = %24, m1 = %25, seed = %26
#define REP8(F) F("%0", "%8", "%16"), F("%1", "%9", "%17"), F("%2", "%10", "%18"), F("%3", "%11", "%19"), \
                F("%4", "%12", "%20"), F("%5", "%13", "%21"), F("%6", "%14", "%22"), F("%7", "%15", "%23")

#define F_MOV(a, b, c) "v_mov_b32 " a ", " b
#define F_ADD(a, b, c) "v_add_u32 " a ", " a ", " b
#define F_SUB(a, b, c) "v_sub_u32 " a ", " a ", " b
#define F_AND(a, b, c) "v_and_b32 " a ", " a ", " b
#define F_XOR(a, b, c) "v_xor_b32 " a ", " a ", " b
#define F_LSHL(a, b, c) "v_lshlrev_b32 " a ", 3, " a
#define F_ALIGNBIT(a, b, c) "v_alignbit_b32 " a ", " a ", " b ", 7"
#define F_MIN(a, b, c) "v_min_u32 " a ", " a ", " b
#define F_ADDCO(a, b, c) "v_add_co_u32 " a ", %24, " a ", " b
#define F_ADDC(a, b, c) "v_addc_co_u32 " a ", %25, " a ", " b ", %24"
#define F_SUBCO(a, b, c) "v_sub_co_u32 " a ", %24, " a ", " b
#define F_CND(a, b, c) "v_cndmask_b32 " a ", " a ", " b ", %24"
#define F_CNDVCC(a, b, c) "v_cndmask_b32 " a ", " a ", " b ", vcc"
#define F_CMP32(a, b, c) "v_cmp_lt_u32 %25, " a ", " b
#define F_CMP64(a, b, c) "v_cmp_lt_u64 %25, " c ", " c
#define F_CMP32VCC(a, b, c) "v_cmp_lt_u32 vcc, " a ", " b
#define F_LSHLADD64(a, b, c) "v_lshl_add_u64 " c ", " c ", 0, " c
#define F_LSHR64(a, b, c) "v_lshrrev_b64 " c ", 1, " c
#define F_MAD64(a, b, c) "v_mad_u64_u32 " c ", %25, " a ", " b ", " c
#define F_MAD64S(a, b, c) "v_mad_u64_u32 " c ", %25, " a ", %26, " c
#define F_MAD64Z(a, b, c) "v_mad_u64_u32 " c ", %25, " a ", " b ", 0"
#define F_MULLO(a, b, c) "v_mul_lo_u32 " a ", " a ", " b
#define F_MULHI(a, b, c) "v_mul_hi_u32 " a ", " a ", " b
#define F_MUL24(a, b, c) "v_mul_u32_u24 " a ", " a ", " b
#define F_MULHI24(a, b, c) "v_mul_hi_u32_u24 " a ", " a ", " b
#define F_MAD24(a, b, c) "v_mad_u32_u24 " a ", " a ", " b ", " a
#define F_ADD3(a, b, c) "v_add3_u32 " a ", " a ", " b ", " a
#define F_LSHLADD32(a, b, c) "v_lshl_add_u32 " a ", " a ", 2, " b
#define F_BFE(a, b, c) "v_bfe_u32 " a ", " a ", 3, 9"
#define F_PKADD16(a, b, c) "v_pk_add_u16 " a ", " a ", " b
#define F_PKMUL16(a, b, c) "v_pk_mul_lo_u16 " a ", " a ", " b
#define F_PKMAD16(a, b, c) "v_pk_mad_u16 " a ", " a ", " b ", " a
#define F_FMA32(a, b, c) "v_fma_f32 " a ", " a ", " b ", " a
#define F_PKFMA32(a, b, c) "v_pk_fma_f32 " c ", " c ", " c ", " c
#define F_FMA64(a, b, c) "v_fma_f64 " c ", " c ", " c ", " c
#define F_MUL64F(a, b, c) "v_mul_f64 " c ", " c ", " c
#define F_DOT4(a, b, c) "v_dot4_u32_u8 " a ", " a ", " b ", " a
#define F_MOVDPP(a, b, c) "v_mov_b32_dpp " a ", " b " row_ror:8 row_mask:0xf bank_mask:0xf"
#define F_SWIZ(a, b, c) "ds_swizzle_b32 " a ", " a " offset:swizzle(BITMASK_PERM,\"0000p\")"
#define F_PERMLANE32(a, b, c) "v_permlane32_swap_b32 " a ", " b
#define F_MOV64(a, b, c) "v_mov_b64 " c ", " c
#define F_ADDCO_VCC(a, b, c) "v_add_co_u32 " a ", vcc, " a ", " b
#define F_CVT(a, b, c) "v_cvt_f64_u32 " c ", " a
#define F_MADU32(a, b, c) "v_mad_u32_u24 " a ", " a ", " b ", " b

#define DEF_(...) DEF(__VA_ARGS__)
#define MK(NAME, F) DEF_(NAME, REP8(F))
MK(k_mov, F_MOV) MK(k_add, F_ADD) MK(k_sub, F_SUB) MK(k_and, F_AND) MK(k_xor, F_XOR) MK(k_lshl, F_LSHL)
MK(k_alignbit, F_ALIGNBIT) MK(k_min, F_MIN) MK(k_addco, F_ADDCO) MK(k_addc, F_ADDC) MK(k_subco, F_SUBCO)
MK(k_cnd, F_CND) MK(k_cndvcc, F_CNDVCC) MK(k_cmp32, F_CMP32) MK(k_cmp64, F_CMP64) MK(k_cmp32vcc, F_CMP32VCC)
MK(k_lshladd64, F_LSHLADD64) MK(k_lshr64, F_LSHR64) MK(k_mad64, F_MAD64) MK(k_mad64s, F_MAD64S)
MK(k_mad64z, F_MAD64Z) MK(k_mullo, F_MULLO) MK(k_mulhi, F_MULHI) MK(k_mul24, F_MUL24) MK(k_mulhi24, F_MULHI24)
MK(k_mad24, F_MAD24) MK(k_add3, F_ADD3) MK(k_lshladd32, F_LSHLADD32) MK(k_bfe, F_BFE) MK(k_pkadd16, F_PKADD16)
MK(k_pkmul16, F_PKMUL16) MK(k_pkmad16, F_PKMAD16) MK(k_fma32, F_FMA32) MK(k_pkfma32, F_PKFMA32)
MK(k_fma64, F_FMA64) MK(k_mul64f, F_MUL64F) MK(k_dot4, F_DOT4) MK(k_movdpp, F_MOVDPP)
MK(k_permlane32, F_PERMLANE32) MK(k_mov64, F_MOV64) MK(k_addco_vcc, F_ADDCO_VCC) MK(k_cvt, F_CVT)

template <class K> static int timeit(const char *name, K kernel, uint32_t *d, int wavesPerSimd) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 256 * wavesPerSimd;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // wave-instructions per SIMD: wavesPerSimd * 8 * ITER ; time in cycles at 2.4 GHz nominal
  double cyc = best * 1e-3 * 2.4e9 / ((double)wavesPerSimd * 8 * ITER);
  printf("%-12s w/SIMD=%d %7.3f ms  %6.2f cyc/instr(2.4GHz)\n", name, wavesPerSimd, best, cyc);
  return 0;
}

int main() {
  uint32_t *d;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  for (int w : {4, 8}) {
#define RUN(K) timeit(#K, K, d, w);
    RUN(k_mov) RUN(k_add) RUN(k_sub) RUN(k_and) RUN(k_xor) RUN(k_lshl) RUN(k_alignbit) RUN(k_min)
    RUN(k_addco) RUN(k_addc) RUN(k_subco) RUN(k_cnd) RUN(k_cndvcc) RUN(k_cmp32) RUN(k_cmp64) RUN(k_cmp32vcc)
    RUN(k_lshladd64) RUN(k_lshr64) RUN(k_mad64) RUN(k_mad64s) RUN(k_mad64z) RUN(k_mullo) RUN(k_mulhi)
    RUN(k_mul24) RUN(k_mulhi24) RUN(k_mad24) RUN(k_add3) RUN(k_lshladd32) RUN(k_bfe) RUN(k_pkadd16)
    RUN(k_pkmul16) RUN(k_pkmad16) RUN(k_fma32) RUN(k_pkfma32) RUN(k_fma64) RUN(k_mul64f) RUN(k_dot4)
    RUN(k_movdpp) RUN(k_permlane32) RUN(k_mov64) RUN(k_addco_vcc) RUN(k_cvt)
    printf("\n");
  }
  return 0;
}
