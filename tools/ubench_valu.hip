// tools/ubench_valu.hip -- VALU issue-rate probe for the integer instructions the
// 64-bit Montgomery butterfly is made of (gfx950).  Not part of the product:
// it calibrates the VALU side of the roofline discussion in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

#define DEF_KERNEL(NAME, ASM)                                                   \
  __global__ void NAME(uint32_t *out, uint32_t seed) {                          \
    uint32_t a[8], b[8];                                                        \
    uint64_t c[8];                                                              \
    for (int i = 0; i < 8; ++i) {                                               \
      a[i] = seed * (threadIdx.x + 1) + i; b[i] = a[i] ^ 0x9e3779b9u;           \
      c[i] = ((uint64_t)a[i] << 32) | b[i];                                     \
    }                                                                           \
    for (int it = 0; it < ITER; ++it) {                                         \
      ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                   \
    }                                                                           \
    uint32_t r = 0;                                                             \
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i] ^ (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                             \
  }

#define A_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define A_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define A_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define A_ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define A_ADD3(i)  asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define A_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i + 1) & 7]));
#define A_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(b[i]), "+v"(a[(i+1)&7]) :: "vcc");
#define A_CMP64(i) asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(c[i]), "v"(c[(i + 1) & 7]), "v"(b[i]) : "vcc");
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define A_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(c[i]));
#define A_MULLO_S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(seed));
#define A_DOT2(i) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i+1)&7]));
#define A_MULF64(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(c[i]));

DEF_KERNEL(k_mad64, A_MAD64)
DEF_KERNEL(k_mullo, A_MULLO)
DEF_KERNEL(k_mulhi, A_MULHI)
DEF_KERNEL(k_mul24, A_MUL24)
DEF_KERNEL(k_mad24, A_MAD24)
DEF_KERNEL(k_add32, A_ADD32)
DEF_KERNEL(k_add3, A_ADD3)
DEF_KERNEL(k_lshladd64, A_LSHLADD64)
DEF_KERNEL(k_addco, A_ADDCO)
DEF_KERNEL(k_cmp64, A_CMP64)
DEF_KERNEL(k_cndmask, A_CNDMASK)
DEF_KERNEL(k_mov, A_MOV)
DEF_KERNEL(k_lshl64, A_LSHL64)
DEF_KERNEL(k_mullo_s, A_MULLO_S)
DEF_KERNEL(k_dot2, A_DOT2)
DEF_KERNEL(k_fmaf64, A_MULF64)

// Compiler-generated Montgomery butterfly, 8 independent chains per thread.
__device__ __forceinline__ uint64_t mad32(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
__device__ __forceinline__ uint64_t montmul(uint64_t a, uint64_t b, uint64_t N, uint64_t Ninv) {
  uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  uint64_t m0 = mad32(a0, b0, 0);
  uint64_t m1 = mad32(a0, b1, m0 >> 32);
  uint64_t m2 = mad32(a1, b0, (uint32_t)m1);
  uint64_t thi = mad32(a1, b1, m1 >> 32) + (m2 >> 32);
  uint32_t t0 = (uint32_t)m0, t1 = (uint32_t)m2;
  uint32_t ni0 = (uint32_t)Ninv, ni1 = (uint32_t)(Ninv >> 32);
  uint64_t r0 = mad32(t0, ni0, 0);
  uint32_t q0 = (uint32_t)r0;
  uint32_t q1 = (uint32_t)(r0 >> 32) + t0 * ni1 + t1 * ni0;
  uint32_t n0 = (uint32_t)N, n1 = (uint32_t)(N >> 32);
  uint64_t k0 = mad32(q0, n0, 0);
  uint64_t k1 = mad32(q0, n1, k0 >> 32);
  uint64_t k2 = mad32(q1, n0, (uint32_t)k1);
  uint64_t g = mad32(q1, n1, k1 >> 32) + (k2 >> 32);
  uint64_t c = thi - g;
  if (thi < g) c += N;
  return c;
}
__device__ __forceinline__ uint64_t addmod(uint64_t a, uint64_t b, uint64_t N) {
  uint64_t s = a + b;
  return (s < a || s >= N) ? s - N : s;
}
__device__ __forceinline__ uint64_t submod(uint64_t a, uint64_t b, uint64_t N) {
  uint64_t d = a - b;
  return (a < b) ? d + N : d;
}
__global__ void k_montmul(uint64_t *out, uint64_t N, uint64_t Ninv, uint64_t seed) {
  uint64_t x[8], w[8];
  for (int i = 0; i < 8; ++i) { x[i] = (seed * (threadIdx.x + 3 + i)) % N; w[i] = (x[i] * 7 + i) % N; }
  for (int it = 0; it < ITER / 8; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = montmul(x[i], w[i], N, Ninv);
  }
  uint64_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_butterfly(uint64_t *out, uint64_t N, uint64_t Ninv, uint64_t seed) {
  uint64_t x[8], w[4];
  for (int i = 0; i < 8; ++i) x[i] = (seed * (threadIdx.x + 3 + i)) % N;
  for (int i = 0; i < 4; ++i) w[i] = (x[i] * 7 + i) % N;
  for (int it = 0; it < ITER / 8; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint64_t a = x[i], b = x[i + 4];
      x[i] = addmod(a, b, N);
      x[i + 4] = montmul(submod(a, b, N), w[i], N, Ninv);
    }
    // rotate roles so values keep mixing
    uint64_t t = x[0]; x[0] = x[5]; x[5] = x[2]; x[2] = x[7]; x[7] = t;
  }
  uint64_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class F> static int timeit(const char *name, double ops_per_thread, int wavesPerSimd, F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 256 * wavesPerSimd;  // 256 threads = 4 waves -> 1 wave per SIMD per block
  launch(blocks);  // warmup
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    launch(blocks);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double total = ops_per_thread * 256.0 * blocks;
  double rate = total / (best * 1e-3);
  // cycles per wave-instruction per SIMD, assuming 2.4 GHz, 1024 SIMDs
  double cyc = (2.4e9 * 1024.0) / (rate / 64.0);
  printf("%-14s waves/SIMD=%d  %8.3f ms  %8.2f Tlane-op/s  ~%5.2f cyc/wave-instr/SIMD (at 2.4GHz)\n",
         name, wavesPerSimd, best, rate / 1e12, cyc);
  return 0;
}

int main() {
  uint32_t *d32; uint64_t *d64;
  CHECK(hipMalloc(&d32, 256 * 8 * 256 * 4 * 4));
  CHECK(hipMalloc(&d64, 256 * 8 * 256 * 8 * 4));
  const uint64_t N = 0xfffffc6e80000001ull, Ninv = 0x4000039180000001ull;
  for (int w : {1, 2, 4, 8}) {
#define RUN(K, OPS) timeit(#K, OPS, w, [&](int blocks) { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, d32, 12345u); });
    RUN(k_mad64, 8.0 * ITER) RUN(k_mullo, 8.0 * ITER) RUN(k_mulhi, 8.0 * ITER) RUN(k_mul24, 8.0 * ITER)
    RUN(k_mad24, 8.0 * ITER) RUN(k_add32, 8.0 * ITER) RUN(k_add3, 8.0 * ITER) RUN(k_lshladd64, 8.0 * ITER)
    RUN(k_addco, 16.0 * ITER) RUN(k_cmp64, 16.0 * ITER) RUN(k_cndmask, 8.0 * ITER) RUN(k_mov, 8.0 * ITER)
    RUN(k_lshl64, 8.0 * ITER) RUN(k_mullo_s, 8.0 * ITER) RUN(k_dot2, 8.0 * ITER) RUN(k_fmaf64, 8.0 * ITER)
    timeit("montmul", (double)ITER, w, [&](int blocks) { hipLaunchKernelGGL(k_montmul, dim3(blocks), dim3(256), 0, 0, d64, N, Ninv, 0x1234567ull); });
    timeit("butterfly", (double)ITER / 2, w, [&](int blocks) { hipLaunchKernelGGL(k_butterfly, dim3(blocks), dim3(256), 0, 0, d64, N, Ninv, 0x1234567ull); });
    printf("\n");
  }
  return 0;
}
