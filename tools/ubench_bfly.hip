// tools/ubench_bfly.hip -- A/B of butterfly formulations (cycles per wave-butterfly per SIMD).
// Variant 0: conditional +N through v_cndmask pairs (what hipcc makes of `c ? N : 0`).
// Variant 1: conditional +N under an EXEC mask (s_and_saveexec / one v_lshl_add_u64).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64; typedef uint32_t u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)
struct Field { u64 N, Ninv, negN; };
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
__device__ __forceinline__ bool sub64(u64 a, u64 b, u64 &d) {
  u32 c0, c1; u32 lo = __builtin_subc((u32)a, (u32)b, 0u, &c0);
  u32 hi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), c0, &c1); d = ((u64)hi << 32) | lo; return c1 != 0; }
__device__ __forceinline__ bool add64(u64 a, u64 b, u64 &d) {
  u32 c0, c1; u32 lo = __builtin_addc((u32)a, (u32)b, 0u, &c0);
  u32 hi = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1); d = ((u64)hi << 32) | lo; return c1 != 0; }
template <int V> __device__ __forceinline__ u64 cond_add(u64 d, bool c, u64 N) {
  if constexpr (V == 0) return d + (c ? N : 0);
  else if constexpr (V == 1) { if (c) { asm volatile("" : "+v"(d)); d += N; } return d; }
  else {
    // Variant 2/3: EXEC-masked add in one asm block, no branch (3 = without the s_nop)
    const u64 mask = __builtin_amdgcn_ballot_w64(c);
    u64 save;
    if constexpr (V == 2)
      asm volatile("s_and_saveexec_b64 %1, %2\n\ts_nop 0\n\tv_lshl_add_u64 %0, %0, 0, %3\n\ts_mov_b64 exec, %1"
                   : "+v"(d), "=&s"(save) : "s"(mask), "s"(N) : "scc");
    else
      asm volatile("s_and_saveexec_b64 %1, %2\n\tv_lshl_add_u64 %0, %0, 0, %3\n\ts_mov_b64 exec, %1"
                   : "+v"(d), "=&s"(save) : "s"(mask), "s"(N) : "scc");
    return d;
  }
}
__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 m0h = __umulhi(a0, b0); const u64 m1 = mad32(a0, b1, m0h); const u64 m2 = mad32(a1, b0, (u32)m1);
  return mad32(a1, b1, m1 >> 32) + (m2 >> 32); }
// Variants 4/5: the cross terms of the 64x64 product through the CARRY-OUT of v_mad_u64_u32
// (R = a0*b1 + Q as a full 64-bit add, the carry goes into the top word with one v_addc),
// instead of zero-extending every 32-bit partial into a 64-bit addend.  4: a*w only; 5: also hi(q*N).
__device__ __forceinline__ void mul_carry(u32 a0, u32 a1, u32 b0, u32 b1, u64 p0hi_or_full, bool have_lo,
                                          u64 &hi, u64 &lo) {
  // p0hi_or_full: a0*b0 (have_lo) or just its high word
  const u64 P = p0hi_or_full;
  const u64 Q = mad32(a1, b0, have_lo ? (P >> 32) : P);
  u64 R, c, c2;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(R), "=s"(c) : "v"(a0), "v"(b1), "v"(Q));
  const u64 S = mad32(a1, b1, R >> 32);
  u32 shi = (u32)(S >> 32), shi2;
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(shi2), "=s"(c2) : "v"(shi), "s"(c));
  hi = ((u64)shi2 << 32) | (u32)S;
  lo = (R << 32) | (u32)P;
}
template <int V> __device__ __forceinline__ u64 montmul(u64 a, u64 w, const Field &f);
template <int V> __device__ __forceinline__ u64 montmul_asm(u64 a, u64 w, const Field &f) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)w, b1 = (u32)(w >> 32);
  u64 thi, tlo;
  mul_carry(a0, a1, b0, b1, mad32(a0, b0, 0), true, thi, tlo);
  const u32 t0 = (u32)tlo, t1 = (u32)(tlo >> 32); const u32 ni0 = (u32)f.Ninv, ni1 = (u32)(f.Ninv >> 32);
  const u64 r0 = mad32(t0, ni0, 0); const u32 q0 = (u32)r0; const u32 q1 = (u32)(r0 >> 32) + t0 * ni1 + t1 * ni0;
  u64 g;
  if constexpr (V == 5) {
    u64 dummy;
    mul_carry(q0, q1, (u32)f.N, (u32)(f.N >> 32), (u64)__umulhi(q0, (u32)f.N), false, g, dummy);
  } else {
    g = mulhi64(((u64)q1 << 32) | q0, f.N);
  }
  u64 c; const bool b = sub64(thi, g, c); return cond_add<0>(c, b, f.N); }
template <int V> __device__ __forceinline__ u64 montmul(u64 a, u64 w, const Field &f) {
  if constexpr (V >= 4) return montmul_asm<V>(a, w, f);
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)w, b1 = (u32)(w >> 32);
  const u64 m0 = mad32(a0, b0, 0); const u64 m1 = mad32(a0, b1, m0 >> 32); const u64 m2 = mad32(a1, b0, (u32)m1);
  const u64 thi = mad32(a1, b1, m1 >> 32) + (m2 >> 32);
  const u32 t0 = (u32)m0, t1 = (u32)m2; const u32 ni0 = (u32)f.Ninv, ni1 = (u32)(f.Ninv >> 32);
  const u64 r0 = mad32(t0, ni0, 0); const u32 q0 = (u32)r0; const u32 q1 = (u32)(r0 >> 32) + t0 * ni1 + t1 * ni0;
  const u64 g = mulhi64(((u64)q1 << 32) | q0, f.N);
  u64 c; const bool b = sub64(thi, g, c); return cond_add<V>(c, b, f.N); }
template <int V> __device__ __forceinline__ u64 addmod(u64 a, u64 b, const Field &f) {
  u64 e; const bool k = add64(a, b + f.negN, e); return cond_add<(V >= 4 ? 0 : V)>(e, !k, f.N); }
template <int V> __device__ __forceinline__ u64 submod(u64 a, u64 b, const Field &f) {
  u64 d; const bool br = sub64(a, b, d); return cond_add<(V >= 4 ? 0 : V)>(d, br, f.N); }
constexpr int ITER = 512;
template <int V> __global__ void k_butterfly(u64 *out, Field f, u64 seed) {
  u64 x[16], w[8];
  for (int i = 0; i < 16; ++i) x[i] = (seed * (threadIdx.x + 3 + i) * 0x9e3779b97f4a7c15ull) % f.N;
  for (int i = 0; i < 8; ++i) w[i] = (x[i] * 7 + i) % f.N;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u64 a = x[i], b = x[i + 8];
      x[i] = addmod<V>(a, b, f);
      x[i + 8] = montmul<V>(submod<V>(a, b, f), w[i], f);
    }
    u64 t = x[0]; x[0] = x[9]; x[9] = x[2]; x[2] = x[11]; x[11] = x[4]; x[4] = x[13]; x[13] = t;
  }
  u64 r = 0; for (int i = 0; i < 16; ++i) r ^= x[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class K> static int timeit(const char *name, K kernel, u64 *d, int wavesPerSimd) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const Field f{0xfffffc6e80000001ull, 0x4000039180000001ull, 0ull - 0xfffffc6e80000001ull};
  const int blocks = 256 * wavesPerSimd;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, f, 0x1234567ull); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, f, 0x1234567ull);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  u64 h0 = 0; CHECK(hipMemcpy(&h0, d, 8, hipMemcpyDeviceToHost));
  double cyc = best * 1e-3 * 2.4e9 / ((double)wavesPerSimd * 8 * ITER);
  printf("[out0 %016llx] ", (unsigned long long)h0);
  printf("%-22s w/SIMD=%d %7.3f ms  %7.2f cyc/butterfly (2.4GHz nominal)\n", name, wavesPerSimd, best, cyc);
  return 0;
}
int main() {
  u64 *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 8));
  for (int w : {2, 4, 8}) { timeit("cndmask  (variant 0)", k_butterfly<0>, d, w); timeit("exec-mask (variant 1)", k_butterfly<1>, d, w);
    timeit("exec asm  (variant 2)", k_butterfly<2>, d, w); timeit("exec asm nonop (v3)", k_butterfly<3>, d, w);
    timeit("mad carry a*w (v4)", k_butterfly<4>, d, w); timeit("mad carry both (v5)", k_butterfly<5>, d, w); }
  return 0;
}
