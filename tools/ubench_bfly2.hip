// tools/ubench_bfly2.hip -- radix-16 register step (4 stages x 8 Montgomery butterflies on 16
// values per thread): hipcc's code for csrc/field64.h against hand-written gfx950 assembly on
// fixed registers, where every conditional +N is one v_lshl_add_u64 under an EXEC mask taken
// straight from the borrow / carry SGPR pair (no v_cndmask, no mask round trip through a VGPR).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_bfly2.hip -o tools/ubench_bfly2
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint64_t u64;
typedef uint32_t u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Field { u64 N, Ninv, negN; };

// ---- variant 0: what the shipped field64.h compiles to ---------------------------------
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
__device__ __forceinline__ bool sub64(u64 a, u64 b, u64 &d) {
  u32 c0, c1; u32 lo = __builtin_subc((u32)a, (u32)b, 0u, &c0);
  u32 hi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), c0, &c1); d = ((u64)hi << 32) | lo; return c1 != 0; }
__device__ __forceinline__ bool add64(u64 a, u64 b, u64 &d) {
  u32 c0, c1; u32 lo = __builtin_addc((u32)a, (u32)b, 0u, &c0);
  u32 hi = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1); d = ((u64)hi << 32) | lo; return c1 != 0; }
__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 m0h = __umulhi(a0, b0); const u64 m1 = mad32(a0, b1, m0h); const u64 m2 = mad32(a1, b0, (u32)m1);
  return mad32(a1, b1, m1 >> 32) + (m2 >> 32); }
__device__ __forceinline__ u64 montmul(u64 a, u64 w, const Field &f) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)w, b1 = (u32)(w >> 32);
  const u64 m0 = mad32(a0, b0, 0); const u64 m1 = mad32(a0, b1, m0 >> 32); const u64 m2 = mad32(a1, b0, (u32)m1);
  const u64 thi = mad32(a1, b1, m1 >> 32) + (m2 >> 32);
  const u32 t0 = (u32)m0, t1 = (u32)m2; const u32 ni0 = (u32)f.Ninv, ni1 = (u32)(f.Ninv >> 32);
  const u64 r0 = mad32(t0, ni0, 0); const u32 q0 = (u32)r0; const u32 q1 = (u32)(r0 >> 32) + t0 * ni1 + t1 * ni0;
  const u64 g = mulhi64(((u64)q1 << 32) | q0, f.N);
  u64 c; const bool b = sub64(thi, g, c); return c + (b ? f.N : 0); }
__device__ __forceinline__ u64 addmod(u64 a, u64 b, const Field &f) {
  u64 e; const bool k = add64(a, b + f.negN, e); return e + (k ? 0 : f.N); }
__device__ __forceinline__ u64 submod(u64 a, u64 b, const Field &f) {
  u64 d; const bool br = sub64(a, b, d); return d + (br ? f.N : 0); }

constexpr int ITER = 128;

__device__ __forceinline__ void init(u64 *x, u64 *w, const Field &f, u64 seed) {
  for (int i = 0; i < 16; ++i) x[i] = (seed * (threadIdx.x + 3 + i) * 0x9e3779b97f4a7c15ull) % f.N;
  for (int i = 0; i < 8; ++i) w[i] = (x[i] * 7 + i + blockIdx.x) % f.N;
}

__global__ __launch_bounds__(512) void k_cxx(u64 *out, Field f, u64 seed) {
  u64 x[16], w[8];
  init(x, w, f, seed);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 3; r >= 0; --r) {
#pragma unroll
      for (int v = 0, j = 0; v < 16; ++v) {
        if (v & (1 << r)) continue;
        u64 a = x[v], b = x[v + (1 << r)];
        x[v] = addmod(a, b, f);
        x[v + (1 << r)] = montmul(submod(a, b, f), w[j++], f);
      }
    }
  }
  u64 *o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  for (int i = 0; i < 16; ++i) o[i] = x[i];
}

// ---- variant 1: fixed registers, EXEC-masked corrections ---------------------------------
// data x_i = v[2i:2i+1]; temporaries v[32:47]; v41 = 0 (high half of the zero-extension pair).
#define S_(x) #x
#define S(x) S_(x)
#define MONT_CORE(DL, DH, W0, W1)                                                             \
  "v_mad_u64_u32 v[32:33], vcc, v" DL ", " W0 ", 0\n\t"                                      \
  "v_mov_b32 v40, v33\n\t"                                                                    \
  "v_mad_u64_u32 v[34:35], vcc, v" DL ", " W1 ", v[40:41]\n\t"                              \
  "v_mov_b32 v40, v35\n\t"                                                                    \
  "v_mad_u64_u32 v[38:39], vcc, v" DH ", " W1 ", v[40:41]\n\t"                              \
  "v_mov_b32 v40, v34\n\t"                                                                    \
  "v_mad_u64_u32 v[36:37], vcc, v" DH ", " W0 ", v[40:41]\n\t"                              \
  "v_mov_b32 v40, v37\n\t"                                                                    \
  "v_mad_u64_u32 v[34:35], vcc, v32, %[ni0], 0\n\t"                                           \
  "v_mul_lo_u32 v33, v32, %[ni1]\n\t"                                                         \
  "v_mul_lo_u32 v36, v36, %[ni0]\n\t"                                                         \
  "v_lshl_add_u64 v[38:39], v[38:39], 0, v[40:41]\n\t"                                        \
  "v_add3_u32 v35, v35, v33, v36\n\t"                                                         \
  "v_mul_hi_u32 v40, v34, %[n0]\n\t"                                                          \
  "v_mad_u64_u32 v[32:33], vcc, v34, %[n1], v[40:41]\n\t"                                     \
  "v_mov_b32 v40, v33\n\t"                                                                    \
  "v_mad_u64_u32 v[36:37], vcc, v35, %[n1], v[40:41]\n\t"                                     \
  "v_mov_b32 v40, v32\n\t"                                                                    \
  "v_mad_u64_u32 v[32:33], vcc, v35, %[n0], v[40:41]\n\t"                                     \
  "v_mov_b32 v40, v33\n\t"                                                                    \
  "v_lshl_add_u64 v[36:37], v[36:37], 0, v[40:41]\n\t"

#define BF_FWD(XL, XH, YL, YH, X, Y, W)                                                        \
  asm volatile(                                                                                \
      "v_mov_b32 v41, 0\n\t"                                                                   \
      "v_sub_co_u32 v44, %[sb], v" S(XL) ", v" S(YL) "\n\t"                                  \
      "v_lshl_add_u64 v[46:47], v[" S(YL) ":" S(YH) "], 0, %[negN]\n\t"                      \
      "v_subb_co_u32 v45, %[sb], v" S(XH) ", v" S(YH) ", %[sb]\n\t"                          \
      "v_add_co_u32 v" S(XL) ", %[sc], v" S(XL) ", v46\n\t"                                  \
      "s_mov_b64 exec, %[sb]\n\t"                                                              \
      "v_lshl_add_u64 v[44:45], v[44:45], 0, %[N]\n\t"                                         \
      "s_mov_b64 exec, %[save]\n\t"                                                            \
      "v_addc_co_u32 v" S(XH) ", %[sc], v" S(XH) ", v47, %[sc]\n\t"                          \
      "s_nop 0\n\t"                                                                            \
      "s_andn2_b64 exec, %[save], %[sc]\n\t"                                                   \
      "v_lshl_add_u64 v[" S(XL) ":" S(XH) "], v[" S(XL) ":" S(XH) "], 0, %[N]\n\t"          \
      "s_mov_b64 exec, %[save]\n\t"                                                            \
      MONT_CORE("44", "45", "%[w0]", "%[w1]")                                                  \
      "v_sub_co_u32 v" S(YL) ", %[sb], v38, v36\n\t"                                          \
      "s_nop 0\n\t"                                                                            \
      "v_subb_co_u32 v" S(YH) ", %[sb], v39, v37, %[sb]\n\t"                                  \
      "s_nop 0\n\t"                                                                            \
      "s_mov_b64 exec, %[sb]\n\t"                                                              \
      "v_lshl_add_u64 v[" S(YL) ":" S(YH) "], v[" S(YL) ":" S(YH) "], 0, %[N]\n\t"          \
      "s_mov_b64 exec, %[save]\n\t"                                                            \
      : "+{v[" S(XL) ":" S(XH) "]}"(X), "+{v[" S(YL) ":" S(YH) "]}"(Y), [sb] "=&s"(sb),       \
        [sc] "=&s"(sc)                                                                         \
      : [w0] "v"((u32)(W)), [w1] "v"((u32)((W) >> 32)), [N] "s"(f.N), [negN] "s"(f.negN),     \
        [n0] "s"((u32)f.N), [n1] "s"((u32)(f.N >> 32)), [ni0] "s"((u32)f.Ninv),               \
        [ni1] "s"((u32)(f.Ninv >> 32)), [save] "s"(save)                                       \
      : "vcc", "scc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41",   \
        "v44", "v45", "v46", "v47")

__global__ __launch_bounds__(512) void k_asm(u64 *out, Field f, u64 seed) {
  u64 x[16], w[8];
  init(x, w, f, seed);
  u64 x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4], x5 = x[5], x6 = x[6], x7 = x[7];
  u64 x8 = x[8], x9 = x[9], x10 = x[10], x11 = x[11], x12 = x[12], x13 = x[13], x14 = x[14], x15 = x[15];
  u64 sb, sc;
  const u64 save = __builtin_amdgcn_read_exec();
  for (int it = 0; it < ITER; ++it) {
    // r = 3
    BF_FWD(0, 1, 16, 17, x0, x8, w[0]);   BF_FWD(2, 3, 18, 19, x1, x9, w[1]);
    BF_FWD(4, 5, 20, 21, x2, x10, w[2]);  BF_FWD(6, 7, 22, 23, x3, x11, w[3]);
    BF_FWD(8, 9, 24, 25, x4, x12, w[4]);  BF_FWD(10, 11, 26, 27, x5, x13, w[5]);
    BF_FWD(12, 13, 28, 29, x6, x14, w[6]); BF_FWD(14, 15, 30, 31, x7, x15, w[7]);
    // r = 2
    BF_FWD(0, 1, 8, 9, x0, x4, w[0]);     BF_FWD(2, 3, 10, 11, x1, x5, w[1]);
    BF_FWD(4, 5, 12, 13, x2, x6, w[2]);   BF_FWD(6, 7, 14, 15, x3, x7, w[3]);
    BF_FWD(16, 17, 24, 25, x8, x12, w[4]); BF_FWD(18, 19, 26, 27, x9, x13, w[5]);
    BF_FWD(20, 21, 28, 29, x10, x14, w[6]); BF_FWD(22, 23, 30, 31, x11, x15, w[7]);
    // r = 1
    BF_FWD(0, 1, 4, 5, x0, x2, w[0]);     BF_FWD(2, 3, 6, 7, x1, x3, w[1]);
    BF_FWD(8, 9, 12, 13, x4, x6, w[2]);   BF_FWD(10, 11, 14, 15, x5, x7, w[3]);
    BF_FWD(16, 17, 20, 21, x8, x10, w[4]); BF_FWD(18, 19, 22, 23, x9, x11, w[5]);
    BF_FWD(24, 25, 28, 29, x12, x14, w[6]); BF_FWD(26, 27, 30, 31, x13, x15, w[7]);
    // r = 0
    BF_FWD(0, 1, 2, 3, x0, x1, w[0]);     BF_FWD(4, 5, 6, 7, x2, x3, w[1]);
    BF_FWD(8, 9, 10, 11, x4, x5, w[2]);   BF_FWD(12, 13, 14, 15, x6, x7, w[3]);
    BF_FWD(16, 17, 18, 19, x8, x9, w[4]); BF_FWD(20, 21, 22, 23, x10, x11, w[5]);
    BF_FWD(24, 25, 26, 27, x12, x13, w[6]); BF_FWD(28, 29, 30, 31, x14, x15, w[7]);
  }
  u64 *o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3; o[4] = x4; o[5] = x5; o[6] = x6; o[7] = x7;
  o[8] = x8; o[9] = x9; o[10] = x10; o[11] = x11; o[12] = x12; o[13] = x13; o[14] = x14; o[15] = x15;
}

template <class K> static int timeit(const char *name, K kernel, u64 *d, int blocksPerCU, size_t lds,
                                     std::vector<u64> *keep) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const Field f{0xfffffc6e80000001ull, 0x4000039180000001ull, 0ull - 0xfffffc6e80000001ull};
  const int blocks = 256 * blocksPerCU;
  if (lds > 48 * 1024)
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), lds, 0, d, f, 0x1234567ull); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), lds, 0, d, f, 0x1234567ull);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  if (keep) { keep->resize((size_t)blocks * 512 * 16); CHECK(hipMemcpy(keep->data(), d, keep->size() * 8, hipMemcpyDeviceToHost)); }
  // blocksPerCU blocks of 8 waves per CU = 2*blocksPerCU waves per SIMD, each ITER*32 butterflies
  const double cyc = best * 1e-3 * 2.4e9 / ((double)blocksPerCU * 2 * 32 * ITER);
  printf("%-28s blocks/CU=%d lds=%3zuK %7.3f ms  %7.2f cyc/butterfly/SIMD (2.4GHz nominal)\n", name, blocksPerCU,
         lds >> 10, best, cyc);
  return 0;
}

int main() {
  u64 *d; CHECK(hipMalloc(&d, (size_t)256 * 4 * 512 * 16 * 8));
  std::vector<u64> ref, got;
  for (int pass = 0; pass < 2; ++pass) {
    const size_t lds = pass == 0 ? 64 * 1024 : 0;   // 64 KiB per block: two blocks (4 waves/SIMD) per CU as in the tile kernels
    const int bpc = pass == 0 ? 2 : 4;
    if (timeit("hipcc C++ (cndmask)", k_cxx, d, bpc, lds, &ref)) return 1;
    if (timeit("asm fixed regs, EXEC fix", k_asm, d, bpc, lds, &got)) return 1;
    size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
    printf("   asm vs C++: %zu mismatches of %zu\n", bad, ref.size());
  }
  return 0;
}
