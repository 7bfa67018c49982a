// sve_ntt_amd/csrc/kernels.hip -- gfx950 kernels and their launch registry.
//
// tile_kernel<TN> runs TN's steps (tile_ntt.h) for one workgroup: the HIP
// counterpart of one OpenMP iteration of the reference (one block of columns in
// layer/sve/blocked-generic.hpp:139-154, or one row in kernel/recursive.hpp:69-74).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "tile_ntt.h"

namespace sventt_hip {

template <class TN, int IDX>
__device__ __forceinline__ void run_steps(const PassArgs &a, const typename TN::Tile &t, u32 tid,
                                          u64 *lds) {
  constexpr int SI = (TN::MODE == MODE_FWD) ? IDX : TN::NSTEPS - 1 - IDX;
  if constexpr (IDX > 0) __syncthreads();
  TN::template step<SI>(a, t, tid, lds);
  if constexpr (IDX + 1 < TN::NSTEPS) run_steps<TN, IDX + 1>(a, t, tid, lds);
}

template <class TN>
__global__ __launch_bounds__(TN::NT) void tile_kernel(const PassArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *lds = reinterpret_cast<u64 *>(smem);
  const typename TN::Tile t = TN::locate(a, blockIdx.x);
  if (!t.live) return;  // whole workgroup: no barrier is skipped by part of it
  run_steps<TN, 0>(a, t, threadIdx.x, lds);
}

template <class TN>
static hipError_t launch_tile(const PassArgs &a, u32 grid, hipStream_t stream) {
  constexpr size_t lds_bytes = (TN::NSTEPS > 1) ? (sizeof(u64) << TN::LOGT) : 0;
  static bool attr_set = false;
  if (lds_bytes > 48 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_kernel<TN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(tile_kernel<TN>, dim3(grid), dim3(TN::NT), lds_bytes, stream, a);
  return hipGetLastError();
}

// dst[i] = a[i]*b[i] mod p (both plain residues): montmul(a, b) = a*b/R, then *R^2/R.
__global__ __launch_bounds__(256) void pointwise_kernel(u64 *dst, const u64 *a, const u64 *b,
                                                        u64 count, Field f, u64 r2) {
  for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (u64)gridDim.x * 256ull)
    dst[i] = montmul(montmul(a[i], b[i] % f.N, f), r2, f);
}

hipError_t launch_pointwise(u64 *dst, const u64 *a, const u64 *b, u64 count, const Field &f,
                            u64 r2, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  u64 blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pointwise_kernel, dim3((u32)blocks), dim3(256), 0, stream, dst, a, b, count,
                     f, r2);
  return hipGetLastError();
}

// ---- registry -------------------------------------------------------------------
template <class TN> struct HipLauncher {
  static hipError_t launch(const PassArgs &a, u32 grid, hipStream_t stream) {
    return launch_tile<TN>(a, grid, stream);
  }
};

const KernelEntry *find_kernel(int kind, int logl, int dir, int flag, int f0) {
  return find_kernel_in_registry<KernelEntry, HipLauncher>(kind, logl, dir, flag, f0);
}

}  // namespace sventt_hip
