// sve_ntt_amd/csrc/tile_launch.h -- the tile kernel and its launcher, shared by the translation
// units that instantiate the registry (kernels.hip: Montgomery back end, all tile families;
// kernels_gold.hip, kernels_shoup.hip: the E = 16 tiles of the other back ends).
//
// tile_kernel<TN> runs TN's steps (tile_ntt.h) for one workgroup: the HIP counterpart of one
// OpenMP iteration of the reference (one block of columns in layer/sve/blocked-generic.hpp:
// 139-154, or one row in kernel/recursive.hpp:69-74).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include "kernels.h"
#include "tile_ntt.h"

namespace sventt_hip {

template <class TN, int IDX>
__device__ __forceinline__ void run_steps(const PassArgs &a, const typename TN::Tile &t, u32 tid,
                                          u64 *lds) {
  constexpr int SI = (TN::MODE == MODE_FWD) ? IDX : TN::NSTEPS - 1 - IDX;
  TN::template step<SI, (IDX > 0 ? TN::template sync_before<SI>() : TN::SYNC_NONE)>(a, t, tid, lds);
  if constexpr (IDX + 1 < TN::NSTEPS) run_steps<TN, IDX + 1>(a, t, tid, lds);
}

// E = 16 tiles are sized for four waves per SIMD (two 512-thread workgroups per CU with their
// 64 KiB tiles, or four 256-thread ones): keep the register allocator inside 128 VGPRs.
#ifndef SVENTT_KERNEL_ALIGN
#define SVENTT_KERNEL_ALIGN 256  // (the toolchain's default for kernel entry points)
#endif
template <class TN>
__global__ __launch_bounds__(TN::NT, (TN::LOGE == 4 && TN::NT >= 64) ? 4 : 1)
__attribute__((aligned(SVENTT_KERNEL_ALIGN))) void tile_kernel(const PassArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *lds = reinterpret_cast<u64 *>(smem);
  const typename TN::Tile t = TN::locate(a, blockIdx.x);
  if (!t.live) return;  // whole workgroup: no barrier is skipped by part of it
#if defined(SVENTT_TRACE)
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, the same on every XCD
#endif
  run_steps<TN, 0>(a, t, threadIdx.x, lds);
#if defined(SVENTT_TRACE)
  if constexpr (TN::LOGE == 4 && TN::NSTEPS > 1) {
    // slot 30: end of the instruction stream (stores issued, not acknowledged); slot 31: where it ran
    const unsigned long long tm = __builtin_amdgcn_s_memtime();
    const u32 hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    u64 *tr = lds + TN::TRACE_LDS_WORD + wave * TRACE_SLOTS;
    if (lane == 0) {
      tr[28] = rt0;
      tr[29] = __builtin_amdgcn_s_memrealtime();
      tr[30] = tm;
      tr[31] = ((u64)blockIdx.x << 32) | hw;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const size_t w = (size_t)blockIdx.x * (TN::NT / 64) + wave;
    if (lane < (u32)TRACE_SLOTS && w < (size_t)TRACE_MAX_WAVES) g_trace[w * TRACE_SLOTS + lane] = tr[lane];
  }
#endif
}

template <class TN>
inline hipError_t launch_tile(const PassArgs &a, u32 grid, hipStream_t stream) {
  constexpr size_t lds_bytes = TN::LDS_BYTES;  // tile image + the lower steps' twiddles
  if constexpr (lds_bytes > 48 * 1024) {
    // the opt-in to more than 48 KiB of dynamic LDS is a per-device property of the function:
    // one bit per device ordinal, set once the attribute call succeeded there
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_kernel<TN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(tile_kernel<TN>, dim3(grid), dim3(TN::NT), lds_bytes, stream, a);
  return hipGetLastError();
}


template <class TN> struct HipLauncher {
  static hipError_t launch(const PassArgs &a, u32 grid, hipStream_t stream) {
    return launch_tile<TN>(a, grid, stream);
  }
};

}  // namespace sventt_hip
