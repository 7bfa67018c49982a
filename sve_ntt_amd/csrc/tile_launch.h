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
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "tile_ntt.h"

namespace sventt_hip {

template <class TN, int IDX>
__device__ __forceinline__ void run_steps(const PassArgs &a, const typename TN::Tile &t, u32 tid,
                                          u64 *lds) {
  constexpr int SI = (TN::MODE == MODE_FWD) ? IDX : TN::NSTEPS - 1 - IDX;
  TN::template step<SI, (IDX > 0)>(a, t, tid, lds);
  if constexpr (IDX + 1 < TN::NSTEPS) run_steps<TN, IDX + 1>(a, t, tid, lds);
}

// E = 16 tiles are sized for four waves per SIMD (two 512-thread workgroups per CU with their
// 64 KiB tiles, or four 256-thread ones): keep the register allocator inside 128 VGPRs.
template <class TN>
__global__ __launch_bounds__(TN::NT, (TN::LOGE == 4 && TN::NT >= 64) ? 4 : 1) void tile_kernel(const PassArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *lds = reinterpret_cast<u64 *>(smem);
  const typename TN::Tile t = TN::locate(a, blockIdx.x);
  if (!t.live) return;  // whole workgroup: no barrier is skipped by part of it
  run_steps<TN, 0>(a, t, threadIdx.x, lds);
}

// Persistent form for TN::DMA_OK tiles: gridDim.x workgroups share a.grid tiles round robin;
// each fetches its next tile into LDS while it finishes the current one (tile_ntt.h: dma_tile).
template <class TN, int IDX>
__device__ __forceinline__ void run_steps_dma(const PassArgs &a, const typename TN::Tile &t, u32 tid, u64 *lds,
                                              const typename TN::Tile &next, bool has_next) {
  constexpr int SI = (TN::MODE == MODE_FWD) ? IDX : TN::NSTEPS - 1 - IDX;
  TN::template step<SI, (IDX > 0), true>(a, t, tid, lds, next, has_next);
  if constexpr (IDX + 1 < TN::NSTEPS) run_steps_dma<TN, IDX + 1>(a, t, tid, lds, next, has_next);
}

template <class TN>
__global__ __launch_bounds__(TN::NT, (TN::LOGE == 4 && TN::NT >= 64) ? 4 : 1) void tile_kernel_dma(const PassArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only needs the symbol)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *lds = reinterpret_cast<u64 *>(smem);
  const u32 tid = threadIdx.x;
  u32 vb = blockIdx.x;
  typename TN::Tile t = TN::locate(a, vb);
#if defined(SVENTT_PERSIST_NODMA)  // experiment: the persistent loop alone, tiles loaded straight from HBM
  for (;; vb += gridDim.x) {
    if (vb >= a.grid) break;
    t = TN::locate(a, vb);
    u32 tid_j = tid;
    asm volatile("" : "+v"(tid_j));
    run_steps<TN, 0>(a, t, tid_j, lds);
    __syncthreads();
  }
  return;
#endif
  TN::dma_tile(a, t, tid, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first image (later ones: TileNTT::dma_wait)
  for (;;) {
    const u32 vnext = vb + gridDim.x;
    const bool more = vnext < a.grid;
    const typename TN::Tile tn = TN::locate(a, more ? vnext : vb);
    // Everything a step derives from the thread index (LDS and table addresses, a few dozen
    // registers) is invariant across tiles; hoisted out of this loop it would be spilled to
    // scratch.  An opaque copy of the index per iteration keeps those values short-lived.
    u32 tid_i = tid;
    asm volatile("" : "+v"(tid_i));
    run_steps_dma<TN, 0>(a, t, tid_i, lds, tn, more);
    if (!more) break;
    t = tn;
    vb = vnext;
  }
#endif
}

// how many workgroups of a kernel stay resident on the device (cached per kernel and device)
template <class K> inline u32 resident_workgroups(K kernel, int threads, size_t lds_bytes) {
  static std::atomic<u32> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  u32 v = cache[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1)
      return 0;
    v = (u32)per_cu * (u32)cus;
    cache[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

inline bool dma_pipeline_enabled() {
  static const bool on = [] {
    const char *e = std::getenv("SVENTT_DMA");  // SVENTT_DMA=0: one workgroup per tile, loads straight from HBM
    return e ? std::atoi(e) != 0 : true;
  }();
  return on;
}

template <class TN>
inline hipError_t launch_tile(const PassArgs &a, u32 grid, hipStream_t stream) {
  constexpr size_t lds_bytes = (TN::NSTEPS > 1) ? (sizeof(u64) << TN::LOGT) : 0;
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
  if constexpr (TN::DMA_OK) {
    // the fetch moves 16-byte pieces: the source has to be 16-byte aligned in every row
    const bool aligned =
        (reinterpret_cast<uintptr_t>(a.src) & 15u) == 0 &&
        (!TN::COL || (((a.src_istride * 8) | (a.src_ostride * 8) | ((u64)a.src_col_bias * 8)) & 15u) == 0);
    if (aligned && dma_pipeline_enabled() && a.grid == grid) {
      if constexpr (lds_bytes > 48 * 1024) {
        static std::atomic<uint64_t> done_dma{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(done_dma.load(std::memory_order_acquire) & bit)) {
          e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_kernel_dma<TN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
          if (e != hipSuccess) return e;
          done_dma.fetch_or(bit, std::memory_order_release);
        }
      }
      u32 resident = resident_workgroups(tile_kernel_dma<TN>, TN::NT, lds_bytes);
      // (a multiple of 8 keeps workgroup b on XCD b mod 8 for every tile it takes: TileNTT::locate)
      if (resident >= 8) resident &= ~7u;
      if (std::getenv("SVENTT_DEBUG_LAUNCH"))
        fprintf(stderr, "dma launch: LOGT=%d F0=%d MODE=%d tiles=%u resident=%u\n", TN::LOGT, TN::F0, TN::MODE, grid, resident);
      if (resident > 0 && grid > resident) {
        hipLaunchKernelGGL(tile_kernel_dma<TN>, dim3(resident), dim3(TN::NT), lds_bytes, stream, a);
        return hipGetLastError();
      }
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
