// sve_ntt_amd/csrc/kernels.hip -- gfx950 kernels and their launch registry.
//
// tile_kernel<TN> runs TN's steps (tile_ntt.h) for one workgroup: the HIP
// counterpart of one OpenMP iteration of the reference (one block of columns in
// layer/sve/blocked-generic.hpp:139-154, or one row in kernel/recursive.hpp:69-74).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "tile_launch.h"

#if defined(SVENTT_TRACE)
// analysis builds only (tools/trace_tiles.py): the time stamps of the last tile-kernel launch
extern "C" int sventt_debug_trace_read(unsigned long long *host, size_t words) {
  const size_t all = (size_t)sventt_hip::TRACE_MAX_WAVES * sventt_hip::TRACE_SLOTS;
  if (words > all) words = all;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(sventt_hip::g_trace), words * sizeof(unsigned long long), 0,
                             hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

namespace sventt_hip {

// dst[i] = a[i]*b[i] mod p (both plain residues < p): montmul(a, b) = a*b/R, then *R^2/R.
// HBM-bound (24 B per element): two elements per thread through 16-byte accesses
// when the three pointers allow it.
__global__ __launch_bounds__(256) void pointwise_kernel(u64 *dst, const u64 *a, const u64 *b,
                                                        u64 count, Field f, u64 r2, int vec) {
  const u64 stride = (u64)gridDim.x * 256ull;
  if (vec) {
    const u64 pairs = count >> 1;
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < pairs; i += stride) {
      const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(a)[i];
      const ulonglong2 y = reinterpret_cast<const ulonglong2 *>(b)[i];
      ulonglong2 z;
      z.x = montmul(montmul(x.x, y.x, f), r2, f);
      z.y = montmul(montmul(x.y, y.y, f), r2, f);
      reinterpret_cast<ulonglong2 *>(dst)[i] = z;
    }
    if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0)
      dst[count - 1] = montmul(montmul(a[count - 1], b[count - 1], f), r2, f);
  } else {
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < count; i += stride)
      dst[i] = montmul(montmul(a[i], b[i], f), r2, f);
  }
}

hipError_t launch_pointwise(u64 *dst, const u64 *a, const u64 *b, u64 count, const Field &f,
                            u64 r2, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const int vec = (((uintptr_t)dst | (uintptr_t)a | (uintptr_t)b) & 15) == 0;
  u64 blocks = ((vec ? (count + 1) / 2 : count) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(pointwise_kernel, dim3((u32)blocks), dim3(256), 0, stream, dst, a, b, count,
                     f, r2, vec);
  return hipGetLastError();
}

// dst[i] = montmul(a[i], b ? b[i] : c): domain conversions (c = R^2 -> to Montgomery form,
// c = 1 -> back; modmul/sve/p-adic-64.hpp:64-74 of the reference) and the length-1 case
// of the fused forward-multiply.
__global__ __launch_bounds__(256) void montmul_kernel(u64 *dst, const u64 *a, const u64 *b, u64 c,
                                                      u64 count, Field f) {
  for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (u64)gridDim.x * 256ull)
    dst[i] = montmul(a[i], b ? b[i] : c, f);
}

hipError_t launch_montmul(u64 *dst, const u64 *a, const u64 *b, u64 c, u64 count, const Field &f,
                          hipStream_t stream) {
  if (count == 0) return hipSuccess;
  u64 blocks = (count + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(montmul_kernel, dim3((u32)blocks), dim3(256), 0, stream, dst, a, b, c, count, f);
  return hipGetLastError();
}

// ---- stand-alone transposition ---------------------------------------------------
// dst[ld_dst*c + r] = src[ld_src*r + c]: the GPU counterpart of the reference's
// transposition kernels (transposition/sve/in-register.hpp:111-206, in place :215-375).
// The transforms themselves never transpose (a COL tile reads its columns where they
// lie); this exists for callers that used the reference's transposes directly and as
// the bandwidth yardstick of tests/bench-transpose.cpp.
//
// One workgroup moves a 64 x 64 tile through LDS: rows are read and written as 512-byte
// runs, and the tile is stored with column index c ^ r so that both the row-wise fill
// and the column-wise drain touch all 64 banks (no padding words).
constexpr int TR_TILE = 64;

__device__ __forceinline__ void tile_in(u64 *tile, const u64 *src, u64 r0, u64 c0, u64 rows,
                                        u64 cols, u64 ld, u32 tx, u32 ty) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const u32 r = ty + 4u * k;
    if (r0 + r < rows && c0 + tx < cols) tile[r * TR_TILE + (tx ^ r)] = src[(r0 + r) * ld + c0 + tx];
  }
}

// writes the transposed tile: element (r, c) of the tile goes to dst row c0 + c, column r0 + r
__device__ __forceinline__ void tile_out_transposed(const u64 *tile, u64 *dst, u64 r0, u64 c0,
                                                    u64 rows, u64 cols, u64 ld, u32 tx, u32 ty) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const u32 c = ty + 4u * k;
    if (c0 + c < cols && r0 + tx < rows) dst[(c0 + c) * ld + r0 + tx] = tile[tx * TR_TILE + (c ^ tx)];
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(u64 *dst, const u64 *src, u64 rows, u64 cols,
                                                        u64 ld_dst, u64 ld_src, u32 tiles_c) {
  __shared__ u64 tile[TR_TILE * TR_TILE];
  const u32 tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const u64 r0 = (u64)(blockIdx.x / tiles_c) * TR_TILE, c0 = (u64)(blockIdx.x % tiles_c) * TR_TILE;
  tile_in(tile, src, r0, c0, rows, cols, ld_src, tx, ty);
  __syncthreads();
  tile_out_transposed(tile, dst, r0, c0, rows, cols, ld_dst, tx, ty);
}

// In place, square, leading dimension = dim: tile (i, j), i <= j, is exchanged with
// (j, i).  32 x 32 tiles (two of them = 16 KiB of LDS, so ten workgroups fit a CU) and a
// triangular grid: block b -> j = the largest integer with j(j+1)/2 <= b, i = b - j(j+1)/2.
constexpr int TI_TILE = 32;

__device__ __forceinline__ void tile32_in(u64 *tile, const u64 *src, u64 r0, u64 c0, u64 dim, u32 tx,
                                          u32 ty) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32 r = ty + 8u * k;
    if (r0 + r < dim && c0 + tx < dim) tile[r * TI_TILE + (tx ^ r)] = src[(r0 + r) * dim + c0 + tx];
  }
}

__device__ __forceinline__ void tile32_out_transposed(const u64 *tile, u64 *dst, u64 r0, u64 c0,
                                                      u64 dim, u32 tx, u32 ty) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32 c = ty + 8u * k;
    if (c0 + c < dim && r0 + tx < dim) dst[(c0 + c) * dim + r0 + tx] = tile[tx * TI_TILE + (c ^ tx)];
  }
}

__global__ __launch_bounds__(256) void transpose_inplace_kernel(u64 *m, u64 dim) {
  __shared__ u64 upper[TI_TILE * TI_TILE];
  __shared__ u64 lower[TI_TILE * TI_TILE];
  const u64 b = blockIdx.x;
  u64 bj = (u64)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while (bj * (bj + 1) / 2 > b) --bj;
  while ((bj + 1) * (bj + 2) / 2 <= b) ++bj;
  const u64 bi = b - bj * (bj + 1) / 2;  // bi <= bj
  const u32 tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const u64 r0 = bi * TI_TILE, c0 = bj * TI_TILE;
  tile32_in(upper, m, r0, c0, dim, tx, ty);
  if (bi != bj) tile32_in(lower, m, c0, r0, dim, tx, ty);
  __syncthreads();
  tile32_out_transposed(upper, m, r0, c0, dim, tx, ty);
  if (bi != bj) tile32_out_transposed(lower, m, c0, r0, dim, tx, ty);
}

hipError_t launch_transpose(u64 *dst, const u64 *src, u64 rows, u64 cols, u64 ld_dst, u64 ld_src,
                            hipStream_t stream) {
  if (rows == 0 || cols == 0) return hipSuccess;
  const u64 tr = (rows + TR_TILE - 1) / TR_TILE, tc = (cols + TR_TILE - 1) / TR_TILE;
  if (tr * tc > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_kernel, dim3((u32)(tr * tc)), dim3(256), 0, stream, dst, src, rows,
                     cols, ld_dst, ld_src, (u32)tc);
  return hipGetLastError();
}

hipError_t launch_transpose_inplace(u64 *m, u64 dim, hipStream_t stream) {
  if (dim == 0) return hipSuccess;
  const u64 t = (dim + TI_TILE - 1) / TI_TILE;
  const u64 blocks = t * (t + 1) / 2;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_inplace_kernel, dim3((u32)blocks), dim3(256), 0, stream, m, dim);
  return hipGetLastError();
}

// ---- registry -------------------------------------------------------------------
const KernelEntry *find_kernel(int kind, int logl, int dir, int flag, int f0, int loge, int arith,
                               int two_level) {
  switch (arith) {
    case ARITH_MONT:
      return find_kernel_in_registry<KernelEntry, HipLauncher>(kind, logl, dir, flag, f0, loge, two_level);
    case ARITH_GOLD:
      return find_kernel_gold(kind, logl, dir, flag, f0, loge, two_level);
    case ARITH_SHOUP:
      return find_kernel_shoup(kind, logl, dir, flag, f0, loge, two_level);
  }
  return nullptr;
}

}  // namespace sventt_hip
