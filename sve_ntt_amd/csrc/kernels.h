// sve_ntt_amd/csrc/kernels.h -- launch registry shared by kernels.hip and plan.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "registry.h"

namespace sventt_hip {

using KernelEntry = KernelEntryT<hipError_t, hipStream_t>;

// arith: ARITH_MONT / ARITH_GOLD / ARITH_SHOUP (field64.h); the last two have E = 16 tiles only
// two_level: the TWOLVL variant of a column tile (tile_ntt.h)
const KernelEntry *find_kernel(int kind, int logl, int dir, int flag, int f0, int loge,
                               int arith = ARITH_MONT, int two_level = 0);
const KernelEntry *find_kernel_gold(int kind, int logl, int dir, int flag, int f0, int loge, int two_level);    // kernels_gold.hip
const KernelEntry *find_kernel_shoup(int kind, int logl, int dir, int flag, int f0, int loge, int two_level);   // kernels_shoup.hip

hipError_t launch_pointwise(u64 *dst, const u64 *a, const u64 *b, u64 count, const Field &f,
                            u64 r2, hipStream_t stream);

// dst[i] = montmul(a[i], b ? b[i] : c)
hipError_t launch_montmul(u64 *dst, const u64 *a, const u64 *b, u64 c, u64 count, const Field &f,
                          hipStream_t stream);

// dst[ld_dst*c + r] = src[ld_src*r + c], r < rows, c < cols (out of place)
hipError_t launch_transpose(u64 *dst, const u64 *src, u64 rows, u64 cols, u64 ld_dst, u64 ld_src,
                            hipStream_t stream);
// square, in place, leading dimension = dim
hipError_t launch_transpose_inplace(u64 *m, u64 dim, hipStream_t stream);

}  // namespace sventt_hip
