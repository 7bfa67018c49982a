// sve_ntt_amd/csrc/kernels_gold.hip -- the E = 16 tile kernels of the Goldilocks back end (N = 2^64 - 2^32 + 1: folding reduction, plain twiddles).
// A translation unit of its own so that the three registries compile in parallel.
#include "tile_launch.h"

namespace sventt_hip {

const KernelEntry *find_kernel_gold(int kind, int logl, int dir, int flag, int f0, int loge, int two_level) {
  return find_arith_kernel_in_registry<ARITH_GOLD, KernelEntry, HipLauncher>(kind, logl, dir, flag, f0, loge, two_level);
}

}  // namespace sventt_hip
