// sve_ntt_amd/csrc/kernels_shoup.hip -- the E = 16 tile kernels of the FixedPoint64 back end (Shoup multiplication with a stored companion per twiddle, N < 2^63).
// A translation unit of its own so that the three registries compile in parallel.
#include "tile_launch.h"

namespace sventt_hip {

const KernelEntry *find_kernel_shoup(int kind, int logl, int dir, int flag, int f0, int loge, int two_level) {
  return find_arith_kernel_in_registry<ARITH_SHOUP, KernelEntry, HipLauncher>(kind, logl, dir, flag, f0, loge, two_level);
}

}  // namespace sventt_hip
