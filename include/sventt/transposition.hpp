// include/sventt/transposition.hpp -- the reference's transposition classes on the GPU.
//
// The reference tunes a dozen SVE transposes (transposition/sve/*.hpp there), all with
// the same two static entry points:
//   transpose(dst, src, src_rows, src_cols, ld_dst, ld_src)   dst[ld_dst*c + r] = src[ld_src*r + c]
//   transpose(dst, dim)                                       square, in place
// (e.g. transposition/sve/in-register.hpp:115-119 and :215).  Here every one of those
// names forwards to the one LDS-tiled kernel behind sventt_transpose /
// sventt_transpose_inplace (include/sventt_hip.h); the block-shape template arguments
// keep their divisibility contract (std::invalid_argument, in-register.hpp:121-124) but
// no longer select code.  As layer template arguments they are plain tags: the GPU
// passes read columns where they lie and never transpose.
#ifndef SVENTT_GPU_TRANSPOSITION_HPP_INCLUDED
#define SVENTT_GPU_TRANSPOSITION_HPP_INCLUDED

#include <cstdint>
#include <stdexcept>

#include "sventt/status.hpp"
#include "sventt_hip.h"

namespace sventt {

namespace detail {

template <std::uint64_t block_rows, std::uint64_t block_cols> class transposition {

public:
  static void transpose(std::uint64_t *const dst, const std::uint64_t *const src,
                        const std::uint64_t src_rows, const std::uint64_t src_cols,
                        const std::uint64_t ld_dst, const std::uint64_t ld_src) {
    if (src_rows % block_rows != 0 || src_cols % block_cols != 0) {
      throw std::invalid_argument{"Matrix dimensions are not divisible by block dimensions"};
    }
    throw_on_error(sventt_transpose(dst, src, src_rows, src_cols, ld_dst, ld_src, nullptr));
  }

  static void transpose(std::uint64_t *const dst, const std::uint64_t dim) {
    if (dim % block_rows != 0 || dim % block_cols != 0) {
      throw std::invalid_argument{"Matrix dimensions are not divisible by block dimensions"};
    }
    throw_on_error(sventt_transpose_inplace(dst, dim, nullptr));
  }
};

} // namespace detail

// <block_rows, block_cols>  (in-register.hpp:16,111; gather-*.hpp)
template <std::uint64_t br, std::uint64_t bc>
class TransposeSVEInRegister : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeParallelSVEInRegister : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeSVEGatherRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeParallelSVEGatherRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeParallelSVEGatherColumnFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeSVEGatherVectorIndexRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeParallelSVEGatherVectorIndexRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeParallelSVEGatherVectorIndexColumnFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeSVEGatherCombinedColumnVectorIndexRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc>
class TransposeParallelSVEGatherCombinedColumnVectorIndexRowFirst : public detail::transposition<br, bc> {};
// <block_rows, block_columns, num_shuffle_stages>  (in-register-row-first.hpp:265,351)
template <std::uint64_t br, std::uint64_t bc, std::uint64_t>
class TransposeSVEInRegisterRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc, std::uint64_t>
class TransposeParallelSVEInRegisterRowFirst : public detail::transposition<br, bc> {};
// <block_rows, block_columns, ld_block, num_shuffle_stages>
// (in-register-explicit-blocking-row-first.hpp:18,127)
template <std::uint64_t br, std::uint64_t bc, std::uint64_t, std::uint64_t>
class TransposeSVEInRegisterExplicitBlockingRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc, std::uint64_t, std::uint64_t>
class TransposeParallelSVEInRegisterExplicitBlockingRowFirst : public detail::transposition<br, bc> {};
// <block_rows, block_columns, ld_block_row, ld_block_column, num_shuffle_stages>
// (in-register-full-blocking-row-first.hpp:16,160)
template <std::uint64_t br, std::uint64_t bc, std::uint64_t, std::uint64_t, std::uint64_t>
class TransposeSVEInRegisterFullBlockingRowFirst : public detail::transposition<br, bc> {};
template <std::uint64_t br, std::uint64_t bc, std::uint64_t, std::uint64_t, std::uint64_t>
class TransposeParallelSVEInRegisterFullBlockingRowFirst : public detail::transposition<br, bc> {};

} // namespace sventt

#endif /* SVENTT_GPU_TRANSPOSITION_HPP_INCLUDED */
