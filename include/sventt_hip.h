/*
 * include/sventt_hip.h -- C ABI of the MI355X (gfx950) NTT engine.
 *
 * The reference (Terminus-IMRC/sve-ntt) is a header-only C++20 template
 * library with no FFI of its own; its boundary for the transform path is
 *
 *     sventt::NTT<kernel_type>                       include/sventt/wrapper.hpp:13-83
 *       NTT(enable_forward, enable_inverse, huge)    wrapper.hpp:34-46
 *       compute_forward(dst, src) / (dst)            wrapper.hpp:50-65
 *       compute_inverse(dst, src) / (dst)            wrapper.hpp:67-82
 *
 * with kernel_type = IterativeNTT<...> / RecursiveNTT<...> describing the
 * decomposition at compile time (kernel/iterative.hpp:17-18,
 * kernel/recursive.hpp:15-17).  This header is the C-ABI those entry points
 * bind to when the transform runs on the GPU: the C++ facade under
 * include/sventt/ lowers kernel_type to (modulus, generator, n, n0) and calls
 * the functions below; any other host language binds the same symbols
 * (INTEGRATION.md shows the ctypes and C++ bindings).
 *
 * Semantics are those of the reference's scalar oracle
 * (tests/ntt-reference.hpp:43-83):
 *   forward: natural-order input, bit-reversed output, no scaling;
 *   inverse: bit-reversed input, natural-order output, scaled by n^{-1};
 *   every output is the canonical residue in [0, p)  (the reference's SVE
 *   kernels only guarantee congruence, tests/bench-ntt.cpp:61).
 * Inputs must be < p (tests/bench-ntt.cpp:31-33 guarantees that there).
 *
 * All functions return SVENTT_OK (0) or a negative status; the message of the
 * last failure on the calling thread is sventt_last_error().
 * The library needs a HIP device: there is no CPU fallback.
 */
#ifndef SVENTT_HIP_H_INCLUDED
#define SVENTT_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sventt_plan sventt_plan;

enum {
  SVENTT_OK = 0,
  SVENTT_ERR_INVALID_ARGUMENT = -1, /* std::invalid_argument in the reference
                                       (modulus.hpp:118-120, layer/sve/blocked-generic.hpp:111-116) */
  SVENTT_ERR_ALLOC = -2,            /* std::bad_alloc (vector.hpp:130-132) */
  SVENTT_ERR_HIP = -3,              /* a HIP runtime call failed */
  SVENTT_ERR_LOGIC = -4,            /* std::logic_error (wrapper.hpp:55) */
  SVENTT_ERR_NO_DEVICE = -5,        /* no gfx950 device visible */
  SVENTT_ERR_COMM = -6              /* the exchange of a sharded transform failed (RCCL / transport) */
};

/* sventt_plan_create flags */
enum {
  SVENTT_FORWARD = 1u, /* NTT(enable_forward=true, ...)   wrapper.hpp:34 */
  SVENTT_INVERSE = 2u, /* NTT(..., enable_inverse=true)   wrapper.hpp:34 */
  SVENTT_BOTH = 3u,
  SVENTT_DEVICE_POINTERS = 4u, /* every dst/src/operand ever passed with this plan is memory of the
                                  plan's device: the calls skip hipPointerGetAttributes (two driver
                                  queries per transform, a visible share of a 10 us small transform) */
  /* Arithmetic back end of the kernels (results are the same canonical residues with every one):
   * default = Montgomery multiplication (PAdic64SVE, modmul/sve/p-adic-64.hpp of the reference),
   * except for p = 2^64 - 2^32 + 1, which gets kernels with that prime's folding reduction. */
  SVENTT_GENERIC_ARITHMETIC = 8u, /* Montgomery kernels for every modulus */
  SVENTT_FIXED_POINT = 16u        /* FixedPoint64SVE / FixedPoint64Scalar (modmul/sve/fixed-point-64.hpp:
                                     33-68): Shoup multiplication c = a*w - hi64(a*w')*N with w' stored
                                     beside every twiddle; needs p < 2^63 */
};

/*
 * Threads and devices.  A plan belongs to the HIP device that was current when it was
 * created (sventt_plan_device); calls with it must come from threads whose current device is
 * that one and take memory of that device, otherwise they return SVENTT_ERR_INVALID_ARGUMENT.
 * A plan is immutable once created: any number of threads may use one plan concurrently with
 * device pointers (the reference's compute_* are const and re-entrant the same way,
 * wrapper.hpp:50-82).  Host-pointer calls on the SAME plan take turns on its one staging
 * buffer (they block each other, results stay correct); use one plan per thread to overlap them.
 * p must be prime (checked: deterministic Miller-Rabin).
 */

/*
 * Replaces the construction of sventt::NTT<kernel_type> (wrapper.hpp:34-46:
 * sizes and fills the twiddle blob).  p: prime modulus (any odd prime < 2^64
 * with n | p-1), g: a generator of (Z/p)^* (Modulus<p,g>, modulus.hpp:14),
 * n: transform length (power of two), batch: number of independent transforms
 * stored back to back (1 for the reference's API).
 * n0_log2: log2 of the column length R of the six-step split n = R x C
 * (RecursiveNTT<..., GenericSVELayer<..., inner_kernel R ...>, ..., true>,
 * kernel/recursive.hpp:61-75); 0 = choose automatically.
 */
int sventt_plan_create(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2,
                       uint64_t batch, uint32_t flags, sventt_plan **plan);

/*
 * The same with the inverse's scaling spelled out: the inverse transform multiplies its
 * result by inverse_divisor^{-1} mod p.  0 = n (what sventt_plan_create does: the oracle's
 * inverse, tests/ntt-reference.hpp:78-82); 1 = no scaling.  This is the product of the
 * `inverse_factor` template arguments of the reference's layers: a layer with
 * inverse_factor != 1 multiplies by its modular inverse (layer/sve/radix-two.hpp:208-235,
 * 307-328 and the radix-4/8 siblings), so README.md:36-68-shaped kernels (no inverse_factor)
 * return the UNSCALED inverse and kernels whose last layer ends in `..., m>`
 * (tests/ntt-tests/iterative-sve-radix8-two12.hpp:17) divide by m.
 */
int sventt_plan_create_ex(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2,
                          uint64_t batch, uint32_t flags, uint64_t inverse_divisor,
                          sventt_plan **plan);

/* HIP device ordinal the plan's tables live on (-1 for a null plan). */
int sventt_plan_device(const sventt_plan *plan);

void sventt_plan_destroy(sventt_plan *plan);

/*
 * Replace NTT::compute_forward / compute_inverse (wrapper.hpp:50-82).
 * dst and src hold n*batch uint64_t each and may alias exactly (dst == src,
 * the in-place overloads) but not partially.  They may be device pointers
 * (hipMalloc / torch) -- then the call is asynchronous on `stream`
 * (a hipStream_t, NULL = default stream) -- or plain host pointers, in which
 * case the data is staged through a plan-owned device buffer and the call
 * returns when dst is complete.
 */
int sventt_forward(const sventt_plan *plan, uint64_t *dst, const uint64_t *src,
                   void *stream);
int sventt_inverse(const sventt_plan *plan, uint64_t *dst, const uint64_t *src,
                   void *stream);

/*
 * One pass of the plan at a time, for callers that interleave their own work
 * (the multi-GPU driver runs the column pass, an all-to-all, then the row
 * pass).  pass_index in [0, sventt_plan_num_passes).  Device pointers only.
 */
int sventt_plan_num_passes(const sventt_plan *plan, int inverse);
int sventt_run_pass(const sventt_plan *plan, int inverse, int pass_index,
                    uint64_t *dst, const uint64_t *src, void *stream);

/*
 * Sharded six-step (no precedent in the reference, which is
 * single-node shared memory).  The global transform has n = R*C points viewed
 * as R rows x C columns (row-major, R = 2^r_log2).  With Cl = C/nranks and
 * Rl = R/nranks, rank k owns
 *   before the exchange: the column block [k*Cl, (k+1)*Cl) as an R x Cl slab
 *                        (row-major, leading dimension Cl);
 *   after the exchange:  the rows [k*Rl, (k+1)*Rl), i.e. the contiguous slice
 *                        [k*n/nranks, (k+1)*n/nranks) of the bit-reversed result.
 * Forward on every rank:
 *   1. sventt_sharded_columns(cols, 0, work, slab): the Cl column transforms of
 *      length R and the six-step twiddle omega_n^(bitrev_R(j)*c) with the GLOBAL
 *      column index c (layer/sve/generic.hpp:95-105 of the reference);
 *   2. all-to-all (RCCL): chunk h of `work` (rows [h*Rl,(h+1)*Rl), contiguous)
 *      goes to rank h; the receive buffer is recv[s][q][c] (source rank s, local
 *      row q, column c < Cl);
 *   3. the passes of the ROWS plan, sventt_run_pass(rows, 0, i, ...): pass 0
 *      reads `recv` and writes whole rows to `out`.  It is the first column pass
 *      of the length-C row transform, nranks * 2^k long: row s * 2^k + i' of its
 *      blocks is row i' of the piece received from rank s (a two-level stride),
 *      so the pieces are gathered on the fly -- no transposition and no sweep of
 *      its own (r02 gathered in a length-nranks pass; N = 2^30 on 8 ranks is now
 *      col 2^11 | exchange | col 2^7 | row 2^12: three sweeps of a rank's data).
 *      The remaining passes run in place on `out`.
 * Inverse = the mirror: rows passes 0..k-2 in place on the rows buffer, the last
 * rows pass writes piece layout, all-to-all back, then
 * sventt_sharded_columns(cols, 1, ...) which also applies the 1/n scaling.
 * Both plans take device pointers only and run through the entry points named
 * here (sventt_forward/inverse refuse them).  nranks is a power of two >= 1; one
 * rank (the exchange is with itself) runs the same pipeline on one GPU and needs
 * C >= 2^14.  r_log2 <= 12.
 */
int sventt_sharded_plan_create(uint64_t p, uint64_t g, uint64_t n,
                               uint32_t r_log2, int rank, int nranks,
                               uint32_t flags, sventt_plan **plan);
int sventt_sharded_rows_plan_create(uint64_t p, uint64_t g, uint64_t n,
                                    uint32_t r_log2, int rank, int nranks,
                                    uint32_t flags, sventt_plan **plan);
int sventt_sharded_columns(const sventt_plan *plan, int inverse,
                           uint64_t *dst, const uint64_t *src, void *stream);

/*
 * Pipelining the exchange.  A column pass (pass_index must name one) can be run on
 * chunk `chunk` of `nchunks` at a time, so that the all-to-all of one chunk overlaps
 * the passes of its neighbours.  The chunks are equal column ranges -- of the whole
 * block, or, for the column plan when the rows plan starts with a 2^k > 1 rows-per-piece
 * pass, of each of the 2^k runs of Cl / 2^k columns (that pass needs columns c, c + Cl/2^k,
 * ... of every piece together; the two plans agree on this by construction).  A side marked
 * compact is a buffer holding only that chunk's columns, run after run (leading dimensions
 * divided by nchunks, column index restarting at 0):
 *   forward: column plan pass 0, dst compact  -> work_k = R x (Cl/nchunks), whose row
 *            blocks are the all-to-all chunks; rows plan pass 0, src compact, reads the
 *            received recv_k[s][q][Cl/nchunks] and writes whole rows;
 *   inverse: rows plan last pass with dst compact, column plan pass 0 with src compact.
 */
/* Column tiles per block (per run, see above) of a column pass, 0 for a row pass: nchunks
 * must divide it (a power of two). */
uint64_t sventt_plan_pass_tiles_per_block(const sventt_plan *plan, int inverse,
                                          int pass_index);
int sventt_run_pass_chunk(const sventt_plan *plan, int inverse, int pass_index,
                          uint64_t *dst, const uint64_t *src, uint32_t chunk,
                          uint32_t nchunks, int dst_compact, int src_compact,
                          void *stream);

/*
 * The whole sharded transform in one call, for C/C++ hosts (no torch): the column pass in
 * `chunks` pieces, the all-to-all of each piece on an internal second stream as soon as the
 * piece is written, the gather pass of a piece as soon as it has arrived, the remaining row
 * passes.  `cols` / `rows` are this rank's plans from sventt_sharded_plan_create /
 * sventt_sharded_rows_plan_create (same p, g, n, r_log2, rank, nranks).  All buffers hold
 * n/nranks device words: forward reads the column slab `src` (left intact) and writes the row
 * block `dst`; inverse reads the row block `src` (left intact) and writes the column slab
 * `dst`; `work` and `recv` are scratch.  chunks >= 1 must divide
 * sventt_plan_pass_tiles_per_block of both plans' pass 0 (1 always does; 4 is what the Python
 * driver uses).  Work is enqueued on `stream` and on a plan-owned communication stream; the
 * call returns without waiting.  Calls with the same `rows` plan take turns.
 *
 * sventt_sharded_forward / _inverse: the exchange is ncclSend / ncclRecv between
 * ncclGroupStart / ncclGroupEnd on `nccl_comm` (an ncclComm_t of `nranks` ranks in which this
 * process has the plans' rank); librccl is loaded on first use (dlopen), the library neither
 * links it nor needs its headers.  WHAT HAS RUN: this path has executed on hardware with a
 * ONE-rank communicator only (tests/cpp/rccl_one_rank.cpp: the group of ncclSend/ncclRecv to
 * itself on the communication stream, ordered against the kernels by the same events, results
 * == oracle); with several ranks it is unmeasured -- no multi-GPU node was available -- and
 * only the loopback transport has carried N = 2^30 over 8 ranks (tests/cpp/sharded_driver.cpp).
 * The *_transport variants take the exchange as a callback instead (MPI, a test
 * loopback, another collective library):
 *   all_to_all(ctx, send, recv, count, stream): piece h of `send` (count words each) goes to
 *   rank h, piece s of `recv` comes from rank s; enqueue on `stream` (it may also block).
 *   Return 0 on success.
 */
typedef struct sventt_transport {
  void *ctx;
  int (*all_to_all)(void *ctx, const uint64_t *send, uint64_t *recv, uint64_t count_per_peer,
                    void *stream);
} sventt_transport;

int sventt_sharded_forward(const sventt_plan *cols, const sventt_plan *rows, void *nccl_comm,
                           uint64_t *dst, const uint64_t *src, uint64_t *work, uint64_t *recv,
                           uint32_t chunks, void *stream);
int sventt_sharded_inverse(const sventt_plan *cols, const sventt_plan *rows, void *nccl_comm,
                           uint64_t *dst, const uint64_t *src, uint64_t *work, uint64_t *recv,
                           uint32_t chunks, void *stream);
int sventt_sharded_forward_transport(const sventt_plan *cols, const sventt_plan *rows,
                                     const sventt_transport *transport, uint64_t *dst,
                                     const uint64_t *src, uint64_t *work, uint64_t *recv,
                                     uint32_t chunks, void *stream);
int sventt_sharded_inverse_transport(const sventt_plan *cols, const sventt_plan *rows,
                                     const sventt_transport *transport, uint64_t *dst,
                                     const uint64_t *src, uint64_t *work, uint64_t *recv,
                                     uint32_t chunks, void *stream);

/* Introspection (get_m(): wrapper.hpp:48; modulus_type: wrapper.hpp:32). */
uint64_t sventt_plan_n(const sventt_plan *plan);
uint64_t sventt_plan_batch(const sventt_plan *plan);
uint64_t sventt_plan_modulus(const sventt_plan *plan);
/* Human-readable description of the passes ("col 2^11 x T8 | row 2^13"). */
const char *sventt_plan_describe(const sventt_plan *plan);

/* Element-wise product of device arrays of `count` residues in [0, p):
 * dst[i] = a[i]*b[i] mod p.  The caller of the reference does this between a
 * forward and an inverse transform
 * (examples/magic-series/gaussian-polynomial.hpp:201-212). */
int sventt_pointwise_multiply(const sventt_plan *plan, uint64_t *dst,
                              const uint64_t *a, const uint64_t *b,
                              uint64_t count, void *stream);

/* Domain conversion of `count` device residues: dst[i] = src[i] * 2^64 mod p and back
 * (PAdic64SVE::to_montgomery / from_montgomery, modmul/sve/p-adic-64.hpp:64-74; the
 * reference's caller converts a spectrum once so that every later product with it is a
 * single Montgomery multiplication, examples/magic-series/gaussian-polynomial.hpp:177-179). */
int sventt_to_montgomery(const sventt_plan *plan, uint64_t *dst, const uint64_t *src,
                         uint64_t count, void *stream);
int sventt_from_montgomery(const sventt_plan *plan, uint64_t *dst, const uint64_t *src,
                           uint64_t count, void *stream);

/* Forward transform with the pointwise product fused into its last pass:
 *   dst = forward(src) (.) operand        (device pointers)
 * `operand` holds n*batch residues in MONTGOMERY form (sventt_to_montgomery), indexed
 * like the bit-reversed output; it may alias src but not dst.  This is the pair
 * ntt.compute_forward(x); x[j] = multiply_normalize(x[j], operand[j])
 * of examples/magic-series/gaussian-polynomial.hpp:199-212 without the extra pass over
 * memory; follow it with sventt_inverse for a cyclic convolution. */
int sventt_forward_multiply(const sventt_plan *plan, uint64_t *dst, const uint64_t *src,
                            const uint64_t *operand_montgomery, void *stream);

/* Stand-alone transposition of a matrix of 64-bit words,
 *   dst[ld_dst*c + r] = src[ld_src*r + c],  r < src_rows, c < src_cols,
 * replacing TransposeParallelSVEInRegister<br,bc>::transpose(dst, src, src_rows,
 * src_cols, ld_dst, ld_src) (transposition/sve/in-register.hpp:115-206) and its
 * relatives; leading dimensions may carry padding (ld_src >= src_cols,
 * ld_dst >= src_rows), any sizes (no block-divisibility rule).  dst and src must
 * not overlap, except that dst == src with src_rows == src_cols == ld_dst == ld_src
 * is the in-place case, as in the reference (:130-134).  Host or device pointers
 * (both of the same kind); the transforms never need this (they read columns in
 * place), it is for callers of the reference's transposes and as a bandwidth
 * yardstick (tests/bench-transpose.cpp:17-103). */
int sventt_transpose(uint64_t *dst, const uint64_t *src, uint64_t src_rows,
                     uint64_t src_cols, uint64_t ld_dst, uint64_t ld_src,
                     void *stream);
/* transpose(dst, dim): square, in place (transposition/sve/in-register.hpp:215-375). */
int sventt_transpose_inplace(uint64_t *dst, uint64_t dim, void *stream);

/* Page-lock (pin) a caller-owned host buffer so that the host-pointer path of
 * sventt_forward/sventt_inverse moves it at the PCIe rate (N = 2^24: 4.96 ms per
 * transform against 6.01 ms from pageable memory).  The reference keeps its data in
 * PageMemory<T> (vector.hpp:61-168: mmap'ed, optionally huge pages); the facade's
 * PageMemory registers its mapping with these two, best effort.  Unregister before the
 * memory is unmapped or freed. */
int sventt_host_register(void *host, size_t bytes);
int sventt_host_unregister(void *host);

const char *sventt_last_error(void);
const char *sventt_version(void);

#ifdef __cplusplus
}
#endif

#endif /* SVENTT_HIP_H_INCLUDED */
