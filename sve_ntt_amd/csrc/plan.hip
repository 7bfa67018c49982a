// sve_ntt_amd/csrc/plan.hip -- device side of a plan (table upload, pass launches)
// and the C ABI declared in include/sventt_hip.h.  The pass list itself is built
// by plan_core.h (host-only).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/sventt_hip.h"
#include "kernels.h"
#include "plan_core.h"

using namespace sventt_hip;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                            \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess)                                                        \
      return fail(e_ == hipErrorOutOfMemory ? SVENTT_ERR_ALLOC : SVENTT_ERR_HIP, \
                  std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)

struct DevicePass {
  const KernelEntry *kernel = nullptr;
  const KernelEntry *kernel_multiply = nullptr;  // final forward ROW pass with the product epilogue
  u64 *stage = nullptr, *twist_lo = nullptr, *twist_hi = nullptr;
};

int upload(const std::vector<u64> &host, u64 *&dev) {
  dev = nullptr;
  if (host.empty()) return SVENTT_OK;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dev), host.size() * sizeof(u64)));
  HIP_TRY(hipMemcpy(dev, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice));
  return SVENTT_OK;
}

}  // namespace

struct sventt_plan {
  HostPlan host;
  std::vector<DevicePass> fwd, inv;
  std::string description;
  int device = -1;              // the HIP device whose memory holds the tables
  bool device_pointers = false; // SVENTT_DEVICE_POINTERS: never classify dst/src
  // staging for host-pointer calls: one buffer per plan, so such calls on one plan take turns
  // (the mutex is held from the upload to the end of the download)
  mutable std::mutex staging_mutex;
  mutable u64 *staging = nullptr;
  mutable size_t staging_elems = 0;
  // sventt_sharded_forward/inverse (rows plan): the communication stream and per-chunk events
  mutable std::mutex shard_mutex;
  mutable hipStream_t comm_stream = nullptr;
  mutable std::vector<hipEvent_t> piece_ready, piece_arrived;
};

namespace {

int check_device() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(SVENTT_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
  return SVENTT_OK;
}

int realize(const std::vector<HostPass> &host, std::vector<DevicePass> &dev) {
  dev.resize(host.size());
  for (size_t i = 0; i < host.size(); ++i) {
    const HostPass &h = host[i];
    DevicePass &d = dev[i];
    d.kernel = find_kernel(h.kind, h.logl, h.inverse ? MODE_INV : MODE_FWD, h.flag ? 1 : 0, h.f0, h.loge, h.arith,
                           h.two_level);
    if (!d.kernel) return fail(SVENTT_ERR_LOGIC, "no kernel instantiated for this pass shape");
    if (d.kernel->f0 != h.f0 || d.kernel->logt != h.logt)
      return fail(SVENTT_ERR_LOGIC, "planner and kernel registry disagree on the tile shape");
    if (h.kind == KIND_ROW && !h.inverse && !h.flag)
      d.kernel_multiply = find_kernel(KIND_ROW, h.logl, MODE_FWD, 1, h.f0, h.loge, h.arith);
    int rc;
    if ((rc = upload(h.stage, d.stage))) return rc;
    if ((rc = upload(h.twist_lo, d.twist_lo))) return rc;
    if ((rc = upload(h.twist_hi, d.twist_hi))) return rc;
  }
  return SVENTT_OK;
}

int finish_plan(sventt_plan *pl, int rc, const std::string &err, sventt_plan **out) {
  if (rc) {
    delete pl;
    return fail(rc == PLAN_ERR_INVALID_ARGUMENT ? SVENTT_ERR_INVALID_ARGUMENT : SVENTT_ERR_LOGIC, err);
  }
  if (hipGetDevice(&pl->device) != hipSuccess) {
    delete pl;
    return fail(SVENTT_ERR_HIP, "hipGetDevice failed");
  }
  pl->device_pointers = (pl->host.flags & SVENTT_DEVICE_POINTERS) != 0;
  if ((rc = realize(pl->host.fwd, pl->fwd)) || (rc = realize(pl->host.inv, pl->inv))) {
    sventt_plan_destroy(pl);
    return rc;
  }
  pl->description = describe_plan(pl->host);
  // the tables now live on the device
  for (HostPass &h : pl->host.fwd) h.stage = {}, h.twist_lo = {}, h.twist_hi = {};
  for (HostPass &h : pl->host.inv) h.stage = {}, h.twist_lo = {}, h.twist_hi = {};
  *out = pl;
  return SVENTT_OK;
}

// device: if non-null, receives the ordinal of the device that owns the memory (-1: host)
bool is_device_pointer(const void *p, int *device = nullptr) {
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (device) *device = -1;
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain host memory: clear the sticky error
    return false;
  }
  const bool dev = attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged ||
                   attr.type == hipMemoryTypeArray;
  if (dev && device) *device = attr.device;
  return dev;
}

// The kernels run on the current device and read the plan's tables there.
int check_current_device(const sventt_plan *pl) {
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != pl->device)
    return fail(SVENTT_ERR_INVALID_ARGUMENT,
                "the plan was created on HIP device " + std::to_string(pl->device) +
                    " but the calling thread's current device is " + std::to_string(cur));
  return SVENTT_OK;
}

// One pass launch.  Every entry point that launches on behalf of a plan comes through here or
// through run_chunk / check_current_device itself: a launch from a thread whose current device is
// not the plan's would read tables that live elsewhere.
int run_pass(const sventt_plan *pl, bool inverse, size_t index, u64 *dst, const u64 *src,
             hipStream_t stream) {
  int rc = check_current_device(pl);
  if (rc) return rc;
  const HostPass &h = (inverse ? pl->host.inv : pl->host.fwd)[index];
  const DevicePass &d = (inverse ? pl->inv : pl->fwd)[index];
  const PassArgs a = make_args(pl->host, h, dst, src, d.stage, d.twist_lo, d.twist_hi);
  HIP_TRY(d.kernel->launch(a, (u32)h.grid, stream));
  return SVENTT_OK;
}

int transform_device(const sventt_plan *pl, bool inverse, u64 *dst, const u64 *src,
                     hipStream_t stream) {
  const size_t npass = (inverse ? pl->inv : pl->fwd).size();
  if (pl->host.n == 1) {
    if (inverse && pl->host.inverse_scale != 1) {
      HIP_TRY(launch_montmul(dst, src, nullptr, h_to_montgomery(pl->host.inverse_scale, pl->host.f.N),
                             pl->host.total, pl->host.f, stream));
    } else if (dst != src) {
      HIP_TRY(hipMemcpyAsync(dst, src, pl->host.total * sizeof(u64), hipMemcpyDeviceToDevice, stream));
    }
    return SVENTT_OK;
  }
  const u64 *in = src;
  for (size_t i = 0; i < npass; ++i) {
    int rc = run_pass(pl, inverse, i, dst, in, stream);
    if (rc) return rc;
    in = dst;
  }
  return SVENTT_OK;
}

int transform(const sventt_plan *pl, bool inverse, u64 *dst, const u64 *src, void *stream_) {
  if (!pl || !dst || !src) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (pl->host.sharded) return fail(SVENTT_ERR_LOGIC, "sharded plans run through sventt_sharded_columns");
  if (!(pl->host.flags & (inverse ? PLAN_INVERSE : PLAN_FORWARD)))
    return fail(SVENTT_ERR_LOGIC, inverse ? "plan was created without SVENTT_INVERSE"
                                          : "plan was created without SVENTT_FORWARD");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_current_device(pl);
  if (rc) return rc;
  if (pl->device_pointers) return transform_device(pl, inverse, dst, src, stream);
  int dev_dst = -1, dev_src = -1;
  const bool ddst = is_device_pointer(dst, &dev_dst), dsrc = is_device_pointer(src, &dev_src);
  if (ddst && dsrc) {
    if (dev_dst != pl->device || dev_src != pl->device)
      return fail(SVENTT_ERR_INVALID_ARGUMENT, "dst/src live on another HIP device than the plan");
    return transform_device(pl, inverse, dst, src, stream);
  }
  if (ddst != dsrc)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "dst and src must both be device or both be host pointers");
  // Host pointers (what NTT::compute_* of the reference take, wrapper.hpp:50-82):
  // stage through a plan-owned device buffer; returns when dst is complete.  Calls on the
  // same plan serialise here (one staging buffer per plan).
  std::lock_guard<std::mutex> lock(pl->staging_mutex);
  const size_t bytes = pl->host.total * sizeof(u64);
  if (pl->staging_elems < pl->host.total) {
    if (pl->staging) (void)hipFree(pl->staging);
    pl->staging = nullptr;
    pl->staging_elems = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->staging), bytes));
    pl->staging_elems = pl->host.total;
  }
  HIP_TRY(hipMemcpyAsync(pl->staging, src, bytes, hipMemcpyHostToDevice, stream));
  rc = transform_device(pl, inverse, pl->staging, pl->staging, stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(dst, pl->staging, bytes, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return SVENTT_OK;
}

}  // namespace

extern "C" {

int sventt_plan_create_ex(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2, uint64_t batch,
                          uint32_t flags, uint64_t inverse_divisor, sventt_plan **out) {
  if (!out) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null plan pointer");
  *out = nullptr;
  sventt_plan *pl = new (std::nothrow) sventt_plan;
  if (!pl) return fail(SVENTT_ERR_ALLOC, "out of host memory");
  std::string err;
  int rc = build_plan(pl->host, p, g, n, n0_log2, batch, flags, err, inverse_divisor);
  if (!rc && check_device()) {
    delete pl;
    return SVENTT_ERR_NO_DEVICE;
  }
  return finish_plan(pl, rc, err, out);
}

int sventt_plan_create(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2, uint64_t batch,
                       uint32_t flags, sventt_plan **out) {
  return sventt_plan_create_ex(p, g, n, n0_log2, batch, flags, 0, out);
}

int sventt_plan_device(const sventt_plan *pl) { return pl ? pl->device : -1; }

int sventt_sharded_plan_create(uint64_t p, uint64_t g, uint64_t n, uint32_t r_log2, int rank,
                               int nranks, uint32_t flags, sventt_plan **out) {
  if (!out) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null plan pointer");
  *out = nullptr;
  sventt_plan *pl = new (std::nothrow) sventt_plan;
  if (!pl) return fail(SVENTT_ERR_ALLOC, "out of host memory");
  std::string err;
  int rc = build_sharded_plan(pl->host, p, g, n, r_log2, rank, nranks, flags, err);
  if (!rc && check_device()) {
    delete pl;
    return SVENTT_ERR_NO_DEVICE;
  }
  return finish_plan(pl, rc, err, out);
}

void sventt_plan_destroy(sventt_plan *pl) {
  if (!pl) return;
  for (std::vector<DevicePass> *v : {&pl->fwd, &pl->inv})
    for (DevicePass &d : *v) {
      if (d.stage) (void)hipFree(d.stage);
      if (d.twist_lo) (void)hipFree(d.twist_lo);
      if (d.twist_hi) (void)hipFree(d.twist_hi);
    }
  if (pl->staging) (void)hipFree(pl->staging);
  for (hipEvent_t e : pl->piece_ready) (void)hipEventDestroy(e);
  for (hipEvent_t e : pl->piece_arrived) (void)hipEventDestroy(e);
  if (pl->comm_stream) (void)hipStreamDestroy(pl->comm_stream);
  delete pl;
}

int sventt_forward(const sventt_plan *pl, uint64_t *dst, const uint64_t *src, void *stream) {
  return transform(pl, false, dst, src, stream);
}

int sventt_inverse(const sventt_plan *pl, uint64_t *dst, const uint64_t *src, void *stream) {
  return transform(pl, true, dst, src, stream);
}

int sventt_plan_num_passes(const sventt_plan *pl, int inverse) {
  if (!pl) return 0;
  return (int)(inverse ? pl->inv.size() : pl->fwd.size());
}

int sventt_run_pass(const sventt_plan *pl, int inverse, int pass_index, uint64_t *dst,
                    const uint64_t *src, void *stream) {
  if (!pl || !dst || !src) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  const size_t npass = (inverse ? pl->inv : pl->fwd).size();
  if (pass_index < 0 || (size_t)pass_index >= npass)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "pass index out of range");
  return run_pass(pl, inverse != 0, (size_t)pass_index, dst, src, static_cast<hipStream_t>(stream));
}

uint64_t sventt_plan_pass_tiles_per_block(const sventt_plan *pl, int inverse, int pass_index) {
  if (!pl) return 0;
  const std::vector<HostPass> &passes = inverse ? pl->host.inv : pl->host.fwd;
  if (pass_index < 0 || (size_t)pass_index >= passes.size()) return 0;
  return pass_chunk_tiles(passes[(size_t)pass_index]);
}

int sventt_run_pass_chunk(const sventt_plan *pl, int inverse, int pass_index, uint64_t *dst,
                          const uint64_t *src, uint32_t chunk, uint32_t nchunks, int dst_compact,
                          int src_compact, void *stream) {
  if (!pl || !dst || !src) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  const size_t npass = (inverse ? pl->inv : pl->fwd).size();
  if (pass_index < 0 || (size_t)pass_index >= npass)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "pass index out of range");
  int rc = check_current_device(pl);
  if (rc) return rc;
  const HostPass &h = (inverse ? pl->host.inv : pl->host.fwd)[(size_t)pass_index];
  const DevicePass &d = (inverse ? pl->inv : pl->fwd)[(size_t)pass_index];
  PassArgs a;
  u32 grid = 0;
  std::string err;
  if (make_chunk_args(pl->host, h, dst, src, d.stage, d.twist_lo, d.twist_hi, chunk, nchunks,
                      dst_compact != 0, src_compact != 0, a, grid, err))
    return fail(SVENTT_ERR_INVALID_ARGUMENT, err);
  HIP_TRY(d.kernel->launch(a, grid, static_cast<hipStream_t>(stream)));
  return SVENTT_OK;
}

int sventt_sharded_rows_plan_create(uint64_t p, uint64_t g, uint64_t n, uint32_t r_log2, int rank,
                                    int nranks, uint32_t flags, sventt_plan **out) {
  if (!out) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null plan pointer");
  *out = nullptr;
  sventt_plan *pl = new (std::nothrow) sventt_plan;
  if (!pl) return fail(SVENTT_ERR_ALLOC, "out of host memory");
  std::string err;
  int rc = build_sharded_rows_plan(pl->host, p, g, n, r_log2, rank, nranks, flags, err);
  pl->host.sharded = true;  // driven pass by pass, never through sventt_forward/inverse
  if (!rc && check_device()) {
    delete pl;
    return SVENTT_ERR_NO_DEVICE;
  }
  return finish_plan(pl, rc, err, out);
}

int sventt_sharded_columns(const sventt_plan *pl, int inverse, uint64_t *dst, const uint64_t *src,
                           void *stream) {
  if (!pl || !dst || !src) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (!pl->host.sharded || pl->host.local_cols == 0)
    return fail(SVENTT_ERR_LOGIC, "not a sharded column plan");
  if ((inverse ? pl->inv : pl->fwd).empty())
    return fail(SVENTT_ERR_LOGIC, "direction not enabled in this plan");
  return run_pass(pl, inverse != 0, 0, dst, src, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---- the sharded transform as one call ------------------------------------------------
namespace {

int run_chunk(const sventt_plan *pl, bool inverse, size_t index, u64 *dst, const u64 *src, u32 chunk,
              u32 nchunks, bool dst_compact, bool src_compact, hipStream_t stream) {
  if (int rc = check_current_device(pl)) return rc;
  const HostPass &h = (inverse ? pl->host.inv : pl->host.fwd)[index];
  const DevicePass &d = (inverse ? pl->inv : pl->fwd)[index];
  PassArgs a;
  u32 grid = 0;
  std::string err;
  if (make_chunk_args(pl->host, h, dst, src, d.stage, d.twist_lo, d.twist_hi, chunk, nchunks,
                      dst_compact, src_compact, a, grid, err))
    return fail(SVENTT_ERR_INVALID_ARGUMENT, err);
  HIP_TRY(d.kernel->launch(a, grid, stream));
  return SVENTT_OK;
}

int check_shard_pair(const sventt_plan *cols, const sventt_plan *rows, bool inverse, u32 chunks) {
  if (!cols->host.sharded || cols->host.local_cols == 0)
    return fail(SVENTT_ERR_LOGIC, "`cols` is not a sharded column plan");
  if (!rows->host.sharded || rows->host.local_cols != 0 || rows->host.nranks < 1)
    return fail(SVENTT_ERR_LOGIC, "`rows` is not a sharded rows plan");
  if (cols->host.nranks != rows->host.nranks || cols->host.rank != rows->host.rank ||
      cols->host.f.N != rows->host.f.N || cols->host.total != rows->host.total)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "the two plans do not describe the same rank of the same transform");
  if ((inverse ? cols->inv : cols->fwd).empty() || (inverse ? rows->inv : rows->fwd).empty())
    return fail(SVENTT_ERR_LOGIC, "direction not enabled in these plans");
  const HostPass &cp = (inverse ? cols->host.inv : cols->host.fwd)[0];
  const std::vector<HostPass> &rp = inverse ? rows->host.inv : rows->host.fwd;
  const HostPass &xp = inverse ? rp.back() : rp.front();  // the pass next to the exchange
  if (chunks == 0 || pass_chunk_tiles(cp) % chunks != 0 || pass_chunk_tiles(xp) % chunks != 0)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "chunks must divide the column tile counts of both plans");
  if (cols->host.total % ((u64)chunks * (u64)cols->host.nranks) != 0)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "chunks do not divide the local data");
  return SVENTT_OK;
}

int ensure_shard_ctx(const sventt_plan *rows, u32 chunks) {
  if (!rows->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&rows->comm_stream, hipStreamNonBlocking));
  while (rows->piece_ready.size() < chunks) {
    hipEvent_t a = nullptr, b = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    rows->piece_ready.push_back(a);
    HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    rows->piece_arrived.push_back(b);
  }
  return SVENTT_OK;
}

int sharded_transform(const sventt_plan *cols, const sventt_plan *rows, const sventt_transport *tr,
                      bool inverse, u64 *dst, const u64 *src, u64 *work, u64 *recv, u32 chunks,
                      void *stream_) {
  if (!cols || !rows || !tr || !tr->all_to_all || !dst || !src || !work || !recv)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (dst == src || work == recv || work == dst || recv == dst || work == src || recv == src)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "dst, src, work and recv must be four distinct buffers");
  int rc = check_shard_pair(cols, rows, inverse, chunks);
  if (rc) return rc;
  if ((rc = check_current_device(cols)) || (rc = check_current_device(rows))) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  std::lock_guard<std::mutex> lock(rows->shard_mutex);
  if ((rc = ensure_shard_ctx(rows, chunks))) return rc;
  hipStream_t comm = rows->comm_stream;
  const u32 K = chunks;
  const u64 piece = cols->host.total / K;               // words per chunk
  const u64 per_peer = piece / (u64)cols->host.nranks;  // words per chunk and peer
  const size_t rows_passes = (inverse ? rows->inv : rows->fwd).size();
  auto exchange = [&](u32 k, u64 *out, const u64 *in) -> int {
    // `in` is complete once the launches enqueued so far on `stream` have run
    HIP_TRY(hipEventRecord(rows->piece_ready[k], stream));
    HIP_TRY(hipStreamWaitEvent(comm, rows->piece_ready[k], 0));
    if (tr->all_to_all(tr->ctx, in, out, per_peer, comm) != 0)
      return fail(SVENTT_ERR_COMM, "the transport's all_to_all failed");
    HIP_TRY(hipEventRecord(rows->piece_arrived[k], comm));
    return SVENTT_OK;
  };
  if (!inverse) {
    for (u32 k = 0; k < K; ++k) {
      // chunk k of the column pass, written compactly: row block h of the piece is what rank h needs
      if ((rc = run_chunk(cols, false, 0, work + k * piece, src, k, K, true, false, stream))) return rc;
      if ((rc = exchange(k, recv + k * piece, work + k * piece))) return rc;
    }
    for (u32 k = 0; k < K; ++k) {
      HIP_TRY(hipStreamWaitEvent(stream, rows->piece_arrived[k], 0));
      // first pass of the row transform: reads the received pieces in place of a transposition
      if ((rc = run_chunk(rows, false, 0, dst, recv + k * piece, k, K, false, true, stream))) return rc;
    }
    for (size_t i = 1; i < rows_passes; ++i)
      if ((rc = run_pass(rows, false, i, dst, dst, stream))) return rc;
  } else {
    // rows passes but the last run on `dst` (free until the column pass writes it; every read of it
    // is enqueued before the first such write); the last one scatters chunk k into piece layout
    const u64 *cur = src;
    for (size_t i = 0; i + 1 < rows_passes; ++i) {
      if ((rc = run_pass(rows, true, i, dst, cur, stream))) return rc;
      cur = dst;
    }
    for (u32 k = 0; k < K; ++k) {
      if ((rc = run_chunk(rows, true, rows_passes - 1, work + k * piece, cur, k, K, true, false, stream)))
        return rc;
      if ((rc = exchange(k, recv + k * piece, work + k * piece))) return rc;
    }
    for (u32 k = 0; k < K; ++k) {
      HIP_TRY(hipStreamWaitEvent(stream, rows->piece_arrived[k], 0));
      if ((rc = run_chunk(cols, true, 0, dst, recv + k * piece, k, K, false, true, stream))) return rc;
    }
  }
  // the next call may reuse work/recv on `stream`: order it after this call's exchanges
  return SVENTT_OK;
}

// RCCL, loaded on first use so that single-GPU users need neither the library nor its headers: the
// handful of types and values of <rccl/rccl.h> (nccl.h 2.x ABI) that the exchange uses are restated here.
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;    // ncclSuccess == 0
typedef int ncclDataType_t;  // ncclUint64 == 5
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclUint64 = 5;

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

const Rccl &rccl() {
  static const Rccl r = [] {
    Rccl x;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.handle) break;
    }
    if (!x.handle) {
      x.error = std::string("cannot load librccl: ") + dlerror();
      return x;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(x.handle, n);
      if (!p && x.error.empty()) x.error = std::string("librccl lacks ") + n;
      return p;
    };
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
    x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
    x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(sym("ncclCommUserRank"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    return x;
  }();
  return r;
}

struct RcclCtx {
  ncclComm_t comm;
  int nranks;
};

int rccl_all_to_all(void *ctx_, const uint64_t *send, uint64_t *recv, uint64_t count, void *stream_) {
  const RcclCtx *ctx = static_cast<const RcclCtx *>(ctx_);
  const Rccl &r = rccl();
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ncclResult_t e = r.GroupStart();
  for (int peer = 0; peer < ctx->nranks && e == ncclSuccess; ++peer) {
    e = r.Send(send + (size_t)peer * count, count, ncclUint64, peer, ctx->comm, stream);
    if (e == ncclSuccess) e = r.Recv(recv + (size_t)peer * count, count, ncclUint64, peer, ctx->comm, stream);
  }
  const ncclResult_t e2 = r.GroupEnd();
  if (e == ncclSuccess) e = e2;
  if (e != ncclSuccess) {
    g_last_error = std::string("RCCL: ") + r.GetErrorString(e);
    return 1;
  }
  return 0;
}

int sharded_rccl(const sventt_plan *cols, const sventt_plan *rows, void *nccl_comm, bool inverse,
                 u64 *dst, const u64 *src, u64 *work, u64 *recv, u32 chunks, void *stream) {
  if (!cols || !rows || !nccl_comm) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  const Rccl &r = rccl();
  if (!r.error.empty()) return fail(SVENTT_ERR_COMM, r.error);
  RcclCtx ctx{static_cast<ncclComm_t>(nccl_comm), 0};
  int rank = -1;
  if (r.CommCount(ctx.comm, &ctx.nranks) != ncclSuccess || r.CommUserRank(ctx.comm, &rank) != ncclSuccess)
    return fail(SVENTT_ERR_COMM, "cannot query the communicator");
  if (ctx.nranks != rows->host.nranks || rank != rows->host.rank)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "the communicator's size/rank differ from the plans'");
  const sventt_transport tr{&ctx, &rccl_all_to_all};
  const int rc = sharded_transform(cols, rows, &tr, inverse, dst, src, work, recv, chunks, stream);
  if (rc == SVENTT_ERR_COMM && g_last_error.rfind("RCCL", 0) != 0) g_last_error = "RCCL exchange failed";
  return rc;
}

}  // namespace

extern "C" {

int sventt_sharded_forward_transport(const sventt_plan *cols, const sventt_plan *rows,
                                     const sventt_transport *transport, uint64_t *dst,
                                     const uint64_t *src, uint64_t *work, uint64_t *recv,
                                     uint32_t chunks, void *stream) {
  return sharded_transform(cols, rows, transport, false, dst, src, work, recv, chunks, stream);
}

int sventt_sharded_inverse_transport(const sventt_plan *cols, const sventt_plan *rows,
                                     const sventt_transport *transport, uint64_t *dst,
                                     const uint64_t *src, uint64_t *work, uint64_t *recv,
                                     uint32_t chunks, void *stream) {
  return sharded_transform(cols, rows, transport, true, dst, src, work, recv, chunks, stream);
}

int sventt_sharded_forward(const sventt_plan *cols, const sventt_plan *rows, void *nccl_comm,
                           uint64_t *dst, const uint64_t *src, uint64_t *work, uint64_t *recv,
                           uint32_t chunks, void *stream) {
  return sharded_rccl(cols, rows, nccl_comm, false, dst, src, work, recv, chunks, stream);
}

int sventt_sharded_inverse(const sventt_plan *cols, const sventt_plan *rows, void *nccl_comm,
                           uint64_t *dst, const uint64_t *src, uint64_t *work, uint64_t *recv,
                           uint32_t chunks, void *stream) {
  return sharded_rccl(cols, rows, nccl_comm, true, dst, src, work, recv, chunks, stream);
}

uint64_t sventt_plan_n(const sventt_plan *pl) { return pl ? pl->host.n : 0; }
uint64_t sventt_plan_batch(const sventt_plan *pl) { return pl ? pl->host.batch : 0; }
uint64_t sventt_plan_modulus(const sventt_plan *pl) { return pl ? pl->host.f.N : 0; }
const char *sventt_plan_describe(const sventt_plan *pl) { return pl ? pl->description.c_str() : ""; }

int sventt_pointwise_multiply(const sventt_plan *pl, uint64_t *dst, const uint64_t *a,
                              const uint64_t *b, uint64_t count, void *stream) {
  if (!pl || !dst || !a || !b) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (!pl->device_pointers && (!is_device_pointer(dst) || !is_device_pointer(a) || !is_device_pointer(b)))
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "pointwise multiply takes device pointers");
  if (int rc = check_current_device(pl)) return rc;
  HIP_TRY(launch_pointwise(dst, a, b, count, pl->host.f, pl->host.r2,
                           static_cast<hipStream_t>(stream)));
  return SVENTT_OK;
}

int sventt_forward_multiply(const sventt_plan *pl, uint64_t *dst, const uint64_t *src,
                            const uint64_t *operand, void *stream_) {
  if (!pl || !dst || !src || !operand) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (pl->host.sharded) return fail(SVENTT_ERR_LOGIC, "not available on sharded plans");
  if (!(pl->host.flags & PLAN_FORWARD)) return fail(SVENTT_ERR_LOGIC, "plan was created without SVENTT_FORWARD");
  if (!pl->device_pointers &&
      (!is_device_pointer(dst) || !is_device_pointer(src) || !is_device_pointer(operand)))
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "the fused forward-multiply takes device pointers");
  if (operand == dst) return fail(SVENTT_ERR_INVALID_ARGUMENT, "operand must not alias dst");
  if (int rc = check_current_device(pl)) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (pl->host.n == 1) {
    HIP_TRY(launch_montmul(dst, src, operand, 0, pl->host.total, pl->host.f, stream));
    return SVENTT_OK;
  }
  const size_t npass = pl->fwd.size();
  const u64 *in = src;
  for (size_t i = 0; i + 1 < npass; ++i) {
    int rc = run_pass(pl, false, i, dst, in, stream);
    if (rc) return rc;
    in = dst;
  }
  const HostPass &h = pl->host.fwd[npass - 1];
  const DevicePass &d = pl->fwd[npass - 1];
  if (!d.kernel_multiply) return fail(SVENTT_ERR_LOGIC, "no fused kernel for this plan's final pass");
  PassArgs a = make_args(pl->host, h, dst, in, d.stage, d.twist_lo, d.twist_hi);
  a.epilogue = operand;
  HIP_TRY(d.kernel_multiply->launch(a, (u32)h.grid, stream));
  return SVENTT_OK;
}

static int convert_domain(const sventt_plan *pl, uint64_t *dst, const uint64_t *src, uint64_t count,
                          u64 factor, void *stream) {
  if (!pl || !dst || !src) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (!pl->device_pointers && (!is_device_pointer(dst) || !is_device_pointer(src)))
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "domain conversion takes device pointers");
  if (int rc = check_current_device(pl)) return rc;
  HIP_TRY(launch_montmul(dst, src, nullptr, factor, count, pl->host.f, static_cast<hipStream_t>(stream)));
  return SVENTT_OK;
}

int sventt_to_montgomery(const sventt_plan *pl, uint64_t *dst, const uint64_t *src, uint64_t count,
                         void *stream) {
  return convert_domain(pl, dst, src, count, pl ? pl->host.r2 : 0, stream);
}

int sventt_from_montgomery(const sventt_plan *pl, uint64_t *dst, const uint64_t *src, uint64_t count,
                           void *stream) {
  return convert_domain(pl, dst, src, count, 1, stream);
}

int sventt_transpose(uint64_t *dst, const uint64_t *src, uint64_t rows, uint64_t cols,
                     uint64_t ld_dst, uint64_t ld_src, void *stream_) {
  if (!dst || !src) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (ld_src < cols || ld_dst < rows)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "leading dimension shorter than the row it holds");
  if (rows == 0 || cols == 0) return SVENTT_OK;
  if ((rows | cols | ld_dst | ld_src) >> 31)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "matrix dimensions must stay below 2^31");
  if (dst == src) {
    if (rows == cols && ld_dst == rows && ld_src == cols) return sventt_transpose_inplace(dst, rows, stream_);
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "only a square unpadded matrix can be transposed in place");
  }
  const size_t src_elems = (size_t)(ld_src * (rows - 1) + cols), dst_elems = (size_t)(ld_dst * (cols - 1) + rows);
  if (dst < src + src_elems && src < dst + dst_elems)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "dst and src overlap");
  if (check_device()) return SVENTT_ERR_NO_DEVICE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool ddst = is_device_pointer(dst), dsrc = is_device_pointer(src);
  if (ddst && dsrc) {
    HIP_TRY(launch_transpose(dst, src, rows, cols, ld_dst, ld_src, stream));
    return SVENTT_OK;
  }
  if (ddst != dsrc)
    return fail(SVENTT_ERR_INVALID_ARGUMENT, "dst and src must both be device or both be host pointers");
  // host matrices: stage both through device memory (padding words of dst are preserved)
  u64 *buf = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), (src_elems + dst_elems) * sizeof(u64)));
  hipError_t e = hipMemcpyAsync(buf, src, src_elems * sizeof(u64), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess && ld_dst != rows)
    e = hipMemcpyAsync(buf + src_elems, dst, dst_elems * sizeof(u64), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = launch_transpose(buf + src_elems, buf, rows, cols, ld_dst, ld_src, stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(dst, buf + src_elems, dst_elems * sizeof(u64), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(buf);
  HIP_TRY(e);
  return SVENTT_OK;
}

int sventt_transpose_inplace(uint64_t *dst, uint64_t dim, void *stream_) {
  if (!dst) return fail(SVENTT_ERR_INVALID_ARGUMENT, "null argument");
  if (dim == 0) return SVENTT_OK;
  if (dim >> 31) return fail(SVENTT_ERR_INVALID_ARGUMENT, "matrix dimensions must stay below 2^31");
  if (check_device()) return SVENTT_ERR_NO_DEVICE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (is_device_pointer(dst)) {
    HIP_TRY(launch_transpose_inplace(dst, dim, stream));
    return SVENTT_OK;
  }
  u64 *buf = nullptr;
  const size_t bytes = (size_t)dim * dim * sizeof(u64);
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), bytes));
  hipError_t e = hipMemcpyAsync(buf, dst, bytes, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = launch_transpose_inplace(buf, dim, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dst, buf, bytes, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(buf);
  HIP_TRY(e);
  return SVENTT_OK;
}

int sventt_host_register(void *host, size_t bytes) {
  if (!host || bytes == 0) return fail(SVENTT_ERR_INVALID_ARGUMENT, "sventt_host_register: empty buffer");
  int rc = check_device();
  if (rc) return rc;
  const hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    // callers use this best-effort (PageMemory): do not leave HIP's per-thread sticky error behind,
    // the next launch on this thread would report it as its own (launch_tile returns hipGetLastError)
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? SVENTT_ERR_ALLOC : SVENTT_ERR_HIP,
                std::string("hipHostRegister: ") + hipGetErrorString(e));
  }
  return SVENTT_OK;
}

int sventt_host_unregister(void *host) {
  if (!host) return fail(SVENTT_ERR_INVALID_ARGUMENT, "sventt_host_unregister: null pointer");
  const hipError_t e = hipHostUnregister(host);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(SVENTT_ERR_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e));
  }
  return SVENTT_OK;
}

const char *sventt_last_error(void) { return g_last_error.c_str(); }
const char *sventt_version(void) { return "sventt-hip 0.1 (gfx950)"; }

}  // extern "C"
