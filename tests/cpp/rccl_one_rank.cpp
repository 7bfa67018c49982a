// tests/cpp/rccl_one_rank.cpp -- sventt_sharded_forward / _inverse with a REAL RCCL communicator.
//
// A one-GPU box cannot host two RCCL ranks (RCCL refuses several ranks on one device), so this is
// as far as the library's own RCCL path can be executed without a multi-GPU node: a ONE-rank
// communicator (ncclCommInitAll on device 0).  The sharded plans of one rank run the whole
// pipeline of include/sventt_hip.h -- column pass in chunks, every chunk's exchange on the plan's
// non-blocking communication stream between the same two events as on N ranks, the (two-level)
// first pass of the row phase per chunk, the remaining row passes -- and the exchange is the
// library's rccl_all_to_all (csrc/plan.hip): librccl found by dlopen, ncclGroupStart, ncclSend and
// ncclRecv with peer 0 (itself), ncclGroupEnd.  What this executes that nothing else on one GPU
// does: the dlopen/dlsym table, the group call with a self send/recv, and the ordering of RCCL's
// work on the communication stream against the kernels on the caller's stream.
// What it cannot show: more than one peer, and any throughput.
//
// Results are checked against the scalar oracle (tests/ntt-reference.hpp:43-83 of the reference,
// via oracle/ntt_oracle.c) at 2^22 and against the closed form of the transform of a[i] = s + i
// at 2^27 (tests/test-ntt-reference.cpp:45-63 of the reference check the same identity).
//
//   ./rccl_one_rank --compile-only-check   exits 0 without touching the GPU
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../oracle/ntt_oracle.h"
#include "sventt_hip.h"

namespace {

constexpr std::uint64_t P{UINT64_C(0xfffffc6e80000001)}, G_ROOT{3}, IOTA_START{UINT64_C(0x0123456789abcdef)};
using u128 = unsigned __int128;

std::uint64_t mulmod(std::uint64_t a, std::uint64_t b) { return static_cast<std::uint64_t>(u128{a} * b % P); }
std::uint64_t powmod(std::uint64_t a, std::uint64_t e) {
  std::uint64_t r{1};
  for (; e; e >>= 1, a = mulmod(a, a)) {
    if (e & 1) r = mulmod(r, a);
  }
  return r;
}
std::uint64_t bitrev(std::uint64_t x, unsigned bits) {
  std::uint64_t r{};
  for (unsigned b{}; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
  return r;
}
std::uint64_t closed_form(std::uint64_t k, unsigned log2n) {
  const std::uint64_t n{std::uint64_t{1} << log2n}, nm{n % P};
  if (k == 0) {
    const u128 tri{u128{n} * (n - 1) / 2};
    return static_cast<std::uint64_t>((u128{mulmod(nm, IOTA_START % P)} + static_cast<std::uint64_t>(tri % P)) % P);
  }
  const std::uint64_t w{powmod(G_ROOT, (P - 1) >> log2n)};
  const std::uint64_t wk{powmod(w, k)}, d{wk == 0 ? P - 1 : wk - 1};  // omega^k - 1 (mod P)
  return mulmod(nm, powmod(d, P - 2));
}

#define HIP_OK(x)                                                \
  do {                                                           \
    if ((x) != hipSuccess) {                                     \
      std::printf("HIP failure at %s:%d\n", __FILE__, __LINE__); \
      return false;                                              \
    }                                                            \
  } while (0)

bool run(ncclComm_t comm, unsigned log2n, unsigned r_log2, unsigned chunks, bool closed) {
  const std::uint64_t n{std::uint64_t{1} << log2n};
  sventt_plan *cols{}, *rows{};
  if (sventt_sharded_plan_create(P, G_ROOT, n, r_log2, 0, 1, SVENTT_BOTH, &cols) ||
      sventt_sharded_rows_plan_create(P, G_ROOT, n, r_log2, 0, 1, SVENTT_BOTH, &rows)) {
    std::printf("plan: %s\n", sventt_last_error());
    return false;
  }
  std::vector<std::uint64_t> input(n), want, out(n);
  if (closed) {
    oracle_fill_iota(input.data(), n, IOTA_START);
  } else {
    want.resize(n);
    oracle_fill_splitmix(input.data(), n, 99 + log2n, P);
    oracle_ntt_forward(want.data(), input.data(), n, P, G_ROOT);
  }
  const std::size_t bytes{n * sizeof(std::uint64_t)};
  std::uint64_t *d_src{}, *d_dst{}, *d_work{}, *d_recv{}, *d_back{};
  for (std::uint64_t **p : {&d_src, &d_dst, &d_work, &d_recv, &d_back}) HIP_OK(hipMalloc(p, bytes));
  HIP_OK(hipMemcpy(d_src, input.data(), bytes, hipMemcpyHostToDevice));
  hipStream_t stream{};
  HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  bool ok{true};
  for (int rep{}; rep < 2 && ok; ++rep) {
    HIP_OK(hipMemsetAsync(d_dst, 0x55, bytes, stream));
    HIP_OK(hipMemsetAsync(d_recv, 0x33, bytes, stream));
    if (sventt_sharded_forward(cols, rows, comm, d_dst, d_src, d_work, d_recv, chunks, stream)) {
      std::printf("forward: %s\n", sventt_last_error());
      return false;
    }
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out.data(), d_dst, bytes, hipMemcpyDeviceToHost));
    if (closed) {
      std::uint64_t x{0x9e3779b97f4a7c15ull};
      for (int smp{}; smp < 4096 && ok; ++smp) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        const std::uint64_t j{smp < 64 ? static_cast<std::uint64_t>(smp) : x % n};
        ok = out[j] == closed_form(bitrev(j, log2n), log2n);
      }
    } else {
      ok = out == want;
    }
    if (!ok) std::printf("MISMATCH forward (rep %d)\n", rep);
    HIP_OK(hipMemsetAsync(d_back, 0x55, bytes, stream));
    if (sventt_sharded_inverse(cols, rows, comm, d_back, d_dst, d_work, d_recv, chunks, stream)) {
      std::printf("inverse: %s\n", sventt_last_error());
      return false;
    }
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out.data(), d_back, bytes, hipMemcpyDeviceToHost));
    if (out != input) {
      std::printf("MISMATCH inverse(forward(x)) != x (rep %d)\n", rep);
      ok = false;
    }
  }
  // timing of the pipeline (closed-form cases only): the exchange is a copy of the rank's data to itself, so
  // what this shows is whether the chunks' exchanges run beside the passes of their neighbours
  if (closed && ok) {
    hipEvent_t e0{}, e1{};
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int i{}; i < 3; ++i) (void)sventt_sharded_forward(cols, rows, comm, d_dst, d_src, d_work, d_recv, chunks, stream);
    HIP_OK(hipEventRecord(e0, stream));
    const int reps{10};
    for (int i{}; i < reps; ++i) {
      if (sventt_sharded_forward(cols, rows, comm, d_dst, d_src, d_work, d_recv, chunks, stream)) return false;
    }
    HIP_OK(hipEventRecord(e1, stream));
    HIP_OK(hipEventSynchronize(e1));
    float ms{};
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("   timing: %.3f ms per forward transform of 2^%u points with %u chunk(s), exchange = RCCL self send/recv\n",
                ms / reps, log2n, chunks);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  // a communicator of the wrong size is refused before anything is enqueued
  std::printf("%s RCCL one-rank sharded transform: n=2^%u, R=2^%u, %u chunks (%u RCCL group calls per direction), %s check\n"
              "   plan: %s | all-to-all (RCCL, self) | %s\n",
              ok ? "ok" : "MISMATCH", log2n, r_log2, chunks, chunks, closed ? "closed-form" : "oracle",
              sventt_plan_describe(cols), sventt_plan_describe(rows));
  (void)hipStreamDestroy(stream);
  for (std::uint64_t *p : {d_src, d_dst, d_work, d_recv, d_back}) (void)hipFree(p);
  sventt_plan_destroy(cols);
  sventt_plan_destroy(rows);
  return ok;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc > 1 && std::string{argv[1]} == "--compile-only-check") {
    std::printf("RCCL one-rank harness compiled and linked\n");
    return 0;
  }
  if (hipSetDevice(0) != hipSuccess) {
    std::printf("no HIP device\n");
    return 1;
  }
  ncclComm_t comm{};
  const int devs[1]{0};
  const ncclResult_t rc{ncclCommInitAll(&comm, 1, devs)};
  if (rc != ncclSuccess) {
    std::printf("ncclCommInitAll: %s\n", ncclGetErrorString(rc));
    return 1;
  }
  bool ok{true};
  ok &= run(comm, 22, 8, 4, false);   // C = 2^14: col 2^2 (two-level form) | row 2^12
  ok &= run(comm, 22, 6, 1, false);   // C = 2^16: col 2^3 | row 2^13, one chunk
  ok &= run(comm, 27, 11, 4, true);   // the per-rank size of BASELINE configs[4]
  ok &= run(comm, 27, 11, 1, true);   // the same unpipelined (timing comparison)
  // plans of two ranks against the one-rank communicator: refused, nothing enqueued
  {
    sventt_plan *cols{}, *rows{};
    std::uint64_t *buf{};
    if (sventt_sharded_plan_create(P, G_ROOT, 1 << 20, 8, 0, 2, SVENTT_BOTH, &cols) == 0 &&
        sventt_sharded_rows_plan_create(P, G_ROOT, 1 << 20, 8, 0, 2, SVENTT_BOTH, &rows) == 0 &&
        hipMalloc(&buf, std::size_t{4} << 22) == hipSuccess) {
      const int r{sventt_sharded_forward(cols, rows, comm, buf, buf + (1 << 19), buf + (2 << 19), buf + (3 << 19), 1,
                                         nullptr)};
      if (r != SVENTT_ERR_INVALID_ARGUMENT) {
        std::printf("MISMATCH: a communicator of the wrong size was accepted (%d)\n", r);
        ok = false;
      }
      (void)hipFree(buf);
    } else {
      ok = false;
    }
    sventt_plan_destroy(cols);
    sventt_plan_destroy(rows);
  }
  (void)ncclCommDestroy(comm);
  std::printf(ok ? "ALL OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
