// tests/cpp/sharded_driver.cpp -- the C entry points of the sharded six-step driven from C++.
//
// sventt_sharded_forward_transport / _inverse_transport (include/sventt_hip.h) run one rank's
// share of a transform: column pass in chunks, the exchange of each chunk on the plan's
// communication stream, gather pass, row passes.  Here G ranks are G host threads sharing the
// one GPU of the test box and the exchange is a loopback transport (device-to-device copies
// between the ranks' buffers, host barriers) -- RCCL refuses several ranks on one device, so
// what is under test is the driver's sequencing, chunking and buffer use, not RCCL itself.
// Every rank's output is compared with its slice of the scalar oracle's result
// (tests/ntt-reference.hpp:43-83 of the reference, via oracle/ntt_oracle.c).
//
// Closed-form cases (Case::closed) verify sizes no oracle run fits, BASELINE configs[4] among them
// (N = 2^30 over 8 ranks -- here 8 thread-ranks x 5 buffers x 1 GiB on the one GPU, the exact plans,
// chunking and event graph an 8-GPU node runs, only the wire replaced by the loopback): the input is
// a[i] = s + i (the reference harness's recipe, tests/bench-ntt.cpp:31-33), whose transform is
// X[0] = n*s + n(n-1)/2, X[k] = n / (omega^k - 1) (tests/test-ntt-reference.cpp:45-63 of the
// reference check k = 0, 1 the same way); sampled outputs of every rank are compared with it and the
// inverse must return the input bit for bit.
//
//   g++ -std=c++20 -pthread -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__
//       tests/cpp/sharded_driver.cpp -Lsve_ntt_amd -lsventt_hip -Loracle -lntt_oracle -lamdhip64 ...
//   ./sharded_driver --compile-only-check   exits 0 without touching the GPU
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <barrier>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../oracle/ntt_oracle.h"
#include "sventt_hip.h"

namespace {

constexpr std::uint64_t P{UINT64_C(0xfffffc6e80000001)}, G_ROOT{3};

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    if ((x) != hipSuccess) {                                                       \
      std::printf("HIP failure at %s:%d\n", __FILE__, __LINE__);                   \
      return false;                                                                \
    }                                                                              \
  } while (0)

struct Shared {
  explicit Shared(int g) : ranks{g}, send(g, nullptr), sync{g} {}
  int ranks;
  std::vector<const std::uint64_t *> send;
  std::barrier<> sync;
  std::atomic<int> calls{0};
};

struct Loopback {
  Shared *shared;
  int rank;
};

// piece h of `send` -> rank h; piece s of `recv` <- rank s
int loopback_all_to_all(void *ctx_, const std::uint64_t *send, std::uint64_t *recv, std::uint64_t count,
                        void *stream_) {
  auto *ctx = static_cast<Loopback *>(ctx_);
  auto stream = static_cast<hipStream_t>(stream_);
  Shared &sh = *ctx->shared;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;  // my pieces are written
  sh.send[ctx->rank] = send;
  sh.sync.arrive_and_wait();
  for (int s = 0; s < sh.ranks; ++s) {
    if (hipMemcpyAsync(recv + static_cast<std::size_t>(s) * count,
                       sh.send[s] + static_cast<std::size_t>(ctx->rank) * count, count * sizeof(std::uint64_t),
                       hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return 1;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  sh.sync.arrive_and_wait();  // nobody reuses a send buffer before every peer has copied from it
  ++sh.calls;
  return 0;
}

struct Case {
  int ranks;
  unsigned log2n, r_log2, chunks;
  bool closed{false};  // iota input, closed-form check of sampled outputs (no oracle run)
};

using u128 = unsigned __int128;
constexpr std::uint64_t IOTA_START{UINT64_C(0x0123456789abcdef)};  // oracle.INPUT_I1_START

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
// X[k] of the transform of a[i] = IOTA_START + i, i < n = 2^log2n
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

bool run_rank(const Case &c, int rank, Shared &shared, const std::vector<std::uint64_t> &input,
              const std::vector<std::uint64_t> &want, std::string &why, std::string &plan_text) {
  // a rank that leaves (finished, or failed early) stops counting at the loopback's barrier: every
  // rank makes the same number of exchanges, so a failure cannot leave the others waiting for ever
  struct Leave {
    Shared &s;
    ~Leave() { s.sync.arrive_and_drop(); }
  } leave{shared};
  HIP_OK(hipSetDevice(0));
  const std::uint64_t n{std::uint64_t{1} << c.log2n}, R{std::uint64_t{1} << c.r_log2}, C{n / R};
  const std::uint64_t Cl{C / c.ranks}, local{n / c.ranks};
  sventt_plan *cols{}, *rows{};
  if (sventt_sharded_plan_create(P, G_ROOT, n, c.r_log2, rank, c.ranks, SVENTT_BOTH, &cols) ||
      sventt_sharded_rows_plan_create(P, G_ROOT, n, c.r_log2, rank, c.ranks, SVENTT_BOTH, &rows)) {
    why = std::string{"plan: "} + sventt_last_error();
    return false;
  }
  if (rank == 0) {
    plan_text = std::string{sventt_plan_describe(cols)} + " | all-to-all | " + sventt_plan_describe(rows) + " (" +
                std::to_string(sventt_plan_num_passes(rows, 0)) + " row-phase passes)";
  }
  // this rank's column block of the R x C input matrix
  std::vector<std::uint64_t> slab(local), out(local);
  for (std::uint64_t r{}; r < R; ++r) {
    if (c.closed) {
      for (std::uint64_t k{}; k < Cl; ++k) slab[r * Cl + k] = IOTA_START + r * C + rank * Cl + k;
    } else {
      std::memcpy(&slab[r * Cl], &input[r * C + rank * Cl], Cl * sizeof(std::uint64_t));
    }
  }
  std::uint64_t *d_src{}, *d_dst{}, *d_work{}, *d_recv{}, *d_back{};
  const std::size_t bytes{local * sizeof(std::uint64_t)};
  HIP_OK(hipMalloc(&d_src, bytes));
  HIP_OK(hipMalloc(&d_dst, bytes));
  HIP_OK(hipMalloc(&d_work, bytes));
  HIP_OK(hipMalloc(&d_recv, bytes));
  HIP_OK(hipMalloc(&d_back, bytes));
  HIP_OK(hipMemcpy(d_src, slab.data(), bytes, hipMemcpyHostToDevice));
  hipStream_t stream{};
  HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  // (fills go on the transform's own stream: it does not synchronise with the null stream, and a
  // null-stream hipMemset still running when the last pass writes its output would overwrite it)
  HIP_OK(hipMemsetAsync(d_dst, 0x55, bytes, stream));
  Loopback loop{&shared, rank};
  const sventt_transport transport{&loop, &loopback_all_to_all};

  bool ok{true};
  const int reps{c.closed && c.log2n >= 28 ? 1 : 2};
  for (int rep{}; rep < reps && ok; ++rep) {  // twice: the second call reuses the plan's stream and events
    if (sventt_sharded_forward_transport(cols, rows, &transport, d_dst, d_src, d_work, d_recv, c.chunks, stream)) {
      why = std::string{"forward: "} + sventt_last_error();
      return false;
    }
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out.data(), d_dst, bytes, hipMemcpyDeviceToHost));
    if (c.closed) {
      // outputs [rank*local, +local) of the bit-reversed result: both ends and a pseudo-random sample
      std::uint64_t x{0x9e3779b97f4a7c15ull * static_cast<std::uint64_t>(rank + 1)};
      for (int smp{}; smp < 2048 + 128 && ok; ++smp) {
        std::uint64_t j;
        if (smp < 64) {
          j = static_cast<std::uint64_t>(smp);
        } else if (smp < 128) {
          j = local - 1 - static_cast<std::uint64_t>(smp - 64);
        } else {
          x ^= x << 13, x ^= x >> 7, x ^= x << 17;  // xorshift64
          j = x % local;
        }
        const std::uint64_t k{bitrev(rank * local + j, c.log2n)};
        if (out[j] != closed_form(k, c.log2n)) {
          why = "forward differs from the closed form at local index " + std::to_string(j);
          ok = false;
        }
      }
    } else if (std::memcmp(out.data(), &want[rank * local], bytes) != 0) {
      why = "forward differs from the oracle's slice";
      ok = false;
    }
    HIP_OK(hipMemcpy(out.data(), d_src, bytes, hipMemcpyDeviceToHost));
    if (out != slab) {
      why = "forward modified its source";
      ok = false;
    }
    HIP_OK(hipMemsetAsync(d_back, 0x55, bytes, stream));
    if (sventt_sharded_inverse_transport(cols, rows, &transport, d_back, d_dst, d_work, d_recv, c.chunks, stream)) {
      why = std::string{"inverse: "} + sventt_last_error();
      return false;
    }
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out.data(), d_back, bytes, hipMemcpyDeviceToHost));
    if (out != slab) {
      why = "inverse(forward(x)) != x";
      ok = false;
    }
  }
  // argument checking on live plans
  ok &= sventt_sharded_forward_transport(cols, rows, &transport, d_dst, d_src, d_work, d_recv, 0, stream) ==
        SVENTT_ERR_INVALID_ARGUMENT;
  ok &= sventt_sharded_forward_transport(cols, rows, &transport, d_dst, d_dst, d_work, d_recv, 1, stream) ==
        SVENTT_ERR_INVALID_ARGUMENT;
  ok &= sventt_sharded_forward_transport(rows, cols, &transport, d_dst, d_src, d_work, d_recv, 1, stream) ==
        SVENTT_ERR_LOGIC;
  ok &= sventt_sharded_forward(cols, rows, nullptr, d_dst, d_src, d_work, d_recv, 1, stream) ==
        SVENTT_ERR_INVALID_ARGUMENT;
  if (!ok && why.empty()) why = "argument checks";
  (void)hipStreamDestroy(stream);
  for (std::uint64_t *p : {d_src, d_dst, d_work, d_recv, d_back}) (void)hipFree(p);
  sventt_plan_destroy(cols);
  sventt_plan_destroy(rows);
  return ok;
}

bool run_case(const Case &c) {
  const std::uint64_t n{std::uint64_t{1} << c.log2n};
  std::vector<std::uint64_t> input, want;
  if (!c.closed) {
    input.resize(n);
    want.resize(n);
    oracle_fill_splitmix(input.data(), n, 4242 + c.log2n, P);
    oracle_ntt_forward(want.data(), input.data(), n, P, G_ROOT);
  }
  std::string plan_text;
  Shared shared{c.ranks};
  std::vector<std::thread> threads;
  std::vector<int> ok(c.ranks, 0);
  std::vector<std::string> why(c.ranks);
  for (int r{}; r < c.ranks; ++r) {
    threads.emplace_back([&, r] { ok[r] = run_rank(c, r, shared, input, want, why[r], plan_text); });
  }
  for (auto &t : threads) t.join();
  bool all{true};
  for (int r{}; r < c.ranks; ++r) {
    if (!ok[r]) {
      std::printf("MISMATCH rank %d: %s\n", r, why[r].c_str());
      all = false;
    }
  }
  std::printf("%s sharded C driver: %d ranks, n=2^%u, R=2^%u, %u chunks (%d exchanges), %s check\n   plan: %s\n",
              all ? "ok" : "MISMATCH", c.ranks, c.log2n, c.r_log2, c.chunks, shared.calls.load(),
              c.closed ? "closed-form" : "oracle", plan_text.c_str());
  std::fflush(stdout);
  return all;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc > 1 && std::string{argv[1]} == "--compile-only-check") {
    std::printf("sharded driver harness compiled and linked\n");
    return 0;
  }
  bool ok{true};
  if (argc > 1 && std::string{argv[1]} == "--config5") {
    // BASELINE configs[4] in its own shape: N = 2^30, 8 ranks, R = 2^11, exchange pipelined in 4 chunks
    ok = run_case(Case{8, 30, 11, 4, true});
    std::printf(ok ? "ALL OK\n" : "FAILED\n");
    return ok ? 0 : 1;
  }
  // chunks must divide the tile counts next to the exchange (sventt_plan_pass_tiles_per_block)
  for (const Case &c : {Case{2, 20, 8, 1}, Case{2, 24, 10, 4}, Case{4, 24, 10, 2}, Case{4, 20, 6, 4},
                        Case{2, 25, 11, 4},
                        Case{8, 22, 3, 4},          // the row phase of config #5 (C = 2^19: col 2^6 two-level | row 2^13)
                        Case{8, 27, 11, 4, true},   // 2^24 per rank on 8 ranks, closed form
                        Case{4, 26, 12, 2, true}}) {  // R = 2^12
    ok &= run_case(c);
  }
  std::printf(ok ? "ALL OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
