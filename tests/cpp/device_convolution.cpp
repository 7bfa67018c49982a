// tests/cpp/device_convolution.cpp -- C++ host code on DEVICE pointers through the facade.
//
// The shape of the reference's benchmark harness (tests/bench-ntt.cpp:17-65: build the
// NTT object, fill src with start + i, run, compare with NTTReference, time the calls)
// and of its only caller (examples/magic-series/gaussian-polynomial.hpp:176-241: spectrum
// to Montgomery form once, forward -> product -> inverse), with the buffers resident in
// HBM: hipMalloc'd memory goes straight into NTT::compute_forward / compute_inverse and
// the device-side extensions (to_montgomery, compute_forward_multiply).
//
//   1. N = 2^16: cyclic convolution checked element by element against the oracle
//      (oracle forward -> oracle_modmul -> oracle inverse);
//   2. N = 2^24 (the BASELINE configuration): forward checked against the oracle's
//      digest-free closed forms, then forward and the fused convolution timed with HIP
//      events -- the C++ counterpart of bench.py's number.
//
// TEST/BENCH ONLY: links the oracle as the checker.  Build: see tests/test_cpp_facade.py.
#include <sventt/sventt.hpp>

#include <hip/hip_runtime.h>

#include <bit>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../oracle/ntt_oracle.h"

using namespace sventt;

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    const hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) throw std::runtime_error{std::string{#expr} + ": " + hipGetErrorString(e_)}; \
  } while (0)

template <class T> class DeviceBuffer {
  T *p{};

public:
  explicit DeviceBuffer(const std::uint64_t n) { HIP_OK(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T))); }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  ~DeviceBuffer() { (void)hipFree(p); }
  T *data(void) const { return p; }
  void upload(const std::vector<T> &v) { HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> download(const std::uint64_t n) const {
    std::vector<T> v(n);
    HIP_OK(hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return v;
  }
};

using modulus_type = Modulus<UINT64_C(0xfffffc6e80000001), 3>;
using modmul_type = PAdic64SVE<modulus_type>;
constexpr std::uint64_t N{modulus_type::get_modulus()}, g{modulus_type::get_generator()};

// A kernel type that is valid by the reference's own rules (kernel/iterative.hpp:24-27):
// log2(m) radix-2 layers of span m, m/2, ..., 2; the last one carries inverse_factor = m so that
// the inverse divides by m (layer/sve/radix-two.hpp:208-235 of the reference).  Modulus, length
// and that factor reach the planner.
template <std::uint64_t m, std::size_t... I>
auto radix_two_chain(std::index_sequence<I...>)
    -> IterativeNTT<modulus_type, m,
                    RadixTwoSVELayer<modmul_type, m, (m >> I), ((m >> I) == 2 ? m : 1)>...>;
template <std::uint64_t m>
using iterative = decltype(radix_two_chain<m>(std::make_index_sequence<std::bit_width(m) - 1>{}));
static_assert(iterative<1024>::get_m() == 1024);
static_assert(iterative<1024>::get_inverse_factor() == 1024);

template <std::uint64_t m> static bool convolution_matches_oracle(void) {
  std::vector<std::uint64_t> a(m), b(m), fa(m), fb(m), prod(m), want(m);
  oracle_fill_splitmix(a.data(), m, 1, N);
  oracle_fill_splitmix(b.data(), m, 2, N);
  oracle_ntt_forward(fa.data(), a.data(), m, N, g);
  oracle_ntt_forward(fb.data(), b.data(), m, N, g);
  for (std::uint64_t i{}; i < m; ++i) {
    prod[i] = oracle_modmul(fa[i], fb[i], N);
  }
  oracle_ntt_inverse(want.data(), prod.data(), m, N, g);

  const NTT<iterative<m>> ntt;
  DeviceBuffer<std::uint64_t> da{m}, db{m}, dc{m};
  da.upload(a);
  db.upload(b);
  ntt.compute_forward(db.data());              // spectrum of b ...
  ntt.to_montgomery(db.data(), db.data());     // ... in Montgomery form, once
  ntt.compute_forward_multiply(dc.data(), da.data(), db.data());
  ntt.compute_inverse(dc.data());
  const bool fused_ok{dc.download(m) == want};

  // the same through the three separate operations
  db.upload(b);
  ntt.compute_forward(db.data());
  ntt.compute_forward(dc.data(), da.data());
  ntt.pointwise_multiply(dc.data(), dc.data(), db.data());
  ntt.compute_inverse(dc.data());
  const bool plain_ok{dc.download(m) == want};
  std::printf("%s cyclic convolution N=%llu on device pointers (fused %d, three-step %d)\n",
              fused_ok && plain_ok ? "ok" : "MISMATCH", static_cast<unsigned long long>(m), fused_ok, plain_ok);
  return fused_ok && plain_ok;
}

static double time_us(const int reps, const auto &fn) {
  hipEvent_t t0, t1;
  HIP_OK(hipEventCreate(&t0));
  HIP_OK(hipEventCreate(&t1));
  for (int i{}; i < 600; ++i) {  // steady-state clocks first (tools/clock_ramp.py)
    fn();
  }
  HIP_OK(hipEventRecord(t0, nullptr));
  for (int i{}; i < reps; ++i) {
    fn();
  }
  HIP_OK(hipEventRecord(t1, nullptr));
  HIP_OK(hipEventSynchronize(t1));
  float ms{};
  HIP_OK(hipEventElapsedTime(&ms, t0, t1));
  HIP_OK(hipEventDestroy(t0));
  HIP_OK(hipEventDestroy(t1));
  return ms * 1e3 / reps;
}

static bool baseline_size(void) {
  constexpr std::uint64_t m{std::uint64_t{1} << 24};
  const std::uint64_t start{UINT64_C(0x0123456789abcdef)};
  std::vector<std::uint64_t> src(m);
  oracle_fill_iota(src.data(), m, start);
  const NTT<iterative<m>> ntt;
  DeviceBuffer<std::uint64_t> dsrc{m}, ddst{m}, dspec{m};
  dsrc.upload(src);
  ntt.compute_forward(ddst.data(), dsrc.data());
  const std::vector<std::uint64_t> out{ddst.download(m)};
  // closed forms of tests/test-ntt-reference.cpp:45-63 for the iota input
  const unsigned __int128 sum{static_cast<unsigned __int128>(m) * start +
                              static_cast<unsigned __int128>(m) * (m - 1) / 2};
  bool ok{out[0] == static_cast<std::uint64_t>(sum % N)};
  ok &= out[1] == N - m / 2;  // sum of (-1)^i (start + i) = -m/2
  ntt.compute_inverse(ddst.data());
  ok &= ddst.download(m) == src;
  std::printf("%s forward/inverse N=2^24 on device pointers [%s]\n", ok ? "ok" : "MISMATCH",
              ntt.describe().c_str());

  ntt.compute_forward(dspec.data(), dsrc.data());
  ntt.to_montgomery(dspec.data(), dspec.data());
  const double fwd{time_us(300, [&] { ntt.compute_forward(ddst.data(), dsrc.data()); })};
  const double conv{time_us(150, [&] {
    ntt.compute_forward_multiply(ddst.data(), dsrc.data(), dspec.data());
    ntt.compute_inverse(ddst.data());
  })};
  std::printf("timing N=2^24: forward %.1f us (%.3e elements/s); forward*multiply + inverse %.1f us\n", fwd,
              m / (fwd * 1e-6), conv);
  return ok;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string{argv[1]} == "--compile-only-check") {
    std::printf("device-pointer harness compiled and linked\n");
    return 0;
  }
  try {
    bool ok{convolution_matches_oracle<std::uint64_t{1} << 16>()};
    ok &= convolution_matches_oracle<std::uint64_t{1} << 10>();
    ok &= baseline_size();
    std::printf(ok ? "ALL OK\n" : "FAILED\n");
    return ok ? 0 : 1;
  } catch (const std::exception &e) {
    std::printf("FAILED: %s\n", e.what());
    return 1;
  }
}
