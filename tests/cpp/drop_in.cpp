// tests/cpp/drop_in.cpp -- "switch the include path and relink" check for the C++ facade.
//
// User code written against the reference's template API (kernel_type spelled with
// Modulus / PAdic64SVE / Radix*SVELayer / IterativeNTT / RecursiveNTT /
// [Blocked]GenericSVELayer / Transpose..., run through sventt::NTT<kernel_type>)
// is compiled here against include/sventt/ of THIS repository and linked with
// libsventt_hip.so.  The harness follows the reference's tests/bench-ntt.cpp:17-65:
// src[i] = start + i, dst pre-filled with 0x55.., transform out of place, compare
// every element with the scalar oracle -- here with exact equality, since the
// engine returns canonical residues.
//
// The kernel_type definitions below have the shapes of the reference's own
// configurations: README.md:13-82 (blocked six-step 2^17 = 2^8 x 2^9, 64-bit
// prime) and tests/ntt-tests/{iterative-sve-radix8-two12, recursive-sve-radix248-two13,
// recursive-sve-fourstep-two13}.hpp (62-bit prime).
//
//   g++ -std=c++20 -Iinclude tests/cpp/drop_in.cpp -Lsve_ntt_amd -lsventt_hip
//       -Loracle -lntt_oracle -Wl,-rpath,... -o drop_in && ./drop_in
//   ./drop_in --compile-only-check   exits 0 without touching the GPU
#include <sventt/sventt.hpp>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../oracle/ntt_oracle.h"

using namespace sventt;

namespace readme_blocked_six_step {
using modulus_type = Modulus<UINT64_C(0xfffffc6e80000001), 3>;
using modmul_type = PAdic64SVE<modulus_type>;
using transposition_type =
    TransposeParallelSVEInRegisterExplicitBlockingRowFirst<32, 128, 128 + 32, 3>;
constexpr std::uint64_t n{std::uint64_t{1} << 17}, n0{std::uint64_t{1} << 8},
    n1{std::uint64_t{1} << 9};
using ntt0_type = IterativeNTT<modulus_type, n0, RadixEightSVELayer<modmul_type, n0, n0>,
                               RadixEightSVELayer<modmul_type, n0, (n0 >> 3)>,
                               RadixFourSVELayer<modmul_type, n0, (n0 >> 6)>>;
using ntt1_type =
    RecursiveNTT<modulus_type, n1, RadixEightSVELayer<modmul_type, n1, n1>,
                 IterativeNTT<modulus_type, n1, RadixEightSVELayer<modmul_type, n1, (n1 >> 3)>,
                              RadixEightSVELayer<modmul_type, n1, (n1 >> 6)>>,
                 false>;
using kernel_type =
    RecursiveNTT<modulus_type, n,
                 BlockedGenericSVELayer<modmul_type, n, ntt0_type, 32, 2, 128, transposition_type>,
                 ntt1_type, true>;
} // namespace readme_blocked_six_step

namespace test62 {
using modulus_type = Modulus<UINT64_C(0x3a00000000000001), 3>;
using modmul_type = PAdic64SVE<modulus_type>;

constexpr std::uint64_t m12{std::uint64_t{1} << 12};
using iterative_radix8 =
    IterativeNTT<modulus_type, m12, RadixEightSVELayer<modmul_type, m12, (m12 >> 0), 1, false>,
                 RadixEightSVELayer<modmul_type, m12, (m12 >> 3), 1, true>,
                 RadixEightSVELayer<modmul_type, m12, (m12 >> 6), 1, false>,
                 RadixEightSVELayer<modmul_type, m12, (m12 >> 9), m12, true>>;

constexpr std::uint64_t m13{std::uint64_t{1} << 13}, m10{std::uint64_t{1} << 10},
    m9{std::uint64_t{1} << 9};
using inner9 = IterativeNTT<modulus_type, m9, RadixEightSVELayer<modmul_type, m9, m9, 1, true>,
                            RadixFourSVELayer<modmul_type, m9, (m9 >> 3), 1, true>,
                            RadixTwoSVELayer<modmul_type, m9, (m9 >> 5), 1, false>,
                            RadixEightSVELayer<modmul_type, m9, (m9 >> 6), m13, false>>;
using inner10 =
    RecursiveNTT<modulus_type, m10, RadixTwoSVELayer<modmul_type, m10, m10, 1, false>, inner9, false>;
using recursive_radix248 =
    RecursiveNTT<modulus_type, m13, RadixEightSVELayer<modmul_type, m13, m13, 1, true>, inner10, false>;

constexpr std::uint64_t m15{std::uint64_t{1} << 15}, m6{std::uint64_t{1} << 6};
using column9 = IterativeNTT<modulus_type, m9, RadixEightSVELayer<modmul_type, m9, m9>,
                             RadixFourSVELayer<modmul_type, m9, (m9 >> 3)>,
                             RadixTwoSVELayer<modmul_type, m9, (m9 >> 5)>,
                             RadixEightSVELayer<modmul_type, m9, (m9 >> 6), m15>>;
using row6 = IterativeNTT<modulus_type, m6, RadixEightSVELayer<modmul_type, m6, m6>,
                          RadixEightSVELayer<modmul_type, m6, (m6 >> 3)>>;
using four_step =
    RecursiveNTT<modulus_type, m15,
                 GenericSVELayer<modmul_type, m15, column9, 8, 2, TransposeParallelSVEInRegister<8, 64>>,
                 row6, true>;
} // namespace test62

namespace fixed_point {
// tests/ntt-tests/iterative-scalar-radix8-two12.hpp of the reference: PAdic64 and FixedPoint64
// layers alternate; one FixedPoint64 layer selects the Shoup kernels for the plan
using modulus_type = Modulus<UINT64_C(0x3a00000000000001), 3>;
constexpr std::uint64_t m{std::uint64_t{1} << 12};
using mixed_radix8 =
    IterativeNTT<modulus_type, m, RadixEightScalarLayer<PAdic64Scalar<modulus_type>, m, (m >> 0)>,
                 RadixEightScalarLayer<FixedPoint64Scalar<modulus_type>, m, (m >> 3)>,
                 RadixEightScalarLayer<PAdic64Scalar<modulus_type>, m, (m >> 6)>,
                 RadixEightScalarLayer<FixedPoint64Scalar<modulus_type>, m, (m >> 9), m>>;
// a whole kernel on FixedPoint64SVE layers, six-step shaped, 2^20 points
constexpr std::uint64_t n{std::uint64_t{1} << 20}, r{std::uint64_t{1} << 9}, c{std::uint64_t{1} << 11};
using fp = FixedPoint64SVE<modulus_type>;
using col = IterativeNTT<modulus_type, r, RadixEightSVELayer<fp, r, r>, RadixEightSVELayer<fp, r, (r >> 3)>,
                         RadixEightSVELayer<fp, r, (r >> 6)>>;
using row = IterativeNTT<modulus_type, c, RadixEightSVELayer<fp, c, c>, RadixEightSVELayer<fp, c, (c >> 3)>,
                         RadixEightSVELayer<fp, c, (c >> 6)>, RadixFourSVELayer<fp, c, (c >> 9), n>>;
using six_step = RecursiveNTT<modulus_type, n,
                              GenericSVELayer<fp, n, col, 8, 2, TransposeParallelSVEInRegister<8, 8>>, row, true>;
static_assert(mixed_radix8::uses_fixed_point() && six_step::uses_fixed_point());
static_assert(!test62::iterative_radix8::uses_fixed_point());
static_assert(FixedPoint64SVE<modulus_type>::to_montgomery(12345) == 12345);
} // namespace fixed_point

namespace big {
// BASELINE config #3 spelled as a six-step 2^24 = 2^11 x 2^13
using modulus_type = Modulus<UINT64_C(0xfffffc6e80000001), 3>;
using modmul_type = PAdic64SVE<modulus_type>;
constexpr std::uint64_t n{std::uint64_t{1} << 24}, r{std::uint64_t{1} << 11}, c{std::uint64_t{1} << 13};
using col = IterativeNTT<modulus_type, r, RadixEightSVELayer<modmul_type, r, r>,
                         RadixEightSVELayer<modmul_type, r, (r >> 3)>,
                         RadixEightSVELayer<modmul_type, r, (r >> 6)>,
                         RadixFourSVELayer<modmul_type, r, (r >> 9)>>;
using row = IterativeNTT<modulus_type, c, RadixEightSVELayer<modmul_type, c, c>,
                         RadixEightSVELayer<modmul_type, c, (c >> 3)>,
                         RadixEightSVELayer<modmul_type, c, (c >> 6)>,
                         RadixEightSVELayer<modmul_type, c, (c >> 9)>,
                         RadixTwoSVELayer<modmul_type, c, (c >> 12)>>;
using kernel_type =
    RecursiveNTT<modulus_type, n,
                 BlockedGenericSVELayer<modmul_type, n, col, 32, 2, 128,
                                        TransposeParallelSVEInRegister<8, 8>>,
                 row, true>;
} // namespace big

template <class kernel_type, bool is_inverse> static bool check(const char *name) {
  using ntt_type = NTT<kernel_type>;
  using modulus_type = typename ntt_type::modulus_type;
  constexpr std::uint64_t N{modulus_type::get_modulus()}, g{modulus_type::get_generator()};
  const std::uint64_t m{ntt_type::get_m()};

  PageMemory<std::uint64_t> buffer{m * 3, false};
  std::uint64_t *const src{&buffer[0]}, *const dst{&buffer[m]}, *const ref{&buffer[m * 2]};
  oracle_fill_iota(src, m, UINT64_C(0x0123456789abcdef) % (N - m));
  std::memset(dst, 0x55, sizeof(std::uint64_t) * m);
  std::memset(ref, 0xaa, sizeof(std::uint64_t) * m);
  if (is_inverse) {
    oracle_ntt_inverse(ref, src, m, N, g);
  } else {
    oracle_ntt_forward(ref, src, m, N, g);
  }

  // The oracle's inverse divides by m (tests/ntt-reference.hpp:78-82); a kernel_type divides by
  // the product of its layers' inverse_factor arguments (layer/sve/radix-two.hpp:208-235 of the
  // reference): README-shaped kernels by 1 (unscaled), `..., m>`-terminated ones by m.
  constexpr std::uint64_t divisor{kernel_type::get_inverse_factor()};
  const std::uint64_t rescale{modulus_type::multiply(m % N, modulus_type::invert(divisor))};
  if (is_inverse) {
    for (std::uint64_t i{}; i < m; ++i) {
      ref[i] = modulus_type::multiply(ref[i], rescale);
    }
  }

  const ntt_type ntt{!is_inverse, is_inverse, false};
  if (is_inverse) {
    ntt.compute_inverse(dst, src);
  } else {
    ntt.compute_forward(dst, src);
  }
  for (std::uint64_t i{}; i < m; ++i) {
    if (dst[i] != ref[i]) {
      std::printf("MISMATCH %s %s at %llu\n", is_inverse ? "Inverse," : "Forward,", name,
                  static_cast<unsigned long long>(i));
      return false;
    }
  }

  // the in-place overloads and the round trip
  const ntt_type both;
  std::memcpy(dst, src, sizeof(std::uint64_t) * m);
  both.compute_forward(dst);
  both.compute_inverse(dst);
  for (std::uint64_t i{}; i < m; ++i) {
    if (dst[i] != modulus_type::multiply(src[i], rescale)) {
      std::printf("MISMATCH round trip %s at %llu\n", name, static_cast<unsigned long long>(i));
      return false;
    }
  }
  std::printf("ok %s %s  [%s] (inverse divides by %llu)\n", is_inverse ? "Inverse," : "Forward,", name,
              ntt.describe().c_str(), static_cast<unsigned long long>(divisor));
  return true;
}

static bool check_errors(void) {
  using modulus_type = Modulus<UINT64_C(0xfffffc6e80000001), 3>;
  using modmul_type = PAdic64SVE<modulus_type>;
  using k = IterativeNTT<modulus_type, 4, RadixFourSVELayer<modmul_type, 4, 4>>;
  const NTT<k> forward_only{true, false, false};
  std::uint64_t v[4]{1, 2, 3, 4};
  try {
    forward_only.compute_inverse(v);
  } catch (const std::logic_error &) {
    try {
      (void)modulus_type::get_root_forward(7);  // 7 does not divide p-1
    } catch (const std::invalid_argument &) {
      std::printf("ok error mapping\n");
      return true;
    }
  }
  std::printf("MISMATCH error mapping\n");
  return false;
}

// the transposition classes keep their static entry points (bench-transpose.cpp:17-103 of the
// reference checks itself the same way: 2-D iota in, transpose, transpose back)
static bool check_transposition(void) {
  constexpr std::uint64_t rows{256}, cols{1024}, pad_src{32}, pad_dst{8};
  std::vector<std::uint64_t> src((cols + pad_src) * rows - pad_src),
      dst((rows + pad_dst) * cols - pad_dst, UINT64_C(0x5555555555555555)), back(src.size());
  for (std::uint64_t r{}; r < rows; ++r) {
    for (std::uint64_t c{}; c < cols + pad_src && (cols + pad_src) * r + c < src.size(); ++c) {
      src[(cols + pad_src) * r + c] = c < cols ? UINT64_C(0x0123456789abcdef) + r * cols + c : 0;
    }
  }
  using out_of_place = TransposeParallelSVEInRegisterExplicitBlockingRowFirst<32, 128, 128 + 32, 3>;
  out_of_place::transpose(dst.data(), src.data(), rows, cols, rows + pad_dst, cols + pad_src);
  bool ok{true};
  for (std::uint64_t r{}; r < rows; r += 37) {
    for (std::uint64_t c{}; c < cols; c += 41) {
      ok &= dst[(rows + pad_dst) * c + r] == src[(cols + pad_src) * r + c];
    }
  }
  ok &= dst[rows] == UINT64_C(0x5555555555555555);  // padding untouched
  TransposeParallelSVEInRegister<32, 32>::transpose(back.data(), dst.data(), cols, rows,
                                                    cols + pad_src, rows + pad_dst);
  for (std::uint64_t r{}; r < rows; ++r) {
    for (std::uint64_t c{}; c < cols; ++c) {
      ok &= back[(cols + pad_src) * r + c] == src[(cols + pad_src) * r + c];
    }
  }
  std::vector<std::uint64_t> square(512 * 512);
  for (std::uint64_t i{}; i < square.size(); ++i) {
    square[i] = i;
  }
  TransposeParallelSVEInRegisterRowFirst<64, 64, 3>::transpose(square.data(), 512);
  for (std::uint64_t i{}; i < square.size(); i += 97) {
    ok &= square[i] == (i % 512) * 512 + i / 512;
  }
  try {
    TransposeParallelSVEInRegister<32, 32>::transpose(back.data(), dst.data(), 48, 64, 64, 48);
    ok = false;
  } catch (const std::invalid_argument &) {  // in-register.hpp:121-124 of the reference
  }
  std::printf(ok ? "ok transposition classes\n" : "MISMATCH transposition classes\n");
  return ok;
}

// Driving a kernel_type directly, the way NTT<> of the reference does (wrapper.hpp:13-47,
// 50-82): size the auxiliary blob with a FakeByteVector, fill an AuxiliaryVector, run with a
// cursor that must end exactly at the end of what prepare_* wrote.
static bool check_kernel_concept(void) {
  using kernel_type = test62::recursive_radix248;
  using modulus_type = kernel_type::modulus_type;
  constexpr std::uint64_t N{modulus_type::get_modulus()}, g{modulus_type::get_generator()};
  constexpr std::uint64_t m{kernel_type::get_m()};
  FakeByteVector sizing;
  kernel_type::prepare_forward(sizing);
  const std::uint64_t forward_bytes{sizing.size()};
  kernel_type::prepare_inverse(sizing);
  AuxiliaryVector aux{sizing.size(), false};
  kernel_type::prepare_forward(aux);
  bool ok{aux.size() == forward_bytes};
  kernel_type::prepare_inverse(aux);
  ok &= aux.size() == sizing.size() && aux.size() > 0;

  std::vector<std::uint64_t> src(m), dst(m, UINT64_C(0x5555555555555555)), ref(m);
  oracle_fill_iota(src.data(), m, 12345);
  oracle_ntt_forward(ref.data(), src.data(), m, N, g);
  const std::byte *cursor{aux.data()};
  kernel_type::compute_forward(dst.data(), src.data(), cursor);
  ok &= cursor == aux.data() + forward_bytes && dst == ref;
  kernel_type::compute_inverse(dst.data(), cursor);  // the in-place overload, inverse record
  ok &= cursor == aux.data() + aux.size() && dst == src;

  AuxiliaryVector too_small{forward_bytes - 1, false};
  try {
    kernel_type::prepare_forward(too_small);
    ok = false;
  } catch (const std::bad_alloc &) {  // vector.hpp:245-247 of the reference
  }
  std::printf(ok ? "ok kernel concept (prepare / compute with an auxiliary cursor)\n"
                 : "MISMATCH kernel concept\n");
  return ok;
}

// compile-time facts the reference's API promises
static_assert(readme_blocked_six_step::kernel_type::get_m() == (std::uint64_t{1} << 17));
static_assert(NTT<readme_blocked_six_step::kernel_type>::get_m() == (std::uint64_t{1} << 17));
static_assert(readme_blocked_six_step::ntt0_type::get_m() == 256);
static_assert(Modulus<UINT64_C(0xfffffc6e80000001), 3>::get_montgomery_inverse() ==
              UINT64_C(0x4000039180000001));
static_assert(Modulus<UINT64_C(0xfffffc6e80000001), 3>::get_root_forward(std::uint64_t{1} << 24) ==
              UINT64_C(0x3a215e9b536c5fbc));
static_assert(Modulus<UINT64_C(0xffffffff00000001), 7>::multiply(
                  Modulus<UINT64_C(0xffffffff00000001), 7>::get_root_forward(257),
                  Modulus<UINT64_C(0xffffffff00000001), 7>::get_root_inverse(257)) == 1);
static_assert(PAdic64SVE<Modulus<UINT64_C(0xfffffc6e80000001), 3>>::to_montgomery(1) ==
              UINT64_C(0x000003917fffffff));
static_assert(bitreverse(1) == (std::uint64_t{1} << 63));
static_assert(readme_blocked_six_step::kernel_type::get_inverse_factor() == 1);  // README: unscaled
static_assert(test62::iterative_radix8::get_inverse_factor() == test62::m12);
static_assert(test62::recursive_radix248::get_inverse_factor() == test62::m13);
static_assert(test62::four_step::get_inverse_factor() == test62::m15);

int main(int argc, char **argv) {
  if (argc > 1 && std::string{argv[1]} == "--compile-only-check") {
    std::printf("facade compiled and linked\n");
    return 0;
  }
  bool ok{true};
  ok &= check<readme_blocked_six_step::kernel_type, false>("README blocked six-step 2^17");
  ok &= check<readme_blocked_six_step::kernel_type, true>("README blocked six-step 2^17");
  ok &= check<test62::iterative_radix8, false>("iterative, SVE, radix-8");
  ok &= check<test62::iterative_radix8, true>("iterative, SVE, radix-8");
  ok &= check<test62::recursive_radix248, false>("recursive, SVE, radix-2,4,8");
  ok &= check<test62::recursive_radix248, true>("recursive, SVE, radix-2,4,8");
  ok &= check<test62::four_step, false>("recursive, SVE, four-step");
  ok &= check<test62::four_step, true>("recursive, SVE, four-step");
  ok &= check<fixed_point::mixed_radix8, false>("iterative, scalar, radix-8, PAdic64/FixedPoint64 layers");
  ok &= check<fixed_point::mixed_radix8, true>("iterative, scalar, radix-8, PAdic64/FixedPoint64 layers");
  ok &= check<fixed_point::six_step, false>("six-step 2^20 on FixedPoint64SVE layers");
  ok &= check<fixed_point::six_step, true>("six-step 2^20 on FixedPoint64SVE layers");
  ok &= NTT<fixed_point::six_step>{}.describe().rfind("[fixed-point]", 0) == 0;
  ok &= NTT<test62::four_step>{}.describe().rfind("[fixed-point]", 0) != 0;
  ok &= check<big::kernel_type, false>("six-step 2^24 = 2^11 x 2^13");
  ok &= check<big::kernel_type, true>("six-step 2^24 = 2^11 x 2^13");
  ok &= check_errors();
  ok &= check_transposition();
  ok &= check_kernel_concept();
  std::printf(ok ? "ALL OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
