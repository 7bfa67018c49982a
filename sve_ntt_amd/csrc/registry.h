// sve_ntt_amd/csrc/registry.h -- which TileNTT instantiation serves which pass.
//
// Parameterised by the launcher so that the same table drives the HIP kernels
// (kernels.hip) and the sequential emulation used by the CPU-side tests
// (tests/cpu_sim/sim.cpp; test-only, never linked into the product).
#pragma once

#include <cstdlib>

#include "tile_ntt.h"

namespace sventt_hip {

enum : int { KIND_ROW = 0, KIND_COL = 1 };

constexpr int MAX_ROW_LOGL = 13;  // longest transform one workgroup keeps on chip (64 KiB of LDS)
constexpr int MAX_COL_LOGL = 11;  // longest strided column of the wide (T = 8) tiles: 128 KiB of LDS.  The slim
                                  // (T = 4) tiles go to 2^12 (plan_core.h: Tuning::max_col_logl)

// Step lists: radix-16 steps from the top stage down (E = 16 elements per
// thread), remainder last.
template <int LOGL> struct DefaultSteps;
template <> struct DefaultSteps<1> { using type = Steps<1>; };
template <> struct DefaultSteps<2> { using type = Steps<2>; };
template <> struct DefaultSteps<3> { using type = Steps<3>; };
template <> struct DefaultSteps<4> { using type = Steps<4>; };
template <> struct DefaultSteps<5> { using type = Steps<4, 1>; };
template <> struct DefaultSteps<6> { using type = Steps<4, 2>; };
template <> struct DefaultSteps<7> { using type = Steps<4, 3>; };
template <> struct DefaultSteps<8> { using type = Steps<4, 4>; };
template <> struct DefaultSteps<9> { using type = Steps<4, 4, 1>; };
#if defined(SVENTT_COL10_STEPS)  // A/B builds
template <> struct DefaultSteps<10> { using type = Steps<SVENTT_COL10_STEPS>; };
#else
template <> struct DefaultSteps<10> { using type = Steps<3, 4, 3>; };  // r03: 30.6 against <4,4,2> 31.5 us in the 2^22 plan
#endif
// 2^11 (the column tile of the 2^24 plan): radix-8 in the middle.  r03 A/B over eight alternating rounds and
// a duplicate baseline (profiles/r03/asm_stages_ab.txt (5)): <4,3,4> 99.7-103.0 us against <4,4,3> 103.6-106.0
// forward, 104.5-106.0 against 106.7-107.2 inverse.  -DSVENTT_COL11_STEPS=4,4,3 rebuilds r02's order.
#if defined(SVENTT_COL11_STEPS)
template <> struct DefaultSteps<11> { using type = Steps<SVENTT_COL11_STEPS>; };
#else
template <> struct DefaultSteps<11> { using type = Steps<4, 3, 4>; };
#endif
#if defined(SVENTT_COL12_STEPS)  // A/B builds (also the default of 2^12 rows)
template <> struct DefaultSteps<12> { using type = Steps<SVENTT_COL12_STEPS>; };
#else
template <> struct DefaultSteps<12> { using type = Steps<4, 4, 4>; };
#endif
template <> struct DefaultSteps<13> { using type = Steps<4, 4, 4, 1>; };

// ROW tiles end on a short step where that helps: a last step of k stages leaves each thread
// 2^k consecutive elements (8 * 2^k bytes), and a wave's store then touches 64 separate runs;
// k <= 2 keeps the stores in 16..32-byte lane pieces of one contiguous kilobyte.  Measured on
// batches (2^24 elements): 2^8 as (4,4) 94 us, see tools/batched_rows.py.
template <int LOGL> struct RowSteps { using type = typename DefaultSteps<LOGL>::type; };
template <> struct RowSteps<3> { using type = Steps<2, 1>; };
template <> struct RowSteps<4> { using type = Steps<3, 1>; };
template <> struct RowSteps<7> { using type = Steps<4, 2, 1>; };
template <> struct RowSteps<8> { using type = Steps<4, 3, 1>; };
template <> struct RowSteps<11> { using type = Steps<4, 4, 2, 1>; };
// 2^13 rows end on a radix-4 step (32 contiguous bytes per lane) since r03: <4,4,3,2> 92.4-94.0 us against
// <4,4,4,1> 94.2-95.0 forward, inverse level (profiles/r03/asm_stages_ab.txt (5)); the table prefix the lower
// steps read from LDS stays 511 entries.  (<3,4,4,2> is as fast but doubles that prefix; <4,4,2,3> is 3 %
// slower forward and 2 % faster inverse.)
#if SVENTT_PAIR
// first step radix-8 on two neighbouring elements (16-byte loads), last step radix-4 (32 bytes per lane)
template <> struct RowSteps<13> { using type = Steps<3, 4, 4, 2>; };
#elif defined(SVENTT_ROW13_STEPS)  // A/B builds, e.g. -DSVENTT_ROW13_STEPS=4,4,4,1 (r02)
template <> struct RowSteps<13> { using type = Steps<SVENTT_ROW13_STEPS>; };
#else
template <> struct RowSteps<13> { using type = Steps<4, 4, 3, 2>; };
#endif
// 2^12 rows (BASELINE config #4, and the row pass of 2^22, 2^23 and the large plans): two radix-8 steps in the
// middle and a radix-4 end: <4,3,3,2> 324.0 us against <4,4,4> 335.2 for 2^14 transforms forward (-3.4 %), inverse
// level (profiles/r03/asm_stages_ab.txt (5)).
#if defined(SVENTT_ROW12_STEPS)  // A/B builds
template <> struct RowSteps<12> { using type = Steps<SVENTT_ROW12_STEPS>; };
#else
template <> struct RowSteps<12> { using type = Steps<4, 3, 3, 2>; };
#endif

// The two directions are separate kernels and need not cut their stages alike: the inverse 2^13 row tile runs
// <4,4,2,3> (it starts, from HBM, with the radix-8 step and ends on the radix-16 one): 86.9-89.9 us against
// 88.3-92.3 with the forward order (profiles/r03/asm_stages_ab.txt (5)).  -DSVENTT_ROW13_INV_STEPS=... for A/B.
template <int LOGL, int MODE> struct RowStepsDir { using type = typename RowSteps<LOGL>::type; };
#if defined(SVENTT_ROW13_INV_STEPS)
template <> struct RowStepsDir<13, 1> { using type = Steps<SVENTT_ROW13_INV_STEPS>; };
#elif !SVENTT_PAIR && !defined(SVENTT_ROW13_STEPS)
template <> struct RowStepsDir<13, 1> { using type = Steps<4, 4, 2, 3>; };
#endif

constexpr int REG_LOGE = 4;
// ROW tiles are 2^12 elements (256 threads) unless the row itself is longer.
constexpr int row_logt(int logl) { return logl > 12 ? logl : 12; }
// Wide COL tiles hold T = 8 adjacent columns (64-byte HBM segments), more for short columns so
// that the tile still has 2^12 elements.  Columns of 2^10..2^12 default to the slim T = 4 tiles
// below (plan_core.h: Tuning::col_slim); the wide ones remain for SVENTT_COL_SLIM=0.
constexpr int col_f0(int logl) { return logl >= 9 ? 3 : 12 - logl; }

template <int LOGL, int MODE, bool FLAG, int ARITH = ARITH_MONT>
using RowTile =
    TileNTT<row_logt(LOGL), 0, LOGL, REG_LOGE, MODE, FLAG, typename RowStepsDir<LOGL, MODE>::type, ARITH>;
// The step order of a column tile may depend on the arithmetic back end: the FixedPoint64 kernels (two words per
// twiddle, other register pressure) run the 2^11 tile 2 % faster with r02's <4,4,3> (198.9-200.5 against
// 203.1-204.8 us at N = 2^24, tools/ab_shoup.sh).
template <int LOGL, int ARITH> struct ColSteps { using type = typename DefaultSteps<LOGL>::type; };
#if !defined(SVENTT_COL11_STEPS)
template <> struct ColSteps<11, ARITH_SHOUP> { using type = Steps<4, 4, 3>; };
#endif
template <int LOGL, int MODE, int ARITH = ARITH_MONT, bool TWOLVL = false>
using ColTile = TileNTT<LOGL + col_f0(LOGL), col_f0(LOGL), LOGL, REG_LOGE, MODE, true,
                        typename ColSteps<LOGL, ARITH>::type, ARITH, TWOLVL>;

// Narrow COL tiles (T = 8 whatever the column length) for blocks with fewer
// columns than the wide tile wants; tiny tiles, only met at small n.
constexpr int NARROW_F0 = 3;
template <int LOGL, int MODE, int ARITH = ARITH_MONT>
using ColTileNarrow = TileNTT<LOGL + NARROW_F0, NARROW_F0, LOGL, REG_LOGE, MODE, true,
                              typename ColSteps<LOGL, ARITH>::type, ARITH>;

// Slim COL tiles (T = 4, 32-byte segments; the XCD-aware tile order of
// TileNTT::locate lets one L2 merge the two halves of a 64-byte line): half the
// LDS of the T = 8 tile, so two workgroups share a CU.
constexpr int SLIM_F0 = 2;
template <int LOGL, int MODE, int ARITH = ARITH_MONT, bool TWOLVL = false>
using ColTileSlim = TileNTT<LOGL + SLIM_F0, SLIM_F0, LOGL, REG_LOGE, MODE, true,
                            typename ColSteps<LOGL, ARITH>::type, ARITH, TWOLVL>;
// Thin COL tiles (T = 2, 16-byte segments): a quarter of the wide tile's LDS, 2^11 columns in 256 threads --
// four independent workgroups per CU instead of two (experiment, SVENTT_COL_THIN=1).
constexpr int THIN_F0 = 1;
template <int LOGL, int MODE, int ARITH = ARITH_MONT>
using ColTileThin = TileNTT<LOGL + THIN_F0, THIN_F0, LOGL, REG_LOGE, MODE, true,
                            typename ColSteps<LOGL, ARITH>::type, ARITH>;
// Two-level variants of both (tile_ntt.h: TWOLVL): the first pass of the sharded row transform when it
// is longer than the rank count, i.e. the gather of the received pieces fused with the column pass
// that used to follow it.  Lengths 2^2 .. 2^12 (rank count x inner column length).
template <int LOGL, int MODE, int ARITH = ARITH_MONT> using ColTile2L = ColTile<LOGL, MODE, ARITH, true>;
template <int LOGL, int MODE, int ARITH = ARITH_MONT> using ColTileSlim2L = ColTileSlim<LOGL, MODE, ARITH, true>;

// Fine tiles for transforms too small to fill the chip with 2^12-element tiles
// (n * batch <= 2^20, or 2^21 for single-pass rows; e.g. the reference's README shape 2^17 = 2^8 x 2^9): E = 4
// elements per thread in radix-4 steps, 2^8..2^11-element tiles -- four times the
// workgroups and a quarter of the serial work per thread; these runs are latency
// bound, not bandwidth bound.
constexpr int FINE_LOGE = 2;
constexpr int MAX_FINE_ROW_LOGL = 11;
constexpr int MAX_FINE_COL_LOGL = 10;
constexpr int MAX_FINE_TOTAL_LOG2 = 21;
template <int LOGL> struct FineSteps;
template <> struct FineSteps<1> { using type = Steps<1>; };
template <> struct FineSteps<2> { using type = Steps<2>; };
template <> struct FineSteps<3> { using type = Steps<2, 1>; };
template <> struct FineSteps<4> { using type = Steps<2, 2>; };
template <> struct FineSteps<5> { using type = Steps<2, 2, 1>; };
template <> struct FineSteps<6> { using type = Steps<2, 2, 2>; };
template <> struct FineSteps<7> { using type = Steps<2, 2, 2, 1>; };
template <> struct FineSteps<8> { using type = Steps<2, 2, 2, 2>; };
template <> struct FineSteps<9> { using type = Steps<2, 2, 2, 2, 1>; };
template <> struct FineSteps<10> { using type = Steps<2, 2, 2, 2, 2>; };
template <> struct FineSteps<11> { using type = Steps<2, 2, 2, 2, 2, 1>; };
constexpr int fine_row_logt(int logl) { return logl > 8 ? logl : 8; }
constexpr int fine_col_f0(int logl) { return logl >= 6 ? 2 : 8 - logl; }
template <int LOGL, int MODE, bool FLAG>
using RowTileFine = TileNTT<fine_row_logt(LOGL), 0, LOGL, FINE_LOGE, MODE, FLAG,
                            typename FineSteps<LOGL>::type>;
template <int LOGL, int MODE>
using ColTileFine = TileNTT<LOGL + fine_col_f0(LOGL), fine_col_f0(LOGL), LOGL, FINE_LOGE, MODE, true,
                            typename FineSteps<LOGL>::type>;

template <class Status, class Stream> struct KernelEntryT {
  int kind, logl, dir, flag;
  int logt, f0, threads, loge;
  int two_level;  // TileNTT::TWOLVL
  Status (*launch)(const PassArgs &, u32 grid, Stream);
  bool (*set_mapping_ok)();  // TileNTT::verify_set_mapping (host-side check, test tier)
  int group_barriers;        // workgroup barriers per tile (the other exchanges are wave-local)
};

template <class TN, int SI = 0> constexpr int count_group_barriers() {
  if constexpr (SI == TN::NSTEPS)
    return 0;
  else
    return (TN::template sync_before<SI>() == TN::SYNC_GROUP ? 1 : 0) + count_group_barriers<TN, SI + 1>();
}

template <class TN, class Entry, template <class> class Launcher>
Entry make_entry(int kind, int dir, int flag) {
  Entry e;
  e.kind = kind;
  e.logl = TN::LOGL;
  e.dir = dir;
  e.flag = flag;
  e.logt = TN::LOGT;
  e.f0 = TN::F0;
  e.threads = TN::NT;
  e.loge = TN::LOGE;
  e.two_level = TN::TWOLVL ? 1 : 0;
  e.launch = &Launcher<TN>::launch;
#if !defined(__HIP_DEVICE_COMPILE__)
  e.set_mapping_ok = &TN::template verify_set_mapping<0>;
#else
  e.set_mapping_ok = nullptr;
#endif
  e.group_barriers = count_group_barriers<TN>();
  return e;
}

#define SVENTT_ROW_ENTRIES(L)                                                        \
  make_entry<RowTile<L, MODE_FWD, false>, Entry, Launcher>(KIND_ROW, MODE_FWD, 0),   \
  make_entry<RowTile<L, MODE_FWD, true>, Entry, Launcher>(KIND_ROW, MODE_FWD, 1),    \
  make_entry<RowTile<L, MODE_INV, false>, Entry, Launcher>(KIND_ROW, MODE_INV, 0),   \
  make_entry<RowTile<L, MODE_INV, true>, Entry, Launcher>(KIND_ROW, MODE_INV, 1)
#define SVENTT_COL_ENTRIES(L)                                             \
  make_entry<ColTile<L, MODE_FWD>, Entry, Launcher>(KIND_COL, MODE_FWD, 1), \
  make_entry<ColTile<L, MODE_INV>, Entry, Launcher>(KIND_COL, MODE_INV, 1)

#define SVENTT_NARROW_ENTRIES(L)                                                  \
  make_entry<ColTileNarrow<L, MODE_FWD>, Entry, Launcher>(KIND_COL, MODE_FWD, 1), \
  make_entry<ColTileNarrow<L, MODE_INV>, Entry, Launcher>(KIND_COL, MODE_INV, 1)

#define SVENTT_FINE_ROW_ENTRIES(L)                                                       \
  make_entry<RowTileFine<L, MODE_FWD, false>, Entry, Launcher>(KIND_ROW, MODE_FWD, 0),   \
  make_entry<RowTileFine<L, MODE_FWD, true>, Entry, Launcher>(KIND_ROW, MODE_FWD, 1),    \
  make_entry<RowTileFine<L, MODE_INV, false>, Entry, Launcher>(KIND_ROW, MODE_INV, 0),   \
  make_entry<RowTileFine<L, MODE_INV, true>, Entry, Launcher>(KIND_ROW, MODE_INV, 1)
#define SVENTT_FINE_COL_ENTRIES(L)                                              \
  make_entry<ColTileFine<L, MODE_FWD>, Entry, Launcher>(KIND_COL, MODE_FWD, 1), \
  make_entry<ColTileFine<L, MODE_INV>, Entry, Launcher>(KIND_COL, MODE_INV, 1)

// entries parameterised by the arithmetic back end `ARITH` of the enclosing function
#define SVENTT_A_ROW(L)                                                                   \
  make_entry<RowTile<L, MODE_FWD, false, ARITH>, Entry, Launcher>(KIND_ROW, MODE_FWD, 0), \
  make_entry<RowTile<L, MODE_FWD, true, ARITH>, Entry, Launcher>(KIND_ROW, MODE_FWD, 1),  \
  make_entry<RowTile<L, MODE_INV, false, ARITH>, Entry, Launcher>(KIND_ROW, MODE_INV, 0), \
  make_entry<RowTile<L, MODE_INV, true, ARITH>, Entry, Launcher>(KIND_ROW, MODE_INV, 1)
#define SVENTT_A_COL(T, L)                                                         \
  make_entry<T<L, MODE_FWD, ARITH>, Entry, Launcher>(KIND_COL, MODE_FWD, 1),        \
  make_entry<T<L, MODE_INV, ARITH>, Entry, Launcher>(KIND_COL, MODE_INV, 1)

template <class Entry, template <class> class Launcher>
const Entry *find_kernel_in_registry(int kind, int logl, int dir, int flag, int f0, int loge,
                                     int two_level = 0) {
  constexpr int ARITH = ARITH_MONT;  // (SVENTT_A_COL names it)
  static const Entry table[] = {
      SVENTT_NARROW_ENTRIES(1), SVENTT_NARROW_ENTRIES(2), SVENTT_NARROW_ENTRIES(3),
      SVENTT_NARROW_ENTRIES(4), SVENTT_NARROW_ENTRIES(5), SVENTT_NARROW_ENTRIES(6),
      SVENTT_NARROW_ENTRIES(7), SVENTT_NARROW_ENTRIES(8),
      SVENTT_ROW_ENTRIES(1),  SVENTT_ROW_ENTRIES(2),  SVENTT_ROW_ENTRIES(3),
      SVENTT_ROW_ENTRIES(4),  SVENTT_ROW_ENTRIES(5),  SVENTT_ROW_ENTRIES(6),
      SVENTT_ROW_ENTRIES(7),  SVENTT_ROW_ENTRIES(8),  SVENTT_ROW_ENTRIES(9),
      SVENTT_ROW_ENTRIES(10), SVENTT_ROW_ENTRIES(11), SVENTT_ROW_ENTRIES(12),
      SVENTT_ROW_ENTRIES(13),
      SVENTT_COL_ENTRIES(1),  SVENTT_COL_ENTRIES(2),  SVENTT_COL_ENTRIES(3),
      SVENTT_COL_ENTRIES(4),  SVENTT_COL_ENTRIES(5),  SVENTT_COL_ENTRIES(6),
      SVENTT_COL_ENTRIES(7),  SVENTT_COL_ENTRIES(8),  SVENTT_COL_ENTRIES(9),
      SVENTT_COL_ENTRIES(10), SVENTT_COL_ENTRIES(11),
      make_entry<ColTileSlim<10, MODE_FWD>, Entry, Launcher>(KIND_COL, MODE_FWD, 1),
      make_entry<ColTileSlim<10, MODE_INV>, Entry, Launcher>(KIND_COL, MODE_INV, 1),
      make_entry<ColTileSlim<11, MODE_FWD>, Entry, Launcher>(KIND_COL, MODE_FWD, 1),
      make_entry<ColTileSlim<11, MODE_INV>, Entry, Launcher>(KIND_COL, MODE_INV, 1),
      make_entry<ColTileSlim<12, MODE_FWD>, Entry, Launcher>(KIND_COL, MODE_FWD, 1),
      make_entry<ColTileSlim<12, MODE_INV>, Entry, Launcher>(KIND_COL, MODE_INV, 1),
      SVENTT_A_COL(ColTile2L, 2), SVENTT_A_COL(ColTile2L, 3), SVENTT_A_COL(ColTile2L, 4),
      SVENTT_A_COL(ColTile2L, 5), SVENTT_A_COL(ColTile2L, 6), SVENTT_A_COL(ColTile2L, 7),
      SVENTT_A_COL(ColTile2L, 8), SVENTT_A_COL(ColTile2L, 9), SVENTT_A_COL(ColTile2L, 10),
      SVENTT_A_COL(ColTile2L, 11),
      SVENTT_A_COL(ColTileSlim2L, 10), SVENTT_A_COL(ColTileSlim2L, 11), SVENTT_A_COL(ColTileSlim2L, 12),
#if defined(SVENTT_WITH_THIN)
      SVENTT_A_COL(ColTileThin, 11), SVENTT_A_COL(ColTileThin, 12),
#endif
      SVENTT_FINE_ROW_ENTRIES(1), SVENTT_FINE_ROW_ENTRIES(2), SVENTT_FINE_ROW_ENTRIES(3),
      SVENTT_FINE_ROW_ENTRIES(4), SVENTT_FINE_ROW_ENTRIES(5), SVENTT_FINE_ROW_ENTRIES(6),
      SVENTT_FINE_ROW_ENTRIES(7), SVENTT_FINE_ROW_ENTRIES(8), SVENTT_FINE_ROW_ENTRIES(9),
      SVENTT_FINE_ROW_ENTRIES(10), SVENTT_FINE_ROW_ENTRIES(11),
      SVENTT_FINE_COL_ENTRIES(1), SVENTT_FINE_COL_ENTRIES(2), SVENTT_FINE_COL_ENTRIES(3),
      SVENTT_FINE_COL_ENTRIES(4), SVENTT_FINE_COL_ENTRIES(5), SVENTT_FINE_COL_ENTRIES(6),
      SVENTT_FINE_COL_ENTRIES(7), SVENTT_FINE_COL_ENTRIES(8), SVENTT_FINE_COL_ENTRIES(9),
      SVENTT_FINE_COL_ENTRIES(10),
  };
  for (const Entry &e : table)
    if (e.kind == kind && e.logl == logl && e.dir == dir && e.flag == flag && e.f0 == f0 &&
        e.loge == loge && e.two_level == two_level)
      return &e;
  return nullptr;
}

// The E = 16 tiles of one of the other arithmetic back ends (field64.h: ARITH_GOLD,
// ARITH_SHOUP); plans of those back ends never use the fine tiles.  Same shapes, same lookup.
template <int ARITH, class Entry, template <class> class Launcher>
const Entry *find_arith_kernel_in_registry(int kind, int logl, int dir, int flag, int f0, int loge,
                                           int two_level = 0) {
  static const Entry table[] = {
      SVENTT_A_COL(ColTileNarrow, 1), SVENTT_A_COL(ColTileNarrow, 2), SVENTT_A_COL(ColTileNarrow, 3),
      SVENTT_A_COL(ColTileNarrow, 4), SVENTT_A_COL(ColTileNarrow, 5), SVENTT_A_COL(ColTileNarrow, 6),
      SVENTT_A_COL(ColTileNarrow, 7), SVENTT_A_COL(ColTileNarrow, 8),
      SVENTT_A_ROW(1),  SVENTT_A_ROW(2),  SVENTT_A_ROW(3),  SVENTT_A_ROW(4),  SVENTT_A_ROW(5),
      SVENTT_A_ROW(6),  SVENTT_A_ROW(7),  SVENTT_A_ROW(8),  SVENTT_A_ROW(9),  SVENTT_A_ROW(10),
      SVENTT_A_ROW(11), SVENTT_A_ROW(12), SVENTT_A_ROW(13),
      SVENTT_A_COL(ColTile, 1), SVENTT_A_COL(ColTile, 2), SVENTT_A_COL(ColTile, 3),
      SVENTT_A_COL(ColTile, 4), SVENTT_A_COL(ColTile, 5), SVENTT_A_COL(ColTile, 6),
      SVENTT_A_COL(ColTile, 7), SVENTT_A_COL(ColTile, 8), SVENTT_A_COL(ColTile, 9),
      SVENTT_A_COL(ColTile, 10), SVENTT_A_COL(ColTile, 11),
      SVENTT_A_COL(ColTileSlim, 10), SVENTT_A_COL(ColTileSlim, 11), SVENTT_A_COL(ColTileSlim, 12),
      SVENTT_A_COL(ColTile2L, 2), SVENTT_A_COL(ColTile2L, 3), SVENTT_A_COL(ColTile2L, 4),
      SVENTT_A_COL(ColTile2L, 5), SVENTT_A_COL(ColTile2L, 6), SVENTT_A_COL(ColTile2L, 7),
      SVENTT_A_COL(ColTile2L, 8), SVENTT_A_COL(ColTile2L, 9), SVENTT_A_COL(ColTile2L, 10),
      SVENTT_A_COL(ColTile2L, 11),
      SVENTT_A_COL(ColTileSlim2L, 10), SVENTT_A_COL(ColTileSlim2L, 11), SVENTT_A_COL(ColTileSlim2L, 12),
  };
  for (const Entry &e : table)
    if (e.kind == kind && e.logl == logl && e.dir == dir && e.flag == flag && e.f0 == f0 &&
        e.loge == loge && e.two_level == two_level)
      return &e;
  return nullptr;
}

// Shape facts the host planner needs without instantiating anything.
// log2 of the tile width for a column pass of length 2^logl over 2^logs columns
// (-1: fewer columns than the narrowest tile).
inline int registry_col_f0(int logl, int logs, bool slim = false) {
#if defined(SVENTT_WITH_THIN)
  if (std::getenv("SVENTT_COL_THIN") && logl >= 11 && logl <= 12 && logs >= THIN_F0) return THIN_F0;
#endif
  if (slim && logl >= 10 && logl <= 12 && logs >= SLIM_F0) return SLIM_F0;
  if (logl > 11) return -1;
  if (logs >= col_f0(logl)) return col_f0(logl);
  if (logs >= NARROW_F0 && logl <= 8) return NARROW_F0;
  return -1;
}
// is there a two-level variant (ColTile2L / ColTileSlim2L) of the column tile (logl, f0)?
inline bool registry_has_two_level(int logl, int f0) {
  if (logl >= 10 && logl <= 12 && f0 == SLIM_F0) return true;
  return logl >= 2 && logl <= 11 && f0 == col_f0(logl);
}
inline int registry_row_logt(int logl) { return row_logt(logl); }
inline int registry_fine_col_f0(int logl, int logs) {
  if (logl > MAX_FINE_COL_LOGL || logs < fine_col_f0(logl)) return -1;
  return fine_col_f0(logl);
}
inline int registry_fine_row_logt(int logl) { return fine_row_logt(logl); }

}  // namespace sventt_hip
