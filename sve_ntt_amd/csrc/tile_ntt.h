// sve_ntt_amd/csrc/tile_ntt.h -- one workgroup's share of an NTT pass.
//
// A pass of the plan (plan_core.h) cuts the data into TILES of 2^LOGT elements; one
// workgroup owns one tile for the whole pass: every element is read from HBM
// once, all of the pass's butterfly stages run on chip, every element is
// written once.  This is the GPU shape of the reference's per-thread
// cache-resident block (layer/sve/blocked-generic.hpp:139-154: "transpose a
// block of columns into a padded L1/L2 buffer, run the inner NTTs, transpose
// back") and of its inner IterativeNTT (kernel/iterative.hpp:43-54); there is
// no separate transposition, the tile's addressing *is* the transpose.
//
// Tile index space.  A tile element has a LOGT-bit index I.  The transform
// acts on the bit field [F0, F0+LOGL) of I; the remaining bits number the
// independent sequences in the tile:
//   ROW tile (F0 == 0):  I = b*L + i          2^(LOGT-LOGL) contiguous rows
//   COL tile (F0  > 0):  I = i*T + b          T = 2^F0 adjacent columns, row
//                                             i is `istride` elements away in HBM
//
// Stages are fused into STEPS of k <= LOGE stages (the reference's
// RadixTwo/Four/EightSVELayer are k = 1, 2, 3: layer/sve/radix-*.hpp).  In a
// step every thread holds E = 2^LOGE elements in registers, grouped as
// E/2^k radix-2^k sets, and runs k butterfly stages on them with no exchange;
// between steps the tile is re-distributed through LDS (one ds_write_b64 and
// one ds_read_b64 per element, XOR-swizzled against bank conflicts).  The
// first step reads HBM directly and the last one writes HBM directly.
//
// Forward = decimation in frequency, natural in / bit-reversed out, stage
// twiddle omega_{2h}^(i mod h) as in tests/ntt-reference.hpp:43-61 of the
// reference; inverse = the exact mirror (:63-83).
//
// The same code is compiled for the host by tests/cpu_sim (a sequential
// emulation used to debug index arithmetic without a GPU; test-only).
#pragma once

#include "field64.h"

namespace sventt_hip {

enum : int { MODE_FWD = 0, MODE_INV = 1 };

// Device builds run the butterfly stages of E = 16 tiles as generated gfx950 assembly
// (gen_stage_asm.py: same arithmetic as field64.h, conditional +N under an EXEC mask).
// -DSVENTT_NO_STAGE_ASM keeps hipcc's code for A/B measurements; the host replay
// (tests/cpu_sim) always runs the C++ below.
#ifndef SVENTT_EARLY_STORES
#define SVENTT_EARLY_STORES 1  // 0: A/B builds that store a thread's 16 outputs together at the end
#endif
// Wave priority (s_setprio, 0..3) while a tile's first step issues its HBM loads / its last step its
// stores.  VALU issue goes to the highest priority, then to the oldest wave: a workgroup that has just
// started is the youngest on its SIMDs and computes its load addresses in whatever slots the
// other workgroup's butterflies leave (A/B: profiles/r03/asm_stages_ab.txt).
// Paired sets (experiment, -DSVENTT_PAIR=1): in a radix-8 step that touches HBM the thread's two sets are
// made the two NEIGHBOURING elements (ROW) / columns (COL), so that each of its eight accesses moves
// 16 bytes instead of 8 (half the vector-memory instructions; profiles/r03/asm_stages_ab.txt).
#ifndef SVENTT_PAIR
#define SVENTT_PAIR 0
#endif
#ifndef SVENTT_PRIO_LOAD
#define SVENTT_PRIO_LOAD 0
#endif
#ifndef SVENTT_PRIO_STORE
#define SVENTT_PRIO_STORE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SVENTT_NO_STAGE_ASM)
#define SVENTT_STAGE_ASM 1
#include "stage_asm.inc"
#endif

// Arguments of one pass (passed by value to the kernel: lives in SGPRs).
#if defined(SVENTT_TRACE)  // analysis builds only (tools/trace_tiles.py): per-wave time stamps of a tile's phases
constexpr int TRACE_SLOTS = 32;
constexpr int TRACE_MAX_WAVES = 1 << 15;
__device__ unsigned long long g_trace[(size_t)TRACE_MAX_WAVES * TRACE_SLOTS];
#define SVENTT_TRACE_LDS_BYTES (16 * TRACE_SLOTS * 8)
#else
#define SVENTT_TRACE_LDS_BYTES 0
#endif

struct PassArgs {
  u64 *dst;
  const u64 *src;
  Field f;
  const u64 *stage_tw;  // stage tables of the pass length L: stage bit p (span 2^p) at
                        // [2^p - 1, 2^(p+1) - 1): omega_{2^(p+1)}^j, Montgomery form
  u64 total;            // elements in the whole array (tiles past it are masked)
  // COL geometry: element (o, i, c) is read from o*src_ostride + i*src_istride + c and
  // written to o*ostride + i*istride + c (the two differ only in the gather/scatter
  // passes next to the all-to-all of the sharded transform)
  u64 istride;
  u64 ostride;
  u64 src_istride;
  u64 src_ostride;
  // TWO-LEVEL tiles only (TileNTT's TWOLVL; the row transform of the sharded six-step, whose
  // first pass reads rows that lie in `nranks` received pieces): row i of a block is at
  // (i >> row_split) * istride_hi + (i mod 2^row_split) * istride.  A side that is an ordinary
  // matrix has istride_hi = istride << row_split.
  u64 istride_hi;
  u64 src_istride_hi;
  u32 row_split;
  u32 tiles_per_outer;  // column tiles per block covered by THIS launch
  // A launch may cover only a chunk of the columns (the sharded transform pipelines its
  // all-to-all chunk by chunk).  The chunk is 2^k equal RUNS of adjacent column tiles, one
  // run every run_period columns (one run, run_shift = 31, unless the pass that consumes the
  // exchanged chunk is a two-level one): tile ct of the launch is tile (ct mod 2^run_shift) of
  // run (ct >> run_shift) and starts at column run * run_period + (ct_first + ct mod 2^run_shift) * T.
  // A side whose buffer holds just the chunk (`compact`) numbers its columns ct * T.
  u32 ct_first;
  u32 run_shift;
  u32 run_period;
  u32 compact;          // bit 0: dst is compact, bit 1: src is compact
  u32 grid;             // workgroups in this launch
  // twist of the pass (six-step twiddle, layer/sve/generic.hpp:95-105,169-188):
  // omega_M^e = twist_hi[e >> twist_shift] * twist_lo[e & mask], Montgomery form
  const u64 *twist_lo;
  const u64 *twist_hi;
  u32 twist_shift;
  u64 twist_col_offset; // added to the column index (rank offset of a sharded column pass)
  u64 scale;            // ROW inverse with FLAG: L^{-1} (Montgomery form)
  const u64 *epilogue;  // ROW forward with FLAG: dst[i] = X[i] * epilogue[i] (operand in
                        // Montgomery form), the pointwise product a convolution does next
};

template <int... KS> struct Steps {
  static constexpr int n = sizeof...(KS);
  static constexpr int k[sizeof...(KS)] = {KS...};
  static constexpr int sum(int upto) {
    int s = 0;
    for (int i = 0; i < upto; ++i) s += k[i];
    return s;
  }
};

// XOR swizzle of the LDS image: bank bits 0..4 (8-byte words) are mixed with
// index bits 4..8 so that the strided element sets of every step fall on
// distinct banks (checked by tests/test_host_logic.py::test_lds_swizzle_is_a_bijection_and_conflict_free).
F64_HD u32 lds_phys(u32 I) { return I ^ ((I >> 4) & 31u); }

F64_HD u32 bitrev32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x);
#else
  x = ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
  x = ((x & 0x33333333u) << 2) | ((x >> 2) & 0x33333333u);
  x = ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu);
  x = ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu);
  return (x << 16) | (x >> 16);
#endif
}

// FLAG: COL tiles -> apply the pass twist (store side forward, load side inverse);
//       ROW inverse tiles -> fold the 1/L scaling into the top stage;
//       ROW forward tiles -> multiply by PassArgs::epilogue on the way out.
// ARITH: the arithmetic back end of the butterflies and of the twist (field64.h); the fold of
// 1/L and the fused product always use the Montgomery product (their operands arrive in
// Montgomery form whatever the back end).
// TWOLVL: COL tiles whose rows lie at a two-level stride on one side (PassArgs::row_split): the
// first pass of the sharded row transform, fused with the gather of the received pieces.
template <int LOGT_, int F0_, int LOGL_, int LOGE_, int MODE_, bool FLAG_, class STEPS_,
          int ARITH_ = ARITH_MONT, bool TWOLVL_ = false>
struct TileNTT {
  static constexpr int LOGT = LOGT_, F0 = F0_, LOGL = LOGL_, LOGE = LOGE_, MODE = MODE_;
  static constexpr int ARITH = ARITH_;
  static constexpr bool TWOLVL = TWOLVL_;
  static constexpr int TWW = Arith<ARITH_>::TW_WORDS;  // 64-bit words per table entry
  static constexpr bool FLAG = FLAG_;
  using STEPS = STEPS_;
  static constexpr int E = 1 << LOGE;
  static constexpr int NT = 1 << (LOGT - LOGE);
  static constexpr bool COL = F0 > 0;
  static constexpr int T = 1 << F0;
  static constexpr int NSTEPS = STEPS::n;
  static_assert(STEPS::sum(NSTEPS) == LOGL, "steps must cover every stage once");
  static_assert(LOGT >= LOGE && F0 + LOGL <= LOGT, "tile too small");
  static_assert(!COL || F0 + LOGL == LOGT, "a COL tile holds whole columns");
  static_assert(COL || !TWOLVL, "only COL tiles have row strides");

  // ---- stage twiddles of the lower steps live in LDS ----------------------------------------
  // The vector L1 returns data in order: a twiddle load that hits in L2 still waits behind the
  // HBM loads and stores the CU's other workgroup has in flight (a build without twiddle loads
  // runs 8 % faster, one without HBM traffic barely notices them: profiles/r02/asm_stages_ab.txt).
  // The tables of every step below the top one are a PREFIX of the stage table (stage bit p at
  // [2^p - 1, 2^(p+1) - 1), the top step owns the largest LOGL - k[0] .. LOGL - 1): at most
  // 511 entries.  The first step a tile executes copies that prefix behind the tile image (one
  // load per thread, issued ahead of the data) and the middle steps read their twiddles with
  // ds_read; the lowest step's indices are compile-time constants and stay scalar loads.
  // Measured (profiles/r02/asm_stages_ab.txt): 2^16 x 2^12 forward -10 %, inverse -6 %; N = 2^24 inverse
  // -2.6 %, forward row pass -2.5 %; the forward column pass gains nothing (one step of three reads the
  // copy, and the card runs into its power limit, section 4 of DESIGN.md), so it keeps its table loads.
#if !defined(SVENTT_TW_LDS)  // analysis builds: 0 = none, 1 = all but forward COL tiles, 2 = all
#define SVENTT_TW_LDS 1
#endif
  static constexpr int TW_LDS_P =
      (LOGE == 4 && NSTEPS >= 3 &&
       (SVENTT_TW_LDS == 2 || (SVENTT_TW_LDS == 1 && !(COL && MODE == MODE_FWD && TWW == 1))))  // (two-word twiddles: always, for the registers)
          ? LOGL - STEPS::k[0]
          : 0;
  static constexpr u32 TW_LDS_WORDS = TW_LDS_P > 0 ? ((1u << TW_LDS_P) - 1u) * (u32)TWW : 0u;
  static constexpr u32 TW_LDS_PER_THREAD = (TW_LDS_WORDS + (u32)NT - 1u) / (u32)NT;
  static constexpr size_t LDS_BYTES =
      NSTEPS > 1 ? (sizeof(u64) << LOGT) + (((size_t)TW_LDS_WORDS * sizeof(u64) + 15u) & ~(size_t)15u) +
                       SVENTT_TRACE_LDS_BYTES
                 : 0;
#if defined(SVENTT_TRACE)
  static constexpr size_t TRACE_LDS_WORD = ((size_t)1 << LOGT) + ((TW_LDS_WORDS + 1u) & ~1u);
  __device__ __forceinline__ static void trace_stamp(u64 *lds, int slot) {
    const u64 tm = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0) lds[TRACE_LDS_WORD + (threadIdx.x >> 6) * TRACE_SLOTS + slot] = tm;
  }
#define SVENTT_STAMP(lds, slot) trace_stamp(lds, slot)
#else
#define SVENTT_STAMP(lds, slot) ((void)0)
#endif

  // ---- which thread holds which radix set ----------------------------------------------
  // Set number s (LOGT - k bits) names the 2^k elements I = ((s >> lo) << hi) | (v << lo) |
  // (s mod 2^lo).  Wave w of the workgroup owns the contiguous CHUNK [w, w + 1) * 2^(LOGT - WB)
  // of the tile.  A step whose register bits [lo, hi) lie below the chunk bits is
  // chunk-preserving: with s = w : g : lane (wave, set of the thread, lane) every wave then holds
  // exactly its chunk, as it did in the neighbouring chunk-preserving steps -- the LDS exchange
  // between two such steps stays inside each wave's own region and needs no workgroup barrier
  // (2^13 rows: one barrier per tile instead of three; the waves of a workgroup drift apart and
  // overlap each other's memory phases).  Other steps use s = tid + g * NT.
  static constexpr int WB = (NT > 64) ? LOGT - LOGE - 6 : 0;  // log2(waves per workgroup)
  template <int SI> static constexpr bool chunk_preserving() {
    return NT <= 64 || (F0 + LOGL - STEPS::sum(SI)) <= LOGT - WB;
  }
  // (see SVENTT_PAIR) a radix-8 step of an E = 16 tile whose two sets are bit 0 of the tile index
  template <int SI> static constexpr bool pair_sets() {
    constexpr bool from_hbm = (MODE == MODE_FWD) ? (SI == 0) : (SI == NSTEPS - 1);
    constexpr bool to_hbm = (MODE == MODE_FWD) ? (SI == NSTEPS - 1) : (SI == 0);
    return SVENTT_PAIR != 0 && LOGE == 4 && STEPS::k[SI] == 3 && (from_hbm || to_hbm) && !TWOLVL &&
           (COL || LOGT == LOGL) && F0 + LOGL - STEPS::sum(SI) - 3 >= 1;
  }
  template <int SI> F64_HD static u32 set_number(u32 tid, int g) {
    constexpr int k = STEPS::k[SI];
    if constexpr (pair_sets<SI>()) {
      if constexpr (NT > 64 && chunk_preserving<SI>())
        return ((tid >> 6) << (LOGT - k - WB)) | ((tid & 63u) << 1) | (u32)g;
      else
        return (tid << 1) | (u32)g;
    } else if constexpr (NT > 64 && chunk_preserving<SI>())
      return ((tid >> 6) << (LOGT - k - WB)) | ((u32)g << 6) | (tid & 63u);
    else
      return tid + (u32)g * NT;
  }
  // Host-side proof of the claim above for this tile shape (tests/test_host_logic.py runs it over the
  // whole registry): in every step the sets are a bijection onto the tile, and in a
  // chunk-preserving step every element of wave w lies in chunk w.
#if !defined(__HIP_DEVICE_COMPILE__)
  template <int SI = 0> static bool verify_set_mapping() {
    if constexpr (SI == NSTEPS) {
      return true;
    } else {
      constexpr int k = STEPS::k[SI];
      constexpr int HI = LOGL - STEPS::sum(SI), LO = HI - k, lo = F0 + LO, hi = F0 + HI;
      std::vector<unsigned char> seen((size_t)1 << LOGT, 0);
      for (u32 tid = 0; tid < (u32)NT; ++tid)
        for (int g = 0; g < (E >> k); ++g)
          for (u32 v = 0; v < (1u << k); ++v) {
            const u32 st = set_number<SI>(tid, g);
            const u32 I = ((st >> lo) << hi) | (v << lo) | (st & ((1u << lo) - 1u));
            if (I >> LOGT || seen[I]++) return false;
            if (chunk_preserving<SI>() && NT > 64 && (I >> (LOGT - WB)) != (tid >> 6)) return false;
          }
      return verify_set_mapping<SI + 1>();
    }
  }
#endif
  enum : int { SYNC_NONE = 0, SYNC_WAVE = 1, SYNC_GROUP = 2 };
  // what must separate step SI from the step executed before it
  template <int SI> static constexpr int sync_before() {
    constexpr int prev = (MODE == MODE_FWD) ? SI - 1 : SI + 1;
    if constexpr (prev < 0 || prev >= NSTEPS)
      return SYNC_NONE;
    else
      return (chunk_preserving<SI>() && chunk_preserving<prev>()) ? SYNC_WAVE : SYNC_GROUP;
  }
  template <int KIND> F64_HD static void exchange_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (KIND == SYNC_GROUP) {
      __syncthreads();
    } else if constexpr (KIND == SYNC_WAVE) {
      // same wave wrote what it now reads: LDS executes a wave's operations in order; only the
      // compiler must not move them across this point
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#endif
  }

  struct Tile {
    u64 base;    // element offset of the tile in HBM (destination side)
    u64 sbase;   // same, source side
    u32 c0;      // COL: first column (index inside the M-point sub-transform)
    bool live;   // false: whole tile lies past the end (ROW, ragged batch)
  };

  F64_HD static Tile locate(const PassArgs &a, u32 block) {
    Tile t;
    if constexpr (COL) {
      // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so
      // blocks b and b+8 share an L2.  Give each XCD a contiguous run of tiles: adjacent
      // column tiles (which share 128-byte lines when T*8 < 128) then meet in one L2
      // instead of being fetched by two.  Placement affects speed only.
      if ((a.grid & 7u) == 0) block = (block & 7u) * (a.grid >> 3) + (block >> 3);
      const u32 o = block / a.tiles_per_outer, ct = block - o * a.tiles_per_outer;
      const u32 run = ct >> a.run_shift, within = ct - (run << a.run_shift);
      t.c0 = run * a.run_period + (a.ct_first + within) * (u32)T;
      t.base = (u64)o * a.ostride + ((a.compact & 1u) ? ct * (u32)T : t.c0);
      t.sbase = (u64)o * a.src_ostride + ((a.compact & 2u) ? ct * (u32)T : t.c0);
      t.live = true;
    } else {
      t.c0 = 0;
      t.base = (u64)block << LOGT;
      t.sbase = t.base;
      t.live = t.base < a.total;
    }
    return t;
  }

  // element offset of the row bits `rows` (any subset of a row index's bits: the map is additive
  // over disjoint bit sets) on the destination / source side
  F64_HD static u64 row_offset(const PassArgs &a, u32 rows) {
    if constexpr (TWOLVL)
      return (u64)(rows >> a.row_split) * a.istride_hi + (u64)(rows & ((1u << a.row_split) - 1u)) * a.istride;
    else
      return (u64)rows * a.istride;
  }
  F64_HD static u64 src_row_offset(const PassArgs &a, u32 rows) {
    if constexpr (TWOLVL)
      return (u64)(rows >> a.row_split) * a.src_istride_hi +
             (u64)(rows & ((1u << a.row_split) - 1u)) * a.src_istride;
    else
      return (u64)rows * a.src_istride;
  }

  F64_HD static u64 gaddr(const PassArgs &a, const Tile &t, u32 I) {
    if constexpr (COL)
      return t.base + row_offset(a, I >> F0) + (I & (u32)(T - 1));
    else
      return t.base + I;
  }

  F64_HD static u64 saddr(const PassArgs &a, const Tile &t, u32 I) {
    if constexpr (COL)
      return t.sbase + src_row_offset(a, I >> F0) + (I & (u32)(T - 1));
    else
      return t.sbase + I;
  }
  // offset of register v of a radix set (register bits from row bit LO up); `vstride` is the
  // one-level form v * (stride << LO), which the ordinary tiles keep as it was
  template <int LO> F64_HD static u64 dst_off(const PassArgs &a, int v, u64 vstride) {
    if constexpr (TWOLVL)
      return row_offset(a, (u32)v << LO);
    else
      return (u64)v * vstride;
  }
  template <int LO> F64_HD static u64 src_off(const PassArgs &a, int v, u64 vstride) {
    if constexpr (TWOLVL)
      return src_row_offset(a, (u32)v << LO);
    else
      return (u64)v * vstride;
  }

  F64_HD static bool in_range(const PassArgs &a, const Tile &t, u32 I) {
    // Only a ROW tile holding several rows can straddle the end of a ragged batch; a
    // tile that is one whole row (or whole columns) is either live or not launched.
    if constexpr (COL || LOGT == LOGL)
      return true;
    else
      return t.base + I < a.total;
  }

  // Six-step twiddle of element I = I0 | v<<lo: omega_M^(bitrev_L(i) * c).  bitrev is
  // linear over disjoint bit fields, so the exponent splits into a per-set part
  // (twist_e0) and c times a compile-time constant per element (twist_bv).
  // e < M <= 2^32 (the planner never builds a larger block): 32-bit arithmetic.
  F64_HD static u32 twist_col(const PassArgs &a, const Tile &t, u32 I0) {
    return (u32)a.twist_col_offset + t.c0 + (I0 & (u32)(T - 1));
  }
  F64_HD static u32 twist_e0(u32 col, u32 I0) {
    const u32 i0 = I0 >> F0;
    return col * (LOGL ? (bitrev32(i0) >> (32 - LOGL)) : 0u);
  }
  static constexpr u32 twist_bv(int v, int LO) {
    // bitrev_LOGL(v << LO), by hand so that it folds at compile time
    u32 x = (u32)v << LO, r = 0;
    for (int b = 0; b < LOGL; ++b) r |= ((x >> b) & 1u) << (LOGL - 1 - b);
    return r;
  }
  // x * omega_M^e with omega_M^e = hi[e >> shift] * lo[e & mask] (C++ path)
  F64_HD static u64 twist_apply_cxx(const PassArgs &a, u64 x, u32 e) {
    const u64 *lo = a.twist_lo + (size_t)(e & ((1u << a.twist_shift) - 1u)) * TWW;
    const u64 *hi = a.twist_hi + (size_t)(e >> a.twist_shift) * TWW;
    if constexpr (ARITH == ARITH_SHOUP) {
      // the product hi*lo would need its own precomputed companion: multiply twice instead
      return Arith<ARITH>::mul(Arith<ARITH>::mul(x, hi, a.f), lo, a.f);
    } else {
      const u64 tw = Arith<ARITH>::mul(hi[0], lo, a.f);
      return Arith<ARITH>::mul(x, &tw, a.f);
    }
  }

#if defined(SVENTT_STAGE_ASM)
  // first element of the b-th butterfly (ascending) of stage bit r over 16 registers
  static constexpr int bf_first(int r, int b) { return ((b >> r) << (r + 1)) | (b & ((1 << r) - 1)); }

  // Looked-up halves of the twist factors of TWG elements (x[TWG GRP .. TWG GRP + TWG - 1]).
  static constexpr int TWG = (ARITH == ARITH_SHOUP) ? 2 : 4;
  struct TwistFactors {
    u64 h[TWG], l[TWG];
    u64 hp[TWG], lp[TWG];  // ARITH_SHOUP: the precomputed companions
  };
  template <int k, int LO, int GRP>
  __device__ __forceinline__ static TwistFactors twist_load(const PassArgs &a, const Tile &t,
                                                            const u32 (&I0)[E >> k]) {
    constexpr int R = 1 << k;
    TwistFactors f;
#pragma unroll
    for (int q = 0; q < TWG; ++q) {
      const int i = TWG * GRP + q, g = i >> k, v = i & (R - 1);
      const u32 col = twist_col(a, t, I0[g]);
      const u32 e = twist_e0(col, I0[g]) + col * twist_bv(v, LO);
      const u64 *lo = a.twist_lo + (size_t)(e & ((1u << a.twist_shift) - 1u)) * TWW;
      const u64 *hi = a.twist_hi + (size_t)(e >> a.twist_shift) * TWW;
#if defined(SVENTT_STUB_TWIST)  // analysis builds only: what the scattered twist-table loads cost (wrong results)
      f.l[q] = (u64)(uintptr_t)lo | 1u, f.h[q] = (u64)(uintptr_t)hi | 1u;
      if constexpr (TWW == 2) f.lp[q] = f.l[q], f.hp[q] = f.h[q];
#else
      if constexpr (TWW == 2) {
        const ulonglong2 lv = *reinterpret_cast<const ulonglong2 *>(lo);
        const ulonglong2 hv = *reinterpret_cast<const ulonglong2 *>(hi);
        f.l[q] = lv.x, f.lp[q] = lv.y, f.h[q] = hv.x, f.hp[q] = hv.y;
      } else {
        f.l[q] = lo[0], f.h[q] = hi[0];
      }
#endif
    }
    return f;
  }
  // x[TWG GRP + q] *= h[q] * l[q]   (twist_apply_cxx of the C++ path, one assembly group)
  template <int GRP>
  __device__ __forceinline__ static void twist_apply(u64 (&x)[E], const TwistFactors &f, u32 (&zr)[4],
                                                     const AsmConsts &c) {
    if constexpr (ARITH == ARITH_SHOUP)
      TwistGroup<ARITH, GRP>::run(x, f.h, f.l, f.hp, f.lp, zr, c);
    else
      TwistGroup<ARITH, GRP>::run(x, f.h, f.l, f.h, f.l, zr, c);  // no companions: the arrays are ignored
  }
  // x[FIRST .. FIRST + COUNT) to HBM (elements of a step that writes the pass's output).
  // Neighbouring elements of a lowest step (lo == 0) leave as one 16-byte store.  Every address
  // goes through an empty asm: hipcc's machine scheduler crashes (roc-7.2.0, SIGSEGV) when it
  // tries to cluster plain stores across the assembly statements they are interleaved with.
  template <int k, int LO, int FIRST, int COUNT>
  __device__ __forceinline__ static void store_range(const PassArgs &a, const Tile &t, const u64 (&x)[E],
                                                     const u32 (&I0)[E >> k]) {
    constexpr int R = 1 << k, lo = F0 + LO;
    constexpr bool pairs = !COL && lo == 0 && (FIRST % 2 == 0) && (COUNT % 2 == 0) && in_range_is_static();
    const u64 vstride = COL ? (a.istride << LO) : (1ull << lo);
#pragma unroll
    for (int i = FIRST; i < FIRST + COUNT; i += (pairs ? 2 : 1)) {
      const int g = i >> k, v = i & (R - 1);
      u64 *p = a.dst + gaddr(a, t, I0[g]) + dst_off<LO>(a, v, vstride);
      asm volatile("" : "+v"(p));
#if defined(SVENTT_STUB_HBM) || defined(SVENTT_STUB_STORES)
      if (x[i] == 0x123456789abcdefull) *p = x[i];
#else
      if constexpr (pairs) {
        ulonglong2 two;
        two.x = x[i], two.y = x[i + 1];
        *reinterpret_cast<ulonglong2 *>(p) = two;
      } else {
        if (in_range(a, t, I0[g] | ((u32)v << lo))) *p = x[i];
      }
#endif
    }
  }
  static constexpr bool in_range_is_static() { return COL || LOGT == LOGL; }

  // The twist of all 16 elements; the factors of group g + 1 are requested before group g's ~230
  // VALU instructions run.  `f0` holds group 0's, requested by the caller.
  template <int k, int LO>
  __device__ __forceinline__ static void twist_all(const PassArgs &a, const Tile &t, u64 (&x)[E],
                                                   const u32 (&I0)[E >> k], const TwistFactors &f0,
                                                   u32 (&zr)[4], const AsmConsts &c) {
    if constexpr (ARITH == ARITH_SHOUP) {
      // four words per element: two elements per statement, the next pair requested ahead
      const TwistFactors f1 = twist_load<k, LO, 1>(a, t, I0);
      twist_apply<0>(x, f0, zr, c);
      const TwistFactors f2 = twist_load<k, LO, 2>(a, t, I0);
      twist_apply<1>(x, f1, zr, c);
      const TwistFactors f3 = twist_load<k, LO, 3>(a, t, I0);
      twist_apply<2>(x, f2, zr, c);
      const TwistFactors f4 = twist_load<k, LO, 4>(a, t, I0);
      twist_apply<3>(x, f3, zr, c);
      const TwistFactors f5 = twist_load<k, LO, 5>(a, t, I0);
      twist_apply<4>(x, f4, zr, c);
      const TwistFactors f6 = twist_load<k, LO, 6>(a, t, I0);
      twist_apply<5>(x, f5, zr, c);
      const TwistFactors f7 = twist_load<k, LO, 7>(a, t, I0);
      twist_apply<6>(x, f6, zr, c);
      twist_apply<7>(x, f7, zr, c);
    } else {
      const TwistFactors f1 = twist_load<k, LO, 1>(a, t, I0);
      twist_apply<0>(x, f0, zr, c);
      const TwistFactors f2 = twist_load<k, LO, 2>(a, t, I0);
      twist_apply<1>(x, f1, zr, c);
      const TwistFactors f3 = twist_load<k, LO, 3>(a, t, I0);
      twist_apply<2>(x, f2, zr, c);
      twist_apply<3>(x, f3, zr, c);
    }
  }
  // the same with the first two groups' factors already requested (Montgomery / Goldilocks tables)
  template <int k, int LO>
  __device__ __forceinline__ static void twist_all2(const PassArgs &a, const Tile &t, u64 (&x)[E],
                                                    const u32 (&I0)[E >> k], const TwistFactors &f0,
                                                    const TwistFactors &f1, u32 (&zr)[4], const AsmConsts &c) {
    const TwistFactors f2 = twist_load<k, LO, 2>(a, t, I0);
    twist_apply<0>(x, f0, zr, c);
    const TwistFactors f3 = twist_load<k, LO, 3>(a, t, I0);
    twist_apply<1>(x, f1, zr, c);
    twist_apply<2>(x, f2, zr, c);
    twist_apply<3>(x, f3, zr, c);
  }

  // Operands of the fused pointwise product (ROW forward with FLAG) for x[4 GRP .. 4 GRP + 3].
  struct Operands {
    u64 v[4];
  };
  template <int k, int LO, int GRP>
  __device__ __forceinline__ static Operands epilogue_load(const PassArgs &a, const Tile &t,
                                                           const u32 (&I0)[E >> k]) {
    constexpr int R = 1 << k, lo = F0 + LO;
    Operands o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * GRP + q, g = i >> k, v = i & (R - 1);
      const u32 I = I0[g] | ((u32)v << lo);
      o.v[q] = in_range(a, t, I) ? a.epilogue[gaddr(a, t, I)] : 0;
    }
    return o;
  }

  // The same step with all G sets of the thread held at once (x[g*R + v]) and the stages run
  // by the assembly groups of stage_asm.inc, four butterflies per statement.  Loads are issued
  // in source order between the (volatile) assembly statements, so everything that comes from
  // a table is requested one statement group ahead of its use.  SYNC (sync_before): what separates
  // the step from the one that wrote the LDS image it reads; it comes after the first twiddle requests.
  template <int SI, int SYNC>
  __device__ __forceinline__ static void step_asm(const PassArgs &a, const Tile &t, u32 tid, u64 *lds) {
    constexpr int k = STEPS::k[SI];
    constexpr int HI = LOGL - STEPS::sum(SI);
    constexpr int LO = HI - k;
    constexpr int lo = F0 + LO, hi = F0 + HI;
    constexpr int R = 1 << k;
    constexpr int G = E >> k;
    static_assert(E == 16, "assembly stages are generated for 16 elements per thread");
    constexpr bool from_hbm = (MODE == MODE_FWD) ? (SI == 0) : (SI == NSTEPS - 1);
    constexpr bool to_hbm = (MODE == MODE_FWD) ? (SI == NSTEPS - 1) : (SI == 0);
    constexpr bool twist_in = COL && FLAG && MODE == MODE_INV && from_hbm;
    constexpr bool twist_out = COL && FLAG && MODE == MODE_FWD && to_hbm;
    constexpr bool multiply_out = !COL && FLAG && MODE == MODE_FWD && to_hbm;
    // c.save: EXEC on entry = all lanes of a live workgroup.  The assembly groups restore EXEC from it after
    // their masked corrections, so they must stay in uniform control flow: never inside a divergent branch
    // (gen_stage_asm.py: "EXEC invariant").
    const AsmConsts c{a.f.N, a.f.negN, (u32)a.f.N, (u32)(a.f.N >> 32), (u32)a.f.Ninv,
                      (u32)(a.f.Ninv >> 32), __builtin_amdgcn_read_exec(), (u32)a.f.negN,
                      (u32)(a.f.negN >> 32)};
    u32 zr[4] = {0u, 0u, 0u, 0u};  // the slots' zero-extension high halves (stage_asm.inc)
    u64 x[E];
    u32 I0[G], s_low[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const u32 s = set_number<SI>(tid, g);
      s_low[g] = s & ((1u << lo) - 1u);
      I0[g] = ((s >> lo) << hi) | s_low[g];
    }
    [[maybe_unused]] constexpr int TR = 6 * ((MODE == MODE_FWD) ? SI : NSTEPS - 1 - SI);  // (trace slots of this step)
    if constexpr (from_hbm && SVENTT_PRIO_LOAD > 0) __builtin_amdgcn_s_setprio(SVENTT_PRIO_LOAD);
    SVENTT_STAMP(lds, TR + 0);
    // the table prefix the middle steps read from LDS: asked for first, so that it arrives first
    // (asking for the data first instead was measured: 199.0 against 196.1 us, profiles/r02/asm_stages_ab.txt)
    constexpr bool tw_fill = TW_LDS_WORDS > 0 && from_hbm;
    u64 twl[TW_LDS_PER_THREAD > 0 ? TW_LDS_PER_THREAD : 1];
    if constexpr (tw_fill) {
#pragma unroll
      for (u32 i = 0; i < TW_LDS_PER_THREAD; ++i) {
        const u32 idx = tid + i * (u32)NT;
        twl[i] = idx < TW_LDS_WORDS ? a.stage_tw[idx] : 0;
      }
    }
    // first stage's twiddles: from a table in HBM/L2 they are requested ahead of the barrier and
    // of the data; from the LDS copy after the barrier (which may be what publishes the copy)
    constexpr bool tw_in_lds = TW_LDS_WORDS > 0 && LO > 0 && LO + k <= TW_LDS_P;
    GroupTwiddles w0, w1;
    if constexpr (!tw_in_lds) {
      w0 = group_twiddles<k, LO, lo, 0, 0>(a, s_low, lds);
      w1 = group_twiddles<k, LO, lo, 0, 1>(a, s_low, lds);
    }
#if !defined(SVENTT_TWIST_IN_EARLY)
#define SVENTT_TWIST_IN_EARLY 0  // 1: an inverse column tile asks for its first twist factors BEFORE its data (A/B)
#endif
    TwistFactors fin0, fin1;
    if constexpr (twist_in && SVENTT_TWIST_IN_EARLY != 0) fin0 = twist_load<k, LO, 0>(a, t, I0);
    if constexpr (twist_in && SVENTT_TWIST_IN_EARLY == 2 && ARITH != ARITH_SHOUP) fin1 = twist_load<k, LO, 1>(a, t, I0);
    exchange_sync<SYNC>();
    SVENTT_STAMP(lds, TR + 1);
    if constexpr (tw_in_lds && TWW == 1) {  // (two-word twiddles are read group by group: stages_asm)
      w0 = group_twiddles<k, LO, lo, 0, 0>(a, s_low, lds);
      w1 = group_twiddles<k, LO, lo, 0, 1>(a, s_low, lds);
    }
    // ---- gather ------------------------------------------------------------
    if constexpr (from_hbm && pair_sets<SI>()) {
      // the two sets are neighbours in memory: eight 16-byte loads
      const u64 vstride = COL ? (a.src_istride << LO) : (1ull << lo);
      const u64 *p0 = a.src + saddr(a, t, I0[0]);
#pragma unroll
      for (int v = 0; v < R; ++v) {
        const ulonglong2 two = *reinterpret_cast<const ulonglong2 *>(p0 + src_off<LO>(a, v, vstride));
        x[v] = two.x, x[R + v] = two.y;
      }
    } else
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if constexpr (from_hbm) {
        const u64 vstride = COL ? (a.src_istride << LO) : (1ull << lo);
        const u64 *p0 = a.src + saddr(a, t, I0[g]);
#pragma unroll
        for (int v = 0; v < R; ++v)
#if defined(SVENTT_STUB_HBM) || defined(SVENTT_STUB_LOADS)  // analysis builds only (tools/build_variant.sh)
          x[g * R + v] = (u64)(uintptr_t)(p0 + src_off<LO>(a, v, vstride)) >> 1;
#else
          x[g * R + v] = in_range(a, t, I0[g] | ((u32)v << lo)) ? p0[src_off<LO>(a, v, vstride)] : 0;
#endif
      } else {
        const u32 P0 = lds_phys(I0[g]);
#pragma unroll
        for (int v = 0; v < R; ++v) x[g * R + v] = lds[P0 ^ lds_phys((u32)v << lo)];
      }
    }
    SVENTT_STAMP(lds, TR + 2);
    if constexpr (from_hbm && SVENTT_PRIO_LOAD > 0) __builtin_amdgcn_s_setprio(0);
    if constexpr (tw_fill) {
      // (the data loads above are in flight; every wave is waiting for them anyway)
#pragma unroll
      for (u32 i = 0; i < TW_LDS_PER_THREAD; ++i) {
        const u32 idx = tid + i * (u32)NT;
        if (idx < TW_LDS_WORDS) lds[((size_t)1 << LOGT) + idx] = twl[i];
      }
      // forward: the workgroup barrier behind this (top) step comes before the first reader
      // (step_asm asks for LDS twiddles after its exchange_sync) -- unless every step of the tile
      // exchanges inside its wave.  inverse: the next steps are wave-local, no barrier comes in time.
      if constexpr (MODE == MODE_INV || sync_before<1>() != SYNC_GROUP) __syncthreads();
    }
#if defined(SVENTT_TRACE)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the data (and everything asked for before it) is here
    SVENTT_STAMP(lds, TR + 3);
#endif
    if constexpr (twist_in && SVENTT_TWIST_IN_EARLY == 2 && ARITH != ARITH_SHOUP)
      twist_all2<k, LO>(a, t, x, I0, fin0, fin1, zr, c);
    else if constexpr (twist_in && SVENTT_TWIST_IN_EARLY != 0)
      twist_all<k, LO>(a, t, x, I0, fin0, zr, c);
    else if constexpr (twist_in)
      twist_all<k, LO>(a, t, x, I0, twist_load<k, LO, 0>(a, t, I0), zr, c);
    // ---- k fused stages ------------------------------------------------------
    TwistFactors f0;
    Operands o0;
    // A wave cannot retire before its stores are acknowledged (s_endpgm waits for vmcnt = 0) and the
    // workgroup's LDS is not free before its last wave retires (a build without the stores runs
    // 10-15 % faster).  Storing each element as soon as it is final leaves only the last few stores
    // in flight at the end.
    // Measured (profiles/r02/asm_stages_ab.txt): it pays for the inverse column pass (-4 %), whose last
    // stage is a full butterfly stage; the forward passes end on a twist or a short stage and lose
    // more to the unmerged stores than they gain, so they keep storing at the end.
    constexpr bool early = SVENTT_EARLY_STORES != 0 && to_hbm && COL && MODE == MODE_INV;
    stages_asm<k, LO, lo, 0, twist_out, multiply_out, early>(a, t, lds, x, I0, s_low, w0, w1, f0, o0, zr, c);
    SVENTT_STAMP(lds, TR + 4);
    // ---- scatter -------------------------------------------------------------
    if constexpr (to_hbm) {
      if constexpr (twist_out) twist_all<k, LO>(a, t, x, I0, f0, zr, c);
      if constexpr (SVENTT_PRIO_STORE > 0) __builtin_amdgcn_s_setprio(SVENTT_PRIO_STORE);
      if constexpr (multiply_out) {
        const Operands o1 = epilogue_load<k, LO, 1>(a, t, I0);
        MontGroup<0>::run(x, o0.v[0], o0.v[1], o0.v[2], o0.v[3], zr, c);
        const Operands o2 = epilogue_load<k, LO, 2>(a, t, I0);
        MontGroup<1>::run(x, o1.v[0], o1.v[1], o1.v[2], o1.v[3], zr, c);
        const Operands o3 = epilogue_load<k, LO, 3>(a, t, I0);
        MontGroup<2>::run(x, o2.v[0], o2.v[1], o2.v[2], o2.v[3], zr, c);
        MontGroup<3>::run(x, o3.v[0], o3.v[1], o3.v[2], o3.v[3], zr, c);
      }
      if constexpr (!early && pair_sets<SI>()) {
        const u64 vstride = COL ? (a.istride << LO) : (1ull << lo);
        u64 *p0 = a.dst + gaddr(a, t, I0[0]);
#pragma unroll
        for (int v = 0; v < R; ++v) {
          ulonglong2 two;
          two.x = x[v], two.y = x[R + v];
          *reinterpret_cast<ulonglong2 *>(p0 + dst_off<LO>(a, v, vstride)) = two;
        }
      } else if constexpr (!early) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const u64 vstride = COL ? (a.istride << LO) : (1ull << lo);
          u64 *p0 = a.dst + gaddr(a, t, I0[g]);
#pragma unroll
          for (int v = 0; v < R; ++v)
#if defined(SVENTT_STUB_HBM) || defined(SVENTT_STUB_STORES)
            if (x[g * R + v] == 0x123456789abcdefull) p0[dst_off<LO>(a, v, vstride)] = x[g * R + v];
#else
            if (in_range(a, t, I0[g] | ((u32)v << lo))) p0[dst_off<LO>(a, v, vstride)] = x[g * R + v];
#endif
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const u32 P0 = lds_phys(I0[g]);
#pragma unroll
        for (int v = 0; v < R; ++v) lds[P0 ^ lds_phys((u32)v << lo)] = x[g * R + v];
      }
    }
    SVENTT_STAMP(lds, TR + 5);
  }

  // A lowest step's twiddles with index 0 are omega^0 and their butterflies multiply by nothing
  // -- unless the stage is the top stage of an inverse ROW pass that folds 1/L, whose table
  // holds (1/L) * omega^j (transforms of 2 and 4 points have both in one step).
  template <int LO, int ps> static constexpr bool stage_has_unit_twiddles() {
    return LO == 0 && !(!COL && FLAG && MODE == MODE_INV && ps == LOGL - 1);
  }

  // Twiddles of the GRP-th four butterflies of stage rr of the step (forward walks the stage
  // bits downwards, inverse upwards); omega^0 entries of a lowest step are not loaded.
  struct GroupTwiddles {
    u64 w[4];
    u64 p[4];  // ARITH_SHOUP: the precomputed companions w' = floor(w * 2^64 / N)
  };
  template <int k, int LO, int lo, int rr, int GRP>
  __device__ __forceinline__ static GroupTwiddles group_twiddles(const PassArgs &a, const u32 (&s_low)[E >> k],
                                                                 const u64 *lds) {
    constexpr int r = (MODE == MODE_FWD) ? (k - 1 - rr) : rr;
    constexpr int ps = LO + r;
    constexpr bool triv = stage_has_unit_twiddles<LO, ps>();
    // a middle step (below the top one, indices not compile-time constants) reads the LDS copy
    constexpr bool in_lds = TW_LDS_WORDS > 0 && LO > 0 && LO + k <= TW_LDS_P;
    const u64 *tab = (in_lds ? lds + ((size_t)1 << LOGT) : a.stage_tw) + (size_t)((1u << ps) - 1u) * TWW;
    GroupTwiddles tw;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = bf_first(r, 4 * GRP + q);
      const int g = i >> k;
      const u32 vlow = (u32)(i & ((1 << r) - 1));
      const u32 j = ((vlow << lo) | s_low[g]) >> F0;
      if (triv && vlow == 0) {
        tw.w[q] = 0;
        tw.p[q] = 0;
      } else if constexpr (TWW == 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(tab + (size_t)j * 2);
        tw.w[q] = v.x, tw.p[q] = v.y;
      } else {
        tw.w[q] = tab[j];
        tw.p[q] = 0;
      }
    }
    return tw;
  }

  // the eight elements of the GRP-th four butterflies of stage bit r, to HBM
  template <int k, int LO, int r, int GRP>
  __device__ __forceinline__ static void store_group(const PassArgs &a, const Tile &t, const u64 (&x)[E],
                                                     const u32 (&I0)[E >> k]) {
    constexpr int i0 = bf_first(r, 4 * GRP), i1 = bf_first(r, 4 * GRP + 1), i2 = bf_first(r, 4 * GRP + 2),
                  i3 = bf_first(r, 4 * GRP + 3), h = 1 << r;
    store_range<k, LO, i0, 1>(a, t, x, I0);
    store_range<k, LO, i0 + h, 1>(a, t, x, I0);
    store_range<k, LO, i1, 1>(a, t, x, I0);
    store_range<k, LO, i1 + h, 1>(a, t, x, I0);
    store_range<k, LO, i2, 1>(a, t, x, I0);
    store_range<k, LO, i2 + h, 1>(a, t, x, I0);
    store_range<k, LO, i3, 1>(a, t, x, I0);
    store_range<k, LO, i3 + h, 1>(a, t, x, I0);
  }

  // Stage rr of the step: two assembly groups of four butterflies.  Each group's twiddles were
  // requested a full stage earlier (w0 / w1 on entry): the same group of the NEXT stage is
  // requested right before this stage's group runs, so its L2 latency hides behind ~300 VALU
  // instructions per thread while three groups' worth of twiddle registers are live (24, not 32).
  // Ahead of the step's last group the first quarter of what the scatter multiplies by is
  // requested too (TW: twist factors into f0, MUL: operands of the fused product into o0).
  template <int k, int LO, int lo, int rr, bool TW, bool MUL, bool SCATTER = false>
  __device__ __forceinline__ static void stages_asm(const PassArgs &a, const Tile &t, const u64 *lds, u64 (&x)[E],
                                                    const u32 (&I0)[E >> k], const u32 (&s_low)[E >> k],
                                                    const GroupTwiddles &w0, const GroupTwiddles &w1,
                                                    TwistFactors &f0, Operands &o0, u32 (&zr)[4],
                                                    const AsmConsts &c) {
    constexpr int r = (MODE == MODE_FWD) ? (k - 1 - rr) : rr;
    constexpr int ps = LO + r;
    constexpr bool triv = stage_has_unit_twiddles<LO, ps>();
    constexpr bool more = rr + 1 < k;
    if constexpr (!COL && FLAG && MODE == MODE_INV && ps == LOGL - 1) {
      // fold 1/L into the top stage: (c*x0) +- (c*w)*x1, the table holds c*w
      ScaleGroup<r, 0>::run(x, a.scale, zr, c);
      ScaleGroup<r, 1>::run(x, a.scale, zr, c);
    }
    // Two-word twiddles (ARITH_SHOUP) take 16 registers per group: three groups in flight do not
    // fit beside the data and the temporaries (30-64 spilled VGPRs).  From the LDS copy they are
    // read right before their group (one group live), from the table in L2 one GROUP ahead
    // instead of one stage (two groups live).
    constexpr bool short_lookahead = TWW == 2;
    constexpr bool jit = TWW == 2 && TW_LDS_WORDS > 0 && LO > 0 && LO + k <= TW_LDS_P;
    GroupTwiddles n0, n1, j0, j1;
    if constexpr (jit) j0 = group_twiddles<k, LO, lo, rr, 0>(a, s_low, lds);
    const GroupTwiddles &u0 = jit ? j0 : w0;
    if constexpr (more && !short_lookahead) n0 = group_twiddles<k, LO, lo, rr + 1, 0>(a, s_low, lds);
    BflyGroup<ARITH, MODE, r, 0, triv>::run(x, u0.w[0], u0.w[1], u0.w[2], u0.w[3], u0.p[0], u0.p[1], u0.p[2],
                                            u0.p[3], zr, c);
    if constexpr (jit) j1 = group_twiddles<k, LO, lo, rr, 1>(a, s_low, lds);
    const GroupTwiddles &u1 = jit ? j1 : w1;
    if constexpr (more && short_lookahead && !jit) n0 = group_twiddles<k, LO, lo, rr + 1, 0>(a, s_low, lds);
    if constexpr (more && !short_lookahead) n1 = group_twiddles<k, LO, lo, rr + 1, 1>(a, s_low, lds);
    if constexpr (!more && TW) f0 = twist_load<k, LO, 0>(a, t, I0);
    if constexpr (!more && MUL) o0 = epilogue_load<k, LO, 0>(a, t, I0);
    if constexpr (!more && SCATTER) store_group<k, LO, r, 0>(a, t, x, I0);  // final: out they go
    BflyGroup<ARITH, MODE, r, 1, triv>::run(x, u1.w[0], u1.w[1], u1.w[2], u1.w[3], u1.p[0], u1.p[1], u1.p[2],
                                            u1.p[3], zr, c);
    if constexpr (!more && SCATTER) store_group<k, LO, r, 1>(a, t, x, I0);
    if constexpr (more && short_lookahead && !jit) n1 = group_twiddles<k, LO, lo, rr + 1, 1>(a, s_low, lds);
    if constexpr (more)
      stages_asm<k, LO, lo, rr + 1, TW, MUL, SCATTER>(a, t, lds, x, I0, s_low, n0, n1, f0, o0, zr, c);
  }
#endif  // SVENTT_STAGE_ASM

  // One step for one thread.  `first`/`last` say whether this step touches HBM.
  // SYNC (device only): sync_before<SI>() when the step follows another one of the same tile.
  template <int SI, int SYNC = SYNC_NONE>
  F64_HD static void step(const PassArgs &a, const Tile &t, u32 tid, u64 *lds) {
#if defined(SVENTT_STAGE_ASM)
    if constexpr (LOGE == 4) {
      step_asm<SI, SYNC>(a, t, tid, lds);
      return;
    }
#endif
    exchange_sync<SYNC>();
    constexpr int k = STEPS::k[SI];
    constexpr int HI = LOGL - STEPS::sum(SI);  // field-relative top bit (exclusive)
    constexpr int LO = HI - k;
    constexpr int lo = F0 + LO, hi = F0 + HI;
    constexpr int R = 1 << k;
    constexpr int G = E >> k;
    static_assert(k >= 1 && k <= LOGE, "step radix out of range");
    constexpr bool from_hbm = (MODE == MODE_FWD) ? (SI == 0) : (SI == NSTEPS - 1);
    constexpr bool to_hbm = (MODE == MODE_FWD) ? (SI == NSTEPS - 1) : (SI == 0);

    u64 x[E];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const u32 s = set_number<SI>(tid, g);
      const u32 s_low = s & ((1u << lo) - 1u);
      const u32 I0 = ((s >> lo) << hi) | s_low;
      // lds_phys is XOR-linear and I0 has zeros where v goes, so
      // phys(I0 | v<<lo) = phys(I0) ^ phys(v<<lo): one v_xor with a literal per element
      const u32 P0 = lds_phys(I0);
      // HBM side: the set's elements sit a uniform stride apart (rows of the block for a
      // COL tile), so one 64-bit address per set and constant multiples of the stride.
      constexpr bool twisted = COL && FLAG;
      u32 tw_col = 0, tw_e0 = 0;
      if constexpr (twisted && (from_hbm || to_hbm)) {
        tw_col = twist_col(a, t, I0);
        tw_e0 = twist_e0(tw_col, I0);
      }
      // ---- gather ----------------------------------------------------------
      if constexpr (from_hbm) {
        const u64 vstride = COL ? (a.src_istride << LO) : (1ull << lo);
        const u64 *p0 = a.src + saddr(a, t, I0);
#pragma unroll
        for (int v = 0; v < R; ++v)
          x[g * R + v] = in_range(a, t, I0 | ((u32)v << lo)) ? p0[src_off<LO>(a, v, vstride)] : 0;
        if constexpr (twisted && MODE == MODE_INV) {
#pragma unroll
          for (int v = 0; v < R; ++v)
            x[g * R + v] = twist_apply_cxx(a, x[g * R + v], tw_e0 + tw_col * twist_bv(v, LO));
        }
      } else {
#pragma unroll
        for (int v = 0; v < R; ++v) x[g * R + v] = lds[P0 ^ lds_phys((u32)v << lo)];
      }
      // ---- k fused stages ----------------------------------------------------
#pragma unroll
      for (int rr = 0; rr < k; ++rr) {
        // forward walks the stage bits downwards, inverse upwards
        const int r = (MODE == MODE_FWD) ? (k - 1 - rr) : rr;
        const int ps = LO + r;  // stage bit inside the transform: span 2^ps
        const u64 *tab = a.stage_tw + (size_t)((1u << ps) - 1u) * TWW;
#pragma unroll
        for (int v = 0; v < R; ++v) {
          if (v & (1 << r)) continue;
          const u32 vlow = (u32)v & ((1u << r) - 1u);
          const u32 j = ((vlow << lo) | s_low) >> F0;
          const bool trivial = (LO == 0) && (vlow == 0);  // omega^0, known at compile time
          u64 &x0 = x[g * R + v], &x1 = x[g * R + v + (1 << r)];
          if constexpr (MODE == MODE_FWD) {
            if (trivial)
              butterfly_fwd(x0, x1, a.f);
            else
              butterfly_fwd_tw<ARITH>(x0, x1, tab + (size_t)j * TWW, a.f);
          } else {
            if (!COL && FLAG && ps == LOGL - 1) {
              // fold 1/L into the top stage: (c*x0) +- (c*w)*x1, table holds c*w
              x0 = montmul(x0, a.scale, a.f);
              butterfly_inv_tw<ARITH>(x0, x1, tab + (size_t)j * TWW, a.f);
            } else if (trivial) {
              butterfly_inv(x0, x1, a.f);
            } else {
              butterfly_inv_tw<ARITH>(x0, x1, tab + (size_t)j * TWW, a.f);
            }
          }
        }
      }
      // ---- scatter -----------------------------------------------------------
      if constexpr (to_hbm) {
        const u64 vstride = COL ? (a.istride << LO) : (1ull << lo);
        u64 *p0 = a.dst + gaddr(a, t, I0);
        if constexpr (twisted && MODE == MODE_FWD) {
#pragma unroll
          for (int v = 0; v < R; ++v)
            x[g * R + v] = twist_apply_cxx(a, x[g * R + v], tw_e0 + tw_col * twist_bv(v, LO));
        }
        if constexpr (!COL && FLAG && MODE == MODE_FWD) {
          const u64 *e0 = a.epilogue + gaddr(a, t, I0);
          u64 op[R];
#pragma unroll
          for (int v = 0; v < R; ++v)
            op[v] = in_range(a, t, I0 | ((u32)v << lo)) ? e0[dst_off<LO>(a, v, vstride)] : 0;
#pragma unroll
          for (int v = 0; v < R; ++v) x[g * R + v] = montmul(x[g * R + v], op[v], a.f);
        }
#pragma unroll
        for (int v = 0; v < R; ++v)
          if (in_range(a, t, I0 | ((u32)v << lo))) p0[dst_off<LO>(a, v, vstride)] = x[g * R + v];
      } else {
#pragma unroll
        for (int v = 0; v < R; ++v) lds[P0 ^ lds_phys((u32)v << lo)] = x[g * R + v];
      }
    }
  }
};

}  // namespace sventt_hip
