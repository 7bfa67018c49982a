// sve_ntt_amd/csrc/plan_core.h -- host-only plan description (no HIP calls).
//
// A plan is the runtime form of the reference's compile-time kernel_type
// (kernel/recursive.hpp:15-17 over layer/sve/[blocked-]generic.hpp): a list of
// passes.  Forward:
//     COL(L0, S0) -> COL(L1, S1) -> ... -> ROW(Lk)       n = L0*L1*...*Lk
// COL(L, S): for every contiguous block of M = L*S elements viewed as L rows x
// S columns, run the S column transforms of length L in place and multiply
// row j by omega_M^(bitrev_L(j)*c) (six-step twiddle, layer/sve/generic.hpp:
// 95-105).  ROW(L): transform every contiguous run of L elements.  Each pass
// reads and writes every element exactly once (16 bytes of HBM traffic per
// element per pass).  The inverse plan is the mirror image with inverse roots;
// the 1/n scaling rides on a multiplication that exists anyway (the outermost
// inverse twiddle, or the top stage of a single-pass plan).
//
// plan.hip uploads these tables and launches kernels; tests/cpu_sim replays the
// same passes sequentially on the host (test-only).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "registry.h"

namespace sventt_hip {

struct HostPass {
  int kind = KIND_ROW, logl = 0;
  bool inverse = false, flag = false;
  u64 istride = 1;  // COL: S
  u64 block = 0;    // COL: M = L*S
  // gather/scatter passes of the sharded transform address the two sides differently
  u64 dst_istride = 0, dst_ostride = 0, src_istride = 0, src_ostride = 0;
  // two-level passes (tile_ntt.h: TWOLVL; the sharded row transform's first pass when it is longer
  // than the rank count): row i lies at (i >> row_split) * X_istride_hi + (i mod 2^row_split) * X_istride
  int two_level = 0;
  u32 row_split = 0;
  u64 dst_istride_hi = 0, src_istride_hi = 0;
  // sharded column pass: the columns of exchange chunk k of K are, in every run of chunk_period
  // columns, the k-th K-th (0: one run, i.e. K contiguous column ranges) -- make_chunk_args
  u64 chunk_period = 0;
  u64 grid = 0;
  int f0 = 0, logt = 0;
  int loge = REG_LOGE;  // elements per thread (log2): REG_LOGE, or FINE_LOGE for small totals
  int arith = ARITH_MONT;  // back end of the butterflies and the twist; fixes the table format
  std::vector<u64> stage, twist_lo, twist_hi;
  u32 twist_shift = 0;
  u64 twist_col_offset = 0;
  u64 scale = 0;  // Montgomery form
};

struct HostPlan {
  Field f{};
  u64 g = 0, n = 0, batch = 1, total = 0;
  u32 flags = 0;
  u64 r2 = 0;  // 2^128 mod N
  u64 inverse_scale = 1;  // what the inverse multiplies by (plain residue): n^{-1} unless asked otherwise
  std::vector<HostPass> fwd, inv;
  int arith = ARITH_MONT;  // field64.h: chosen from the modulus and the plan flags
  bool fine = false;  // E = 4 tiles (registry.h): n * batch too small to fill the chip otherwise
  bool sharded = false;
  int rank = 0, nranks = 1;
  u64 local_cols = 0;
};

enum : int {
  PLAN_OK = 0,
  PLAN_ERR_INVALID_ARGUMENT = -1,
  PLAN_ERR_LOGIC = -4,
};
enum : u32 {
  PLAN_FORWARD = 1u,
  PLAN_INVERSE = 2u,
  PLAN_DEVICE_POINTERS = 4u,
  PLAN_GENERIC_ARITHMETIC = 8u,  // Montgomery kernels even where a special back end exists
  PLAN_FIXED_POINT = 16u,        // the FixedPoint64 (Shoup) back end; needs N < 2^63
  PLAN_KNOWN_FLAGS = 31u
};

// Which back end a plan's kernels use (field64.h).  SVENTT_ARITH=mont|gold|shoup overrides the
// choice for A/B measurements when the modulus allows it.
inline int choose_arith(u64 p, u32 flags, std::string &err, int &arith) {
  arith = ARITH_MONT;
  if (flags & PLAN_FIXED_POINT) {
    if (flags & PLAN_GENERIC_ARITHMETIC) {
      err = "SVENTT_GENERIC_ARITHMETIC and SVENTT_FIXED_POINT exclude each other";
      return -1;
    }
    if (p >> 63) {
      err = "the FixedPoint64 back end needs a modulus below 2^63";  // c = a*w - q*N must fit [0, 2N)
      return -1;
    }
    arith = ARITH_SHOUP;
  } else if (!(flags & PLAN_GENERIC_ARITHMETIC) && p == GOLDILOCKS_N) {
    arith = ARITH_GOLD;
  }
  if (const char *e = std::getenv("SVENTT_ARITH")) {
    const std::string v(e);
    if (v == "mont") arith = ARITH_MONT;
    if (v == "gold" && p == GOLDILOCKS_N) arith = ARITH_GOLD;
    if (v == "shoup" && !(p >> 63)) arith = ARITH_SHOUP;
  }
  return 0;
}

// Tuning knobs (environment, read once).  SVENTT_COL_SLIM=0/1: 4-column instead of
// 8-column tiles for column passes of length >= 2^10 (two workgroups per CU).
// (Tried and dropped in r01: keeping the six-step twiddles of a whole block as one
// table in HBM instead of composing them from two small tables per element.  It
// saves ~40 VALU instructions per element but the extra 8 B/element of HBM reads
// made the column pass 8-12 % slower.)
struct Tuning {
  bool col_slim;
  int max_col_logl;
  bool fine;  // SVENTT_FINE=0 disables the E = 4 tiles for small transforms
  int fine_max_total_log2;  // SVENTT_FINE_MAX_LOG2: largest n*batch (log2) that runs on them
  int fine_max_two_pass_log2;  // ... when the transform takes two passes (the same variable sets both)
  int twist_lo_log2;        // SVENTT_TWIST_LO_LOG2: cap on the low twist table (0: balanced split)
  bool sharded_fuse;        // SVENTT_SHARDED_FUSE=0: the sharded row phase gathers in a pass of its own (r02)
};
inline const Tuning &tuning(void) {
  static const Tuning t = [] {
    Tuning x;
    const char *e = std::getenv("SVENTT_COL_SLIM");
    x.col_slim = e ? (std::atoi(e) != 0) : true;  // r01: 2^24 forward column pass 150 -> 138 us
    x.max_col_logl = x.col_slim ? 12 : MAX_COL_LOGL;
    const char *fe = std::getenv("SVENTT_FINE");
    x.fine = fe ? (std::atoi(fe) != 0) : true;
    const char *fm = std::getenv("SVENTT_FINE_MAX_LOG2");
    x.fine_max_total_log2 = fm ? std::atoi(fm) : MAX_FINE_TOTAL_LOG2;
    // r02: at n*batch = 2^21 the 2^12/2^13-element tiles win for two-pass transforms (2^21: 35.0 against
    // 38.4 us, 2^17 x 16: 30.4 against 33.1), the fine ones for single rows (2^10 x 2048: 14.9 against 16.2)
    x.fine_max_two_pass_log2 = fm ? std::atoi(fm) : MAX_FINE_TOTAL_LOG2 - 1;
    const char *tl = std::getenv("SVENTT_TWIST_LO_LOG2");
    // r01: a low table of 2^10 entries (8 KiB) stays in the vector L1 whatever the lanes ask
    // for; the balanced split (2^12..2^14 entries) cost the column pass 2 % at M = 2^24 and
    // 12 % at M = 2^27 (sharded, 8 ranks)
    x.twist_lo_log2 = tl ? std::atoi(tl) : 10;
    const char *sf = std::getenv("SVENTT_SHARDED_FUSE");
    x.sharded_fuse = sf ? (std::atoi(sf) != 0) : true;
    return x;
  }();
  return t;
}

inline bool is_pow2(u64 x) { return x && !(x & (x - 1)); }
inline int ilog2_u64(u64 x) {
  int l = 0;
  while ((x >> l) > 1) ++l;
  return l;
}

// stage tables of a length-2^logl transform: stage bit p at [2^p-1, 2^(p+1)-1),
// entry j = root_{2^(p+1)}^j in Montgomery form; `top_scale` (plain) multiplies
// the top stage's entries (inverse ROW passes that fold 1/L).
inline std::vector<u64> build_stage_table(const Field &f, u64 gen, int logl, bool inverse,
                                          u64 top_scale, int arith = ARITH_MONT) {
  const u64 N = f.N;
  const u64 L = 1ull << logl;
  std::vector<u64> t;  // entry (2^p - 1 + j) in the back end's format (h_push_twiddle)
  t.reserve((size_t)(L > 1 ? L - 1 : 1) * (arith == ARITH_SHOUP ? 2 : 1));
  for (int p = 0; p < logl; ++p) {
    const u64 order = 2ull << p;
    u64 w = h_powmod(gen, (N - 1) / order, N);
    if (inverse) w = h_invmod(w, N);
    u64 cur = (p == logl - 1) ? top_scale % N : 1;
    for (u64 j = 0; j < (1ull << p); ++j) {
      h_push_twiddle(t, arith, cur, N);
      cur = h_mulmod(cur, w, N);
    }
  }
  if (t.empty()) t.push_back(0);
  return t;
}

// omega_M^e = hi[e >> shift] * lo[e & mask]; `scale` (plain) is folded into hi.
inline void build_twist_tables(const Field &f, u64 gen, int logm, bool inverse, u64 scale,
                               std::vector<u64> &lo, std::vector<u64> &hi, u32 &shift,
                               int arith = ARITH_MONT) {
  const u64 N = f.N;
  u64 w = h_powmod(gen, (N - 1) >> logm, N);
  if (inverse) w = h_invmod(w, N);
  shift = (u32)((logm + 1) / 2);
  if (tuning().twist_lo_log2 > 0 && (int)shift > tuning().twist_lo_log2) shift = (u32)tuning().twist_lo_log2;
  const u64 nlo = 1ull << shift, nhi = 1ull << (logm - (int)shift);
  lo.clear();
  hi.clear();
  u64 cur = 1;
  for (u64 i = 0; i < nlo; ++i) {
    h_push_twiddle(lo, arith, cur, N);
    cur = h_mulmod(cur, w, N);
  }
  const u64 wbig = h_powmod(w, 1ull << shift, N);
  cur = scale % N;
  for (u64 i = 0; i < nhi; ++i) {
    h_push_twiddle(hi, arith, cur, N);
    cur = h_mulmod(cur, wbig, N);
  }
}

inline int make_host_pass(const HostPlan &pl, HostPass &ps, int kind, int logl, u64 S, bool inverse,
                          bool flag, u64 scale_plain, u64 col_offset, int twist_order_log2,
                          std::string &err) {
  const Field &f = pl.f;
  ps.kind = kind;
  ps.logl = logl;
  ps.inverse = inverse;
  ps.flag = flag;
  ps.istride = S;
  ps.block = S << logl;
  ps.dst_istride = ps.src_istride = S;
  ps.dst_ostride = ps.src_ostride = ps.block;
  ps.two_level = 0;
  ps.row_split = 0;
  ps.dst_istride_hi = ps.src_istride_hi = 0;
  ps.chunk_period = 0;
  ps.twist_col_offset = col_offset;
  ps.loge = pl.fine ? FINE_LOGE : REG_LOGE;
  ps.arith = pl.arith;
  const bool fold_row_scale = (kind == KIND_ROW) && inverse && flag;
  ps.stage = build_stage_table(f, pl.g, logl, inverse, fold_row_scale ? scale_plain : 1, pl.arith);
  if (fold_row_scale) ps.scale = h_to_montgomery(scale_plain % f.N, f.N);
  u64 tiles;
  if (kind == KIND_COL) {
    if (logl > (pl.fine ? MAX_FINE_COL_LOGL : tuning().max_col_logl)) {
      err = "column pass longer than one workgroup can hold";
      return PLAN_ERR_LOGIC;
    }
    if (twist_order_log2 > 32) {
      err = "transforms longer than 2^32 points are not supported";  // 32-bit twist exponents
      return PLAN_ERR_INVALID_ARGUMENT;
    }
    build_twist_tables(f, pl.g, twist_order_log2, inverse, inverse ? scale_plain : 1, ps.twist_lo,
                       ps.twist_hi, ps.twist_shift, pl.arith);
    ps.f0 = pl.fine ? registry_fine_col_f0(logl, ilog2_u64(S))
                    : registry_col_f0(logl, ilog2_u64(S), tuning().col_slim);
    if (ps.f0 < 0 || !is_pow2(S)) {
      // the reference rejects shapes its blocks do not divide the same way
      // (layer/sve/blocked-generic.hpp:111-116)
      err = "too few columns for a column pass of this length";
      return PLAN_ERR_INVALID_ARGUMENT;
    }
    ps.logt = ps.f0 + logl;
    const u64 T = 1ull << ps.f0;
    tiles = (pl.total / ps.block) * (S / T);
  } else {
    if (logl > (pl.fine ? MAX_FINE_ROW_LOGL : MAX_ROW_LOGL)) {
      err = "row pass longer than one workgroup can hold";
      return PLAN_ERR_LOGIC;
    }
    ps.f0 = 0;
    ps.logt = pl.fine ? registry_fine_row_logt(logl) : registry_row_logt(logl);
    const u64 tile = 1ull << ps.logt;
    tiles = (pl.total + tile - 1) / tile;
  }
  if (tiles > 0x7fffffffull) {
    err = "too many tiles for one launch";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  ps.grid = tiles;
  return PLAN_OK;
}

// Measured time of one pass over data that does not fit the Infinity Cache, in us per 2^24 elements
// on MI355X (tools/split3_search.py, profiles/r03/split3_search.txt: 2^28..2^31 points, every
// admissible three-pass split).  where = 0: the first column pass (rows tens of MiB apart: what
// counts is the width of the tile, T >= 32 columns = 256-byte segments up to 2^7; a 2^9 x T8 pass
// touches 512 different pages with 64 bytes each and takes 140 us), 1: a later column pass (rows
// 32-64 KiB apart; cheap up to 2^8, the 32-byte segments of the slim 2^10..2^12 tiles cost 40 % more),
// 2: the row pass.  r02 split the column stages evenly (2^30 as 9 | 8 | 13: 317 us per 2^24; now
// 7 | 11 | 12: 289).
inline int large_pass_cost(int where, int logl) {
  static const int first[13] = {0, 51, 56, 61, 73, 99, 92, 89, 105, 139, 180, 180, 180};
  static const int later[13] = {0, 66, 68, 70, 72, 74, 76, 78, 82, 98, 124, 118, 133};
  if (where == 2) return logl >= 13 ? 92 : 80;
  return (where == 0 ? first : later)[logl < 1 ? 1 : (logl > 12 ? 12 : logl)];
}

// Split log2(n) into pass lengths: every COL pass <= MAX_COL_LOGL, the final ROW
// pass <= MAX_ROW_LOGL.  n0_log2 (if non-zero) fixes the first COL pass (the R
// of the reference's n = R x C six-step, kernel/recursive.hpp:61-75).
// `large`: the whole array (n * batch) does not fit the Infinity Cache (>= 2^26 elements).
// `mid_rows12`: a single transform of 2^22 points (both directions) or 2^23 points (forward) runs with
// a 2^12 row pass (four workgroups per CU) and the 2^10 / 2^11 x T4 column pass instead of the longest
// row: 60.9 against 62.7 us and 101.8 against 106.7 us (tools/split2_search.py, profiles/r03/
// split2_search.txt; the inverse of 2^23 is 2.5 % slower that way and keeps 2^10 | 2^13 -- the two
// directions of a plan need not use the same split).
inline int choose_split(int logn, u32 n0_log2, std::vector<int> &cols, int &row, std::string &err,
                        bool fine = false, bool large = false, bool mid_rows12 = false) {
  cols.clear();
  if (fine) {  // at most two passes; the caller checked that the fine tiles cover the shape
    const int c = n0_log2 ? (int)n0_log2 : (logn <= MAX_FINE_ROW_LOGL ? 0 : logn / 2);
    if (c) cols.push_back(c);
    row = logn - c;
    return PLAN_OK;
  }
  // SVENTT_SPLIT="c0,c1,...,row" (log2 of every pass length; developer knob, tools/split3_search.py):
  // used when it fits this transform, ignored otherwise
  if (const char *e = std::getenv("SVENTT_SPLIT"); e && n0_log2 == 0) {
    std::vector<int> v;
    for (const char *q = e; *q;) {
      v.push_back(std::atoi(q));
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
    int sum = 0;
    bool ok = !v.empty();
    for (size_t i = 0; i < v.size(); ++i) {
      sum += v[i];
      ok = ok && v[i] >= 1 && v[i] <= (i + 1 == v.size() ? MAX_ROW_LOGL : tuning().max_col_logl);
    }
    if (ok && sum == logn) {
      row = v.back();
      cols.assign(v.begin(), v.end() - 1);
      return PLAN_OK;
    }
  }
  int rem = logn;
  if (n0_log2 != 0) {
    if ((int)n0_log2 >= logn || (int)n0_log2 > tuning().max_col_logl) {
      err = "n0_log2 out of range for this transform length";
      return PLAN_ERR_INVALID_ARGUMENT;
    }
    cols.push_back((int)n0_log2);
    rem -= (int)n0_log2;
  }
  if (rem <= MAX_ROW_LOGL && (n0_log2 != 0 || logn <= MAX_ROW_LOGL)) {
    row = rem;
    return PLAN_OK;
  }
  // Three passes (2^26 points and more: the data no longer fits the 256 MiB Infinity Cache and every
  // pass is a sweep of HBM): the split with the smallest modelled time.
  if (n0_log2 == 0 && rem > MAX_ROW_LOGL + tuning().max_col_logl) {
    int best = 1 << 30, ba = 0, bb = 0, br = 0;
    for (int r = MAX_ROW_LOGL - 1; r <= MAX_ROW_LOGL; ++r)
      for (int a = 1; a <= tuning().max_col_logl; ++a) {
        const int b = rem - r - a;
        if (b < 1 || b > tuning().max_col_logl) continue;
        const int c = large_pass_cost(0, a) + large_pass_cost(1, b) + large_pass_cost(2, r);
        if (c < best) best = c, ba = a, bb = b, br = r;
      }
    if (ba) {
      cols.push_back(ba);
      cols.push_back(bb);
      row = br;
      return PLAN_OK;
    }
  }
  // Widest smallest tile of the column pass is 2^3 columns: the row pass (the
  // block of contiguous elements below the innermost column pass) is >= 2^3.
  // The longest row pass wins (fully contiguous 512-byte wave accesses, the column pass keeps
  // whatever is left): 2^22 as 2^9 x 2^13 runs in 66 us against 73 us for 2^11 x 2^11; batches
  // of 2^14..2^21-point transforms (2^24 elements in all) are 3-12 % faster than with a balanced
  // split (tools/split_search.py, r01).  Columns shorter than 2^3 are not worth a 2^13 row.
  row = (rem - MAX_ROW_LOGL >= 3) ? MAX_ROW_LOGL : MAX_ROW_LOGL - 1;
  // out of the cache a 2^12 row pass is 12 us per 2^24 elements cheaper than a 2^13 one and a column
  // pass of up to 2^8 costs 2-4 us per extra stage (large_pass_cost): rows of 2^19 run as 7 | 12
  if (large && n0_log2 == 0 && rem > MAX_ROW_LOGL && rem - (MAX_ROW_LOGL - 1) <= tuning().max_col_logl &&
      large_pass_cost(1, rem - (MAX_ROW_LOGL - 1)) + large_pass_cost(2, MAX_ROW_LOGL - 1) <
          large_pass_cost(1, rem - MAX_ROW_LOGL) + large_pass_cost(2, MAX_ROW_LOGL))
    row = MAX_ROW_LOGL - 1;
  if (mid_rows12 && n0_log2 == 0) row = MAX_ROW_LOGL - 1;
  if (row > rem - 1) row = rem - 1;
  rem -= row;
  // (2^25 = 2^12 x 2^13 in two passes beats three: 581 vs 624 us forward, r01)
  const int max_col = tuning().max_col_logl;
  while (rem > 0) {
    const int npass = (rem + max_col - 1) / max_col;
    const int c = (rem + npass - 1) / npass;
    cols.push_back(c);
    rem -= c;
  }
  return PLAN_OK;
}

// Deterministic Miller-Rabin for 64-bit integers (the twelve smallest primes as witnesses
// decide every n < 3.3 * 10^24).  The tables are built with Fermat inverses, so a composite
// modulus would yield garbage silently; the reference's Modulus<p, g> is equally only
// meaningful for primes (modulus.hpp:14).
inline bool is_prime_u64(u64 n) {
  if (n < 2) return false;
  for (u64 q : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    if (n % q == 0) return n == q;
  }
  u64 d = n - 1;
  int s = 0;
  while ((d & 1) == 0) d >>= 1, ++s;
  for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    u64 x = h_powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool composite = true;
    for (int i = 1; i < s && composite; ++i) {
      x = h_mulmod(x, x, n);
      if (x == n - 1) composite = false;
    }
    if (composite) return false;
  }
  return true;
}

inline int validate_field(u64 p, u64 g, u64 n, std::string &err) {
  if (p < 3 || (p & 1) == 0 || !is_prime_u64(p)) {
    err = "modulus must be an odd prime";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  if (!is_pow2(n)) {
    err = "Transform length must be a power of two for now";  // tests/ntt-reference.hpp:38-40
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  if ((p - 1) % n != 0) {
    err = "the field has no such root";  // modulus.hpp:118-120
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  if (g == 0 || g >= p) {
    err = "generator out of range";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  if (n > 1) {
    const u64 w = h_powmod(g, (p - 1) / n, p);
    if (h_powmod(w, n / 2, p) != p - 1) {
      err = "g does not generate a root of unity of order n";
      return PLAN_ERR_INVALID_ARGUMENT;
    }
  }
  return PLAN_OK;
}

inline void init_field(HostPlan &pl, u64 p, u64 g) {
  pl.f.N = p;
  pl.f.Ninv = h_montgomery_inverse(p);
  pl.f.negN = 0 - p;
  pl.g = g;
  pl.r2 = h_to_montgomery(h_to_montgomery(1, p), p);
}

// inverse_divisor: the inverse transform multiplies by inverse_divisor^{-1} mod p; 0 stands for n
// (the oracle's 1/n, tests/ntt-reference.hpp:78-82), 1 for the unscaled inverse that the
// reference's layers compute when no layer carries an inverse_factor (layer/sve/radix-two.hpp:
// 208-235: only `inverse_factor != 1` adds the multiplication by its inverse).
inline int build_plan(HostPlan &pl, u64 p, u64 g, u64 n, u32 n0_log2, u64 batch, u32 flags,
                      std::string &err, u64 inverse_divisor = 0) {
  int rc = validate_field(p, g, n, err);
  if (rc) return rc;
  if (batch == 0) {
    err = "batch must be at least 1";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  if ((flags & (PLAN_FORWARD | PLAN_INVERSE)) == 0) {
    err = "neither direction enabled";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  if (flags & ~PLAN_KNOWN_FLAGS) {
    err = "unknown plan flag";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  if (n > (~0ull) / batch / 8) {
    err = "n*batch overflows";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  init_field(pl, p, g);
  if (choose_arith(p, flags, err, pl.arith)) return PLAN_ERR_INVALID_ARGUMENT;
  pl.n = n;
  pl.batch = batch;
  pl.total = n * batch;
  pl.flags = flags;
  const int logn = ilog2_u64(n);
  if (logn == 0) {  // length 1: the transform is the identity (times the inverse's scale)
    const u64 d1 = inverse_divisor ? inverse_divisor % p : 1;
    if (d1 == 0) {
      err = "inverse divisor is a multiple of the modulus";
      return PLAN_ERR_INVALID_ARGUMENT;
    }
    pl.inverse_scale = h_invmod(d1, p);
    return PLAN_OK;
  }
  std::vector<int> cols;
  int row = 0;
  // small totals run on the fine (E = 4) tiles, provided they cover the requested split
  pl.fine = tuning().fine && pl.arith == ARITH_MONT &&
            pl.total <= (1ull << (logn <= MAX_FINE_ROW_LOGL ? tuning().fine_max_total_log2
                                                           : tuning().fine_max_two_pass_log2)) &&
            logn <= MAX_FINE_COL_LOGL + MAX_FINE_ROW_LOGL &&
            // 2^12 and 2^13 are one pass on the 2^12/2^13-element tiles, two on the fine ones
            (n0_log2 != 0 || logn <= MAX_FINE_ROW_LOGL || logn > MAX_ROW_LOGL);
  if (pl.fine && n0_log2 != 0)
    pl.fine = (int)n0_log2 < logn && (int)n0_log2 <= MAX_FINE_COL_LOGL &&
              logn - (int)n0_log2 <= MAX_FINE_ROW_LOGL &&
              registry_fine_col_f0((int)n0_log2, logn - (int)n0_log2) >= 0;
  const bool large = pl.total >= (1ull << 26);
  if ((rc = choose_split(logn, n0_log2, cols, row, err, pl.fine, large,
                         batch == 1 && !pl.fine && (logn == 22 || logn == 23))))
    return rc;
  std::vector<int> icols;  // the inverse plan's own split (mirrored below)
  int irow = 0;
  if ((rc = choose_split(logn, n0_log2, icols, irow, err, pl.fine, large, batch == 1 && !pl.fine && logn == 22)))
    return rc;
  const u64 divisor = inverse_divisor ? inverse_divisor % p : n % p;
  if (divisor == 0) {
    err = "inverse divisor is a multiple of the modulus";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const u64 ninv = h_invmod(divisor, p);
  pl.inverse_scale = ninv;
  if (flags & PLAN_FORWARD) {
    int rem = logn;
    for (size_t i = 0; i < cols.size(); ++i) {
      pl.fwd.emplace_back();
      rem -= cols[i];
      if ((rc = make_host_pass(pl, pl.fwd.back(), KIND_COL, cols[i], 1ull << rem, false, true, 1, 0,
                               rem + cols[i], err)))
        return rc;
    }
    if (row > 0) {
      pl.fwd.emplace_back();
      if ((rc = make_host_pass(pl, pl.fwd.back(), KIND_ROW, row, 1, false, false, 1, 0, 0, err)))
        return rc;
    }
  }
  if (flags & PLAN_INVERSE) {
    // mirror: ROW first, then the COL passes innermost -> outermost
    if (irow > 0) {
      pl.inv.emplace_back();
      if ((rc = make_host_pass(pl, pl.inv.back(), KIND_ROW, irow, 1, true, icols.empty(),
                               icols.empty() ? ninv : 1, 0, 0, err)))
        return rc;
    }
    int rem = irow;
    for (size_t k = icols.size(); k-- > 0;) {
      pl.inv.emplace_back();
      const bool outermost = (k == 0);
      if ((rc = make_host_pass(pl, pl.inv.back(), KIND_COL, icols[k], 1ull << rem, true, true,
                               outermost ? ninv : 1, 0, rem + icols[k], err)))
        return rc;
      rem += icols[k];
    }
  }
  return PLAN_OK;
}

// How one rank's share of the ROW phase of the sharded six-step is cut into passes (both sharded
// planners call this: they must agree on what an exchange chunk is).  The row transform has length
// C = 2^logc and arrives as G = 2^logg pieces of C / G columns, one from every rank.  It runs as the
// ordinary plan of a length-C transform whose first column pass is at least G long: the piece index
// is the top logg bits of that pass's row index, so the pass reads the pieces where they were
// received and the gather costs no sweep of its own (kernel/recursive.hpp:61-75 of the reference:
// one column phase, one row phase).
//   cols[0] == logg: one row per piece, a plain strided pass;
//   cols[0]  > logg: 2^(cols[0] - logg) rows per piece, a two-level pass (TileNTT's TWOLVL).
// r02 always gathered in a length-G pass of its own (N = 2^30 on 8 ranks: col 2^3 | col 2^3 | row 2^13,
// four sweeps per rank with the column phase; now col 2^6 | row 2^13, three).
// A single rank (logg = 0; the whole pipeline on one GPU, exchange with itself) has one piece, which
// IS the row: its first pass is two-level only to run the same code as the multi-rank plans.
inline int sharded_row_split(int logc, int logg, std::vector<int> &cols, int &row, bool &two_level,
                             std::string &err, bool large = false) {
  int rc = choose_split(logc, 0, cols, row, err, false, large);
  if (rc) return rc;
  two_level = false;
  bool ok = tuning().sharded_fuse && !cols.empty() && cols[0] >= logg;
  if (ok && cols[0] > logg) {
    const int f0 = registry_col_f0(cols[0], logc - cols[0], tuning().col_slim);
    ok = two_level = f0 >= 0 && registry_has_two_level(cols[0], f0);
  }
  if (!ok && logg > 0) rc = choose_split(logc, (u32)logg, cols, row, err, false, large);  // a gather pass of its own, then the rest
  if (!rc && cols.empty()) {
    err = "rows too short for a column pass next to the exchange";  // (one rank, C <= 2^13)
    rc = PLAN_ERR_INVALID_ARGUMENT;
  }
  return rc;
}

// One rank's column pass of the sharded six-step (include/sventt_hip.h).
inline int build_sharded_plan(HostPlan &pl, u64 p, u64 g, u64 n, u32 r_log2, int rank, int nranks,
                              u32 flags, std::string &err) {
  int rc = validate_field(p, g, n, err);
  if (rc) return rc;
  if (nranks < 1 || rank < 0 || rank >= nranks || !is_pow2((u64)nranks)) {
    err = "rank/nranks invalid (nranks must be a power of two)";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const int logn = ilog2_u64(n);
  if (r_log2 == 0 || (int)r_log2 >= logn || (int)r_log2 > tuning().max_col_logl) {
    err = "r_log2 out of range";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const u64 C = n >> r_log2;
  if (C % (u64)nranks != 0) {
    err = "columns do not divide over the ranks";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  init_field(pl, p, g);
  if (choose_arith(p, flags, err, pl.arith)) return PLAN_ERR_INVALID_ARGUMENT;
  pl.n = n;
  pl.batch = 1;
  pl.flags = flags;
  pl.sharded = true;
  pl.rank = rank;
  pl.nranks = nranks;
  pl.local_cols = C / (u64)nranks;
  pl.total = pl.local_cols << r_log2;
  const u64 ninv = h_invmod(n % p, p);
  const u64 off = pl.local_cols * (u64)rank;
  // what an exchange chunk is depends on the first pass of the row phase (make_chunk_args)
  u64 chunk_period = 0;
  const int logc = logn - (int)r_log2, logg = ilog2_u64((u64)nranks);
  if (logg < logc && logg <= (int)r_log2) {
    std::vector<int> cols;
    int row = 0;
    bool two_level = false;
    std::string ignored;
    if (sharded_row_split(logc, logg, cols, row, two_level, ignored, pl.total >= (1ull << 26)) == PLAN_OK &&
        cols[0] > logg)
      chunk_period = C >> cols[0];
  }
  if (flags & PLAN_FORWARD) {
    pl.fwd.emplace_back();
    if ((rc = make_host_pass(pl, pl.fwd.back(), KIND_COL, (int)r_log2, pl.local_cols, false, true, 1,
                             off, logn, err)))
      return rc;
    pl.fwd.back().chunk_period = chunk_period;
  }
  if (flags & PLAN_INVERSE) {
    pl.inv.emplace_back();
    if ((rc = make_host_pass(pl, pl.inv.back(), KIND_COL, (int)r_log2, pl.local_cols, true, true,
                             ninv, off, logn, err)))
      return rc;
    pl.inv.back().chunk_period = chunk_period;
  }
  return PLAN_OK;
}

// Row phase of the sharded six-step on one rank (include/sventt_hip.h): the rank
// owns R/nranks rows of length C = nranks * Cl, delivered by the all-to-all as
// nranks pieces per row: piece s of local row q sits at recv[s][q][0..Cl).
// Forward = the passes of a length-C transform over the rank's R/nranks rows
// (sharded_row_split), the first of which reads that layout and writes whole rows.
// Inverse = the mirror, the last pass scattering back into piece layout.
// Never scales: the 1/n of the inverse rides on the column phase.
inline int build_sharded_rows_plan(HostPlan &pl, u64 p, u64 g, u64 n, u32 r_log2, int rank,
                                   int nranks, u32 flags, std::string &err) {
  int rc = validate_field(p, g, n, err);
  if (rc) return rc;
  if (nranks < 1 || rank < 0 || rank >= nranks || !is_pow2((u64)nranks)) {
    err = "rank/nranks invalid (nranks must be a power of two)";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const int logn = ilog2_u64(n), logg = ilog2_u64((u64)nranks);
  if (r_log2 == 0 || (int)r_log2 >= logn || (int)r_log2 > tuning().max_col_logl || logg > (int)r_log2 ||
      logg > MAX_COL_LOGL) {
    err = "r_log2 out of range";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const int logc = logn - (int)r_log2;
  if (logg >= logc) {
    err = "rows are shorter than the rank count";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const int logcl = logc - logg;
  const u64 Cl = 1ull << logcl, C = 1ull << logc, Rl = (1ull << r_log2) >> logg;
  init_field(pl, p, g);
  if (choose_arith(p, flags, err, pl.arith)) return PLAN_ERR_INVALID_ARGUMENT;
  pl.n = C;  // length of the transforms this plan runs
  pl.batch = Rl;
  pl.total = Rl * C;
  pl.flags = flags;
  pl.rank = rank;
  pl.nranks = nranks;
  std::vector<int> cols;
  int row = 0;
  bool two_level = false;
  if ((rc = sharded_row_split(logc, logg, cols, row, two_level, err, pl.total >= (1ull << 26)))) return rc;
  // the pass next to the exchange: row i = s * Lc + i' of a block is row i' of piece s
  const int lc_log = cols[0] - logg;
  auto piece_side = [&](HostPass &ps, u64 &istride, u64 &istride_hi, u64 &ostride, u64 &other_hi) {
    const u64 S = ps.istride;
    if (lc_log > 0 && !two_level) return;  // one rank, no two-level tile of this shape: the piece is the row
    ostride = Cl;  // local row q
    if (lc_log == 0) {
      istride = Rl * Cl;  // piece s
    } else {
      ps.two_level = 1;
      ps.row_split = (u32)lc_log;
      istride = S;
      istride_hi = Rl * Cl;
      other_hi = S << lc_log;  // the ordinary side, in two-level form
    }
  };
  if (flags & PLAN_FORWARD) {
    int rem = logc;
    for (size_t i = 0; i < cols.size(); ++i) {
      pl.fwd.emplace_back();
      rem -= cols[i];
      if ((rc = make_host_pass(pl, pl.fwd.back(), KIND_COL, cols[i], 1ull << rem, false, true, 1, 0,
                               rem + cols[i], err)))
        return rc;
    }
    pl.fwd.emplace_back();
    if ((rc = make_host_pass(pl, pl.fwd.back(), KIND_ROW, row, 1, false, false, 1, 0, 0, err))) return rc;
    HostPass &gp = pl.fwd.front();
    piece_side(gp, gp.src_istride, gp.src_istride_hi, gp.src_ostride, gp.dst_istride_hi);
  }
  if (flags & PLAN_INVERSE) {
    pl.inv.emplace_back();
    if ((rc = make_host_pass(pl, pl.inv.back(), KIND_ROW, row, 1, true, false, 1, 0, 0, err))) return rc;
    int rem = row;
    for (size_t k = cols.size(); k-- > 0;) {
      pl.inv.emplace_back();
      if ((rc = make_host_pass(pl, pl.inv.back(), KIND_COL, cols[k], 1ull << rem, true, true, 1, 0,
                               rem + cols[k], err)))
        return rc;
      rem += cols[k];
    }
    HostPass &sp = pl.inv.back();
    piece_side(sp, sp.dst_istride, sp.dst_istride_hi, sp.dst_ostride, sp.src_istride_hi);
  }
  return PLAN_OK;
}

// Kernel arguments of a pass, given where its tables live.
inline PassArgs make_args(const HostPlan &pl, const HostPass &ps, u64 *dst, const u64 *src,
                          const u64 *stage, const u64 *twist_lo, const u64 *twist_hi) {
  PassArgs a{};
  a.dst = dst;
  a.src = src;
  a.f = pl.f;
  a.stage_tw = stage;
  a.total = pl.total;
  a.istride = ps.dst_istride;
  a.ostride = ps.dst_ostride;
  a.src_istride = ps.src_istride;
  a.src_ostride = ps.src_ostride;
  a.istride_hi = ps.dst_istride_hi;
  a.src_istride_hi = ps.src_istride_hi;
  a.row_split = ps.row_split;
  a.tiles_per_outer = (ps.kind == KIND_COL) ? (u32)(ps.istride >> ps.f0) : 0;
  a.ct_first = 0;
  a.run_shift = 31;  // one run
  a.run_period = 0;
  a.compact = 0;
  a.grid = (u32)ps.grid;
  a.twist_lo = twist_lo;
  a.twist_hi = twist_hi;
  a.twist_shift = ps.twist_shift;
  a.twist_col_offset = ps.twist_col_offset;
  a.scale = ps.scale;
  return a;
}

// Column tiles per run of a column pass (0 for a row pass): the number of exchange chunks must
// divide it (make_chunk_args).
inline u64 pass_chunk_tiles(const HostPass &ps) {
  if (ps.kind != KIND_COL) return 0;
  return (ps.chunk_period ? ps.chunk_period : ps.istride) >> ps.f0;
}

// The same pass restricted to chunk `chunk` of `nchunks`.  A chunk is the `chunk`-th of nchunks equal
// column ranges of every RUN of the block's columns; a run is the whole block (ps.chunk_period == 0)
// or chunk_period columns of it (the sharded column pass when the row phase starts with a two-level
// pass: that pass needs, for its column c, the columns c + i' * period of every received piece, so a
// chunk must hold all of them).  A "compact" side is a buffer that holds only the chunk's columns,
// run after run: its strides shrink by nchunks and tile ct of the launch sits at column ct * T.
// (The pipelined all-to-all of the sharded transform: column pass chunk -> exchange chunk -> gather
// chunk.)  Returns the launch size in `grid`.
inline int make_chunk_args(const HostPlan &pl, const HostPass &ps, u64 *dst, const u64 *src,
                           const u64 *stage, const u64 *twist_lo, const u64 *twist_hi, u32 chunk,
                           u32 nchunks, bool dst_compact, bool src_compact, PassArgs &a, u32 &grid,
                           std::string &err) {
  a = make_args(pl, ps, dst, src, stage, twist_lo, twist_hi);
  const u64 period_tiles = pass_chunk_tiles(ps);
  if (ps.kind != KIND_COL || nchunks == 0 || chunk >= nchunks || period_tiles % nchunks != 0) {
    err = "pass cannot be split into that many column chunks";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const u64 period_cols = ps.chunk_period ? ps.chunk_period : ps.istride;
  const u64 run_tiles = period_tiles / nchunks, runs = ps.istride / period_cols;
  if (!is_pow2(run_tiles) || period_cols >> 32) {
    err = "pass cannot be split into that many column chunks";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const u64 tpc = run_tiles * runs;
  const u64 launch = (pl.total / ps.block) * tpc;
  a.tiles_per_outer = (u32)tpc;
  a.ct_first = (u32)(run_tiles * chunk);
  a.run_shift = (u32)ilog2_u64(run_tiles);
  a.run_period = (u32)period_cols;
  a.grid = (u32)launch;
  if (dst_compact) {
    a.istride /= nchunks;
    a.istride_hi /= nchunks;
    a.ostride /= nchunks;
    a.compact |= 1u;
  }
  if (src_compact) {
    a.src_istride /= nchunks;
    a.src_istride_hi /= nchunks;
    a.src_ostride /= nchunks;
    a.compact |= 2u;
  }
  grid = (u32)launch;
  return PLAN_OK;
}

inline std::string describe_plan(const HostPlan &pl) {
  std::string d;
  if (pl.arith == ARITH_GOLD) d = "[goldilocks] ";
  if (pl.arith == ARITH_SHOUP) d = "[fixed-point] ";
  char buf[96];
  const std::vector<HostPass> &v = pl.fwd.empty() ? pl.inv : pl.fwd;
  bool first = true;
  for (const HostPass &p : v) {
    if (!first) d += " | ";
    first = false;
    if (p.kind == KIND_COL)
      snprintf(buf, sizeof buf, "col 2^%d x T%d (stride %llu%s%s)", p.logl, 1 << p.f0,
               (unsigned long long)p.istride, p.loge == FINE_LOGE ? ", E4" : "",
               p.two_level ? ", two-level" : "");
    else
      snprintf(buf, sizeof buf, "row 2^%d (tile 2^%d%s)", p.logl, p.logt,
               p.loge == FINE_LOGE ? ", E4" : "");
    d += buf;
  }
  return d;
}

}  // namespace sventt_hip
