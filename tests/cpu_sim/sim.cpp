// tests/cpu_sim/sim.cpp -- sequential host replay of the plan's passes.
//
// TEST-ONLY.  Compiles the very same tile code (sve_ntt_amd/csrc/tile_ntt.h) and
// planner (plan_core.h) for the host and replays them one workgroup, one step,
// one thread at a time, so that index arithmetic, twiddle tables and the plan
// split can be checked against the oracle in the no-GPU test tier.  It is not a
// fallback: nothing in sve_ntt_amd/ links or loads it, and it is orders of
// magnitude too slow to be one.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../sve_ntt_amd/csrc/plan_core.h"

using namespace sventt_hip;

namespace {

template <class TN, int IDX> void sim_steps(const PassArgs &a, const typename TN::Tile &t, u64 *lds) {
  constexpr int SI = (TN::MODE == MODE_FWD) ? IDX : TN::NSTEPS - 1 - IDX;
  // a barrier separates steps on the GPU; here each step simply runs for all threads
  for (u32 tid = 0; tid < (u32)TN::NT; ++tid) TN::template step<SI>(a, t, tid, lds);
  if constexpr (IDX + 1 < TN::NSTEPS) sim_steps<TN, IDX + 1>(a, t, lds);
}

template <class TN> struct SimLauncher {
  static int launch(const PassArgs &a, u32 grid, int /*stream*/) {
    std::vector<u64> lds((size_t)1 << TN::LOGT);
    // In-place passes: a tile only ever touches its own elements, so replaying the
    // workgroups one after another is equivalent to running them concurrently.
    for (u32 b = 0; b < grid; ++b) {
      const typename TN::Tile t = TN::locate(a, b);
      if (!t.live) continue;
      sim_steps<TN, 0>(a, t, lds.data());
    }
    return 0;
  }
};

using SimEntry = KernelEntryT<int, int>;

// the tile code of the pass's arithmetic back end (field64.h)
const SimEntry *find_sim_kernel(const HostPass &h, int flag) {
  const int dir = h.inverse ? MODE_INV : MODE_FWD;
  if (h.arith == ARITH_GOLD)
    return find_arith_kernel_in_registry<ARITH_GOLD, SimEntry, SimLauncher>(h.kind, h.logl, dir, flag, h.f0, h.loge,
                                                                            h.two_level);
  if (h.arith == ARITH_SHOUP)
    return find_arith_kernel_in_registry<ARITH_SHOUP, SimEntry, SimLauncher>(h.kind, h.logl, dir, flag, h.f0, h.loge,
                                                                             h.two_level);
  return find_kernel_in_registry<SimEntry, SimLauncher>(h.kind, h.logl, dir, flag, h.f0, h.loge, h.two_level);
}

thread_local std::string g_err;

// `epilogue` (forward only): the fused product of sventt_forward_multiply -- the final
// ROW pass runs as its FLAG variant with PassArgs::epilogue set.
int run_plan(const HostPlan &pl, bool inverse, u64 *dst, const u64 *src, const u64 *epilogue = nullptr) {
  const std::vector<HostPass> &passes = inverse ? pl.inv : pl.fwd;
  if (pl.n == 1) {
    for (u64 i = 0; i < pl.total; ++i) dst[i] = epilogue ? montmul(src[i], epilogue[i], pl.f) : src[i];
    return 0;
  }
  const u64 *in = src;
  for (size_t i = 0; i < passes.size(); ++i) {
    const HostPass &h = passes[i];
    const bool fused = epilogue && i + 1 == passes.size();
    const SimEntry *e = find_sim_kernel(h, (h.flag || fused) ? 1 : 0);
    if (!e || e->f0 != h.f0 || e->logt != h.logt || (fused && h.kind != KIND_ROW)) {
      g_err = "registry mismatch";
      return PLAN_ERR_LOGIC;
    }
    PassArgs a = make_args(pl, h, dst, in, h.stage.data(), h.twist_lo.data(), h.twist_hi.data());
    if (fused) a.epilogue = epilogue;
    e->launch(a, (u32)h.grid, 0);
    in = dst;
  }
  return 0;
}

}  // namespace

// Every tile shape of the three registries: set mapping bijective, chunk-preserving steps wave-local.
// Returns the number of shapes checked, or -(index + 1) of the first bad one.
template <class F> int check_table(F find) {
  int n = 0;
  for (int kind = 0; kind < 2; ++kind)
    for (int logl = 1; logl <= 13; ++logl)
      for (int dir = 0; dir < 2; ++dir)
        for (int flag = 0; flag < 2; ++flag)
          for (int f0 = 0; f0 <= 11; ++f0)
            for (int loge : {2, 4})
              for (int two = 0; two < 2; ++two) {
                const SimEntry *e = find(kind, logl, dir, flag, f0, loge, two);
                if (!e) continue;
                ++n;
                if (!e->set_mapping_ok()) return -n;
              }
  return n;
}

extern "C" {

int sim_check_set_mappings(void) {
  const int a = check_table([](int k, int l, int d, int fl, int f0, int e, int two) {
    return find_kernel_in_registry<SimEntry, SimLauncher>(k, l, d, fl, f0, e, two); });
  if (a < 0) return a;
  const int b = check_table([](int k, int l, int d, int fl, int f0, int e, int two) {
    return find_arith_kernel_in_registry<ARITH_GOLD, SimEntry, SimLauncher>(k, l, d, fl, f0, e, two); });
  if (b < 0) return b - 100000;
  return a + b;
}

// workgroup barriers per tile of the kernel that serves a pass shape (-1: no such kernel)
int sim_group_barriers(int kind, int logl, int dir, int flag, int f0, int loge) {
  const SimEntry *e = find_kernel_in_registry<SimEntry, SimLauncher>(kind, logl, dir, flag, f0, loge);
  return e ? e->group_barriers : -1;
}

const char *sim_last_error(void) { return g_err.c_str(); }

// Whole transform through the planner (same arguments as sventt_plan_create + forward/inverse).
int sim_transform(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2, uint64_t batch, int inverse,
                  uint64_t *dst, const uint64_t *src) {
  HostPlan pl;
  int rc = build_plan(pl, p, g, n, n0_log2, batch, inverse ? PLAN_INVERSE : PLAN_FORWARD, g_err);
  if (rc) return rc;
  return run_plan(pl, inverse != 0, dst, src);
}

// The same with extra plan flags (PLAN_GENERIC_ARITHMETIC, PLAN_FIXED_POINT: the arithmetic back end).
int sim_transform_flags(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2, uint64_t batch, int inverse,
                        uint32_t flags, uint64_t *dst, const uint64_t *src) {
  HostPlan pl;
  int rc = build_plan(pl, p, g, n, n0_log2, batch, (inverse ? PLAN_INVERSE : PLAN_FORWARD) | flags, g_err);
  if (rc) return rc;
  return run_plan(pl, inverse != 0, dst, src);
}

// sventt_forward_multiply: forward transform with the pointwise product (operand in
// Montgomery form) fused into the last pass.
int sim_forward_multiply(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2, uint64_t batch,
                         uint64_t *dst, const uint64_t *src, const uint64_t *operand) {
  HostPlan pl;
  int rc = build_plan(pl, p, g, n, n0_log2, batch, PLAN_FORWARD, g_err);
  if (rc) return rc;
  return run_plan(pl, false, dst, src, operand);
}

// One rank's column pass of the sharded six-step (same arguments as sventt_sharded_columns).
int sim_sharded_columns(uint64_t p, uint64_t g, uint64_t n, uint32_t r_log2, int rank, int nranks,
                        int inverse, uint64_t *dst, const uint64_t *src) {
  HostPlan pl;
  int rc = build_sharded_plan(pl, p, g, n, r_log2, rank, nranks,
                              inverse ? PLAN_INVERSE : PLAN_FORWARD, g_err);
  if (rc) return rc;
  return run_plan(pl, inverse != 0, dst, src);
}

// Row phase of the sharded six-step, one pass at a time (sventt_run_pass on the rows plan).
int sim_sharded_rows_num_passes(uint64_t p, uint64_t g, uint64_t n, uint32_t r_log2, int rank,
                                int nranks) {
  HostPlan pl;
  int rc = build_sharded_rows_plan(pl, p, g, n, r_log2, rank, nranks, PLAN_FORWARD, g_err);
  if (rc) return rc;
  return (int)pl.fwd.size();
}

int sim_sharded_rows_pass(uint64_t p, uint64_t g, uint64_t n, uint32_t r_log2, int rank, int nranks,
                          int inverse, int index, uint64_t *dst, const uint64_t *src) {
  HostPlan pl;
  int rc = build_sharded_rows_plan(pl, p, g, n, r_log2, rank, nranks,
                                   inverse ? PLAN_INVERSE : PLAN_FORWARD, g_err);
  if (rc) return rc;
  const std::vector<HostPass> &passes = inverse ? pl.inv : pl.fwd;
  if (index < 0 || (size_t)index >= passes.size()) return PLAN_ERR_INVALID_ARGUMENT;
  const HostPass &h = passes[(size_t)index];
  const SimEntry *e = find_sim_kernel(h, h.flag ? 1 : 0);
  if (!e) return PLAN_ERR_LOGIC;
  if (dst == src && (h.src_istride != h.dst_istride || h.src_ostride != h.dst_ostride ||
                     h.src_istride_hi != h.dst_istride_hi)) {
    g_err = "gather/scatter passes cannot run in place";
    return PLAN_ERR_INVALID_ARGUMENT;
  }
  const PassArgs a = make_args(pl, h, dst, src, h.stage.data(), h.twist_lo.data(), h.twist_hi.data());
  e->launch(a, (u32)h.grid, 0);
  return 0;
}

// Chunked column passes of the sharded transform (sventt_run_pass_chunk): which = 0 the
// column plan's pass, which = 1 the rows plan's gather (forward pass 0) / scatter
// (inverse last pass).
int sim_sharded_chunk(uint64_t p, uint64_t g, uint64_t n, uint32_t r_log2, int rank, int nranks,
                      int which, int inverse, uint32_t chunk, uint32_t nchunks, int dst_compact,
                      int src_compact, uint64_t *dst, const uint64_t *src) {
  HostPlan pl;
  const u32 flags = inverse ? PLAN_INVERSE : PLAN_FORWARD;
  int rc = which == 0 ? build_sharded_plan(pl, p, g, n, r_log2, rank, nranks, flags, g_err)
                      : build_sharded_rows_plan(pl, p, g, n, r_log2, rank, nranks, flags, g_err);
  if (rc) return rc;
  const std::vector<HostPass> &passes = inverse ? pl.inv : pl.fwd;
  const HostPass &h = (which == 1 && inverse) ? passes.back() : passes.front();
  const SimEntry *e = find_sim_kernel(h, h.flag ? 1 : 0);
  if (!e) return PLAN_ERR_LOGIC;
  PassArgs a;
  u32 grid = 0;
  rc = make_chunk_args(pl, h, dst, src, h.stage.data(), h.twist_lo.data(), h.twist_hi.data(), chunk,
                       nchunks, dst_compact != 0, src_compact != 0, a, grid, g_err);
  if (rc) return rc;
  e->launch(a, grid, 0);
  return 0;
}

int64_t sim_sharded_tiles_per_block(uint64_t p, uint64_t g, uint64_t n, uint32_t r_log2, int rank,
                                    int nranks, int which) {
  HostPlan pl;
  int rc = which == 0 ? build_sharded_plan(pl, p, g, n, r_log2, rank, nranks, PLAN_FORWARD, g_err)
                      : build_sharded_rows_plan(pl, p, g, n, r_log2, rank, nranks, PLAN_FORWARD, g_err);
  if (rc) return rc;
  return (int64_t)pass_chunk_tiles(pl.fwd.front());
}

// Planner introspection: writes up to `cap` entries of (kind, logl, f0, logt, grid, loge) per pass.
int sim_plan_shape(uint64_t p, uint64_t g, uint64_t n, uint32_t n0_log2, uint64_t batch, int inverse,
                   int64_t *out, int cap) {
  HostPlan pl;
  int rc = build_plan(pl, p, g, n, n0_log2, batch, inverse ? PLAN_INVERSE : PLAN_FORWARD, g_err);
  if (rc) return rc;
  const std::vector<HostPass> &passes = inverse ? pl.inv : pl.fwd;
  int k = 0;
  for (const HostPass &h : passes) {
    if (k >= cap) break;
    out[6 * k + 0] = h.kind;
    out[6 * k + 1] = h.logl;
    out[6 * k + 2] = h.f0;
    out[6 * k + 3] = h.logt;
    out[6 * k + 4] = (int64_t)h.grid;
    out[6 * k + 5] = h.loge;
    ++k;
  }
  return k;
}

// Device-function restatements exposed for unit tests (host build of field64.h).
uint64_t sim_montmul(uint64_t a, uint64_t w, uint64_t N) {
  Field f{N, h_montgomery_inverse(N), 0 - N};
  return montmul(a, w, f);
}
uint64_t sim_addmod(uint64_t a, uint64_t b, uint64_t N) {
  Field f{N, 0, 0 - N};
  return addmod(a, b, f);
}
uint64_t sim_submod(uint64_t a, uint64_t b, uint64_t N) {
  Field f{N, 0, 0 - N};
  return submod(a, b, f);
}
uint64_t sim_montgomery_inverse(uint64_t N) { return h_montgomery_inverse(N); }
uint64_t sim_to_montgomery(uint64_t a, uint64_t N) { return h_to_montgomery(a, N); }
uint32_t sim_lds_phys(uint32_t I) { return lds_phys(I); }

}  // extern "C"
