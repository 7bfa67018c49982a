#!/usr/bin/env python3
"""Times the LOCAL phases one rank runs in the sharded six-step (no communication),
for world sizes 2, 4, 8 at 2^L elements per rank (default L = 24; 27 = the per-rank size of
BASELINE configs[4], which bench.py --gpus G runs) -- what a rank does between the exchanges.
Run on a one-GPU box:  python tools/bench_sharded_local.py [L] [worlds, e.g. 8 or 2,4,8] [log2 R, default 11]
SVENTT_SHARDED_FUSE=0 gives r02's plans (a gather pass of its own) for a before/after."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402
from sve_ntt_amd.sharded import HipShardEngine  # noqa: E402

LOG2_LOCAL = int(sys.argv[1]) if len(sys.argv) > 1 else 24
WORLDS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (2, 4, 8)
R_LOG2 = int(sys.argv[3]) if len(sys.argv) > 3 else 11  # rows of the six-step (the length of the column phase)
n_local = 1 << LOG2_LOCAL
rng = np.random.default_rng(0)
src = torch.from_numpy(rng.integers(0, eng.BASELINE_MODULUS.modulus, size=n_local, dtype=np.uint64)
                       .view(np.int64)).cuda()
work = torch.empty_like(src)
recv = torch.empty_like(src)
out = torch.empty_like(src)
iters = 30
for world in WORLDS:
    e = HipShardEngine(eng.BASELINE_MODULUS, n_local * world, R_LOG2, 0, world)
    names = e.describe().split(" | ")

    def run(ev=None):
        k = 0
        if ev: ev[k].record()
        e.columns_chunk(False, work, src, 0, 1); k += 1
        if ev: ev[k].record()
        # (the all-to-all would run here; the gather pass reads `work` as if received)
        e.exchange_side_chunk(False, out, work, 0, 1); k += 1
        if ev: ev[k].record()
        for i in range(1, e.rows_passes):
            e.rows_pass(False, i, out, out); k += 1
            if ev: ev[k].record()
    for _ in range(max(20, 300 >> max(0, LOG2_LOCAL - 24))):  # steady-state clocks, tools/clock_ramp.py
        run()
    torch.cuda.synchronize()
    nph = 1 + e.rows_passes
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(nph + 1)] for _ in range(iters)]
    for it in range(iters):
        run(evs[it])
    torch.cuda.synchronize()
    ms = [float(np.median([evs[it][i].elapsed_time(evs[it][i + 1]) for it in range(iters)])) for i in range(nph)]
    local = [nm for nm in names if nm != "all-to-all"]
    print(f"world={world}: " + ", ".join(f"{nm}: {t * 1e3:.0f} us" for nm, t in zip(local, ms)) +
          f"  | local total {sum(ms) * 1e3:.0f} us per 2^{LOG2_LOCAL} elements"
          f" = {sum(ms) * 1e3 / (1 << max(0, LOG2_LOCAL - 24)):.0f} us per 2^24"
          + (" [SVENTT_SHARDED_FUSE=0]" if os.environ.get("SVENTT_SHARDED_FUSE") == "0" else ""))
