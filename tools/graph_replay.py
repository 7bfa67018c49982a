#!/usr/bin/env python3
"""Launch-bound small transforms under a HIP graph: K forward transforms (default 2^17 as 2^8 x 2^9, the
reference's README configuration) captured once on a stream and replayed, against the same K calls made
one by one.  The library's launches are capturable (no synchronisation, no allocation after a plan's first
call), so a caller with many small transforms can amortise the per-launch CPU cost.
    python tools/graph_replay.py [log2n] [n0_log2] [K]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = 1 << log2n
ntt = eng.NTT(eng.BASELINE_MODULUS, n, n0_log2=n0, device_pointers=True)
src = torch.arange(n, dtype=torch.int64, device="cuda") + 0x0123456789ABCDEF
dst = [torch.empty_like(src) for _ in range(K)]
ntt.compute_forward(dst[0], src)  # first call: code object load, LDS attribute
torch.cuda.synchronize()
want = dst[0].clone()

s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for k in range(K):
            ntt.compute_forward(dst[k], src, stream=s)
for d in dst:
    d.zero_()
g.replay()
torch.cuda.synchronize()
ok = all(torch.equal(d, want) for d in dst)


def timed(fn, reps):
    for _ in range(reps // 4 + 1):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def one_by_one():
    for k in range(K):
        ntt.compute_forward(dst[k], src)


t_calls = timed(one_by_one, 200) / K
t_graph = timed(g.replay, 200) / K
print(f"n=2^{log2n} [{ntt.describe()}], {K} transforms per batch: one call each {t_calls * 1e6:.2f} us per transform, "
      f"graph replay {t_graph * 1e6:.2f} us per transform ({'results equal' if ok else 'MISMATCH'})")
