#!/usr/bin/env python3
"""Column pass of a single-GPU 2^24 plan against the sharded plans' (one launch, and cut into
four pipeline chunks): same tile shape, same stride."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sve_ntt_amd as eng
from sve_ntt_amd.sharded import HipShardEngine
n = 1 << 24
src = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
dst = torch.empty_like(src)
ntt = eng.NTT(eng.BASELINE_MODULUS, n)
engines = {G: HipShardEngine(eng.BASELINE_MODULUS, n * G, 11, 0, G) for G in (2, 8)}
def t(fn, reps=200):
    for _ in range(300): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for rep in range(2):
    print("single col pass   %.1f us" % t(lambda: ntt.run_pass(False, 0, dst, src)))
    for G, e in engines.items():
        print("sharded G=%d col   %.1f us [%s]" % (G, t(lambda: e.columns_chunk(False, dst, src, 0, 1)), e.describe().split(" | ")[0]))
        print("sharded G=%d col K=4 chunks %.1f us" % (G, t(lambda: [e.columns_chunk(False, dst[k*(n//4):(k+1)*(n//4)], src, k, 4) for k in range(4)])))
