#!/usr/bin/env python3
"""Developer loop: per-pass device time of a forward (and inverse) NTT plus a
golden-digest check.  Run on the GPU box:  python tools/quick_bench.py [log2n] [n0_log2] [batch]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker for the digest only)
import sve_ntt_amd as eng  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iters = 50
n = 1 << log2n
port = oracle.port()
src_h = port.fill_iota(n * batch, oracle.INPUT_I1_START)
src = torch.from_numpy(src_h.view(np.int64)).cuda()
dst = torch.empty_like(src)
ntt = eng.NTT(eng.BASELINE_MODULUS, n, n0_log2=n0, batch=batch)
print("plan:", ntt.describe())

ntt.compute_forward(dst, src)
got = dst.cpu().numpy().view(np.uint64)
if batch == 1:
    with open(os.path.join(ROOT, "tests", "golden", "ntt_digests.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["prime"] == "baseline" and c["log2m"] == log2n
                 and c["input"]["kind"] == "iota"]
    if cases:
        ok = [f"{x:016x}" for x in port.digest(got)] == cases[0]["forward_digest"]
        print("forward digest vs reference golden:", "OK" if ok else "MISMATCH")
    elif log2n <= 22:
        print("forward vs oracle:", "OK" if np.array_equal(got, port.forward(src_h, oracle.BASELINE_P, 3)) else "MISMATCH")
back = torch.empty_like(src)
ntt.compute_inverse(back, dst)
print("round trip:", "OK" if torch.equal(back, src) else "MISMATCH")

for inverse in (False, True):
    npass = ntt.num_passes(inverse)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(npass + 1)] for _ in range(iters)]
    a, b = (dst, src) if not inverse else (back, dst)
    for _ in range(max(5, min(1000, (1 << 33) // (n * batch)))):  # clock ramp, tools/clock_ramp.py
        for i in range(npass):
            ntt.run_pass(inverse, i, a, b if i == 0 else None)
    torch.cuda.synchronize()
    for k in range(iters):
        for i in range(npass):
            ev[k][i].record()
            ntt.run_pass(inverse, i, a, b if i == 0 else None)
        ev[k][npass].record()
    torch.cuda.synchronize()
    ms = [float(np.median([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(iters)])) for i in range(npass)]
    tot = sum(ms)
    print(("inverse" if inverse else "forward"), "per-pass us:", [round(x * 1e3, 1) for x in ms],
          "total us: %.1f  -> %.3e elem/s, %.1f%% of 8 TB/s (16 B/elem)" % (
              tot * 1e3, n * batch / (tot * 1e-3), 100 * 16 * n * batch / (tot * 1e-3) / 8e12))
