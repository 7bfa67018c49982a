#!/usr/bin/env python3
"""Two-pass splits of mid-size transforms (2^20..2^25 elements in all): every column length against the
planner's choice, wall clock of a stream of launches at steady-state clocks.
    python tools/split2_search.py [log2n ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

for logn in [int(a) for a in sys.argv[1:]] or [21, 22, 23, 24]:
    n = 1 << logn
    x = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
    y = torch.empty_like(x)
    res = []
    for col in [0] + list(range(max(1, logn - 13), min(12, logn - 8) + 1)):
        try:
            ntt = eng.NTT(eng.BASELINE_MODULUS, n, n0_log2=col, device_pointers=True)
        except Exception as exc:  # noqa: BLE001
            print(f"2^{logn} col 2^{col}: {exc}")
            continue
        reps = max(100, min(3000, (1 << 32) // n))
        for _ in range(reps):
            ntt.compute_forward(y, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ntt.compute_forward(y, x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res.append((dt, col, ntt.describe()))
        del ntt
    for dt, col, d in res:
        print(f"2^{logn} {'planner' if col == 0 else 'col 2^%d' % col:10s} {dt * 1e6:8.1f} us  [{d}]", flush=True)
