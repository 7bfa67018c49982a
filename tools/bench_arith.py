#!/usr/bin/env python3
"""The arithmetic back ends against each other: forward and inverse
device time of the same transform under ARITH_MONT ("generic"), the plan's own choice ("auto") and
ARITH_SHOUP ("fixed_point").  Run on the GPU box:  python tools/bench_arith.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

GOLD = eng.Modulus(0xffffffff00000001, 7)
P62 = eng.Modulus(0x3a00000000000001, 3)  # the reference's own NTT test prime (tests/ntt-tests/*.hpp:4-5)
CASES = [(GOLD, "generic"), (GOLD, "auto"), (P62, "auto"), (P62, "fixed_point")]


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for log2n, batch, warm, reps in ((24, 1, 300, 100), (12, 1 << 16, 30, 20)):
    n = 1 << log2n
    rng = np.random.default_rng(5)
    for mod, arith in CASES:
        src = torch.from_numpy(rng.integers(0, mod.modulus, n * batch, dtype=np.uint64).view(np.int64)).cuda()
        dst, back = torch.empty_like(src), torch.empty_like(src)
        ntt = eng.NTT(mod, n, batch=batch, arithmetic=arith, device_pointers=True)
        for _ in range(warm):
            ntt.compute_forward(dst, src)
        fwd = timed(lambda: ntt.compute_forward(dst, src), reps)
        for _ in range(warm // 3):
            ntt.compute_inverse(back, dst)
        inv = timed(lambda: ntt.compute_inverse(back, dst), reps)
        ok = torch.equal(back, src)
        size = "2^%d" % log2n + ("" if batch == 1 else " x %d" % batch)
        print("%#018x  %-11s  %-14s  forward %8.1f us  inverse %8.1f us  round trip %s  %s" % (
            mod.modulus, arith, size, fwd, inv, "OK" if ok else "MISMATCH", ntt.describe()), flush=True)
        del ntt, src, dst, back
