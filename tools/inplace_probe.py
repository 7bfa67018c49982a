#!/usr/bin/env python3
"""N = 2^24 forward: out of place (the reference harness's compute_forward(dst, src), two 128 MiB buffers =
the whole 256 MiB Infinity Cache) against in place (one buffer), and the round trip with the inverse out
of place (three buffers) or in place (two).  Per-pass HIP-event medians.  python tools/inplace_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

n = 1 << 24
ntt = eng.NTT(eng.BASELINE_MODULUS, n, device_pointers=True)
src = torch.arange(n, dtype=torch.int64, device="cuda") + 0x0123456789ABCDEF
dst = torch.empty_like(src)
back = torch.empty_like(src)
iters = 60


def measure(label, passes):
    """passes: list of (inverse, index, dst, src)"""
    for _ in range(600):
        for inv, i, d, s in passes:
            ntt.run_pass(inv, i, d, s)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(passes) + 1)] for _ in range(iters)]
    for k in range(iters):
        for j, (inv, i, d, s) in enumerate(passes):
            ev[k][j].record()
            ntt.run_pass(inv, i, d, s)
        ev[k][len(passes)].record()
    torch.cuda.synchronize()
    ms = [float(np.median([ev[k][j].elapsed_time(ev[k][j + 1]) for k in range(iters)])) for j in range(len(passes))]
    print(f"{label:58s} {' + '.join('%.1f' % (t * 1e3) for t in ms)} = {sum(ms) * 1e3:.1f} us", flush=True)


work = src.clone()
measure("forward out of place (src -> dst, 2 buffers)", [(False, 0, dst, src), (False, 1, dst, None)])
measure("forward in place (1 buffer; input = previous output)", [(False, 0, work, None), (False, 1, work, None)])
measure("round trip, inverse out of place (3 buffers)",
        [(False, 0, dst, src), (False, 1, dst, None), (True, 0, back, dst), (True, 1, back, None)])
measure("round trip, inverse in place (2 buffers)",
        [(False, 0, dst, src), (False, 1, dst, None), (True, 0, dst, None), (True, 1, dst, None)])
