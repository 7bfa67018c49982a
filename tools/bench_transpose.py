#!/usr/bin/env python3
"""GPU analogue of the reference's tests/bench-transpose.cpp (out-of-place with padded
leading dimensions, in-place square) and of the copy leg of tests/bench-stream-cmg.cpp:
bytes processed = 8*rows*cols per call as there (`SetBytesProcessed`), plus the HBM rate
(read + write).  Self-checking by transposing back.  Run on the GPU box:
    python tools/bench_transpose.py > gpurun_out/transpose.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for i in range(iters):
        ev[i].record()
        fn()
    ev[iters].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(iters)])) * 1e-3


print("%-28s %10s %12s %12s" % ("case", "us", "GB/s moved", "GB/s HBM r+w"))
for rows, cols, ps, pd in [(1 << 11, 1 << 13, 0, 0), (1 << 12, 1 << 12, 0, 0), (1 << 12, 1 << 12, 32, 32),
                           (1 << 13, 1 << 11, 0, 0), (1 << 14, 1 << 14, 0, 0), (1 << 14, 1 << 14, 32, 32),
                           (1 << 15, 1 << 13, 0, 0)]:
    src = torch.arange((cols + ps) * rows, dtype=torch.int64, device="cuda")
    dst = torch.full(((rows + pd) * cols,), 0x55, dtype=torch.int64, device="cuda")
    t = timed(lambda: eng.transpose(dst, src, rows, cols, rows + pd, cols + ps))
    back = torch.zeros_like(src)
    eng.transpose(back, dst, cols, rows, cols + ps, rows + pd)
    ok = torch.equal(back.view(rows, cols + ps)[:, :cols], src.view(rows, cols + ps)[:, :cols])
    b = 8.0 * rows * cols
    print("%-28s %10.1f %12.1f %12.1f %s" % (f"oop {rows}x{cols} pad {ps},{pd}", t * 1e6, b / t / 1e9,
                                             2 * b / t / 1e9, "ok" if ok else "MISMATCH"))
for dim in (1 << 10, 1 << 12, 1 << 13, 1 << 14):
    a = torch.arange(dim * dim, dtype=torch.int64, device="cuda")
    ref = a.clone()
    t = timed(lambda: eng.transpose_inplace(a, dim), iters=30)  # 3 + 30 calls
    eng.transpose_inplace(a, dim)  # odd count so far -> even
    ok = torch.equal(a, ref)
    b = 8.0 * dim * dim
    print("%-28s %10.1f %12.1f %12.1f %s" % (f"in place {dim}x{dim}", t * 1e6, b / t / 1e9, 2 * b / t / 1e9,
                                             "ok" if ok else "MISMATCH"))
n = 1 << 27
x = torch.arange(n, dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
t = timed(lambda: y.copy_(x))
print("%-28s %10.1f %12.1f %12.1f" % ("device copy 1 GiB (hipMemcpy)", t * 1e6, 8.0 * n / t / 1e9, 16.0 * n / t / 1e9))
p = eng.NTT(eng.BASELINE_MODULUS, 1 << 24)
a = torch.randint(0, 2**62, (1 << 24,), dtype=torch.int64, device="cuda")
b = torch.randint(0, 2**62, (1 << 24,), dtype=torch.int64, device="cuda")
c = torch.empty_like(a)
t = timed(lambda: p.pointwise_multiply(c, a, b))
print("%-28s %10.1f %12.1f %12.1f" % ("pointwise multiply 2^24", t * 1e6, 8.0 * (1 << 24) / t / 1e9,
                                      24.0 * (1 << 24) / t / 1e9))
