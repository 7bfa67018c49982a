#!/usr/bin/env python3
"""Cyclic convolution pipeline at N = 2^24 with data resident on the device (SURVEY.md 8f
rank 1): forward -> pointwise -> inverse as three operations, and with the product fused
into the forward transform's last pass (sventt_forward_multiply)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng

for log2n in (15, 20, 24):
    n = 1 << log2n
    ntt = eng.NTT(eng.BASELINE_MODULUS, n)
    a = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
    spec = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
    spec_m = torch.empty_like(spec)
    ntt.to_montgomery(spec_m, spec)
    x, y = torch.empty_like(a), torch.empty_like(a)

    def three():
        ntt.compute_forward(x, a)
        ntt.pointwise_multiply(x, x, spec)
        ntt.compute_inverse(x)

    def fused():
        ntt.compute_forward_multiply(y, a, spec_m)
        ntt.compute_inverse(y)

    three(); fused(); torch.cuda.synchronize()
    assert torch.equal(x, y)
    for name, fn in (("forward | pointwise | inverse", three), ("forward*multiply | inverse", fused)):
        reps = max(20, min(300, (1 << 27) // n))
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("2^%-3d %-32s %9.1f us  %.3e elem/s" % (log2n, name, dt * 1e6, n / dt), flush=True)
