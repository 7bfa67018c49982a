#!/usr/bin/env python3
"""The largest transforms the fields allow on one GPU: 2^31 points modulo the BASELINE prime
(2-adicity 31) and 2^32 modulo Goldilocks (2-adicity 32), in place; iota input, sampled
outputs against the closed form, inverse round trip compared on the device."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng

S0 = 0x0123456789ABCDE
for P, G, log2n in ((0xFFFFFC6E80000001, 3, 31), (0xFFFFFFFF00000001, 7, 32)):
    m = 1 << log2n
    t0 = time.time()
    ntt = eng.NTT(eng.Modulus(P, G), m)
    buf = torch.empty(m, dtype=torch.int64, device="cuda")
    step = 1 << 28
    for lo in range(0, m, step):  # (one torch.arange of 2^32 elements exceeds torch's own launch limits)
        buf[lo:lo + step] = torch.arange(S0 + lo, S0 + lo + step, dtype=torch.int64, device="cuda")
    # one untimed round trip first: the first launch of each kernel loads its code object and sets its
    # LDS attribute (r01 timed cold calls here and reported the inverse at twice the forward time)
    ntt.compute_forward(buf)
    ntt.compute_inverse(buf)
    torch.cuda.synchronize(); t1 = time.time()
    ntt.compute_forward(buf)
    torch.cuda.synchronize(); t2 = time.time()
    w = pow(G, (P - 1) // m, P)
    rng = np.random.default_rng(log2n)
    where = np.unique(np.concatenate([[0, 1, 2, m - 1, m // 2], rng.integers(0, m, size=1024)]))
    got = buf[torch.from_numpy(where).cuda()].cpu().numpy().view(np.uint64)
    bad = 0
    for j, x in zip(where.tolist(), got.tolist()):
        k = int(format(j, f"0{log2n}b")[::-1], 2)
        want = (m * S0 + m * (m - 1) // 2) % P if k == 0 else m * pow(pow(w, k, P) - 1, -1, P) % P
        bad += x != want
    fwd = t2 - t1
    torch.cuda.synchronize(); t_inv0 = time.time()  # the host-side checks above are not timed
    ntt.compute_inverse(buf)
    torch.cuda.synchronize(); t3 = time.time()
    ok = bad == 0
    # compare in slices to bound the temporary
    for lo in range(0, m, step):
        ok = ok and bool(torch.equal(buf[lo:lo + step], torch.arange(S0 + lo, S0 + lo + step, dtype=torch.int64, device="cuda")))
    print(f"p={P:#x} n=2^{log2n} [{ntt.describe()}]: plan {t1 - t0:.1f} s, forward {1e3 * fwd:.1f} ms "
          f"({m / fwd:.3e} elem/s), inverse {1e3 * (t3 - t_inv0):.1f} ms, sampled mismatches {bad}, "
          f"{'OK' if ok else 'MISMATCH'}", flush=True)
    del buf, ntt
    torch.cuda.empty_cache()
