#!/usr/bin/env python3
"""Large transforms with random explicit splits: iota input (every output has a closed
form, see tests/test_gpu_parity.py::test_iota_closed_form_up_to_2p30), sampled outputs
checked exactly, then the inverse must return the input (compared on the device).
    python tools/stress_large.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
P, G = 0xFFFFFC6E80000001, 3
S0 = 0x0123456789ABCDEF
t_end = time.time() + budget
cases = fails = 0
last_report = time.time()
plans = set()
while time.time() < t_end:
    log2n = int(rng.integers(20, 29))
    n0 = 0 if rng.random() < 0.3 else int(rng.integers(1, 13))
    batch = 1 if log2n > 24 else int(rng.integers(1, 4))
    m = 1 << log2n
    try:
        ntt = eng.NTT(eng.Modulus(P, G), m, n0_log2=n0, batch=batch)
    except ValueError:
        continue
    plans.add(ntt.describe())
    src = torch.arange(S0, S0 + m * batch, dtype=torch.int64, device="cuda")
    dst = torch.full_like(src, 0x5555555555555555)
    ntt.compute_forward(dst, src)
    w = pow(G, (P - 1) // m, P)
    ok = True
    for b in range(batch):
        where = np.unique(np.concatenate([[0, 1, m - 1], rng.integers(0, m, size=256)]))
        got = dst[b * m + torch.from_numpy(where).cuda()].cpu().numpy().view(np.uint64)
        s0 = S0 + b * m
        for j, x in zip(where.tolist(), got.tolist()):
            k = int(format(j, f"0{log2n}b")[::-1], 2)
            want = (m * s0 + m * (m - 1) // 2) % P if k == 0 else m * pow(pow(w, k, P) - 1, -1, P) % P
            if x != want:
                ok = False
    ntt.compute_inverse(dst)
    ok = ok and bool(torch.equal(dst, src))
    cases += 1
    if time.time() - last_report > 30:
        last_report = time.time()
        print(f"... {cases} cases, {fails} mismatches so far", flush=True)
    if not ok:
        fails += 1
        print(f"MISMATCH n=2^{log2n} n0={n0} batch={batch} [{ntt.describe()}]", flush=True)
    del src, dst, ntt
print(f"{cases} cases, {fails} mismatches, {len(plans)} distinct plans")
sys.exit(1 if fails else 0)
