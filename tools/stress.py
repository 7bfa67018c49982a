#!/usr/bin/env python3
"""Randomised parity stress on the GPU: random (prime, length, split, batch, direction,
in/out of place, fused product) against the CPU oracle for a fixed wall-clock budget.
    python tools/stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
import sve_ntt_amd as eng

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
port = oracle.port()
PRIMES = [(0xFFFFFC6E80000001, 3, 31), (0x3A00000000000001, 3, 57), (0xFFFFFFFF00000001, 7, 32),
          (0xFFFFFFFF00000001, 0xF44872F5EC1C4CC0, 32), (0xA3B25F400C7A8001, 5, 15),
          (0x41D33D0D1FBF8001, 6, 15), (0x08AA90297F870001, 3, 16), (0x10001, 3, 16)]
t_end = time.time() + budget
cases = fails = 0
last_report = time.time()
kinds = {}
while time.time() < t_end:
    p, g, adic = PRIMES[rng.integers(len(PRIMES))]
    log2n = int(rng.integers(0, min(adic, 22) + 1))
    n = 1 << log2n
    batch = int(rng.integers(1, max(2, min(300, (1 << 22) // n) + 1)))
    n0 = 0
    if log2n >= 2 and rng.random() < 0.5:
        n0 = int(rng.integers(1, min(12, log2n - 1) + 1))
    inverse = bool(rng.integers(2))
    in_place = bool(rng.integers(2))
    fused = (not inverse) and rng.random() < 0.25
    # arithmetic back end: the plan's own choice, or forced (fixed_point needs p < 2^63)
    arith = str(rng.choice(["auto", "auto", "generic", "fixed_point"]))
    if arith == "fixed_point" and p >> 63:
        arith = "generic"
    divisor = int(rng.choice([0, 0, 1, 12345])) if inverse else 0
    try:
        ntt = eng.NTT(eng.Modulus(p, g), n, n0_log2=n0, batch=batch, arithmetic=arith, inverse_divisor=divisor)
    except ValueError:
        continue  # a split the tiles do not cover; the planner said so
    src = rng.integers(0, p, size=n * batch, dtype=np.uint64)
    if rng.random() < 0.1:
        src[rng.integers(0, src.size, size=min(8, src.size))] = p - 1  # edge values
    s = torch.from_numpy(src.view(np.int64)).cuda()
    d = s.clone() if in_place else torch.full_like(s, 0x5555555555555555)
    if fused:
        op = rng.integers(0, p, size=n * batch, dtype=np.uint64)
        om = torch.from_numpy(op.view(np.int64)).cuda()
        ntt.to_montgomery(om)
        ntt.compute_forward_multiply(d, None if in_place else s, om)
    elif inverse:
        ntt.compute_inverse(d, None if in_place else s)
    else:
        ntt.compute_forward(d, None if in_place else s)
    got = d.cpu().numpy().view(np.uint64)
    ok = True
    for b in range(batch):
        a = src[b * n:(b + 1) * n]
        want = port.inverse(a, p, g) if inverse else (port.forward(a, p, g) if n > 1 else a.copy())
        if inverse and divisor:  # the oracle divides by n; the plan by `divisor`
            factor = n * pow(divisor, -1, p) % p
            want = np.array((want.astype(object) * factor) % p, dtype=np.uint64)
        if fused:
            want = np.array((want.astype(object) * op[b * n:(b + 1) * n].astype(object)) % p, dtype=np.uint64)
        if not np.array_equal(got[b * n:(b + 1) * n], want):
            ok = False
            break
    cases += 1
    if time.time() - last_report > 30:
        last_report = time.time()
        print(f"... {cases} cases, {fails} mismatches so far", flush=True)
    key = ntt.describe()
    kinds[key] = kinds.get(key, 0) + 1
    if not ok:
        fails += 1
        print(f"MISMATCH p={p:#x} g={g:#x} n=2^{log2n} n0={n0} batch={batch} inverse={inverse} arith={arith} divisor={divisor} "
              f"in_place={in_place} fused={fused} plan=[{key}] batch_index={b}", flush=True)
print(f"{cases} cases, {fails} mismatches, {len(kinds)} distinct plans, seed {seed}")
sys.exit(1 if fails else 0)
