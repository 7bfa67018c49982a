#!/usr/bin/env python3
"""Per-pass HIP-event times of explicit pass splits of large single-GPU transforms (SVENTT_SPLIT,
plan_core.h: choose_split) against the planner's own choice.
    python tools/split3_search.py [log2n ...]        (default 26 27 28 30)
Every plan is checked first: iota input, sampled outputs against the closed form."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

P, G, S0 = 0xFFFFFC6E80000001, 3, 0x0123456789ABCDEF


def candidates(logn):
    out = [None]  # the planner's choice
    if os.environ.get("SPLIT3_WIDE"):
        for row in (13, 12):
            for a in range(4, 10):
                b = logn - row - a
                if 5 <= b <= 12:
                    out.append(f"{a},{b},{row}")
        return out
    rem = logn - 13
    for c_last in (11, 10, 12):
        first = rem - c_last
        if 1 <= first <= 12:
            out.append(f"{first},{c_last},13")
    if rem - 12 >= 1 and rem - 12 + 1 <= 12:
        out.append(f"{rem - 12 + 1},12,12")
    half = rem // 2
    out.append(f"{rem - half},{half},13")
    seen, res = set(), []
    for c in out:
        if c not in seen:
            seen.add(c)
            res.append(c)
    return res


def check(buf, logn):
    m = 1 << logn
    w = pow(G, (P - 1) // m, P)
    rng = np.random.default_rng(logn)
    where = np.unique(np.concatenate([[0, 1, 2, m - 1, m // 2], rng.integers(0, m, size=256)]))
    got = buf[torch.from_numpy(where).cuda()].cpu().numpy().view(np.uint64)
    for j, x in zip(where.tolist(), got.tolist()):
        k = int(format(j, f"0{logn}b")[::-1], 2)
        want = (m * S0 + m * (m - 1) // 2) % P if k == 0 else m * pow(pow(w, k, P) - 1, -1, P) % P
        if x != want:
            return False
    return True


for logn in [int(a) for a in sys.argv[1:]] or [26, 27, 28, 30]:
    m = 1 << logn
    src = torch.empty(m, dtype=torch.int64, device="cuda")
    step = 1 << 28
    for lo in range(0, m, step):
        src[lo:lo + min(step, m)] = torch.arange(S0 + lo, S0 + lo + min(step, m), dtype=torch.int64, device="cuda")
    dst = torch.empty_like(src)
    for split in candidates(logn):
        if split is None:
            os.environ.pop("SVENTT_SPLIT", None)
        else:
            os.environ["SVENTT_SPLIT"] = split
        try:
            ntt = eng.NTT(eng.BASELINE_MODULUS, m, device_pointers=True)
        except Exception as exc:  # noqa: BLE001
            print(f"2^{logn} split {split}: {exc}")
            continue
        npass = ntt.num_passes(False)
        ntt.compute_forward(dst, src)
        ok = check(dst, logn)
        iters = max(3, min(20, (1 << 31) // m))
        for _ in range(max(2, (1 << 30) // m)):
            ntt.compute_forward(dst, src)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(npass + 1)] for _ in range(iters)]
        for k in range(iters):
            for i in range(npass):
                ev[k][i].record()
                ntt.run_pass(False, i, dst, src if i == 0 else None)
            ev[k][npass].record()
        torch.cuda.synchronize()
        ms = [float(np.median([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(iters)])) for i in range(npass)]
        per24 = [t * 1e3 / (m >> 24) for t in ms]
        print(f"2^{logn} split {split or 'planner':12s} {'OK ' if ok else 'MISMATCH'} [{ntt.describe()}]: "
              f"{' + '.join('%.0f' % (t * 1e3) for t in ms)} = {sum(ms) * 1e3:.0f} us; per 2^24: "
              f"{' + '.join('%.0f' % t for t in per24)} = {sum(per24):.0f} us", flush=True)
        del ntt
    del src, dst
    torch.cuda.empty_cache()
os.environ.pop("SVENTT_SPLIT", None)
