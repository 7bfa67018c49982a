#!/usr/bin/env python3
"""Where a tile's time goes: per-wave time stamps (s_memtime) of every phase of every step, from a library built
with -DSVENTT_TRACE (tools/build_variant.sh trace -DSVENTT_TRACE).  Run on the GPU box:
    SVENTT_HIP_LIBRARY=sve_ntt_amd/build/lib_trace.so python tools/trace_tiles.py [log2n]
Phases of a step: wait = barrier / wave fence before the step, issue = address arithmetic + load or ds_read issue,
data = until the step's inputs are in registers, compute = the butterfly stages, out = twist / product + stores or
ds_writes issued.  "end" is the end of the instruction stream (the stores are still in flight)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lib = ctypes.CDLL(os.environ["SVENTT_HIP_LIBRARY"])
lib.sventt_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
SLOTS = 32
n = 1 << log2n
src = torch.from_numpy(np.random.default_rng(1).integers(0, eng.BASELINE_MODULUS.modulus, n, dtype=np.uint64).view(np.int64)).cuda()
dst = torch.empty_like(src)
back = torch.empty_like(src)
ntt = eng.NTT(eng.BASELINE_MODULUS, n, device_pointers=True)
print("plan:", ntt.describe())
for _ in range(300):
    ntt.compute_forward(dst, src)
torch.cuda.synchronize()

for inverse in (False, True):
    names = ntt.describe().split(" | ")
    if inverse:
        names = names[::-1]
        ntt.compute_forward(dst, src)
    for i in range(ntt.num_passes(inverse)):
        a, b = (dst, src) if not inverse else (back, dst)
        for _ in range(20):  # keep the clocks where they are under load
            ntt.compute_inverse(back, dst) if inverse else ntt.compute_forward(dst, src)
        for j in range(i):
            ntt.run_pass(inverse, j, a, b if j == 0 else None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ntt.run_pass(inverse, i, a, b if i == 0 else None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        waves = (n >> 13) * 8
        buf = np.zeros(waves * SLOTS, dtype=np.uint64)
        assert lib.sventt_debug_trace_read(buf.ctypes.data, buf.size) == 0
        t = buf.reshape(waves, SLOTS).astype(np.int64)
        steps = 0
        while steps < 5 and t[:, 6 * steps].min() > 0:
            steps += 1
        if os.environ.get("TRACE_DEBUG"):
            print("zeros per slot:", (t == 0).sum(axis=0).tolist())
            print("first waves:", t[:2, :8].tolist(), t[:2, 28:].tolist())
        # s_memtime counters are not comparable across the chip; the 100 MHz s_memrealtime is (slots 28/29:
        # start and end of each wave): calibrate s_memtime against it wave by wave
        rt = (t[:, 29] - t[:, 28]).astype(np.float64) / 100.0  # us
        mt = (t[:, 30] - t[:, 0]).astype(np.float64)
        tick = float(np.median(rt / mt))  # us per s_memtime tick
        span = (t[:, 29].max() - t[:, 28].min()) / 100.0
        print(f"\n== {'inverse' if inverse else 'forward'} pass {i}: {names[i]}: {us:.1f} us by events, "
              f"{span:.1f} us from first to last wave by the 100 MHz clock, {1 / tick:.0f} s_memtime ticks/us, {steps} steps, {waves} waves")
        tot = np.zeros(waves)
        for s in range(steps):
            p = t[:, 6 * s:6 * s + 6]
            seg = {"wait": p[:, 1] - p[:, 0], "issue": p[:, 2] - p[:, 1], "data": p[:, 3] - p[:, 2],
                   "compute": p[:, 4] - p[:, 3], "out": p[:, 5] - p[:, 4]}
            line = "  ".join(f"{k} {v.mean() * tick:6.2f}" for k, v in seg.items())
            gap = (t[:, 6 * (s + 1)] - p[:, 5]).mean() * tick if s + 1 < steps else (t[:, 30] - p[:, 5]).mean() * tick
            print(f"  step {s}: {line}   -> next {gap:5.2f}   (us per wave, mean)")
        life = (t[:, 30] - t[:, 0]) * tick
        print(f"  wave lifetime (first stamp to end of stream): mean {life.mean():.2f} us, p10 {np.percentile(life, 10):.2f}, "
              f"p90 {np.percentile(life, 90):.2f}")
        # per workgroup and per CU slot: what happens between two workgroups of the same CU
        wg_start = t[:, 28].reshape(-1, 8).min(axis=1) / 100.0  # us, chip-wide clock (10 ns resolution)
        wg_end = t[:, 29].reshape(-1, 8).max(axis=1) / 100.0
        hw = (buf.reshape(waves, SLOTS)[:, 31] & np.uint64(0xffffffff)).astype(np.int64).reshape(-1, 8)[:, 0]
        cu = (hw >> 8) & 0xf
        sh = (hw >> 12) & 0x1
        se = (hw >> 13) & 0x7
        xcd = np.arange(len(hw)) & 7
        key = ((xcd * 8 + se) * 2 + sh) * 16 + cu
        gaps, conc = [], []
        for k in np.unique(key):
            idx = np.where(key == k)[0]
            order = idx[np.argsort(wg_start[idx])]
            ends = np.sort(wg_end[idx])
            starts = wg_start[order]
            # two workgroups share the CU: the k-th start (k >= 2) follows the (k-2)-th end
            for j in range(2, len(starts)):
                gaps.append(starts[j] - ends[j - 2])
        gaps = np.array(gaps)
        print(f"  CUs seen: {len(np.unique(key))}; workgroup duration mean {(wg_end - wg_start).mean():.2f} us; "
              f"end of a workgroup's stream -> first stamp of its successor on that CU: median {np.median(gaps):.2f} us, "
              f"p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f}")

        # How many of a CU's 16 resident waves are in a butterfly phase (compute + out) at a time: s_memtime is
        # common to the CUs of an XCD, so the stamps of a CU's waves are comparable.  Phase-locked workgroups
        # (both loading, then both computing) show up as time with few computing waves.
        wave_key = np.repeat(key, 8)
        hist = np.zeros(17)
        span_ticks = 0.0
        for k in np.unique(key):
            w = np.where(wave_key == k)[0]
            lo, hi = t[w, 0].min(), t[w, 30].max()
            grid = np.linspace(lo, hi, 400)
            busy = np.zeros(len(grid), dtype=np.int64)
            for s in range(steps):
                a0, a1 = t[w, 6 * s + 3], t[w, 6 * s + 5]
                busy += ((grid[None, :] >= a0[:, None]) & (grid[None, :] < a1[:, None])).sum(axis=0)
            hist += np.bincount(np.minimum(busy, 16), minlength=17)
            span_ticks += hi - lo
        hist /= hist.sum()
        print("  share of a CU's time with n waves in a butterfly phase (of 16 resident): "
              f"n=0: {hist[0]:.3f}, 1-4: {hist[1:5].sum():.3f}, 5-8: {hist[5:9].sum():.3f}, 9-12: {hist[9:13].sum():.3f}, "
              f"13-16: {hist[13:].sum():.3f}; mean {np.dot(hist, np.arange(17)):.2f}")
