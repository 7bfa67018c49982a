#!/usr/bin/env python3
"""Samples rocm-smi (power, sclk) once per 0.5 s while the N = 2^24 forward transform runs back to back for
~12 s, then for a few seconds of idling: is the card power- or clock-limited under this load?
Run on the GPU box:  python tools/power_probe.py"""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

stop = False
samples = []


def sampler():
    while not stop:
        t = time.time()
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True,
                                 text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out = "error %r" % (e,)
        samples.append((t, out))
        time.sleep(0.5)


n = 1 << 24
src = torch.from_numpy(np.random.default_rng(1).integers(0, eng.BASELINE_MODULUS.modulus, n, dtype=np.uint64).view(np.int64)).cuda()
dst = torch.empty_like(src)
ntt = eng.NTT(eng.BASELINE_MODULUS, n, device_pointers=True)
ntt.compute_forward(dst, src)
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
time.sleep(2.0)
t0 = time.time()
marks = []
while time.time() - t0 < 12.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        ntt.compute_forward(dst, src)
    e1.record()
    torch.cuda.synchronize()
    marks.append((time.time() - t0, e0.elapsed_time(e1) / 500 * 1e3))
time.sleep(3.0)
stop = True
th.join()
print("us per transform over time:", " ".join("%.1f@%.1fs" % (us, t) for t, us in marks[::3]))
for t, out in samples:
    lines = [ln for ln in out.splitlines() if ln and not ln.startswith("WARNING")]
    print("%6.1f s | %s" % (t - t0, " | ".join(lines[-2:])[:400]))
