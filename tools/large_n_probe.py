#!/usr/bin/env python3
"""Three forward transforms of 2^L points (L = argv[1], default 30), for rocprofv3 --pmc passes over the
large-N plans (tools/pmc_large_n.sh).  SVENTT_SPLIT selects an explicit split (plan_core.h)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = 1 << logn
ntt = eng.NTT(eng.BASELINE_MODULUS, m, device_pointers=True)
print("plan:", ntt.describe(), flush=True)
src = torch.empty(m, dtype=torch.int64, device="cuda")
step = 1 << 28
for lo in range(0, m, step):
    src[lo:lo + min(step, m)] = torch.arange(lo, lo + min(step, m), dtype=torch.int64, device="cuda")
dst = torch.empty_like(src)
for _ in range(3):
    ntt.compute_forward(dst, src)
torch.cuda.synchronize()
