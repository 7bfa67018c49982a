#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer path (what NTT::compute_forward on PageMemory
costs): pageable numpy buffers vs pinned (page-locked) buffers, N = 2^24."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng
n = 1 << 24
ntt = eng.NTT(eng.BASELINE_MODULUS, n)
rng = np.random.default_rng(0)
src = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
dst = np.empty_like(src)
def timeit(fn, reps=10):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
t = timeit(lambda: ntt.compute_forward(dst, src))
print("pageable host buffers: %.2f ms per transform, %.2e elem/s, %.1f GB/s each way" % (t * 1e3, n / t, 8 * n / t / 1e9 * 2))
ps = torch.from_numpy(src.view(np.int64)).pin_memory()
pd = torch.empty_like(ps).pin_memory()
t = timeit(lambda: ntt.compute_forward(pd.data_ptr(), ps.data_ptr()))
print("pinned host buffers:   %.2f ms per transform, %.2e elem/s" % (t * 1e3, n / t))
assert np.array_equal(pd.numpy().view(np.uint64), dst)

# what the facade's PageMemory does: the caller's own (numpy) buffers page-locked through the engine
import ctypes
from sve_ntt_amd import _lib
lib = _lib.load()
for arr in (src, dst):
    assert lib.sventt_host_register(ctypes.c_void_p(arr.ctypes.data), arr.nbytes) == 0
t = timeit(lambda: ntt.compute_forward(dst, src))
print("registered (sventt_host_register) buffers: %.2f ms per transform, %.2e elem/s" % (t * 1e3, n / t))
for arr in (src, dst):
    assert lib.sventt_host_unregister(ctypes.c_void_p(arr.ctypes.data)) == 0
