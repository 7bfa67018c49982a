"""Loader for tests/cpu_sim/libsim.so -- the sequential host replay of the tile
kernels and planner.  TEST-ONLY: nothing in sve_ntt_amd/ imports this."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "cpu_sim", "sim.cpp")
_SO = os.path.join(_HERE, "cpu_sim", "libsim.so")
_CSRC = os.path.join(os.path.dirname(_HERE), "sve_ntt_amd", "csrc")

_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_p64 = ctypes.POINTER(ctypes.c_uint64)
_int = ctypes.c_int

_lib = None


def _stale() -> bool:
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    deps = [_SRC] + [os.path.join(_CSRC, f) for f in ("field64.h", "tile_ntt.h", "registry.h", "plan_core.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if _stale():
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                            "-o", _SO, _SRC], check=True)
        L = ctypes.CDLL(_SO)
        L.sim_last_error.restype = ctypes.c_char_p
        L.sim_transform.restype = _int
        L.sim_transform.argtypes = [_u64, _u64, _u64, _u32, _u64, _int, _p64, _p64]
        L.sim_check_set_mappings.restype = _int
        L.sim_check_set_mappings.argtypes = []
        L.sim_group_barriers.restype = _int
        L.sim_group_barriers.argtypes = [_int] * 6
        L.sim_transform_flags.restype = _int
        L.sim_transform_flags.argtypes = [_u64, _u64, _u64, _u32, _u64, _int, _u32, _p64, _p64]
        L.sim_forward_multiply.restype = _int
        L.sim_forward_multiply.argtypes = [_u64, _u64, _u64, _u32, _u64, _p64, _p64, _p64]
        L.sim_sharded_columns.restype = _int
        L.sim_sharded_columns.argtypes = [_u64, _u64, _u64, _u32, _int, _int, _int, _p64, _p64]
        L.sim_sharded_rows_num_passes.restype = _int
        L.sim_sharded_rows_num_passes.argtypes = [_u64, _u64, _u64, _u32, _int, _int]
        L.sim_sharded_rows_pass.restype = _int
        L.sim_sharded_rows_pass.argtypes = [_u64, _u64, _u64, _u32, _int, _int, _int, _int, _p64, _p64]
        L.sim_sharded_chunk.restype = _int
        L.sim_sharded_chunk.argtypes = [_u64, _u64, _u64, _u32, _int, _int, _int, _int, _u32, _u32,
                                        _int, _int, _p64, _p64]
        L.sim_sharded_tiles_per_block.restype = ctypes.c_int64
        L.sim_sharded_tiles_per_block.argtypes = [_u64, _u64, _u64, _u32, _int, _int, _int]
        L.sim_plan_shape.restype = _int
        L.sim_plan_shape.argtypes = [_u64, _u64, _u64, _u32, _u64, _int,
                                     ctypes.POINTER(ctypes.c_int64), _int]
        for f in (L.sim_montmul, L.sim_addmod, L.sim_submod):
            f.restype = _u64
            f.argtypes = [_u64, _u64, _u64]
        L.sim_montgomery_inverse.restype = _u64
        L.sim_montgomery_inverse.argtypes = [_u64]
        L.sim_to_montgomery.restype = _u64
        L.sim_to_montgomery.argtypes = [_u64, _u64]
        L.sim_lds_phys.restype = _u32
        L.sim_lds_phys.argtypes = [_u32]
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p64)


class SimError(ValueError):
    pass


def transform(src: np.ndarray, p: int, g: int, n: int, n0_log2: int = 0, batch: int = 1,
              inverse: bool = False, flags: int = 0) -> np.ndarray:
    """flags: extra plan flags (8 = generic arithmetic, 16 = FixedPoint64 back end)"""
    L = load()
    src = np.ascontiguousarray(src, dtype=np.uint64)
    dst = np.full_like(src, 0x5555555555555555)
    rc = L.sim_transform_flags(p, g, n, n0_log2, batch, int(inverse), flags, _ptr(dst), _ptr(src))
    if rc != 0:
        raise SimError((rc, L.sim_last_error().decode()))
    return dst


def forward_multiply(src: np.ndarray, operand_montgomery: np.ndarray, p: int, g: int, n: int,
                     n0_log2: int = 0, batch: int = 1) -> np.ndarray:
    L = load()
    src = np.ascontiguousarray(src, dtype=np.uint64)
    op = np.ascontiguousarray(operand_montgomery, dtype=np.uint64)
    dst = np.full_like(src, 0x5555555555555555)
    rc = L.sim_forward_multiply(p, g, n, n0_log2, batch, _ptr(dst), _ptr(src), _ptr(op))
    if rc != 0:
        raise SimError((rc, L.sim_last_error().decode()))
    return dst


def plan_shape(p: int, g: int, n: int, n0_log2: int = 0, batch: int = 1, inverse: bool = False):
    L = load()
    out = (ctypes.c_int64 * 48)()
    k = L.sim_plan_shape(p, g, n, n0_log2, batch, int(inverse), out, 8)
    if k < 0:
        raise SimError((k, L.sim_last_error().decode()))
    names = ("kind", "logl", "f0", "logt", "grid", "loge")
    return [dict(zip(names, out[6 * i:6 * i + 6])) for i in range(k)]


class SimShardEngine:
    """Host stand-in with the interface of sve_ntt_amd.sharded.HipShardEngine, so that
    the distributed driver's index logic runs under gloo without a GPU."""

    def __init__(self, modulus, n: int, r_log2: int, rank: int, nranks: int):
        self.L = load()
        self.args = (modulus.modulus, modulus.generator, n, r_log2, rank, nranks)
        self.n_local = n // nranks
        self.rows_passes = self.L.sim_sharded_rows_num_passes(*self.args)
        if self.rows_passes < 0:
            raise SimError(self.L.sim_last_error().decode())
        self.chunk_limits = (int(self.L.sim_sharded_tiles_per_block(*self.args, 0)),
                             int(self.L.sim_sharded_tiles_per_block(*self.args, 1)))

    def describe(self) -> str:
        return "host replay"

    @staticmethod
    def _np(t):
        return t.numpy().view(np.uint64)

    def _chunk(self, which, inverse, dst, src, k, nchunks, dst_compact, src_compact):
        rc = self.L.sim_sharded_chunk(*self.args, which, int(inverse), k, nchunks, int(dst_compact),
                                      int(src_compact), _ptr(self._np(dst)), _ptr(self._np(src)))
        if rc:
            raise SimError(self.L.sim_last_error().decode())

    def columns_chunk(self, inverse, dst, src, k, nchunks):
        self._chunk(0, inverse, dst, src, k, nchunks, not inverse, inverse)

    def exchange_side_chunk(self, inverse, dst, src, k, nchunks):
        self._chunk(1, inverse, dst, src, k, nchunks, inverse, not inverse)

    def rows_pass(self, inverse, index, dst, src, stream=None):
        rc = self.L.sim_sharded_rows_pass(*self.args, int(inverse), index, _ptr(self._np(dst)),
                                          _ptr(self._np(src)))
        if rc:
            raise SimError(self.L.sim_last_error().decode())
