"""CPU checkers for the NTT path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product (``sve_ntt_amd``) never does.

Two back ends, same call signatures:

* ``port``      -- ``oracle/ntt_oracle.c`` (our C restatement of
                   ``tests/ntt-reference.hpp`` and the scalar field helpers)
* ``reference`` -- ``oracle/_ref/libntt_ref.so``: the upstream headers
                   themselves, compiled in place by ``oracle/Makefile``
                   (present only if built where /root/reference exists; the
                   built library travels to the GPU box with the snapshot)

Parity status: PINNED (see tests/test_oracle.py and tests/golden/).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "libntt_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libntt_ref.so")

_u64 = ctypes.c_uint64
_p64 = ctypes.POINTER(ctypes.c_uint64)
_pint = ctypes.POINTER(ctypes.c_int)

BASELINE_P = 0xFFFFFC6E80000001  # README.md:19 of the reference
BASELINE_G = 3
TEST62_P = 0x3A00000000000001  # tests/ntt-tests/*.hpp:4-5
TEST62_G = 3
GOLDILOCKS_P = 0xFFFFFFFF00000001  # tests/test-modulus.cpp:13
GOLDILOCKS_G = 7
INPUT_I1_START = 0x0123456789ABCDEF  # SURVEY.md 8(d), input I1


def build(force: bool = False) -> None:
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_PORT_SO) or (
        os.path.getmtime(_PORT_SO) < os.path.getmtime(os.path.join(_HERE, "ntt_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-s", os.path.join(_HERE, "libntt_oracle.so")],
                       check=True)
    if os.path.isdir(os.environ.get("REFERENCE_ROOT", "/root/reference")) and (
        force or not os.path.exists(_REF_SO)
    ):
        subprocess.run(["make", "-C", _HERE, "-s", "ref"], check=True)


def _ptr(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p64)


class _Port:
    kind = "port"

    def __init__(self):
        build()
        L = ctypes.CDLL(_PORT_SO)
        L.oracle_modmul.restype = _u64
        L.oracle_modmul.argtypes = [_u64, _u64, _u64]
        L.oracle_modpow.restype = _u64
        L.oracle_modpow.argtypes = [_u64, _u64, _u64]
        L.oracle_modadd.restype = _u64
        L.oracle_modadd.argtypes = [_u64, _u64, _u64]
        L.oracle_modsub.restype = _u64
        L.oracle_modsub.argtypes = [_u64, _u64, _u64]
        L.oracle_montgomery_inverse.restype = _u64
        L.oracle_montgomery_inverse.argtypes = [_u64]
        for f in (L.oracle_root_forward, L.oracle_root_inverse):
            f.restype = _u64
            f.argtypes = [_u64, _u64, _u64, _pint]
        for f in (L.oracle_to_montgomery, L.oracle_from_montgomery,
                  L.oracle_padic_precompute):
            f.restype = _u64
            f.argtypes = [_u64, _u64]
        L.oracle_padic_multiply_normalize.restype = _u64
        L.oracle_padic_multiply_normalize.argtypes = [_u64, _u64, _u64, _u64]
        L.oracle_bitreverse64.restype = _u64
        L.oracle_bitreverse64.argtypes = [_u64]
        for f in (L.oracle_ntt_forward, L.oracle_ntt_inverse):
            f.restype = ctypes.c_int
            f.argtypes = [_p64, _p64, _u64, _u64, _u64]
        for f in (L.oracle_ntt_forward_sixstep, L.oracle_ntt_inverse_sixstep):
            f.restype = ctypes.c_int
            f.argtypes = [_p64, _p64, _u64, _u64, _u64, _u64]
        L.oracle_fill_iota.restype = None
        L.oracle_fill_iota.argtypes = [_p64, _u64, _u64]
        L.oracle_fill_splitmix.restype = None
        L.oracle_fill_splitmix.argtypes = [_p64, _u64, _u64, _u64]
        L.oracle_digest.restype = None
        L.oracle_digest.argtypes = [_p64, _u64, _p64]
        self.L = L

    # -- transforms ---------------------------------------------------------
    def forward(self, src: np.ndarray, N: int, g: int) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint64)
        dst = np.empty_like(src)
        if src.size == 1:
            dst[:] = src  # the reference leaves dst untouched for m == 1
        rc = self.L.oracle_ntt_forward(_ptr(dst), _ptr(src), src.size, N, g)
        if rc != 0:
            raise ValueError("Transform length must be a power of two for now")
        return dst

    def inverse(self, src: np.ndarray, N: int, g: int) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint64)
        dst = np.empty_like(src)
        rc = self.L.oracle_ntt_inverse(_ptr(dst), _ptr(src), src.size, N, g)
        if rc != 0:
            raise ValueError("Transform length must be a power of two for now")
        return dst

    def forward_sixstep(self, src, R: int, N: int, g: int) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint64)
        dst = np.empty_like(src)
        rc = self.L.oracle_ntt_forward_sixstep(_ptr(dst), _ptr(src), src.size, R, N, g)
        if rc != 0:
            raise ValueError("bad six-step shape")
        return dst

    def inverse_sixstep(self, src, R: int, N: int, g: int) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint64)
        dst = np.empty_like(src)
        rc = self.L.oracle_ntt_inverse_sixstep(_ptr(dst), _ptr(src), src.size, R, N, g)
        if rc != 0:
            raise ValueError("bad six-step shape")
        return dst

    # -- field helpers ------------------------------------------------------
    def modmul(self, x, y, N):
        return int(self.L.oracle_modmul(x, y, N))

    def modpow(self, x, e, N):
        return int(self.L.oracle_modpow(x, e, N))

    def modadd(self, a, b, N):
        return int(self.L.oracle_modadd(a, b, N))

    def modsub(self, a, b, N):
        return int(self.L.oracle_modsub(a, b, N))

    def montgomery_inverse(self, N):
        return int(self.L.oracle_montgomery_inverse(N))

    def root_forward(self, N, g, order):
        ok = ctypes.c_int(0)
        r = int(self.L.oracle_root_forward(N, g, order, ctypes.byref(ok)))
        if not ok.value:
            raise ValueError("the field has no such root")
        return r

    def root_inverse(self, N, g, order):
        ok = ctypes.c_int(0)
        r = int(self.L.oracle_root_inverse(N, g, order, ctypes.byref(ok)))
        if not ok.value:
            raise ValueError("the field has no such root")
        return r

    def to_montgomery(self, b, N):
        return int(self.L.oracle_to_montgomery(b, N))

    def from_montgomery(self, b, N):
        return int(self.L.oracle_from_montgomery(b, N))

    def padic_precompute(self, b, N):
        return int(self.L.oracle_padic_precompute(b, N))

    def padic_multiply_normalize(self, a, b, bp, N):
        return int(self.L.oracle_padic_multiply_normalize(a, b, bp, N))

    def bitreverse64(self, x):
        return int(self.L.oracle_bitreverse64(x))

    # -- inputs / digests -----------------------------------------------------
    def fill_iota(self, m: int, start: int) -> np.ndarray:
        a = np.empty(m, dtype=np.uint64)
        self.L.oracle_fill_iota(_ptr(a), m, start)
        return a

    def fill_splitmix(self, m: int, seed: int, N: int) -> np.ndarray:
        a = np.empty(m, dtype=np.uint64)
        self.L.oracle_fill_splitmix(_ptr(a), m, seed, N)
        return a

    def digest(self, v: np.ndarray) -> tuple[int, int, int]:
        v = np.ascontiguousarray(v, dtype=np.uint64)
        out = (ctypes.c_uint64 * 3)()
        self.L.oracle_digest(_ptr(v), v.size, out)
        return int(out[0]), int(out[1]), int(out[2])


class _Reference:
    """The upstream headers compiled in place (oracle/_ref)."""

    kind = "reference"

    def __init__(self):
        if not os.path.exists(_REF_SO):
            build()
        if not os.path.exists(_REF_SO):
            raise FileNotFoundError(_REF_SO)
        L = ctypes.CDLL(_REF_SO)
        for f in (L.ref_ntt_forward, L.ref_ntt_inverse):
            f.restype = ctypes.c_int
            f.argtypes = [_p64, _p64, _u64, _u64, _u64]
        for f in (L.ref_root_forward, L.ref_root_inverse):
            f.restype = _u64
            f.argtypes = [_u64, _u64, _pint]
        for f in (L.ref_montgomery_inverse, L.ref_generator):
            f.restype = _u64
            f.argtypes = [_u64, _pint]
        for f in (L.ref_to_montgomery, L.ref_from_montgomery, L.ref_padic_precompute):
            f.restype = _u64
            f.argtypes = [_u64, _u64, _pint]
        L.ref_padic_multiply_lazy.restype = _u64
        L.ref_padic_multiply_lazy.argtypes = [_u64, _u64, _u64, _u64, _pint]
        L.ref_bitreverse.restype = _u64
        L.ref_bitreverse.argtypes = [_u64]
        self.L = L

    def forward(self, src, N, g):
        src = np.ascontiguousarray(src, dtype=np.uint64)
        dst = np.empty_like(src)
        if src.size == 1:
            dst[:] = src
        if self.L.ref_ntt_forward(_ptr(dst), _ptr(src), src.size, N, g) != 0:
            raise ValueError("Transform length must be a power of two for now")
        return dst

    def inverse(self, src, N, g):
        src = np.ascontiguousarray(src, dtype=np.uint64)
        dst = np.empty_like(src)
        if self.L.ref_ntt_inverse(_ptr(dst), _ptr(src), src.size, N, g) != 0:
            raise ValueError("Transform length must be a power of two for now")
        return dst

    def _call(self, fn, *args):
        ok = ctypes.c_int(0)
        r = int(fn(*args, ctypes.byref(ok)))
        if ok.value == -1:
            raise KeyError("prime not instantiated in oracle/ref_shim.cpp")
        if ok.value == 0:
            raise ValueError("the field has no such root")
        return r

    def root_forward(self, N, order):
        return self._call(self.L.ref_root_forward, N, order)

    def root_inverse(self, N, order):
        return self._call(self.L.ref_root_inverse, N, order)

    def montgomery_inverse(self, N):
        return self._call(self.L.ref_montgomery_inverse, N)

    def generator(self, N):
        return self._call(self.L.ref_generator, N)

    def to_montgomery(self, b, N):
        return self._call(self.L.ref_to_montgomery, N, b)

    def from_montgomery(self, b, N):
        return self._call(self.L.ref_from_montgomery, N, b)

    def padic_precompute(self, b, N):
        return self._call(self.L.ref_padic_precompute, N, b)

    def padic_multiply_lazy(self, a, b, bp, N):
        return self._call(self.L.ref_padic_multiply_lazy, N, a, b, bp)

    def bitreverse64(self, x):
        return int(self.L.ref_bitreverse(x))


_port = None
_ref = None


def port() -> _Port:
    global _port
    if _port is None:
        _port = _Port()
    return _port


def reference() -> _Reference:
    global _ref
    if _ref is None:
        _ref = _Reference()
    return _ref


def have_reference() -> bool:
    try:
        reference()
        return True
    except (FileNotFoundError, OSError):
        return False
