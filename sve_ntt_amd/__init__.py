"""sve_ntt_amd -- MI355X (gfx950) 64-bit NTT engine behind sventt's transform API.

Python host mirror of the reference's boundary for this path
(``sventt::NTT<kernel_type>``, include/sventt/wrapper.hpp:13-83 of the
reference): construct once (twiddle tables are built and uploaded), then
``compute_forward`` / ``compute_inverse`` on caller-owned buffers.  All compute
runs in the hand-written HIP kernels of ``csrc/`` through the C ABI of
``include/sventt_hip.h``; there is no CPU path -- a missing library or device
raises.

    ntt = NTT(Modulus(0xfffffc6e80000001, 3), 1 << 24)      # n0_log2 auto
    ntt.compute_forward(dst, src)        # torch.int64 CUDA tensors, or numpy uint64
    ntt.compute_inverse(dst)             # in place

Semantics = tests/ntt-reference.hpp of the reference: forward is natural-order
in / bit-reversed out, inverse the opposite with the 1/n scaling; outputs are
canonical residues in [0, p).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

from . import _lib
from ._lib import SventtError  # noqa: F401

__all__ = ["Modulus", "NTT", "SventtError", "BASELINE_MODULUS", "transpose", "transpose_inplace"]


@dataclass(frozen=True)
class Modulus:
    """Host mirror of ``sventt::Modulus<modulus, generator>`` (modulus.hpp:14-133)."""

    modulus: int
    generator: int = 0

    def get_modulus(self) -> int:
        return self.modulus

    def get_generator(self) -> int:
        return self.generator

    def get_montgomery_inverse(self) -> int:
        return pow(self.modulus, -1, 1 << 64)

    def multiply(self, a: int, b: int) -> int:
        return a * b % self.modulus

    def power(self, a: int, e: int) -> int:
        return pow(a, e, self.modulus)

    def invert(self, a: int) -> int:
        return pow(a, self.modulus - 2, self.modulus)

    def get_root_forward(self, order: int) -> int:
        if self.generator == 0 or (self.modulus - 1) % order != 0:
            raise ValueError("the field has no such root")  # modulus.hpp:118-120
        return pow(self.generator, (self.modulus - 1) // order, self.modulus)

    def get_root_inverse(self, order: int) -> int:
        return self.invert(self.get_root_forward(order))


BASELINE_MODULUS = Modulus(0xFFFFFC6E80000001, 3)  # README.md:19 of the reference


def _buffer(x, count: int):
    """(address, keepalive) of a caller buffer holding `count` 64-bit words."""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(x, torch.Tensor):
        if x.dtype not in (torch.int64, torch.uint64):
            raise ValueError("tensors must be int64/uint64 (bit pattern of uint64 residues)")
        if not x.is_contiguous() or x.numel() < count:
            raise ValueError("tensor must be contiguous and hold n*batch elements")
        return x.data_ptr(), x
    import numpy as np
    if isinstance(x, np.ndarray):
        if x.dtype != np.uint64 or not x.flags["C_CONTIGUOUS"] or x.size < count:
            raise ValueError("arrays must be contiguous uint64 with n*batch elements")
        return x.ctypes.data, x
    if isinstance(x, int):
        return x, None  # raw device/host address
    raise TypeError(f"unsupported buffer type {type(x)!r}")


def _stream_handle(stream) -> int:
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return torch.cuda.current_stream().cuda_stream
        except ImportError:  # pragma: no cover
            pass
        return 0
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


class NTT:
    """Mirror of ``sventt::NTT<kernel_type>`` (wrapper.hpp:13-83).

    ``m`` is the transform length (``get_m()``), ``n0_log2`` the column length of
    the six-step split (0 = automatic), ``batch`` the number of back-to-back
    independent transforms (the reference's API is batch 1).  ``inverse_divisor``:
    the inverse multiplies by its modular inverse -- 0 = m (the oracle's 1/m),
    1 = unscaled (what the reference's layers compute when none carries an
    ``inverse_factor``, layer/sve/radix-two.hpp:208-235).  ``device_pointers``:
    promise that every buffer is memory of the plan's device (skips the
    pointer-kind queries of each call).  ``arithmetic``: "auto", "generic" (the
    reference's PAdic64 modmul for every modulus) or "fixed_point" (its FixedPoint64
    modmul, modmul/sve/fixed-point-64.hpp) -- results are identical, speed is not.
    """

    def __init__(self, modulus: Modulus, m: int, n0_log2: int = 0, batch: int = 1,
                 enable_forward: bool = True, enable_inverse: bool = True,
                 allocate_huge_pages: bool = True, inverse_divisor: int = 0,
                 device_pointers: bool = False, arithmetic: str = "auto"):
        del allocate_huge_pages  # accepted for signature parity; device tables need no huge pages
        self._lib = _lib.load()
        self.modulus_type = modulus
        self._m = m
        self._batch = batch
        flags = (_lib.SVENTT_FORWARD if enable_forward else 0) | (
            _lib.SVENTT_INVERSE if enable_inverse else 0) | (
            _lib.SVENTT_DEVICE_POINTERS if device_pointers else 0) | {
                "auto": 0,  # Montgomery; the Goldilocks prime gets its own folding reduction
                "generic": _lib.SVENTT_GENERIC_ARITHMETIC,  # PAdic64 (Montgomery) for every modulus
                "fixed_point": _lib.SVENTT_FIXED_POINT,  # FixedPoint64 (Shoup), modulus < 2^63
            }[arithmetic]
        h = ctypes.c_void_p()
        _lib.check(self._lib.sventt_plan_create_ex(modulus.modulus, modulus.generator, m, n0_log2,
                                                   batch, flags, inverse_divisor, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.sventt_plan_destroy(h)
            self._h = None

    def get_m(self) -> int:
        return self._m

    @property
    def batch(self) -> int:
        return self._batch

    def describe(self) -> str:
        return self._lib.sventt_plan_describe(self._h).decode()

    def num_passes(self, inverse: bool = False) -> int:
        return self._lib.sventt_plan_num_passes(self._h, int(inverse))

    def _run(self, fn, dst, src, stream):
        count = self._m * self._batch
        d, keep_d = _buffer(dst, count)
        s, keep_s = (d, keep_d) if src is None else _buffer(src, count)
        _lib.check(fn(self._h, d, s, _stream_handle(stream)))
        return dst

    def compute_forward(self, dst, src=None, stream=None):
        """wrapper.hpp:50-65 (``src=None`` = the in-place overload)."""
        return self._run(self._lib.sventt_forward, dst, src, stream)

    def compute_inverse(self, dst, src=None, stream=None):
        """wrapper.hpp:67-82."""
        return self._run(self._lib.sventt_inverse, dst, src, stream)

    def run_pass(self, inverse: bool, index: int, dst, src=None, stream=None):
        count = self._m * self._batch
        d, _k1 = _buffer(dst, count)
        s, _k2 = (d, None) if src is None else _buffer(src, count)
        _lib.check(self._lib.sventt_run_pass(self._h, int(inverse), index, d, s,
                                             _stream_handle(stream)))
        return dst

    def pointwise_multiply(self, dst, a, b, stream=None):
        """dst = a*b mod p element-wise (gaussian-polynomial.hpp:201-212 of the reference)."""
        count = self._m * self._batch
        d, _k0 = _buffer(dst, count)
        pa, _k1 = _buffer(a, count)
        pb, _k2 = _buffer(b, count)
        _lib.check(self._lib.sventt_pointwise_multiply(self._h, d, pa, pb, count,
                                                       _stream_handle(stream)))
        return dst

    def to_montgomery(self, dst, src=None, stream=None):
        """dst = src * 2^64 mod p (PAdic64SVE::to_montgomery, p-adic-64.hpp:64-69)."""
        return self._convert(self._lib.sventt_to_montgomery, dst, src, stream)

    def from_montgomery(self, dst, src=None, stream=None):
        return self._convert(self._lib.sventt_from_montgomery, dst, src, stream)

    def _convert(self, fn, dst, src, stream):
        count = self._m * self._batch
        d, _k1 = _buffer(dst, count)
        s, _k2 = (d, None) if src is None else _buffer(src, count)
        _lib.check(fn(self._h, d, s, count, _stream_handle(stream)))
        return dst

    def compute_forward_multiply(self, dst, src, operand_montgomery, stream=None):
        """dst = forward(src) * operand element-wise, the product fused into the last pass
        (the reference's caller does the two separately, gaussian-polynomial.hpp:199-212).
        ``operand_montgomery``: a spectrum converted once with ``to_montgomery``."""
        count = self._m * self._batch
        d, _k0 = _buffer(dst, count)
        s, _k1 = (d, None) if src is None else _buffer(src, count)
        o, _k2 = _buffer(operand_montgomery, count)
        _lib.check(self._lib.sventt_forward_multiply(self._h, d, s, o, _stream_handle(stream)))
        return dst


def transpose(dst, src, src_rows: int, src_cols: int, ld_dst: int | None = None,
              ld_src: int | None = None, stream=None):
    """``dst[ld_dst*c + r] = src[ld_src*r + c]`` -- the reference's
    ``Transpose...::transpose(dst, src, src_rows, src_cols, ld_dst, ld_src)``
    (transposition/sve/in-register.hpp:115-206) as one LDS-tiled HIP kernel."""
    ld_dst = src_rows if ld_dst is None else ld_dst
    ld_src = src_cols if ld_src is None else ld_src
    lib = _lib.load()
    d, _k1 = _buffer(dst, max(0, ld_dst * (src_cols - 1) + src_rows) if src_cols else 0)
    s, _k2 = _buffer(src, max(0, ld_src * (src_rows - 1) + src_cols) if src_rows else 0)
    _lib.check(lib.sventt_transpose(d, s, src_rows, src_cols, ld_dst, ld_src, _stream_handle(stream)))
    return dst


def transpose_inplace(dst, dim: int, stream=None):
    """Square, in place: ``Transpose...::transpose(dst, dim)`` (in-register.hpp:215-375)."""
    lib = _lib.load()
    d, _k = _buffer(dst, dim * dim)
    _lib.check(lib.sventt_transpose_inplace(d, dim, _stream_handle(stream)))
    return dst
