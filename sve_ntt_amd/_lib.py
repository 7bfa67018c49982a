"""ctypes binding of libsventt_hip.so (the C ABI of include/sventt_hip.h).

There is no fallback: if the library is missing or cannot be loaded this
module raises, and every entry point needs a HIP device.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVENTT_HIP_LIBRARY: load another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("SVENTT_HIP_LIBRARY") or os.path.join(_HERE, "libsventt_hip.so")

SVENTT_OK = 0
SVENTT_ERR_INVALID_ARGUMENT = -1
SVENTT_ERR_ALLOC = -2
SVENTT_ERR_HIP = -3
SVENTT_ERR_LOGIC = -4
SVENTT_ERR_NO_DEVICE = -5
SVENTT_ERR_COMM = -6

SVENTT_FORWARD = 1
SVENTT_INVERSE = 2
SVENTT_BOTH = 3
SVENTT_DEVICE_POINTERS = 4
SVENTT_GENERIC_ARITHMETIC = 8
SVENTT_FIXED_POINT = 16

# name -> (restype, argtypes); must list every symbol include/sventt_hip.h declares
_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_vp = ctypes.c_void_p
_int = ctypes.c_int
SYMBOLS = {
    "sventt_plan_create": (_int, [_u64, _u64, _u64, _u32, _u64, _u32, ctypes.POINTER(_vp)]),
    "sventt_plan_create_ex": (_int, [_u64, _u64, _u64, _u32, _u64, _u32, _u64, ctypes.POINTER(_vp)]),
    "sventt_plan_device": (_int, [_vp]),
    "sventt_plan_destroy": (None, [_vp]),
    "sventt_forward": (_int, [_vp, _vp, _vp, _vp]),
    "sventt_inverse": (_int, [_vp, _vp, _vp, _vp]),
    "sventt_plan_num_passes": (_int, [_vp, _int]),
    "sventt_run_pass": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "sventt_plan_pass_tiles_per_block": (_u64, [_vp, _int, _int]),
    "sventt_run_pass_chunk": (_int, [_vp, _int, _int, _vp, _vp, _u32, _u32, _int, _int, _vp]),
    "sventt_sharded_plan_create": (_int, [_u64, _u64, _u64, _u32, _int, _int, _u32,
                                          ctypes.POINTER(_vp)]),
    "sventt_sharded_rows_plan_create": (_int, [_u64, _u64, _u64, _u32, _int, _int, _u32,
                                               ctypes.POINTER(_vp)]),
    "sventt_sharded_columns": (_int, [_vp, _int, _vp, _vp, _vp]),
    "sventt_sharded_forward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sventt_sharded_inverse": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sventt_sharded_forward_transport": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sventt_sharded_inverse_transport": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "sventt_plan_n": (_u64, [_vp]),
    "sventt_plan_batch": (_u64, [_vp]),
    "sventt_plan_modulus": (_u64, [_vp]),
    "sventt_plan_describe": (ctypes.c_char_p, [_vp]),
    "sventt_pointwise_multiply": (_int, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "sventt_to_montgomery": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "sventt_from_montgomery": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "sventt_forward_multiply": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "sventt_transpose": (_int, [_vp, _vp, _u64, _u64, _u64, _u64, _vp]),
    "sventt_transpose_inplace": (_int, [_vp, _u64, _vp]),
    "sventt_host_register": (_int, [_vp, ctypes.c_size_t]),
    "sventt_host_unregister": (_int, [_vp]),
    "sventt_last_error": (ctypes.c_char_p, []),
    "sventt_version": (ctypes.c_char_p, []),
}

_lib = None


class SventtError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library; raises if it is not built (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SventtError(
                f"{LIB_PATH} is missing: build it with `python -m sve_ntt_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc: int) -> None:
    """Map C status codes to the exception types the reference throws."""
    if rc == SVENTT_OK:
        return
    msg = (load().sventt_last_error() or b"").decode()
    if rc == SVENTT_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)  # std::invalid_argument
    if rc == SVENTT_ERR_ALLOC:
        raise MemoryError(msg)  # std::bad_alloc
    if rc == SVENTT_ERR_LOGIC:
        raise SventtError("logic error: " + msg)  # std::logic_error
    raise SventtError(f"status {rc}: {msg}")
