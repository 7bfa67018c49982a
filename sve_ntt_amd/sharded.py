"""One transform sharded over the GPUs of a node (BASELINE config #5).

The reference has no multi-device code; this is the six-step of its
``RecursiveNTT<..., true>`` driver (kernel/recursive.hpp:61-75) with the column
transforms split by column block, ONE exchange (an all-to-all over xGMI, the
global transpose), and the row transforms split by row block:

    n = R x C  (row-major),  rank k of G:
      in : columns [k*C/G, (k+1)*C/G)  as an R x (C/G) slab
      1. local column pass (length R, six-step twiddle with the global column index)
      2. dist.all_to_all_single          (RCCL; contiguous row blocks, no packing)
      3. local row passes; the first one is the length-G column pass of the row
         transform and reads the received pieces in place of a transposition
      out: rows [k*R/G, (k+1)*R/G) = slice k of the bit-reversed result

Every rank moves (G-1)/G of its data through the exchange once; nothing else
crosses GPUs.  One process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL).  The local compute goes through the C ABI (include/sventt_hip.h); tests
inject a host stand-in to exercise this file's index logic with gloo.
"""
from __future__ import annotations

import ctypes

from . import _lib
from . import Modulus, _buffer, _stream_handle


class HipShardEngine:
    """Local passes on the GPU through libsventt_hip.so (no fallback)."""

    def __init__(self, modulus: Modulus, n: int, r_log2: int, rank: int, nranks: int):
        self._lib = _lib.load()
        self._cols = ctypes.c_void_p()
        self._rows = ctypes.c_void_p()
        _lib.check(self._lib.sventt_sharded_plan_create(
            modulus.modulus, modulus.generator, n, r_log2, rank, nranks, _lib.SVENTT_BOTH,
            ctypes.byref(self._cols)))
        _lib.check(self._lib.sventt_sharded_rows_plan_create(
            modulus.modulus, modulus.generator, n, r_log2, rank, nranks, _lib.SVENTT_BOTH,
            ctypes.byref(self._rows)))
        self.n_local = n // nranks
        self.rows_passes = self._lib.sventt_plan_num_passes(self._rows, 0)

    def __del__(self):
        for name in ("_cols", "_rows"):
            h = getattr(self, name, None)
            if h is not None and h.value:
                self._lib.sventt_plan_destroy(h)
                setattr(self, name, None)

    def describe(self) -> str:
        return (self._lib.sventt_plan_describe(self._cols).decode() + " | all-to-all | " +
                self._lib.sventt_plan_describe(self._rows).decode())

    def columns(self, inverse: bool, dst, src, stream=None) -> None:
        d, _k1 = _buffer(dst, self.n_local)
        s, _k2 = _buffer(src, self.n_local)
        _lib.check(self._lib.sventt_sharded_columns(self._cols, int(inverse), d, s,
                                                    _stream_handle(stream)))

    def rows_pass(self, inverse: bool, index: int, dst, src, stream=None) -> None:
        d, _k1 = _buffer(dst, self.n_local)
        s, _k2 = _buffer(src, self.n_local)
        _lib.check(self._lib.sventt_run_pass(self._rows, int(inverse), index, d, s,
                                             _stream_handle(stream)))


class ShardedNTT:
    """Forward/inverse NTT of ``n`` points over ``dist.get_world_size()`` ranks.

    ``src``/``dst`` are this rank's ``n / world`` elements (int64 tensors holding
    uint64 residues): the column slab on the natural-order side, the row block on
    the bit-reversed side.
    """

    def __init__(self, modulus: Modulus, n: int, dist, r_log2: int = 11, engine=None,
                 device=None):
        import torch
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        if self.world < 2:
            raise ValueError("ShardedNTT needs at least two ranks; use NTT on one GPU")
        if n % self.world:
            raise ValueError("n must divide over the ranks")
        self.n = n
        self.n_local = n // self.world
        self.r_log2 = r_log2
        self.engine = engine or HipShardEngine(modulus, n, r_log2, self.rank, self.world)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if engine is None else "cpu"
        self._work = torch.empty(self.n_local, dtype=torch.int64, device=device)
        self._recv = torch.empty(self.n_local, dtype=torch.int64, device=device)
        # phases: column pass, exchange, row passes
        self.num_local_phases = 2 + self.engine.rows_passes

    def describe(self) -> str:
        return self.engine.describe()

    def _mark(self, events, i):
        if events is not None:
            events[i].record()

    def _exchange(self, out, inp):
        """The one collective of the transform.  RCCL moves device buffers directly;
        a gloo group (rehearsals on a box with fewer GPUs than ranks) cannot, so the
        chunks are bounced through host memory there -- transport only, the
        transforms on either side still run in the HIP kernels."""
        if inp.is_cuda and self.dist.get_backend() == "gloo":
            h_in, h_out = inp.cpu(), out.cpu()
            self.dist.all_to_all_single(h_out, h_in)
            out.copy_(h_out)
        else:
            self.dist.all_to_all_single(out, inp)

    def forward(self, dst, src, events=None):
        """natural-order column slab ``src`` -> bit-reversed row block ``dst``."""
        e = self.engine
        self._mark(events, 0)
        e.columns(False, self._work, src)
        self._mark(events, 1)
        # chunk h of _work = rows [h*R/G, (h+1)*R/G) of the slab -> rank h
        self._exchange(self._recv, self._work)
        self._mark(events, 2)
        for i in range(e.rows_passes):
            e.rows_pass(False, i, dst, self._recv if i == 0 else dst)
            self._mark(events, 3 + i)
        return dst

    def inverse(self, dst, src, events=None):
        """bit-reversed row block ``src`` -> natural-order column slab ``dst``."""
        e = self.engine
        k = e.rows_passes
        cur = src
        self._mark(events, 0)
        for i in range(k - 1):
            e.rows_pass(True, i, self._work, cur)
            cur = self._work
            self._mark(events, 1 + i)
        e.rows_pass(True, k - 1, self._recv, cur)  # scatters into piece layout
        self._mark(events, k)
        self._exchange(self._work, self._recv)
        self._mark(events, k + 1)
        e.columns(True, dst, self._work)
        self._mark(events, k + 2)
        return dst
