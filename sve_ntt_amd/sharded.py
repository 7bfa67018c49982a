"""One transform sharded over the GPUs of a node (BASELINE config #5).

The reference has no multi-device code; this is the six-step of its
``RecursiveNTT<..., true>`` driver (kernel/recursive.hpp:61-75) with the column
transforms split by column block, ONE exchange (an all-to-all over xGMI, the
global transpose), and the row transforms split by row block:

    n = R x C  (row-major),  rank k of G:
      in : columns [k*C/G, (k+1)*C/G)  as an R x (C/G) slab
      1. local column pass (length R, six-step twiddle with the global column index)
      2. dist.all_to_all_single          (RCCL; contiguous row blocks, no packing)
      3. local row passes; the first one is the first column pass of the length-C row
         transform (G * 2^k long) and reads the received pieces through a two-level
         stride in place of a transposition -- no gather sweep (csrc/plan_core.h:
         sharded_row_split); N = 2^30 on 8 ranks: col 2^11 | exchange | col 2^7 | row 2^12
      out: rows [k*R/G, (k+1)*R/G) = slice k of the bit-reversed result

Callers that hold the natural-order vector as contiguous chunks use forward_natural /
inverse_natural (one more exchange up front).

Steps 1-3 are PIPELINED over ``chunks`` column ranges: the column pass writes
chunk j compactly, its all-to-all is started asynchronously, the column pass of
chunk j+1 runs meanwhile; the gather pass of chunk j waits only for exchange j.
Every rank moves (G-1)/G of its data through the exchange once; nothing else
crosses GPUs.  One process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL).  The local compute goes through the C ABI (include/sventt_hip.h); tests
inject a host stand-in to exercise this file's index logic with gloo.
"""
from __future__ import annotations

import ctypes
import os

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
        # the exchange can be cut into any number of chunks dividing both tile counts
        self.chunk_limits = (
            int(self._lib.sventt_plan_pass_tiles_per_block(self._cols, 0, 0)),
            int(self._lib.sventt_plan_pass_tiles_per_block(self._rows, 0, 0)))

    def __del__(self):
        for name in ("_cols", "_rows"):
            h = getattr(self, name, None)
            if h is not None and h.value:
                self._lib.sventt_plan_destroy(h)
                setattr(self, name, None)

    def describe(self) -> str:
        return (self._lib.sventt_plan_describe(self._cols).decode() + " | all-to-all | " +
                self._lib.sventt_plan_describe(self._rows).decode())

    def _chunk(self, plan, inverse, index, dst, src, k, nchunks, dst_compact, src_compact):
        d, _k1 = _buffer(dst, 1)
        s, _k2 = _buffer(src, 1)
        _lib.check(self._lib.sventt_run_pass_chunk(plan, int(inverse), index, d, s, k, nchunks,
                                                   int(dst_compact), int(src_compact),
                                                   _stream_handle(None)))

    def columns_chunk(self, inverse: bool, dst, src, k: int, nchunks: int) -> None:
        """forward: slab -> compact chunk k;  inverse: compact chunk k -> slab."""
        self._chunk(self._cols, inverse, 0, dst, src, k, nchunks, not inverse, inverse)

    def exchange_side_chunk(self, inverse: bool, dst, src, k: int, nchunks: int) -> None:
        """forward: received chunk k -> whole rows (gather);  inverse: rows -> send chunk k."""
        index = self.rows_passes - 1 if inverse else 0
        self._chunk(self._rows, inverse, index, dst, src, k, nchunks, inverse, not inverse)

    def rows_pass(self, inverse: bool, index: int, dst, src, stream=None) -> None:
        d, _k1 = _buffer(dst, self.n_local)
        s, _k2 = _buffer(src, self.n_local)
        _lib.check(self._lib.sventt_run_pass(self._rows, int(inverse), index, d, s,
                                             _stream_handle(stream)))


class ShardedNTT:
    """Forward/inverse NTT of ``n`` points over ``dist.get_world_size()`` ranks.

    ``src``/``dst`` are this rank's ``n / world`` elements (int64 tensors holding
    uint64 residues): the column slab on the natural-order side, the row block on
    the bit-reversed side.  ``chunks``: pieces the exchange is pipelined in
    (default: env SVENTT_A2A_CHUNKS, else 4; reduced to what
    the tile counts allow).
    """

    def __init__(self, modulus: Modulus, n: int, dist, r_log2: int = 11, engine=None,
                 device=None, chunks: int | None = None):
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
        # Four chunks whatever the rank count: the exchange is the long pole (two ranks move half of
        # their data over ONE xGMI link, eight ranks 7/8 over seven), and with K chunks the local passes
        # cost local/K + (what the exchange does not cover) instead of all of local.  Cutting the column
        # pass into four launches costs 139 against 120 us per 2^24 elements
        # (profiles/r01/ubench_copy_and_sharded_local.txt), far less than it hides.  (r01/r02 ran two
        # ranks unpipelined.)
        default_chunks = "4"
        want = chunks if chunks is not None else int(os.environ.get("SVENTT_A2A_CHUNKS", default_chunks))
        k = max(1, want)
        while k > 1 and any(lim % k for lim in self.engine.chunk_limits):
            k -= 1
        self.chunks = k
        self._work = torch.empty(self.n_local, dtype=torch.int64, device=device)
        self._recv = torch.empty(self.n_local, dtype=torch.int64, device=device)
        self._back = None  # third buffer, inverse only (allocated on first use)
        # phases bench.py times with events: [column pass + exchanges started],
        # [exchange waits + gather pass], then each remaining row pass
        local = [nm for nm in self.engine.describe().split(" | ") if nm != "all-to-all"]
        if len(local) == 1 + self.engine.rows_passes:
            self.phase_names = ([local[0] + " (+ all-to-all started)",
                                 "all-to-all wait + " + local[1]] + local[2:])
        else:
            self.phase_names = [f"phase {i}" for i in range(1 + self.engine.rows_passes)]
        self.num_local_phases = 1 + self.engine.rows_passes

    def describe(self) -> str:
        return self.engine.describe()

    def _mark(self, events, i):
        if events is not None:
            events[i].record()

    def _piece(self, buf, k):
        m = self.n_local // self.chunks
        return buf[k * m:(k + 1) * m]

    def _exchange(self, out, inp):
        """The one collective of the transform (per chunk).  Returns a waitable or None.
        RCCL moves device buffers directly and asynchronously; a gloo group (rehearsals
        on a box with fewer GPUs than ranks) cannot, so there the chunk is bounced
        through host memory -- transport only, the transforms on either side still run
        in the HIP kernels."""
        if inp.is_cuda and self.dist.get_backend() == "gloo":
            h_in, h_out = inp.cpu(), out.cpu()
            self.dist.all_to_all_single(h_out, h_in)
            out.copy_(h_out)
            return None
        if inp.is_cuda:
            return self.dist.all_to_all_single(out, inp, async_op=True)
        self.dist.all_to_all_single(out, inp)
        return None

    def forward(self, dst, src, events=None):
        """natural-order column slab ``src`` -> bit-reversed row block ``dst``."""
        e, K = self.engine, self.chunks
        pending = []
        self._mark(events, 0)
        for k in range(K):
            # chunk k of the column pass, written compactly: row block h of the piece
            # is what rank h needs, so the piece is exchanged as is
            e.columns_chunk(False, self._piece(self._work, k), src, k, K)
            pending.append(self._exchange(self._piece(self._recv, k), self._piece(self._work, k)))
        self._mark(events, 1)
        for k in range(K):
            if pending[k] is not None:
                pending[k].wait()
            e.exchange_side_chunk(False, dst, self._piece(self._recv, k), k, K)
        self._mark(events, 2)
        for i in range(1, e.rows_passes):
            e.rows_pass(False, i, dst, dst)
            self._mark(events, 2 + i)
        return dst

    def inverse(self, dst, src, events=None):
        """bit-reversed row block ``src`` -> natural-order column slab ``dst``."""
        import torch
        e, K = self.engine, self.chunks
        n_in_place = e.rows_passes - 1
        rows, send = self._recv, self._work
        if self._back is None:
            self._back = torch.empty_like(send)
        cur = src
        self._mark(events, 0)
        for i in range(n_in_place):
            e.rows_pass(True, i, rows, cur)  # first one out of place, then in place on `rows`
            cur = rows
            self._mark(events, 1 + i)
        pending = []
        for k in range(K):
            # the last rows pass scatters chunk k into piece layout; its exchange starts at once
            e.exchange_side_chunk(True, self._piece(send, k), cur, k, K)
            pending.append(self._exchange(self._piece(self._back, k), self._piece(send, k)))
        self._mark(events, 1 + n_in_place)
        for k in range(K):
            if pending[k] is not None:
                pending[k].wait()
            e.columns_chunk(True, dst, self._piece(self._back, k), k, K)
        self._mark(events, 2 + n_in_place)
        return dst


    # ---- callers whose data is sharded the natural way -------------------------------------------
    # forward()/inverse() take the natural-order side as a COLUMN slab (what the six-step's first
    # phase needs).  A caller that holds the natural-order vector as contiguous chunks -- rank k has
    # elements [k*n/G, (k+1)*n/G), i.e. rows [k*R/G, (k+1)*R/G) of the R x C matrix -- pays one more
    # exchange ("if the caller's input is row-sharded a second all-to-all is needed up
    # front"): every rank cuts its rows into G column blocks, block h goes to rank h, and what arrives
    # IS the slab (piece s = rows [s*R/G, (s+1)*R/G) of it), no reordering on the receiving side.
    def slab_from_natural(self, chunk, out=None):
        """this rank's contiguous n/G elements of the natural-order vector -> its R x (C/G) column slab"""
        import torch
        G = self.world
        R = 1 << self.r_log2
        Rl, Cl = R // G, (self.n // R) // G
        packed = chunk.view(Rl, G, Cl).permute(1, 0, 2).contiguous().view(-1)   # [h][q][c']
        out = torch.empty_like(packed) if out is None else out
        w = self._exchange(out, packed)
        if w is not None:
            w.wait()
        return out

    def natural_from_slab(self, slab, out=None):
        """the inverse redistribution: column slab -> this rank's contiguous chunk of the natural-order vector"""
        import torch
        G = self.world
        R = 1 << self.r_log2
        Rl, Cl = R // G, (self.n // R) // G
        recv = torch.empty_like(slab)
        w = self._exchange(recv, slab.contiguous())      # piece h of the slab (rows of rank h) -> rank h
        if w is not None:
            w.wait()
        res = recv.view(G, Rl, Cl).permute(1, 0, 2).contiguous().view(-1)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def forward_natural(self, dst, src_chunk):
        """natural-order contiguous chunk -> bit-reversed contiguous chunk (two exchanges)"""
        return self.forward(dst, self.slab_from_natural(src_chunk))

    def inverse_natural(self, dst_chunk, src):
        """bit-reversed contiguous chunk -> natural-order contiguous chunk (two exchanges)"""
        import torch
        slab = torch.empty_like(src)
        self.inverse(slab, src)
        return self.natural_from_slab(slab, dst_chunk)


def batch_partition(batch: int, world: int, rank: int) -> tuple[int, int]:
    """(first transform, count) of rank ``rank`` when ``batch`` independent transforms are
    dealt over ``world`` ranks in contiguous, near-equal shares (the first ``batch % world``
    ranks take one more)."""
    if batch < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError("bad partition arguments")
    base, extra = divmod(batch, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class ReplicaNTT:
    """Many independent short transforms over the GPUs of a node (SURVEY.md 8e "small N:
    replicas only", BASELINE config #4): the batch is partitioned by ``batch_partition``,
    every rank transforms its own share with a local batched plan, nothing is exchanged.
    ``src``/``dst`` are the rank's ``count * n`` elements."""

    def __init__(self, modulus: Modulus, n: int, batch: int, dist=None, rank: int = 0, world: int = 1):
        from . import NTT
        if dist is not None:
            rank, world = dist.get_rank(), dist.get_world_size()
        self.first, self.count = batch_partition(batch, world, rank)
        self.n = n
        self._ntt = NTT(modulus, n, batch=self.count) if self.count else None

    def forward(self, dst, src=None):
        if self._ntt is not None:
            self._ntt.compute_forward(dst, src)
        return dst

    def inverse(self, dst, src=None):
        if self._ntt is not None:
            self._ntt.compute_inverse(dst, src)
        return dst
