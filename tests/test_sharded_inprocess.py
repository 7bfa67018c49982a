"""Many sharded shapes without spawning processes: the ranks of sve_ntt_amd.sharded.ShardedNTT
run as threads of this process, the collective is an in-memory all-to-all with the semantics of
torch.distributed.all_to_all_single (equal splits), the local passes are the host replay of the
tile code.  Complements tests/test_sharded.py (real gloo processes, fewer shapes): here the
world size, the transform length, the row count R and the number of pipeline chunks are drawn
at random."""
import threading

import numpy as np
import pytest
import torch

import oracle
import sve_ntt_amd as eng
from sve_ntt_amd.sharded import ShardedNTT
from tests.simlib import SimShardEngine, SimError

P, G = oracle.BASELINE_P, oracle.BASELINE_G


class ThreadDist:
    """The slice of torch.distributed that ShardedNTT uses, for `world` threads."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    def bind(self, rank):
        self.local.rank = rank

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.local.rank

    def get_backend(self):
        return "threads"

    def all_to_all_single(self, out, inp, async_op=False):
        r, w = self.local.rank, self.world
        self.slots[r] = inp
        self.barrier.wait()
        m = inp.numel() // w
        for src in range(w):  # piece r of every rank's input, in rank order
            out[src * m:(src + 1) * m] = self.slots[src][r * m:(r + 1) * m]
        self.barrier.wait()
        return None


def run_case(world, log2n, r_log2, chunks, seed):
    port = oracle.port()
    n, R = 1 << log2n, 1 << r_log2
    C = n // R
    Cl = C // world
    full = port.fill_splitmix(n, seed, P)
    want = port.forward(full, P, G)
    dist = ThreadDist(world)
    results, errors = {}, []

    def rank_main(rank):
        try:
            dist.bind(rank)
            engine = SimShardEngine(eng.BASELINE_MODULUS, n, r_log2, rank, world)
            sh = ShardedNTT(eng.BASELINE_MODULUS, n, dist, r_log2=r_log2, engine=engine, device="cpu",
                            chunks=chunks)
            slab = np.ascontiguousarray(full.reshape(R, C)[:, rank * Cl:(rank + 1) * Cl]).reshape(-1)
            src = torch.from_numpy(slab.view(np.int64).copy())
            dst = torch.full_like(src, 0x5555555555555555)
            sh.forward(dst, src)
            back = torch.full_like(src, 0x5555555555555555)
            sh.inverse(back, dst)
            results[rank] = (dst.numpy().view(np.uint64), back.numpy().view(np.uint64), slab, sh.chunks)
        except Exception as exc:  # noqa: BLE001 - re-raised by the caller
            errors.append(exc)
            dist.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    share = n // world
    for rank in range(world):
        fwd, back, slab, _ = results[rank]
        assert np.array_equal(fwd, want[rank * share:(rank + 1) * share]), ("forward", rank)
        assert np.array_equal(back, slab), ("inverse", rank)
    return results[0][3]


@pytest.mark.parametrize("world,log2n,r_log2,chunks", [(2, 14, 3, 2), (4, 16, 6, 1), (8, 18, 6, 4)])
def test_naturally_sharded_callers(world, log2n, r_log2, chunks):
    """forward_natural / inverse_natural: every rank holds a CONTIGUOUS chunk of the natural-order vector
    (a row-sharded caller needs one more exchange up front) and gets its contiguous chunk
    of the bit-reversed result; the inverse returns the input chunk."""
    port = oracle.port()
    n = 1 << log2n
    full = port.fill_splitmix(n, 77, P)
    want = port.forward(full, P, G)
    dist = ThreadDist(world)
    share = n // world
    results, errors = {}, []

    def rank_main(rank):
        try:
            dist.bind(rank)
            engine = SimShardEngine(eng.BASELINE_MODULUS, n, r_log2, rank, world)
            sh = ShardedNTT(eng.BASELINE_MODULUS, n, dist, r_log2=r_log2, engine=engine, device="cpu", chunks=chunks)
            mine = torch.from_numpy(full[rank * share:(rank + 1) * share].view(np.int64).copy())
            out = torch.full_like(mine, 0x5555555555555555)
            sh.forward_natural(out, mine)
            back = torch.full_like(mine, 0x5555555555555555)
            sh.inverse_natural(back, out)
            results[rank] = (out.numpy().view(np.uint64), back.numpy().view(np.uint64))
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)
            dist.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    for rank in range(world):
        fwd, back = results[rank]
        assert np.array_equal(fwd, want[rank * share:(rank + 1) * share]), ("forward", rank)
        assert np.array_equal(back, full[rank * share:(rank + 1) * share]), ("inverse", rank)


def test_random_sharded_shapes_in_process():
    rng = np.random.default_rng(2024)
    done = 0
    seen = set()
    for attempt in range(200):
        world = int(rng.choice([2, 4, 8]))
        logw = world.bit_length() - 1
        log2n = int(rng.integers(2 * logw + 4, 19))
        r_log2 = int(rng.integers(max(logw, 1), min(11, log2n - logw - 3) + 1))
        chunks = int(rng.choice([1, 2, 3, 4, 8]))
        key = (world, log2n, r_log2, chunks)
        if key in seen:
            continue
        seen.add(key)
        try:
            used = run_case(world, log2n, r_log2, chunks, seed=attempt)
        except (SimError, ValueError) as exc:
            # shapes the planner declines (too few columns for a tile, ...) are fine -- but only those
            assert "column" in str(exc) or "r_log2" in str(exc) or "tile" in str(exc), (key, exc)
            continue
        assert 1 <= used <= max(1, chunks)
        done += 1
        if done >= 40:
            break
    assert done >= 25, done


@pytest.mark.parametrize("world,log2n,r_log2,chunks", [(8, 20, 11, 4), (2, 14, 1, 2), (4, 12, 2, 1),
                                                        (2, 20, 12, 2)])  # R = 2^12: the slim tiles' longest column
def test_edge_sharded_shapes_in_process(world, log2n, r_log2, chunks):
    run_case(world, log2n, r_log2, chunks, seed=99)


@pytest.mark.parametrize("world,log2n,r_log2,chunks", [
    (8, 22, 3, 4),   # rows of C = 2^19 on 8 ranks: col 2^6 (two-level) | row 2^13 -- BASELINE config #5's row phase
    (4, 21, 4, 2),   # C = 2^17 on 4 ranks: col 2^4 (two-level, 4 rows per piece) | row 2^13
    (2, 18, 3, 4),   # C = 2^15 on 2 ranks: col 2^3 (two-level) | row 2^12
    (8, 22, 3, 1)])  # the same as the first without pipelining
def test_fused_gather_shapes_in_process(world, log2n, r_log2, chunks):
    """The first pass of the row phase is longer than the rank count: it reads `rows per piece`
    rows from each received piece (two-level strides) and the column phase cuts its exchange
    chunks run by run (plan_core.h: sharded_row_split, make_chunk_args)."""
    from tests import simlib
    L = simlib.load()
    args = (P, G, 1 << log2n, r_log2, 0, world)
    assert L.sim_sharded_rows_num_passes(*args) == 2
    used = run_case(world, log2n, r_log2, chunks, seed=7)
    assert used == chunks


def test_config5_row_phase_is_two_sweeps():
    """N = 2^30 on 8 ranks, R = 2^11 (BASELINE configs[4]): one column sweep, the exchange, then
    col 2^6 | row 2^13 -- three sweeps of the rank's 1 GiB, not four."""
    from tests import simlib
    L = simlib.load()
    for rank in (0, 7):
        assert L.sim_sharded_rows_num_passes(P, G, 1 << 30, 11, rank, 8) == 2
    assert L.sim_sharded_rows_num_passes(P, G, 1 << 30, 12, 0, 8) == 2      # R = 2^12 is allowed now
    assert L.sim_sharded_rows_num_passes(P, G, 1 << 27, 11, 0, 8) == 2      # 2^24 per rank: col 2^3 | row 2^13
    # rows of 2^19 run as col 2^7 | row 2^12 out of the cache: the column phase cuts its chunks per run of 2^12 columns
    assert L.sim_sharded_tiles_per_block(P, G, 1 << 30, 11, 0, 8, 0) == 1024
    assert L.sim_sharded_tiles_per_block(P, G, 1 << 30, 11, 0, 8, 1) == 128  # col 2^7 x T32 over 2^12 columns


@pytest.mark.parametrize("log2n,r_log2,chunks", [(18, 4, 4), (18, 2, 1), (20, 5, 2)])
def test_one_rank_sharded_pipeline_on_the_host_replay(log2n, r_log2, chunks):
    """A single rank runs the same pipeline (tests/cpp/rccl_one_rank.cpp executes it over a one-rank
    RCCL communicator on the GPU): column pass chunk -> exchange with itself -> first row-phase pass
    on the chunk -> remaining row passes; inverse mirrored."""
    from tests import simlib
    L = simlib.load()
    port = oracle.port()
    n = 1 << log2n
    args = (P, G, n, r_log2, 0, 1)
    full = port.fill_splitmix(n, 31, P)
    want = port.forward(full, P, G)
    npass = L.sim_sharded_rows_num_passes(*args)
    assert npass >= 2
    for which in (0, 1):
        assert L.sim_sharded_tiles_per_block(*args, which) % chunks == 0
    piece = n // chunks
    work = np.zeros(n, dtype=np.uint64)
    dst = np.full(n, 0x5555555555555555, dtype=np.uint64)
    for k in range(chunks):
        wk = work[k * piece:(k + 1) * piece]
        assert L.sim_sharded_chunk(*args, 0, 0, k, chunks, 1, 0, simlib._ptr(wk), simlib._ptr(full)) == 0
    recv = work.copy()  # the exchange of one rank with itself
    for k in range(chunks):
        rk = recv[k * piece:(k + 1) * piece]
        assert L.sim_sharded_chunk(*args, 1, 0, k, chunks, 0, 1, simlib._ptr(dst), simlib._ptr(rk)) == 0
    for i in range(1, npass):
        assert L.sim_sharded_rows_pass(*args, 0, i, simlib._ptr(dst), simlib._ptr(dst)) == 0
    assert np.array_equal(dst, want)
    # inverse: row passes but the last, the last one scatters per chunk, exchange, column pass per chunk
    cur = dst.copy()
    for i in range(npass - 1):
        assert L.sim_sharded_rows_pass(*args, 1, i, simlib._ptr(cur), simlib._ptr(cur)) == 0
    for k in range(chunks):
        wk = work[k * piece:(k + 1) * piece]
        assert L.sim_sharded_chunk(*args, 1, 1, k, chunks, 1, 0, simlib._ptr(wk), simlib._ptr(cur)) == 0
    recv = work.copy()
    back = np.full(n, 0x5555555555555555, dtype=np.uint64)
    for k in range(chunks):
        rk = recv[k * piece:(k + 1) * piece]
        assert L.sim_sharded_chunk(*args, 0, 1, k, chunks, 0, 1, simlib._ptr(back), simlib._ptr(rk)) == 0
    assert np.array_equal(back, full)
