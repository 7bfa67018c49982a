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


@pytest.mark.parametrize("world,log2n,r_log2,chunks", [(8, 20, 11, 4), (2, 14, 1, 2), (4, 12, 2, 1)])
def test_edge_sharded_shapes_in_process(world, log2n, r_log2, chunks):
    run_case(world, log2n, r_log2, chunks, seed=99)
