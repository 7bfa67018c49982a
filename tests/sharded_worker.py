"""Worker for tests/test_sharded.py: runs the sharded six-step driver
(sve_ntt_amd/sharded.py) on every rank and checks it against the oracle.

    ENGINE=sim : gloo, CPU tensors, local passes replayed on the host (tests/cpu_sim)
    ENGINE=hip : nccl (RCCL), one GPU per rank, local passes in the HIP kernels

    CHECK=oracle (default): every rank builds the whole input and the oracle's whole output --
                 sizes the oracle finishes in seconds.
    CHECK=closed : the input is a[i] = s + i (the reference harness's recipe, tests/bench-ntt.cpp:
                 31-33); the forward result has the closed form X[0] = m*s + m(m-1)/2,
                 X[k] = m / (omega^k - 1) (tests/test-ntt-reference.cpp:45-63 of the reference check
                 k = 0, 1 the same way), checked on sampled outputs of every rank, and the inverse
                 must return the input.  No rank ever holds more than its own n/world elements:
                 this is how BASELINE config #5 (N = 2^30 on 8 GPUs) is verified without a 16 GiB
                 oracle run.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
import sve_ntt_amd as eng  # noqa: E402
from sve_ntt_amd.sharded import ShardedNTT  # noqa: E402


def main() -> None:
    engine_kind = os.environ.get("ENGINE", "sim")
    log2n = int(os.environ.get("LOG2N", "16"))
    r_log2 = int(os.environ.get("R_LOG2", "6"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if engine_kind == "hip":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local % ndev)
        # RCCL refuses two ranks on one device; on a one-GPU box the exchange goes
        # over gloo (host bounce) while every transform still runs in the HIP kernels
        dist.init_process_group("nccl" if ndev >= int(os.environ["WORLD_SIZE"]) else "gloo")
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        dist.init_process_group("gloo")
        device = torch.device("cpu")
    rank, world = dist.get_rank(), dist.get_world_size()
    P, G = oracle.BASELINE_P, oracle.BASELINE_G
    if os.environ.get("MODULUS") == "goldilocks":  # its own kernel family (field64.h: ARITH_GOLD)
        P, G = oracle.GOLDILOCKS_P, 7
    modulus = eng.Modulus(P, G)
    port = oracle.port()
    n = 1 << log2n
    R = 1 << r_log2
    C = n // R
    Cl, Rl = C // world, R // world
    closed = os.environ.get("CHECK", "oracle") == "closed"
    s0 = oracle.INPUT_I1_START
    if closed:
        # element (r, c) of the R x C input is s0 + r*C + c; this rank holds columns [rank*Cl, +Cl)
        rows_i = np.arange(R, dtype=np.uint64)[:, None] * np.uint64(C)
        cols_i = np.arange(rank * Cl, (rank + 1) * Cl, dtype=np.uint64)[None, :]
        slab = (np.uint64(s0) + rows_i + cols_i).reshape(-1)
        want = None
    else:
        full = port.fill_splitmix(n, 4242, P)          # every rank builds the same global input
        want = port.forward(full, P, G)
        slab = np.ascontiguousarray(full.reshape(R, C)[:, rank * Cl:(rank + 1) * Cl]).reshape(-1)

    engine = None
    if engine_kind == "sim":
        from tests.simlib import SimShardEngine
        engine = SimShardEngine(modulus, n, r_log2, rank, world)
    chunks = int(os.environ.get("CHUNKS", "4"))
    sh = ShardedNTT(modulus, n, dist, r_log2=r_log2, engine=engine, device=device,
                    chunks=chunks)

    src = torch.from_numpy(slab.view(np.int64).copy()).to(device)
    dst = torch.full_like(src, 0x5555555555555555)
    sh.forward(dst, src)
    got = dst.cpu().numpy().view(np.uint64)
    n_local = n // world
    if closed:
        # this rank's outputs are positions [rank*n_local, +n_local) of the bit-reversed result
        w = pow(G, (P - 1) // n, P)
        rng = np.random.default_rng(1000 + rank)
        where = np.unique(np.concatenate([np.arange(min(64, n_local)), n_local - 1 - np.arange(min(64, n_local)),
                                          rng.integers(0, n_local, size=int(os.environ.get("SAMPLES", "1024")))]))
        ok_f = True
        for j in where.tolist():
            k = int(format(rank * n_local + j, f"0{log2n}b")[::-1], 2)  # position -> frequency
            exp = (n * s0 + n * (n - 1) // 2) % P if k == 0 else n * pow(pow(w, k, P) - 1, -1, P) % P
            if int(got[j]) != exp:
                ok_f = False
                break
    else:
        mine = want[rank * n_local:(rank + 1) * n_local]   # rows [rank*Rl, (rank+1)*Rl)
        ok_f = bool(np.array_equal(got, mine))
    assert np.array_equal(src.cpu().numpy().view(np.uint64), slab), "forward modified its source"

    back = torch.full_like(src, 0x5555555555555555)
    sh.inverse(back, dst)
    ok_i = bool(np.array_equal(back.cpu().numpy().view(np.uint64), slab))

    flags = torch.tensor([int(ok_f), int(ok_i)], dtype=torch.int64, device=device)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"SHARDED world={world} n=2^{log2n} R=2^{r_log2} Rl={Rl} Cl={Cl} chunks={sh.chunks} "
              f"check={'closed-form' if closed else 'oracle'} "
              f"forward={'OK' if flags[0].item() else 'MISMATCH'} "
              f"inverse={'OK' if flags[1].item() else 'MISMATCH'}", flush=True)
    dist.destroy_process_group()
    if not (flags[0].item() and flags[1].item()):
        sys.exit(1)


if __name__ == "__main__":
    main()
