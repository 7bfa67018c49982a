"""Worker for tests/test_sharded.py: runs the sharded six-step driver
(sve_ntt_amd/sharded.py) on every rank and checks it against the oracle.

    ENGINE=sim : gloo, CPU tensors, local passes replayed on the host (tests/cpu_sim)
    ENGINE=hip : nccl (RCCL), one GPU per rank, local passes in the HIP kernels
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
    port = oracle.port()
    n = 1 << log2n
    R = 1 << r_log2
    C = n // R
    Cl, Rl = C // world, R // world
    full = port.fill_splitmix(n, 4242, P)          # every rank builds the same global input
    want = port.forward(full, P, G)
    slab = np.ascontiguousarray(full.reshape(R, C)[:, rank * Cl:(rank + 1) * Cl]).reshape(-1)

    engine = None
    if engine_kind == "sim":
        from tests.simlib import SimShardEngine
        engine = SimShardEngine(eng.BASELINE_MODULUS, n, r_log2, rank, world)
    chunks = int(os.environ.get("CHUNKS", "4"))
    sh = ShardedNTT(eng.BASELINE_MODULUS, n, dist, r_log2=r_log2, engine=engine, device=device,
                    chunks=chunks)

    src = torch.from_numpy(slab.view(np.int64).copy()).to(device)
    dst = torch.full_like(src, 0x5555555555555555)
    sh.forward(dst, src)
    got = dst.cpu().numpy().view(np.uint64)
    mine = want[rank * (n // world):(rank + 1) * (n // world)]   # rows [rank*Rl, (rank+1)*Rl)
    ok_f = bool(np.array_equal(got, mine))
    assert np.array_equal(src.cpu().numpy().view(np.uint64), slab), "forward modified its source"

    back = torch.full_like(src, 0x5555555555555555)
    sh.inverse(back, dst)
    ok_i = bool(np.array_equal(back.cpu().numpy().view(np.uint64), slab))

    flags = torch.tensor([int(ok_f), int(ok_i)], dtype=torch.int64, device=device)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"SHARDED world={world} n=2^{log2n} R=2^{r_log2} Rl={Rl} Cl={Cl} chunks={sh.chunks} "
              f"forward={'OK' if flags[0].item() else 'MISMATCH'} "
              f"inverse={'OK' if flags[1].item() else 'MISMATCH'}", flush=True)
    dist.destroy_process_group()
    if not (flags[0].item() and flags[1].item()):
        sys.exit(1)


if __name__ == "__main__":
    main()
