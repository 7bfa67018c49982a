#!/usr/bin/env python3
"""bench.py -- forward-NTT throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: one step = one out-of-place forward NTT of 2^24 uint64 elements,
p = 0xfffffc6e80000001 (BASELINE configs[2], the configuration the metric is
quoted on), inputs resident in HBM, as the reference's harness does it
(tests/bench-ntt.cpp:47-56: timed compute_forward(dst, src)).
N > 1 (launched by torch.distributed.run, one rank per GPU): one step = one
forward NTT of 2^24 * N elements sharded over the N GPUs (six-step: local
column pass, RCCL all-to-all, local row passes); weak scaling.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on
the launch stream around every pass of the timed region; `cpu_baseline` times
the reference's scalar path (oracle/_ref if built, else the C port) on a host
core in the same run (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P = 0xFFFFFC6E80000001
G = 3
LOG2N = 24
HBM_PEAK = 8.0e12  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_ELEMENT = 16  # 8 B compulsory read + 8 B compulsory write


def splitmix_fill(n: int, seed: int) -> np.ndarray:
    """Input I2 of SURVEY.md 8(d) without the oracle: uniform residues < P."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, P, size=n, dtype=np.uint64)


def cpu_baseline(budget_s: float = 10.0) -> dict:
    """Reference scalar path (tests/ntt-reference.hpp:43-61) on ONE host core."""
    import oracle  # checker only: timed here as the reported CPU baseline
    try:
        impl = oracle.reference()
    except (FileNotFoundError, OSError):
        impl = oracle.port()
    n = 1 << LOG2N
    src = oracle.port().fill_iota(n, oracle.INPUT_I1_START)
    reps, t_total = 0, 0.0
    while reps < 5 and (reps == 0 or t_total < budget_s):
        t0 = time.perf_counter()
        impl.forward(src, P, G)
        t_total += time.perf_counter() - t0
        reps += 1
    return {
        "value": n * reps / t_total,
        "unit": "elements/s",
        "cores": 1,
        "kind": impl.kind,
        "sample": f"{reps} x forward NTT N=2^{LOG2N}, p=0xfffffc6e80000001, input start+i, "
                  f"{t_total:.1f} s on 1 of {os.cpu_count()} host cores",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="untimed clock-ramp steps before the warmup (default 1500, sharded 300)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU path)")
    ndev = torch.cuda.device_count()
    # One rank per GPU.  SVENTT_BENCH_REHEARSE=1 lets more ranks than GPUs share a card
    # over gloo (RCCL refuses that) to rehearse this script on a one-GPU box; such a
    # run is marked in `config` and is not a scaling measurement.
    rehearsal = world > ndev and os.environ.get("SVENTT_BENCH_REHEARSE") == "1"
    if world > ndev and not rehearsal:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPUs (one rank per GPU)")
    torch.cuda.set_device(local_rank % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    import sve_ntt_amd as eng

    n_local = 1 << LOG2N
    src = torch.from_numpy(splitmix_fill(n_local, 42 + rank).view(np.int64)).cuda()
    dst = torch.full_like(src, 0x5555555555555555)

    if world == 1:
        ntt = eng.NTT(eng.BASELINE_MODULUS, n_local, enable_inverse=False)
        npass = ntt.num_passes()
        desc = ntt.describe()

        def step(events=None):
            for i in range(npass):
                if events is not None:
                    events[i].record()
                ntt.run_pass(False, i, dst, src if i == 0 else None)
            if events is not None:
                events[npass].record()
        n_total = n_local
        parallelism = "1 GPU: " + desc
    else:
        from sve_ntt_amd.sharded import ShardedNTT
        sh = ShardedNTT(eng.BASELINE_MODULUS, n_local * world, dist)
        npass = sh.num_local_phases
        desc = " | ".join(sh.phase_names)

        def step(events=None):
            sh.forward(dst, src, events)
        n_total = n_local * world
        parallelism = f"{world} GPUs: {sh.describe()} (exchange pipelined in {sh.chunks} chunks)"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The GPU ramps its clocks over the first tens of milliseconds of load (from idle the
    # first ~100 transforms run at ~300 us instead of ~230 us, tools/clock_ramp.py): bring
    # it to its steady state with a fixed number of untimed steps (the same count on every
    # rank -- a step contains collectives when sharded) before the W warmup steps.
    for _ in range(args.prewarm if args.prewarm >= 0 else (1500 if world == 1 else 300)):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # per-pass HIP events on every 4th step of the timed region (an event pair around every
    # launch costs 3-7 % of the step, tools/event_overhead.py; a quarter of them ~1 %)
    sampled = [k for k in range(args.steps) if k % 4 == 0]
    events = {k: [torch.cuda.Event(enable_timing=True) for _ in range(npass + 1)] for k in sampled}
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events.get(k))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-phase device time from the HIP events of the timed region
    phase_ms = [float(np.mean([events[k][i].elapsed_time(events[k][i + 1])
                               for k in sampled])) for i in range(npass)]
    names = desc.split(" | ")
    if len(names) != npass:
        names = [f"phase {i}" for i in range(npass)]
    kernels = [i for i in range(npass) if "all-to-all wait" not in names[i]] or list(range(npass))
    dom = max(kernels, key=lambda i: phase_ms[i])  # dominant KERNEL (the exchange is not one)
    dom_bytes = ALGO_BYTES_PER_ELEMENT * n_local  # each pass reads+writes every local element once
    achieved = dom_bytes / (phase_ms[dom] * 1e-3)
    device_ms = float(sum(phase_ms))
    # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/
    # (tools/pmc_run.sh; FETCH_SIZE corrected per MI355X_MICROARCH.md), keyed by kernel name
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            traffic = json.load(f).get("bytes_per_launch", {}).get(names[dom])
    except (OSError, ValueError):
        pass

    out = {
        "metric": "forward-NTT uint64 elements/s at N=2^24; achieved HBM GB/s vs peak",
        "value": n_total * args.steps / elapsed,
        "unit": "elements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"forward NTT, N=2^{LOG2N}{'' if world == 1 else ' per GPU, one sharded transform of 2^%d' % (LOG2N + int(np.log2(world)))}, "
                        "p=0xfffffc6e80000001, g=3, out-of-place, bit-reversed output",
            "plan": parallelism + (" [REHEARSAL: ranks share one GPU, gloo]" if rehearsal else ""),
            "elements_per_step": n_total,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": names[dom],
            "achieved": achieved / 1e9,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": dom_bytes,
            "kernel_ms": phase_ms[dom],
            "all_phases_ms": [[names[i], phase_ms[i]] for i in range(npass)],
            "event_sampled_steps": len(sampled),
            "transform_device_ms": device_ms,
            "transform_frac": (ALGO_BYTES_PER_ELEMENT * n_local / (device_ms * 1e-3)) / HBM_PEAK,
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
