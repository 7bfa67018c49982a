#!/usr/bin/env python3
"""bench.py -- forward-NTT throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|cfg4|roundtrip]
                    [--log2n-total T]

Default (the driver's run), N = 1: one step = one out-of-place forward NTT of 2^24 uint64
elements, p = 0xfffffc6e80000001 (BASELINE configs[2], the configuration the metric is quoted on),
inputs resident in HBM, as the reference's harness does it (tests/bench-ntt.cpp:47-56: timed
compute_forward(dst, src)).
N > 1 (launched by torch.distributed.run, one rank per GPU): one step = one forward NTT of
2^24 * N elements (or 2^T with --log2n-total T, e.g. 30 for BASELINE configs[4]) sharded over the
N GPUs (six-step: local column pass, RCCL all-to-all, local row passes); weak scaling.

--config selects the other BASELINE configurations on one GPU (same JSON schema):
  cfg2       N = 2^17 as 2^8 x 2^9 (README.md:30-32 of the reference), latency bound
  cfg4       2^16 independent N = 2^12 transforms, in place; cpu_baseline on every host core
  roundtrip  N = 2^24 forward then inverse (32 algorithmic bytes per element)

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the launch stream
around every pass of sampled steps of the timed region: `frac` is the fraction of the HBM peak the
WHOLE transform reaches on its algorithmic bytes (16 B per element per transform, SURVEY.md 8d --
the quantity the north-star target is stated on); the per-launch figure of the slowest kernel is
`dominant_kernel_frac`.  `cpu_baseline` times the reference's scalar path (oracle/_ref if built,
else the C port) on the host in the same run (N = 1 only).
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

CONFIGS = {
    # name: (log2 n, batch, n0_log2, in_place, inverse too, prewarm steps)
    "cfg3": (24, 1, 0, False, False, 1500),
    "cfg2": (17, 1, 8, False, False, 20000),
    "cfg4": (12, 1 << 16, 0, True, False, 200),
    "roundtrip": (24, 1, 0, False, True, 800),
}


def uniform_residues(n: int, seed: int) -> np.ndarray:
    """Input I2 of SURVEY.md 8(d) without the oracle: uniform residues < P (numpy's PCG64)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, P, size=n, dtype=np.uint64)


def _reference_impl():
    import oracle  # checker only: timed here as the reported CPU baseline
    try:
        return oracle, oracle.reference()
    except (FileNotFoundError, OSError):
        return oracle, oracle.port()


def cpu_baseline(log2n: int = LOG2N, batch: int = 1, budget_s: float = 10.0) -> dict:
    """Reference scalar path (tests/ntt-reference.hpp:43-61).  One transform is serial, so a single
    transform is timed on ONE host core; independent transforms (cfg4) are spread over all cores
    (BASELINE.md section 3)."""
    oracle, impl = _reference_impl()
    n = 1 << log2n
    if batch == 1:
        src = oracle.port().fill_iota(n, oracle.INPUT_I1_START)
        reps, t_total = 0, 0.0
        while reps < 5 and (reps == 0 or t_total < budget_s):
            t0 = time.perf_counter()
            impl.forward(src, P, G)
            t_total += time.perf_counter() - t0
            reps += 1
        return {"value": n * reps / t_total, "unit": "elements/s", "cores": 1, "kind": impl.kind,
                "sample": f"{reps} x forward NTT N=2^{log2n}, p=0xfffffc6e80000001, input start+i, "
                          f"{t_total:.1f} s on 1 of {os.cpu_count()} host cores"}
    # independent transforms: a thread per core (the C library releases the GIL), bounded sample
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())
    src = oracle.port().fill_splitmix(n, 7, P)
    t1 = time.perf_counter()
    impl.forward(src, P, G)
    per = max(time.perf_counter() - t1, 1e-6)
    each = max(8, min(batch // cores + 1, int(budget_s / per)))

    def work(_):
        for _i in range(each):
            impl.forward(src, P, G)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    t_total = time.perf_counter() - t0
    return {"value": n * each * cores / t_total, "unit": "elements/s", "cores": cores, "kind": impl.kind,
            "sample": f"{each * cores} of the {batch} independent forward NTTs N=2^{log2n} "
                      f"({each} per thread, {cores} threads), {t_total:.1f} s"}


def _load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 200; cfg2 2000, cfg4 30)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3")
    ap.add_argument("--log2n-total", type=int, default=0,
                    help="sharded runs: total transform length 2^T (default 2^24 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="untimed clock-ramp steps before the warmup (default per config, sharded 300)")
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
        if args.config != "cfg3":
            raise SystemExit("bench.py: --config other than cfg3 runs on one GPU")
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    import sve_ntt_amd as eng

    log2n, batch, n0_log2, in_place, with_inverse, prewarm_default = CONFIGS[args.config]
    steps = args.steps or {"cfg2": 2000, "cfg4": 30}.get(args.config, 200)
    if world > 1 and args.log2n_total:
        log2_world = int(np.log2(world))
        if (1 << log2_world) != world or args.log2n_total - log2_world < 16:
            raise SystemExit("bench.py: --log2n-total needs a power-of-two rank count and >= 2^16 per rank")
        n_local = 1 << (args.log2n_total - log2_world)
    else:
        n_local = (1 << log2n) * batch
    src = torch.from_numpy(uniform_residues(n_local, 42 + rank).view(np.int64)).cuda()
    dst = src if in_place else torch.full_like(src, 0x5555555555555555)

    if world == 1:
        ntt = eng.NTT(eng.BASELINE_MODULUS, 1 << log2n, n0_log2=n0_log2, batch=batch,
                      enable_inverse=with_inverse, device_pointers=True)
        nf = ntt.num_passes(False)
        ni = ntt.num_passes(True) if with_inverse else 0
        npass = nf + ni
        names = ntt.describe().split(" | ")
        if with_inverse:
            names = [f"forward {x}" for x in names] + [f"inverse {x}" for x in reversed(names)]
        back = torch.empty_like(src) if with_inverse else None

        def step(events=None):
            if events is None and not with_inverse:
                # one call per transform, as a user makes it (the per-pass entry point costs two
                # Python->C round trips, visible on the 10 us transform of cfg2)
                ntt.compute_forward(dst, None if in_place else src)
                return
            for i in range(nf):
                if events is not None:
                    events[i].record()
                ntt.run_pass(False, i, dst, src if (i == 0 and not in_place) else None)
            for i in range(ni):
                if events is not None:
                    events[nf + i].record()
                ntt.run_pass(True, i, back, dst if i == 0 else None)
            if events is not None:
                events[npass].record()
        n_total = n_local
        transforms_per_step = 2 if with_inverse else 1
        parallelism = "1 GPU: " + ntt.describe()
    else:
        from sve_ntt_amd.sharded import ShardedNTT
        sh = ShardedNTT(eng.BASELINE_MODULUS, n_local * world, dist)
        npass = sh.num_local_phases
        names = list(sh.phase_names)

        def step(events=None):
            sh.forward(dst, src, events)
        n_total = n_local * world
        transforms_per_step = 1
        parallelism = f"{world} GPUs: {sh.describe()} (exchange pipelined in {sh.chunks} chunks)"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The GPU ramps its clocks over the first tens of milliseconds of load (from idle the
    # first ~100 transforms run ~30 % slower, tools/clock_ramp.py): bring it to its steady
    # state with a fixed number of untimed steps (the same count on every rank -- a step
    # contains collectives when sharded) before the W warmup steps.
    for _ in range(args.prewarm if args.prewarm >= 0 else (prewarm_default if world == 1 else 300)):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # per-pass HIP events on every 4th step of the timed region (an event pair around every
    # launch costs 3-7 % of the step, tools/event_overhead.py; a quarter of them ~1 %)
    stride = 64 if args.config == "cfg2" else 4  # (a 10 us transform would mostly measure the events)
    sampled = [k for k in range(steps) if k % stride == 0]
    events = {k: [torch.cuda.Event(enable_timing=True) for _ in range(npass + 1)] for k in sampled}
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
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
    if len(names) != npass:
        names = [f"phase {i}" for i in range(npass)]
    kernels = [i for i in range(npass) if "all-to-all wait" not in names[i]] or list(range(npass))
    dom = max(kernels, key=lambda i: phase_ms[i])  # dominant KERNEL (the exchange is not one)
    launch_bytes = ALGO_BYTES_PER_ELEMENT * n_local  # each pass reads+writes every local element once
    dom_achieved = launch_bytes / (phase_ms[dom] * 1e-3)
    device_ms = float(sum(phase_ms))
    transform_bytes = ALGO_BYTES_PER_ELEMENT * n_local * transforms_per_step
    achieved = transform_bytes / (device_ms * 1e-3)
    # HBM bytes per launch and VALU issue of the kernels, from the rocprofv3 PMC passes committed
    # under profiles/ (tools/pmc_run.sh; FETCH_SIZE corrected per MI355X_MICROARCH.md)
    plain = names[dom].replace("forward ", "").replace("inverse ", "")
    traffic = _load_json("traffic.json").get("bytes_per_launch", {}).get(plain)
    valu = _load_json("valu.json").get("kernels", {})

    out = {
        "metric": "forward-NTT uint64 elements/s at N=2^24; achieved HBM GB/s vs peak",
        "value": n_total * transforms_per_step * steps / elapsed,
        "unit": "elements/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": {
                "cfg3": f"forward NTT, N=2^{LOG2N}" + ("" if world == 1 else
                        " per GPU, one sharded transform of 2^%d" % int(np.log2(n_total))),
                "cfg2": "forward NTT, N=2^17 as 2^8 x 2^9 (README configuration)",
                "cfg4": "2^16 independent forward NTTs of N=2^12, in place",
                "roundtrip": "forward then inverse NTT, N=2^24 (elements/s counts both transforms)",
            }[args.config] + ", p=0xfffffc6e80000001, g=3, bit-reversed output",
            "name": args.config,
            "plan": parallelism + (" [REHEARSAL: ranks share one GPU, gloo]" if rehearsal else ""),
            "elements_per_step": n_total,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "whole transform: " + " + ".join(names[i] for i in kernels),
            "achieved": achieved / 1e9,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK,
            "traffic": traffic,
            "traffic_kernel": names[dom],
            "algorithmic_bytes_per_transform": ALGO_BYTES_PER_ELEMENT * n_local,
            "algorithmic_bytes_per_launch": launch_bytes,
            "transform_device_ms": device_ms,
            "dominant_kernel": names[dom],
            "dominant_kernel_ms": phase_ms[dom],
            "dominant_kernel_achieved": dom_achieved / 1e9,
            "dominant_kernel_frac": dom_achieved / HBM_PEAK,
            "all_phases_ms": [[names[i], phase_ms[i]] for i in range(npass)],
            "event_sampled_steps": len(sampled),
        },
    }
    if valu:
        # what actually bounds the kernels: VALU issue, from SQ_INSTS_VALU
        # and GRBM_GUI_ACTIVE of the committed PMC passes
        out["valu"] = {
            "source": _load_json("valu.json").get("source"),
            "kernels": {k: v for k, v in valu.items() if any(k in nm for nm in names)} or valu,
        }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(log2n, batch)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
