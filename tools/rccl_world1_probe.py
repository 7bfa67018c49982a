#!/usr/bin/env python3
"""Probe (one GPU, world size 1): the RCCL calls the sharded driver makes --
all_to_all_single on int64 device tensors with async_op=True, wait() ordering against
kernels on the current stream -- exist and behave on this image.  A one-GPU box cannot
host two RCCL ranks, so this is as far as the real backend can be exercised here."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(1 << 20, dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
for _ in range(3):
    x.add_(1)                                   # producer on the current stream
    w = dist.all_to_all_single(y, x, async_op=True)
    w.wait()                                    # current stream now waits for the exchange
    z = y.clone()
torch.cuda.synchronize()
print("backend", dist.get_backend(), "ok" if torch.equal(z, x) else "MISMATCH")
t = torch.zeros(1, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("done")
