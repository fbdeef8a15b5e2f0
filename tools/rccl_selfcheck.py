#!/usr/bin/env python3
"""One-rank RCCL group on the GPU at hand: the collectives bench.py --gpus N issues (all_reduce MIN of int64, reduce SUM
of float64, all_reduce MAX of float64, barrier) with the dtypes it issues them in.  A box with one GPU cannot run two
RCCL ranks, but it can show that the backend takes these calls.
usage (on a GPU box): python tools/rccl_selfcheck.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([12345], dtype=torch.int64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MIN)
s = torch.arange(3 * 1920 * 1080, dtype=torch.float64, device="cuda:0")
ref = s.clone()
dist.reduce(s, dst=0, op=dist.ReduceOp.SUM)
e = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(e, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert int(t.item()) == 12345 and bool((s == ref).all()) and float(e.item()) == 1.5
dist.destroy_process_group()
print("rccl one-rank group: all_reduce(MIN, int64), reduce(SUM, float64), all_reduce(MAX, float64), barrier ok")
