#!/usr/bin/env python3
"""Single-process RCCL probe (world_size = 1): the collectives and options the N > 1 path of bench.py uses must at least
be accepted by this ROCm build - init with device_id, all_reduce AVG / MAX on float32 / float64, barrier."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("backend", dist.get_backend())
t = torch.arange(6208, dtype=torch.float32, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.AVG)
u = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(u, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("AVG ok", float(t[5]), "MAX ok", float(u))
dist.destroy_process_group()
