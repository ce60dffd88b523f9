"""Batch-sharded data parallelism for the hot path: one process per GPU, RCCL over xGMI.

The path shards over the batch (every sphere sample is convolved independently; operator and
weights are replicated), so forward and backward need no collective.  The only exchange is ONE
all-reduce of the parameter gradients per step.  The payload is tiny (24.8 KB for a 32->64 K=3
layer, 7.08 MB for the whole UNetSpherical), i.e. latency-bound on the fully connected 7-link xGMI
topology: everything goes in a single flat bucket so RCCL runs one collective, not one per tensor.
The reference itself has no multi-GPU path.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / MASTER_*).

    Returns ``(rank, world_size, local_rank)``; a no-op single-process world if WORLD_SIZE is unset.
    backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("DSW_FORCE_GRAD_SYNC") == "1"   # probe: run the collective path in a 1-rank world
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:   # DSW_DIST_BACKEND=gloo: run the N>1 path of a GPU script on a single-GPU box (tests)
            backend = os.environ.get("DSW_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous [start, stop) slice of the batch owned by ``rank`` (ragged tails allowed)."""
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_module_state(module, src: int = 0, group=None):
    """Make every rank's parameters AND buffers bit-identical to rank ``src``'s.

    Replicas must convolve with the same operator: ``prepare_torch_laplacian`` rescales by an ARPACK estimate of
    lambda_max that differs by ~1e-3 between calls (random start vector), so Laplacians built independently on each
    rank are NOT the same matrix.  Sparse buffers are broadcast as their (indices, values) pair; everything travels in
    ONE flat bucket per dtype (a handful of collectives, not one per tensor).  No-op in a single-process world."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return module
    dense = [p.data for p in module.parameters()]
    for name, buf in module.named_buffers():
        if buf is None:
            continue
        if buf.is_sparse:
            if not buf.is_coalesced():
                raise ValueError(f"sparse buffer '{name}' must be coalesced before it can be broadcast")
            dense += [buf._indices(), buf._values()]
        else:
            dense.append(buf.data)
    by_dtype = {}
    for t in dense:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), tensors in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
    from . import functional

    functional.invalidate_operator_caches()   # derived CSR / plans of the pre-broadcast operators are stale
    return module


class FlatGradAllReduce:
    """Average the gradients of ``params`` across ranks with a single flat-bucket all-reduce."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.bucket = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.bucket[off : off + p.numel()].view_as(p))
            off += p.numel()

    def __call__(self):
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1
                                         and os.environ.get("DSW_FORCE_GRAD_SYNC") != "1"):
            return
        have = [p.grad is not None for p in self.params]
        if all(have):
            torch._foreach_copy_(self.views, [p.grad for p in self.params])   # one launch for the whole bucket
        else:
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
        if dist.get_backend(self.group) == "nccl":     # RCCL averages in the collective: no separate scale kernel
            dist.all_reduce(self.bucket, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
            self.bucket.div_(dist.get_world_size(self.group))
        if all(have):
            torch._foreach_copy_([p.grad for p in self.params], self.views)
        else:
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    p.grad = v.clone()
                else:
                    p.grad.copy_(v)
