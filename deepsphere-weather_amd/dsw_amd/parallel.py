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
        if rank == 0 and (force or os.environ.get("DSW_DIST_BACKEND")):     # never silently: these change what runs
            import sys

            print("dsw_amd.parallel: %s" % ", ".join(
                ([("DSW_DIST_BACKEND=%s (process-group backend)" % os.environ["DSW_DIST_BACKEND"])] if os.environ.get("DSW_DIST_BACKEND") else [])
                + (["DSW_FORCE_GRAD_SYNC=1 (gradient exchange in a one-rank world)"] if force else [])), file=sys.stderr, flush=True)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        attempt = int(os.environ.get("DSW_PG_ATTEMPT", "0") or 0)
        if attempt > 0:
            # A rank that re-executed itself (bench.py: the first replay of a graph-captured exchange never came back) joins a
            # NEW process group under the same launcher: the launcher's store still holds the keys of the first group
            # (barrier counts, the communicator id), so this incarnation talks through its own key prefix.  Under torchrun
            # the agent hosts the store; without it rank 0 does (and its old store died with the old process image).
            from datetime import timedelta

            agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world,
                                  is_master=(rank == 0 and not agent_store), timeout=timedelta(seconds=120),
                                  wait_for_workers=False)
            kw["store"] = dist.PrefixStore("dsw_attempt_%d" % attempt, store)
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
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


def _is_permuted_dense(t: torch.Tensor) -> bool:
    """True when `t` covers `t.numel()` consecutive elements of its storage in some dimension order (a transposed /
    permuted contiguous tensor)."""
    expect = 1
    for size, stride in sorted(zip(t.shape, t.stride()), key=lambda ss: ss[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return True


class GradBucket:
    """The parameter gradients of one replica as ONE flat HBM buffer, averaged across ranks in place.

    * ``attach()`` (default) makes every ``p.grad`` a view of the bucket: backward accumulates straight into it, the
      collective runs on it, the optimizer reads it - no pack / unpack copies (the whole UNetSpherical: 7 MB each way).
      ``zero()`` replaces ``optimizer.zero_grad()`` (one memset; ``set_to_none`` would detach the views).
    * The bucket is laid out in REVERSE registration order - the order in which backward finishes the gradients - and cut
      into chunks of ``chunk_bytes``.  With ``overlap`` a chunk's all-reduce is enqueued (``async_op``) by the
      post-accumulate hook of the last of its parameters to become ready, i.e. it travels over xGMI while backward is
      still computing the earlier layers; ``finish()`` enqueues whatever is left and waits.  xGMI is point-to-point and
      the payload small, so a few MB-sized chunks (not per-tensor collectives) keep the collectives bandwidth- rather
      than latency-bound.
    * Hooks only fire when autograd runs: a step replayed from a HIP graph calls ``finish()`` after the replay and
      the chunks go out back to back (``capturing`` suppresses launches while the graph is being recorded) - or, with
      RCCL, ``finish()`` is called INSIDE the capture (``graph_capturable``): the collectives become nodes of the step
      graph, one replay = forward + backward + exchange, and several steps fit one graph.
    * Contract: ONE ``backward()`` per ``finish()``.  A second backward that reaches a chunk whose all-reduce already
      went out would add local-only gradients on top of averaged ones (replicas silently diverge), so the hook RAISES.
      Gradient accumulation (several backwards, one exchange) goes inside ``no_sync()``: hooks only count, ``finish()``
      after the block reduces the accumulated bucket.
    """

    def __init__(self, params, group=None, chunk_bytes=2 << 20, overlap=True, attach=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.overlap = overlap
        self.capturing = False
        self._defer = 0          # depth of no_sync() blocks: hooks count, nothing is launched before finish()
        if not self.params:
            raise ValueError("GradBucket needs at least one parameter that requires grad")
        order = list(reversed(self.params))
        ref = order[0]
        if any(p.dtype != ref.dtype or p.device != ref.device for p in order):
            raise ValueError("GradBucket: all parameters must share one dtype and device (one flat buffer)")
        n = sum(p.numel() for p in order)
        self.bucket = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, self.chunk_of, self.chunks = {}, {}, []
        off = start = 0
        members = []
        for p in order:
            seg = self.bucket[off:off + p.numel()]
            # the gradient view has the parameter's own (dense) strides, e.g. the column-major residual-branch weights:
            # autograd then accumulates into it without a layout change
            dense = not p.is_contiguous() and _is_permuted_dense(p)
            self.views[p] = seg.as_strided(p.shape, p.stride()) if dense else seg.view_as(p)
            off += p.numel()
            members.append(p)
            if (off - start) * self.bucket.element_size() >= chunk_bytes:
                self.chunks.append((start, off, members))
                start, members = off, []
        if members:
            self.chunks.append((start, off, members))
        for ci, (_, _, ps) in enumerate(self.chunks):
            for p in ps:
                self.chunk_of[p] = ci
        self.attached = False
        self._pending = [len(ps) for _, _, ps in self.chunks]
        self._launched = [False] * len(self.chunks)
        self._works = []
        self._hooks = []
        self._direct = False
        self._dirty = False       # a backward ran inside no_sync(): the hooks did not count, finish() sends everything
        # a bucket built earlier on the same parameters may have left its direct-accumulation pointers behind: the
        # kernels would keep adding into THAT buffer while p.grad is a view of this one
        for p in self.params:
            if hasattr(p, "_dsw_grad_acc"):
                del p._dsw_grad_acc
        if attach:
            self.attach()

    # ---- gradients live in the bucket
    def attach(self):
        for p in self.params:
            p.grad = self.views[p]
            if hasattr(p, "_dsw_grad_acc") and p._dsw_grad_acc is not self.views[p]:
                del p._dsw_grad_acc
        if self.overlap and not self._hooks:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._ready))
        self.attached = True
        return self

    def direct_accumulation(self, enable=True):
        """Let the weight-gradient kernels ADD into the bucket themselves (`p._dsw_grad_acc`, read by
        ``dsw_amd.functional``): a parameter that is used several times per backward (an autoregressive step applies every
        layer once per forward) otherwise costs one tiny autograd `add` launch per use - 266 of the 1 530 launches of the
        nside-16 training step.  The layers then return no gradient for such parameters, so their post-accumulate hooks do
        not fire: with ``overlap`` their chunks are simply exchanged by ``finish()`` instead of during backward."""
        if enable and not self.attached:
            raise RuntimeError("direct accumulation needs an attached bucket")
        for p in self.params:
            if enable:
                # call it after the last .to() / re-layout of the model: the view has the strides the parameter had when
                # the bucket was built; `functional.grad_accumulators` re-checks identity and strides on every use and falls
                # back to autograd's accumulation when they no longer match
                p._dsw_grad_acc = self.views[p]
            elif hasattr(p, "_dsw_grad_acc"):
                del p._dsw_grad_acc
        self._direct = bool(enable)
        return self

    def detach(self):
        """Give the parameters ordinary gradient tensors back (copies of the bucket's content) and remove the hooks and the
        direct-accumulation pointers: the inverse of ``attach()`` + ``direct_accumulation()``."""
        self.direct_accumulation(False)
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            if p.grad is self.views[p]:
                p.grad = self.views[p].clone()
        self.attached = False
        return self

    def reset(self):
        """Forget a half-finished exchange (e.g. a graph capture that died inside ``finish()``): no chunk counts as launched,
        nothing is waited for.  The caller synchronises the device first."""
        self._works = []
        self._launched = [False] * len(self.chunks)
        self._pending = [len(ps) for _, _, ps in self.chunks]
        self._dirty = False
        self.capturing = False
        return self

    def zero(self):
        if not self.attached:
            raise RuntimeError("zero() is the zero_grad of an attached bucket")
        for p in self.params:          # an optimizer.zero_grad(set_to_none=True) in between would have detached them
            if p.grad is not self.views[p]:
                p.grad = self.views[p]
        self.bucket.zero_()

    def active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1
                                          or os.environ.get("DSW_FORCE_GRAD_SYNC") == "1")

    # ---- the exchange
    def _ready(self, p):
        ci = self.chunk_of[p]
        if self._launched[ci]:
            raise RuntimeError(
                "GradBucket: a second backward() produced gradients for a chunk whose all-reduce was already launched "
                "(parameter of shape %s). One backward() per finish(); accumulate several backwards inside "
                "`with bucket.no_sync():` and call finish() afterwards." % (tuple(p.shape),))
        if self._defer:
            # accumulation block: count nothing (a later backward outside the block must see every parameter of the chunk
            # again before the chunk may go out); finish() sends whatever has not been launched
            self._dirty = True
            return
        self._pending[ci] -= 1
        if self._pending[ci] == 0 and self.overlap and not self._dirty and not self.capturing and self.active():
            self._launch(ci)

    def no_sync(self):
        """Context manager for gradient accumulation: backwards inside the block only accumulate into the bucket (no
        collective is enqueued by the hooks - nor by a backward AFTER the block, whose hooks would otherwise see a chunk
        "complete" that still holds local-only contributions); the next ``finish()`` averages the accumulated gradients in
        one go.  With ``direct_accumulation`` the same holds for the parameters the kernels add to."""
        import contextlib

        @contextlib.contextmanager
        def _block():
            self._defer += 1
            try:
                yield self
            finally:
                self._defer -= 1

        return _block()

    @property
    def graph_capturable(self):
        """True when ``finish()`` may run inside a HIP graph capture: RCCL collectives are stream-ordered (enqueue +
        event dependencies, no host wait), gloo's are host calls."""
        return self.active() and self.attached and dist.get_backend(self.group) == "nccl"

    def _launch(self, ci):
        start, stop, _ = self.chunks[ci]
        seg = self.bucket[start:stop]
        if dist.get_backend(self.group) == "nccl":     # RCCL averages in the collective: no separate scale kernel
            work = dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._launched[ci] = True
        self._works.append(work)

    def finish(self):
        """All chunks averaged across ranks when this returns (stream-ordered for RCCL: later kernels on the current
        stream see the result).  No-op in a single-process world."""
        if self.active():
            if not self.attached:
                self._pack()
            for ci in range(len(self.chunks)):
                if not self._launched[ci]:
                    self._launch(ci)
            for w in self._works:
                w.wait()
            if dist.get_backend(self.group) != "nccl":
                self.bucket.div_(dist.get_world_size(self.group))
            if not self.attached:
                self._unpack()
        self._works = []
        self._launched = [False] * len(self.chunks)
        self._pending = [len(ps) for _, _, ps in self.chunks]
        self._dirty = False

    __call__ = finish

    # ---- detached mode (gradients are ordinary tensors: copy in, exchange, copy out)
    def _pack(self):
        have = [p.grad is not None for p in self.params]
        views = [self.views[p] for p in self.params]
        if all(have):
            torch._foreach_copy_(views, [p.grad for p in self.params])   # one launch for the whole bucket
        else:
            for p, v in zip(self.params, views):
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)

    def _unpack(self):
        for p in self.params:
            v = self.views[p]
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


class FlatGradAllReduce(GradBucket):
    """Average the gradients of ``params`` across ranks with a single flat-bucket all-reduce (gradients stay ordinary
    tensors: packed before and unpacked after the collective)."""

    def __init__(self, params, group=None):
        super().__init__(params, group=group, chunk_bytes=1 << 62, overlap=False, attach=False)
