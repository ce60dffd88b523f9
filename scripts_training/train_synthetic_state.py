#!/usr/bin/env python3
"""Config-driven training step driver on synthetic fields (HBM-resident), MI355X.

Counterpart of the harness that drives the hot path in the reference - ``main()`` of
``/root/reference/scripts_training/train_predict_state.py:136-632`` with ``get_pytorch_model``
(``/root/reference/modules/utils_config.py:349-372``) - reduced to what does not need the un-vendored
packages (xforecasting / xscaler / zarr data): JSON config (same ``model_settings`` / ``training_settings`` /
``ar_settings`` schema as ``configs/UNetSpherical/*/*.json``) -> model through the ``architecture_name`` +
inspect-filtered kwargs convention -> ``.to(device)`` -> Adam(eps=1e-7) (``train_predict_state.py:334-340``)
-> ``WeightedMSELoss`` with the sampling's area weights (``:325-330``, ``modules/loss.py:118-156``) ->
autoregressive steps: ``ar_iterations`` + 1 forwards per optimisation step, the prediction replacing the most
recent dynamic features of the input window (``stack_most_recent_prediction``), the loss summed over the
iterations, one backward, one flat-bucket gradient all-reduce when launched under torchrun, optimizer step.
The whole step (~77 ConvCheb launches x 7 forwards at the default config) is captured once into a HIP graph and
replayed; under torchrun parameters and operator buffers are broadcast from rank 0 first (ARPACK's lambda_max
estimate is not deterministic, so independently built replicas would not share one operator).

    python scripts_training/train_synthetic_state.py --config_file configs/UNetSpherical/Healpix_400km/InterpPool-Graph_knn.synthetic.json --steps 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts_training/train_synthetic_state.py ...

The AR schedule of xforecasting itself is external to the reference repository (parity unpinned).
"""
import argparse
import inspect
import json
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(REPO, "deepsphere-weather_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

_PRECISIONS = {"float32": torch.float32, "bfloat16": torch.bfloat16}


def read_config(path):
    with open(path) as f:
        cfg = json.load(f)
    for section in ("model_settings", "training_settings"):
        if section not in cfg:
            raise KeyError(f"config file lacks the '{section}' section")
    cfg.setdefault("ar_settings", {"input_k": [-3, -2, -1], "output_k": [0], "ar_iterations": 0,
                                   "stack_most_recent_prediction": True})
    cfg.setdefault("synthetic_settings", {"n_dynamic_features": 2, "n_bc_features": 4, "seed_data": 1234})
    return cfg


def synthetic_tensor_info(cfg):
    """What xforecasting's dataset reports for an ERA5-like state: inputs = len(input_k) time steps of
    (dynamic + boundary-condition) features, outputs = len(output_k) steps of the dynamic features."""
    ms, syn, ar = cfg["model_settings"], cfg["synthetic_settings"], cfg["ar_settings"]
    sampling = ms["sampling"].lower()
    if sampling != "healpix":
        raise NotImplementedError("the synthetic driver builds HEALPix samplings (DESIGN.md section 7)")
    n_node = 12 * int(ms["sampling_kwargs"]["subdivisions"]) ** 2
    n_in = syn["n_dynamic_features"] + syn["n_bc_features"]
    return {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": n_in, "output_n_feature": syn["n_dynamic_features"],
        "input_n_time": len(ar["input_k"]), "output_n_time": len(ar["output_k"]),
        "input_shape_info": {"dynamic": {"node": n_node}}, "output_shape_info": {"dynamic": {"node": n_node}},
    }


def get_pytorch_model(module, model_settings):
    """``architecture_name`` names a class of ``module``; only the settings its ``__init__`` declares are passed."""
    if not isinstance(module, types.ModuleType):
        raise TypeError("'module' must be a preimported module with the architecture definition.")
    cls = getattr(module, model_settings["architecture_name"])
    accepted = inspect.getfullargspec(cls.__init__).args
    return cls(**{k: v for k, v in model_settings.items() if k in accepted})


def ar_training_step(model, x, targets, n_dyn, stack_most_recent_prediction=True, criterion=None, dim_info=None):
    """One autoregressive pass: ``len(targets)`` forwards; returns the summed loss.
    x [B, T, V, F]; each target [B, 1, V, n_dyn]; the dynamic features are the LAST n_dyn features (the model's
    increment path reads ``x[:, -1, :, -2:]``, my_models_graph.py:494).  ``criterion`` = a ``WeightedMSELoss`` applied
    to ``reshape_tensors_4_loss`` views as in the reference's training loop (loss.py:31-54,118-156); None = plain MSE."""
    loss = x.new_zeros(())
    for target in targets:
        y = model(x)
        if criterion is None:
            loss = loss + torch.mean((y - target) ** 2)
        else:
            from modules.loss import reshape_tensors_4_loss

            loss = loss + criterion(*reshape_tensors_4_loss(y, target, dim_info))
        if stack_most_recent_prediction and len(targets) > 1:
            nxt = x[:, -1:].clone()
            nxt[..., -n_dyn:] = y[:, -1:]
            x = torch.cat((x[:, 1:], nxt), dim=1)
    return loss


class Trainer:
    """model + WeightedMSELoss + Adam(eps=1e-7) + gradient bucket, stepping on static HBM-resident tensors.

    The gradients live in ONE flat buffer (``dsw_amd.parallel.GradBucket``): backward accumulates into it, RCCL averages
    it in place, Adam reads it.  Single process: ``step()`` launches the whole optimisation step - zero, the AR forwards,
    backward, optimizer update - as ONE HIP graph replay when ``use_graph``.  N > 1 ranks: the step runs eagerly and the
    bucket's chunks are all-reduced from the post-accumulate hooks WHILE backward is still computing the earlier layers
    (``overlap``); with ``use_graph`` and N > 1 the graph holds forwards + backward, the chunks go out back to back after
    the replay and the optimizer update follows eagerly."""

    def __init__(self, model, x, targets, n_dyn, lr, weights=None, stack=True, use_graph=True, dim_names=None):
        from dsw_amd.parallel import GradBucket
        from modules.loss import WeightedMSELoss

        self.model, self.x, self.targets, self.n_dyn, self.stack = model, x, targets, n_dyn, stack
        dim_names = dim_names or ["sample", "time", "node", "feature"]
        self.dim_info = {n: i for i, n in enumerate(dim_names)}
        self.criterion = WeightedMSELoss(weights=None if weights is None else weights.to(x.device))   # fp32 area weights, as the reference
        # (DSW_FORCE_GRAD_SYNC=1: a one-rank world runs the N > 1 code path - RCCL exchange, in-graph capture - on one GPU)
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and (torch.distributed.get_world_size() > 1 or os.environ.get("DSW_FORCE_GRAD_SYNC") == "1")
        # capturable: the step counter lives on the device, so optimizer.step() can sit inside a HIP graph
        self.optimizer = torch.optim.Adam(model.parameters(), lr=lr, eps=1e-7, weight_decay=0, amsgrad=False,
                                          capturable=bool(use_graph and x.is_cuda))
        self.bucket = GradBucket(model.parameters(), overlap=True)
        # every layer is applied once per AR forward: its weight gradients are ADDED into the bucket by the kernels
        # themselves (no autograd `add` per use).  An eager N > 1 run keeps the hook-driven overlapped exchange instead.
        if x.is_cuda and (not self.distributed or use_graph):
            self.bucket.direct_accumulation()
        self.loss = None
        self.graph = None
        self.sync_in_graph = False
        self.launch = "eager" if not self.distributed else "eager, chunked all-reduce overlapped with backward"
        if use_graph and x.is_cuda:
            self._capture()

    def _fwd_bwd(self):
        self.bucket.zero()
        loss = ar_training_step(self.model, self.x, self.targets, self.n_dyn, self.stack, self.criterion, self.dim_info)
        loss.backward()
        return loss.detach()

    def _eager_step(self):
        loss = self._fwd_bwd()
        self.bucket.finish()
        self.optimizer.step()
        return loss

    def _capture(self):
        # warm-up on a side stream (operator caches, tile plans, allocator pools, Adam state), then capture.  The
        # warm-up steps must not count: parameters and optimizer state are restored afterwards.
        saved = [p.detach().clone() for p in self.model.parameters()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            # RCCL collectives are stream-ordered and can be recorded: the exchange and the optimizer update then
            # sit in the graph as well (N > 1: one replay = the whole step).  gloo's are host calls: they follow the replay.
            self.sync_in_graph = self.distributed and self.bucket.graph_capturable
            self.bucket.capturing = True      # the hooks must not enqueue collectives into the recording
            # "thread_local": RCCL's watchdog thread polls the events of the warm-up collectives (hipEventQuery) while this
            # thread records; under the default global capture mode such a call from ANOTHER thread is an error that ends the
            # process (seen once in ~10 runs of the one-rank RCCL test: ProcessGroupNCCL watchdog abort)
            mode = {"capture_error_mode": "thread_local"} if self.distributed else {}
            with torch.cuda.graph(g, **mode):
                self.loss = self._fwd_bwd()
                if self.sync_in_graph:
                    self.bucket.finish()      # all chunks back to back, as nodes of the graph
                if not self.distributed or self.sync_in_graph:
                    self.optimizer.step()
            self.bucket.capturing = False
            if not self.sync_in_graph:
                self.bucket.finish()          # resets the hook counters of the recorded backward
            self.graph = g
            self.launch = "hip graph: whole step" if not self.distributed else \
                "hip graph: whole step incl. the RCCL gradient all-reduce" if self.sync_in_graph else \
                "hip graph: fwd+bwd; chunked all-reduce + Adam eagerly after the replay"
        except Exception as exc:  # noqa: BLE001 - capture not possible: stay eager
            print("trainer: HIP graph capture unavailable (%s: %s); stepping eagerly" % (type(exc).__name__, exc),
                  file=sys.stderr)
            self.graph = None
            torch.cuda.synchronize()
            # a capture that died inside finish() leaves chunks marked as launched and works of the aborted recording
            # behind: the first eager backward would raise "a second backward()..." 
            self.bucket.reset()
        if self.distributed:
            # ALL ranks replay the graph or NONE does: a rank that fell back issues its chunk all-reduces in hook order,
            # the others replay them in index order from their graphs - mismatched collectives otherwise
            import torch.distributed as dist

            ok = torch.tensor([1 if self.graph is not None else 0], device=self.x.device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and self.graph is not None:
                print("trainer: another rank could not capture its step graph; stepping eagerly on all ranks", file=sys.stderr)
                self.graph = None
                torch.cuda.synchronize()
                self.bucket.reset()
        with torch.no_grad():
            for p, v in zip(self.model.parameters(), saved):
                p.copy_(v)
            for st in self.optimizer.state.values():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.zero_()

    def step(self):
        if self.graph is None:
            return self._eager_step()
        self.graph.replay()
        if self.distributed and not self.sync_in_graph:
            self.bucket.finish()
            self.optimizer.step()
        return self.loss


def build_trainer(cfg, device, dtype=None, batch_size=None, ar_iterations=None, rank=0, use_graph=True, model=None,
                  weights=None):
    """config -> (Trainer, info dict): what ``main`` and the tests share."""
    import modules.my_models_graph as my_architectures
    from dsw_amd.parallel import broadcast_module_state
    from modules.loss import AreaWeights

    ms, ts, ar, syn = cfg["model_settings"], cfg["training_settings"], cfg["ar_settings"], cfg["synthetic_settings"]
    dtype = dtype or _PRECISIONS.get(ts.get("numeric_precision", "float32"))
    if dtype is None:
        raise ValueError("numeric_precision: the HIP path implements 'float32' and 'bfloat16'")
    ms = dict(ms)
    ms["tensor_info"] = synthetic_tensor_info(cfg)
    if model is None:
        torch.manual_seed(ts.get("seed_model_weights", 10))
        model = get_pytorch_model(my_architectures, ms)
    model = model.to(device).to(dtype)
    # replicas must share ONE set of operators and weights: lambda_max comes from ARPACK with a random start vector
    broadcast_module_state(model, src=0)
    if weights is None:
        weights = AreaWeights(model.graphs[0])                      # train_predict_state.py:327
    B = batch_size or ts.get("training_batch_size", 16)
    info = ms["tensor_info"]
    V, T, F = info["input_shape_info"]["dynamic"]["node"], info["input_n_time"], info["input_n_feature"]
    n_dyn = info["output_n_feature"]
    n_ar = (ar.get("ar_iterations", 0) if ar_iterations is None else ar_iterations) + 1
    g = torch.Generator(device=device).manual_seed(syn.get("seed_data", 1234) + rank)
    x = torch.randn(B, T, V, F, device=device, dtype=dtype, generator=g)
    targets = [torch.randn(B, info["output_n_time"], V, n_dyn, device=device, dtype=dtype, generator=g)
               for _ in range(n_ar)]
    trainer = Trainer(model, x, targets, n_dyn, ts.get("learning_rate", 0.007), weights=weights,
                      stack=ar.get("stack_most_recent_prediction", True), use_graph=use_graph,
                      dim_names=info["dim_order"]["dynamic"])
    return trainer, {"B": B, "V": V, "n_ar": n_ar, "architecture": ms["architecture_name"],
                     "parameters": sum(p.numel() for p in model.parameters())}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config_file", required=True)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch_size", type=int, default=None, help="per GPU; default training_batch_size of the config")
    ap.add_argument("--ar_iterations", type=int, default=None, help="override ar_settings.ar_iterations")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--graph", action="store_true",
                    help="N > 1: replay forwards + backward from a HIP graph (the all-reduce then follows the replay "
                         "instead of overlapping backward); single process: the default")
    args = ap.parse_args(argv)

    from dsw_amd import _native
    from dsw_amd.parallel import init_from_env

    cfg = read_config(args.config_file)
    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("the ConvCheb hot path runs only on a ROCm device (no CPU fallback)")
    _native.load()
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    trainer, info = build_trainer(cfg, device, batch_size=args.batch_size, ar_iterations=args.ar_iterations, rank=rank,
                                  use_graph=(not args.no_graph) and (world == 1 or args.graph))

    losses = []
    for _ in range(args.warmup):
        losses.append(trainer.step().clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(trainer.step().clone())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(args.steps, 1)
    losses = [float(v) for v in losses]
    if rank == 0:
        print(json.dumps({
            "architecture": info["architecture"], "parameters": info["parameters"], "n_gpus": world,
            "batch_per_gpu": info["B"], "nodes": info["V"], "forwards_per_step": info["n_ar"], "ms_per_step": dt * 1e3,
            "samples_per_s": info["B"] * world / dt, "loss_first": losses[0], "loss_last": losses[-1],
            "launch": trainer.launch, "loss": "WeightedMSELoss(area weights)",
            "dtype": str(trainer.x.dtype).replace("torch.", ""), "data": "synthetic",
        }))
    return losses


if __name__ == "__main__":
    main()
