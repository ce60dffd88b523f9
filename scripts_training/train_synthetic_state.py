#!/usr/bin/env python3
"""Config-driven training step driver on synthetic fields (HBM-resident), MI355X.

Counterpart of the harness that drives the hot path in the reference - ``main()`` of
``/root/reference/scripts_training/train_predict_state.py:136-632`` with ``get_pytorch_model``
(``/root/reference/modules/utils_config.py:349-372``) - reduced to what does not need the un-vendored
packages (xforecasting / xscaler / zarr data): JSON config (same ``model_settings`` / ``training_settings`` /
``ar_settings`` schema as ``configs/UNetSpherical/*/*.json``) -> model through the ``architecture_name`` +
inspect-filtered kwargs convention -> ``.to(device)`` -> Adam(eps=1e-7) (``train_predict_state.py:334-340``)
-> autoregressive steps: ``ar_iterations`` forwards per optimisation step, the prediction replacing the most
recent dynamic features of the input window (``stack_most_recent_prediction``), MSE summed over the
iterations, one backward, one flat-bucket gradient all-reduce when launched under torchrun, optimizer step.

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


def ar_training_step(model, x, targets, n_dyn, stack_most_recent_prediction=True):
    """One autoregressive pass: ``len(targets)`` forwards; returns the summed MSE.
    x [B, T, V, F]; each target [B, 1, V, n_dyn]; the dynamic features are the LAST n_dyn features (the model's
    increment path reads ``x[:, -1, :, -2:]``, my_models_graph.py:494)."""
    loss = x.new_zeros(())
    for target in targets:
        y = model(x)
        loss = loss + torch.mean((y - target) ** 2)
        if stack_most_recent_prediction and len(targets) > 1:
            nxt = x[:, -1:].clone()
            nxt[..., -n_dyn:] = y[:, -1:]
            x = torch.cat((x[:, 1:], nxt), dim=1)
    return loss


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config_file", required=True)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch_size", type=int, default=None, help="per GPU; default training_batch_size of the config")
    ap.add_argument("--ar_iterations", type=int, default=None, help="override ar_settings.ar_iterations")
    args = ap.parse_args(argv)

    import modules.my_models_graph as my_architectures
    from dsw_amd import _native
    from dsw_amd.parallel import FlatGradAllReduce, init_from_env

    cfg = read_config(args.config_file)
    ms, ts, ar, syn = cfg["model_settings"], cfg["training_settings"], cfg["ar_settings"], cfg["synthetic_settings"]
    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("the ConvCheb hot path runs only on a ROCm device (no CPU fallback)")
    _native.load()
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dtype = _PRECISIONS.get(ts.get("numeric_precision", "float32"))
    if dtype is None:
        raise ValueError("numeric_precision: the HIP path implements 'float32' and 'bfloat16'")

    ms = dict(ms)
    ms["tensor_info"] = synthetic_tensor_info(cfg)
    torch.manual_seed(ts.get("seed_model_weights", 10))
    model = get_pytorch_model(my_architectures, ms).to(device).to(dtype)
    n_params = sum(p.numel() for p in model.parameters())
    optimizer = torch.optim.Adam(model.parameters(), lr=ts.get("learning_rate", 0.007), eps=1e-7,
                                 weight_decay=0, amsgrad=False)
    sync_grads = FlatGradAllReduce(model.parameters())

    B = args.batch_size or ts.get("training_batch_size", 16)
    info = ms["tensor_info"]
    V, T, F = info["input_shape_info"]["dynamic"]["node"], info["input_n_time"], info["input_n_feature"]
    n_dyn = info["output_n_feature"]
    n_ar = (ar.get("ar_iterations", 0) if args.ar_iterations is None else args.ar_iterations) + 1
    g = torch.Generator(device=device).manual_seed(syn.get("seed_data", 1234) + rank)
    x = torch.randn(B, T, V, F, device=device, dtype=dtype, generator=g)
    targets = [torch.randn(B, info["output_n_time"], V, n_dyn, device=device, dtype=dtype, generator=g)
               for _ in range(n_ar)]

    losses = []

    def step():
        optimizer.zero_grad(set_to_none=True)
        loss = ar_training_step(model, x, targets, n_dyn, ar.get("stack_most_recent_prediction", True))
        loss.backward()
        sync_grads()
        optimizer.step()
        return loss.detach()

    for _ in range(args.warmup):
        losses.append(step())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(args.steps, 1)
    losses = [float(v) for v in losses]
    if rank == 0:
        print(json.dumps({
            "architecture": ms["architecture_name"], "parameters": n_params, "n_gpus": world,
            "batch_per_gpu": B, "nodes": V, "forwards_per_step": n_ar, "ms_per_step": dt * 1e3,
            "samples_per_s": B * world / dt, "loss_first": losses[0], "loss_last": losses[-1],
            "dtype": str(dtype).replace("torch.", ""), "data": "synthetic",
        }))
    return losses


if __name__ == "__main__":
    main()
