#!/usr/bin/env python3
"""Benchmark of the ConvCheb hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ns|c3|unet] [--knn 8|20]

One "step" = one forward + backward pass of the hot path over one batch of synthetic HEALPix
fields already resident in HBM (plus, for N > 1, the single flat-bucket RCCL all-reduce of the
parameter gradients).  Metric (BASELINE.json): HEALPix nodes*channels/sec through ConvCheb (fwd+bwd)
= B * V * Fin * n_gpus / t_step.  Scaling is weak: every rank owns a full per-GPU batch.

Workloads
  ns    north-star shape: nside=64 (V=49152), K=3, 32->64 channels, B=16 per GPU, fp32   (default)
  c3    BASELINE configs[2]: nside=64, K=5, 64->128, B=16 per GPU, bf16 storage / fp32 accumulate
  unet  BASELINE configs[1]: UNetSpherical nside=32, K=3, B=8 per GPU, fp32, interp pooling

Rank 0 prints ONE JSON line.  At N=1 it also carries
  roofline      the SpMM kernel: algorithmic bytes per launch / mean launch duration (HIP events on
                the launch stream) against the 8 TB/s HBM peak; "traffic" from profiles/ PMC data
  cpu_baseline  the oracle's torch restatement of the reference CPU path, timed on this box's cores
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "deepsphere-weather_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling

GRAPH_STEPS = 10   # steps per captured graph on a single GPU (see main)

WORKLOADS = {
    "ns": dict(nside=64, K=3, fin=32, fout=64, batch=16, dtype="f32"),
    "c3": dict(nside=64, K=5, fin=64, fout=128, batch=16, dtype="bf16"),
    "unet": dict(nside=32, K=3, fin=18, fout=2, batch=8, dtype="f32"),
    # BASELINE configs[4]: equiangular 200 x 400 (V = 80 000, irregular degree at the poles), k = 20, K = 3, 32 ch,
    # interpolation pooling to the HEALPix nside=32 sampling and back (pooling BETWEEN samplings)
    "c5": dict(nside=0, nlat=200, nlon=400, K=3, fin=32, fout=32, batch=8, dtype="f32"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--knn", type=int, default=8, help="neighbours of the HEALPix k-NN stencil (8 or 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    return ap.parse_args()


def make_layer(wl, knn, device, dtype):
    from dsw_amd import sphere
    from modules.layers import ConvCheb, prepare_torch_laplacian

    graph = sphere.SphereHealpix(wl["nside"], nest=True, k=knn)
    lap = prepare_torch_laplacian(graph.L, lmax=1.95)  # fixed lmax: identical operator on every rank
    torch.manual_seed(10)  # seed_model_weights of the reference configs
    layer = ConvCheb(wl["fin"], wl["fout"], wl["K"], laplacian=lap, bias=True)
    return layer.to(device).to(dtype), lap


class _C5Block(torch.nn.Module):
    """ConvCheb on the equiangular graph -> interp-pool to a HEALPix sampling -> ConvCheb there -> unpool back."""

    def __init__(self, wl):
        super().__init__()
        from dsw_amd import sphere
        from modules.layers import ConvCheb, GeneralAvgPool, GeneralAvgUnpool, prepare_torch_laplacian

        fine = sphere.SphereEquiangular(nlat=wl["nlat"], nlon=wl["nlon"], k=20)
        coarse = sphere.SphereHealpix(32, nest=True, k=20)
        pool_m, unpool_m = sphere.knn_interp_pool_matrices(fine.coords, coarse.coords, k=9)
        torch.manual_seed(10)
        self.conv_fine = ConvCheb(wl["fin"], wl["fout"], wl["K"], laplacian=prepare_torch_laplacian(fine.L, lmax=1.95))
        self.pool = GeneralAvgPool(pool_m)
        self.conv_coarse = ConvCheb(wl["fout"], wl["fout"], wl["K"],
                                    laplacian=prepare_torch_laplacian(coarse.L, lmax=1.95))
        self.unpool = GeneralAvgUnpool(unpool_m)

    def forward(self, x):
        y = self.conv_fine(x)
        z, idx = self.pool(y)
        return y + self.unpool(self.conv_coarse(z), idx)


def make_unet(wl, knn, device):
    import modules.my_models_graph as arch

    V = 12 * wl["nside"] ** 2
    tensor_info = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
    }
    torch.manual_seed(10)
    model = arch.UNetSpherical(tensor_info, sampling="healpix",
                               sampling_kwargs={"subdivisions": wl["nside"], "nest": True},
                               kernel_size_conv=wl["K"], conv_type="graph", graph_type="knn", knn=knn,
                               pool_method="interp")
    with torch.no_grad():  # ReZero starts at 0, which would zero every conv gradient
        for n, p in model.named_parameters():
            if n.endswith("rezero_weight"):
                p.fill_(0.5)
    return model.to(device)


def spmm_algorithmic_bytes(E, Lb, K):
    """forward recurrence [2 + 3(K-2)] E + (K-1) Lb, adjoint [3 + 4(K-2)] E + (K-1) Lb."""
    fwd = (2 + 3 * (K - 2)) * E + (K - 1) * Lb
    bwd = (3 + 4 * (K - 2)) * E + (K - 1) * Lb
    return fwd, bwd


def _native_lib():
    from dsw_amd import _native

    return _native.load()


def roofline_leg(layer, x, steps, warmup, traffic_key=None):
    """Time exactly the SpMM launches of one fwd+bwd (K-1 basis hops + K-1 adjoint hops)."""
    from dsw_amd import functional as F_

    op = F_.get_operator(layer.laplacian)
    opt = op.transpose()
    K = layer.kernel_size
    B, V, C = x.shape
    es = x.element_size()
    E = B * V * C * es
    Lb = op.nnz * 8 + 4 * (V + 1)
    lib = _native_lib()
    dcode = 1 if x.dtype == torch.bfloat16 else 0
    st = torch.cuda.current_stream().cuda_stream
    T = torch.empty((K - 1, B, V, C), dtype=x.dtype, device=x.device)
    G0 = torch.randn_like(x)
    Gr = torch.randn((K - 1, B, V, C), dtype=x.dtype, device=x.device)
    spare = torch.empty((2, B, V, C), dtype=x.dtype, device=x.device)
    pp, _k1 = F_._plan_ptr(op, x) if F_._FWD_FUSED else (None, None)
    ppt, _k2 = F_._plan_ptr(opt, x)

    def fwd():   # K-1 launches: T_1 = L x, T_k = 2 L T_{k-1} - T_{k-2}
        rc = lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz,
                                    x.data_ptr(), T.data_ptr(), B, C, K, dcode, st, pp)
        assert rc == 0

    def adj():   # K-1 launches: G_{j-1} += (2|1) L^T G_j - G_{j+1}
        rc = lib.dsw_cheb_basis_adj(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz,
                                    G0.data_ptr(), Gr.data_ptr(), B, C, K, dcode, st, ppt, spare.data_ptr())
        assert rc == 0

    def launches():
        fwd()
        adj()

    for _ in range(warmup):
        launches()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(stream)
    for _ in range(steps):
        launches()
    t1.record(stream)
    torch.cuda.synchronize()
    hops = 2 * (K - 1)                      # operator applications per step (forward + adjoint)
    pairs = (K - 1 + 1) // 2                # launches of a pairwise-fused recurrence
    n_launch = ((K - 1) if pp is None else pairs) + ((K - 1) if ppt is None else pairs)
    avg_s = t0.elapsed_time(t1) * 1e-3 / (steps * n_launch)
    fwd_b, bwd_b = spmm_algorithmic_bytes(E, Lb, K)
    bytes_per_launch = (fwd_b + bwd_b) / n_launch
    achieved = bytes_per_launch / avg_s / 1e9
    # forward recurrence alone (the north-star gate)
    t0.record(stream)
    for _ in range(steps):
        fwd()
    t1.record(stream)
    torch.cuda.synchronize()
    fwd_s = t0.elapsed_time(t1) * 1e-3 / steps
    traffic = None   # HBM bytes per launch from the PMC passes (tools/prof_pmc.sh + tools/make_traffic_json.py)
    tpath = os.path.join(REPO, "profiles", "spmm_traffic.json")
    if traffic_key and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(traffic_key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {
        "bound": "hbm",
        "kernel": ("SpMM recurrence: forward %s + adjoint %s, %d launches/step" % (
            "spmm2_fused" if pp is not None else "spmm_csr x%d" % (K - 1),
            "spmm2_fused" if ppt is not None else "spmm_csr x%d" % (K - 1), n_launch)),
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "bytes_per_launch": int(bytes_per_launch), "avg_launch_us": round(avg_s * 1e6, 2),
        "fwd_recurrence_us": round(fwd_s * 1e6, 2),
        "fwd_recurrence_frac": round(fwd_b / fwd_s / 1e9 / HBM_PEAK_GBS, 4),
    }


def cpu_baseline_leg(wl, lap, layer):
    """The oracle's torch restatement of the reference CPU path (torch.sparse.mm + matmul), fp32."""
    from oracle import cheb_oracle as orc

    cores = os.cpu_count() or 1
    V = 12 * wl["nside"] ** 2
    B = wl["batch"]
    torch.manual_seed(1234)
    x = torch.randn(B, V, wl["fin"])
    gy = torch.randn(B, V, wl["fout"])
    w = layer.weight.detach().float().cpu()
    b = layer.bias.detach().float().cpu()
    lap = lap.float().coalesce()

    def one():
        t = time.perf_counter()
        orc.conv_cheb_fwd_bwd_torch(lap, x, w, b, gy)
        return time.perf_counter() - t

    # ATen's sparse CPU kernels stop scaling (and then regress) well below the core count of a GPU
    # host: pick the fastest of a few thread counts with one probe each, then time that setting.
    best_thr, best_t = 1, float("inf")
    for thr in sorted({min(cores, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(thr)
        one()
        t = one()
        if t < best_t:
            best_thr, best_t = thr, t
        if t > 6.0:
            break
    torch.set_num_threads(best_thr)
    times = []
    budget = time.perf_counter() + 20.0
    while len(times) < 8 and (time.perf_counter() < budget or len(times) < 2):
        times.append(one())
    med = float(np.median(times))
    return {
        "value": B * V * wl["fin"] / med, "unit": "nodes*channels/s", "cores": torch.get_num_threads(),
        "host_cpus": cores,
        "kind": "port",
        "sample": "oracle conv_cheb torch restatement (sparse COO mm + matmul, fp32), full %s shape "
                  "[B=%d,V=%d,%d->%d,K=%d], median of %d fwd+bwd, %.1f ms" % (
                      "workload", B, V, wl["fin"], wl["fout"], wl["K"], len(times), med * 1e3),
    }


def main():
    args = parse()
    from dsw_amd import _native
    from dsw_amd.parallel import FlatGradAllReduce, init_from_env

    rank, world, local = init_from_env()
    assert torch.cuda.is_available(), "bench.py measures the HIP path; a ROCm device is required"
    _native.load()
    local = local % torch.cuda.device_count()   # (gloo smoke runs of the N>1 path put two ranks on one device)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    wl = WORKLOADS[args.workload]
    dtype = torch.bfloat16 if wl["dtype"] == "bf16" else torch.float32
    V = wl["nlat"] * wl["nlon"] if args.workload == "c5" else 12 * wl["nside"] ** 2
    B = wl["batch"]
    torch.manual_seed(1234 + rank)

    if args.workload == "unet":
        model = make_unet(wl, args.knn if args.knn != 8 else 20, device)
        x = torch.randn(B, 3, V, 6, device=device)
        target = torch.randn(B, 1, V, 2, device=device)
        lap = None

        def step():
            model.zero_grad(set_to_none=True)
            loss = ((model(x) - target) ** 2).mean()
            loss.backward()
    else:
        if args.workload == "c5":
            model, lap = _C5Block(wl).to(device), None
        else:
            model, lap = make_layer(wl, args.knn, device, dtype)
        x = torch.randn(B, V, wl["fin"], device=device, dtype=dtype).requires_grad_(True)
        gy = torch.randn(B, V, wl["fout"], device=device, dtype=dtype)

        def step():
            model.zero_grad(set_to_none=True)
            x.grad = None
            model(x).backward(gy)

    sync_grads = FlatGradAllReduce(model.parameters())

    # The step is ~8 back-to-back kernels of 80-120 us each; launched eagerly from Python the GPU idles ~5 us between
    # them.  Capture one fwd+bwd into a HIP graph and replay it (same kernels, same work, same buffers); the gradient
    # all-reduce stays outside the graph.  Falls back to eager launches if capture is not possible.
    graph = None
    graph_multi = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()          # builds the operator caches / tile plans outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            torch.cuda.synchronize()
            graph = g
            # single GPU: consecutive graph launches leave the GPU idle for ~40 us (hipGraphLaunch latency), 8 % of
            # this step; a second graph holding GRAPH_STEPS whole steps amortises it.  With N > 1 every step is
            # followed by the gradient all-reduce, so steps stay one graph each.
            if world == 1 and args.steps >= GRAPH_STEPS and os.environ.get("DSW_FORCE_GRAD_SYNC") != "1":
                gm = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gm):
                    for _ in range(GRAPH_STEPS):
                        step()
                torch.cuda.synchronize()
                graph_multi = gm
        except Exception as exc:  # noqa: BLE001 - any capture problem -> eager
            if rank == 0:
                print("bench: HIP graph capture unavailable (%s); running eagerly" % type(exc).__name__, file=sys.stderr)
            graph = None
            graph_multi = None
            torch.cuda.synchronize()

    def full_step():
        if graph is not None:
            graph.replay()
        else:
            step()
        sync_grads()

    def run_steps(n):
        """exactly n steps"""
        if graph_multi is not None:
            for _ in range(n // GRAPH_STEPS):
                graph_multi.replay()
            n = n % GRAPH_STEPS
        for _ in range(n):
            full_step()

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms = elapsed / args.steps * 1e3
    units = B * V * wl["fin"] * world  # node-channels through the path per step, whole job
    out = {
        "metric": "HEALPix nodes*channels/sec through ConvCheb K=%d (fwd+bwd)" % wl["K"],
        "value": units / (elapsed / args.steps), "unit": "nodes*channels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": wl["dtype"], "data": "synthetic",
        "config": {
            "workload": {
                "ns": "single ConvCheb layer, HEALPix nside=64 nested (V=49152), K=3, 32->64 ch, B=16/GPU, fp32 (north-star shape)",
                "c3": "single ConvCheb layer, HEALPix nside=64 nested, K=5, 64->128 ch, B=16/GPU, bf16 storage + fp32 accumulate",
                "unet": "UNetSpherical, HEALPix nside=32 nested, K=3, B=8/GPU, fp32, interp pooling (11 ConvCheb + 4 RemapBlock)",
                "c5": "equiangular 200x400 (V=80000, k=20, irregular degree) ConvCheb K=3 32ch + interp pooling to HEALPix nside=32 and back, B=8/GPU, fp32",
            }[args.workload],
            "knn": 20 if args.workload == "c5" else args.knn if args.workload != "unet" else (args.knn if args.knn != 8 else 20),
            "batch_per_gpu": B, "global_batch": B * world, "nodes": V,
            "parallelism": "dp%d (batch shards, flat-bucket RCCL grad all-reduce)" % world,
            "launch": ("hip graph replay, %d steps per graph" % GRAPH_STEPS if graph_multi is not None
                       else "hip graph replay of one fwd+bwd" if graph is not None else "eager"),
        },
    }
    if rank == 0 and world == 1:
        # per-iteration HIP-event times (outside the timed region above): median and p10 / p90
        n_ev = 100
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for a, b in evs:
            a.record()
            full_step()
            b.record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        out["per_step_us"] = {"n": n_ev, "median": round(us[n_ev // 2], 1), "p10": round(us[n_ev // 10], 1),
                              "p90": round(us[(9 * n_ev) // 10], 1)}
    if rank == 0 and world == 1 and args.workload not in ("unet", "c5"):
        if not args.no_roofline:
            out["roofline"] = roofline_leg(model, x.detach(), max(10, args.steps), 5,
                                           traffic_key=args.workload if args.knn == 8 else None)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg(wl, lap, model)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
