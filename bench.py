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

Launch.  `python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (it re-executes
under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one process per GPU,
RCCL) and relays rank 0's line; started BY torchrun it checks that WORLD_SIZE is the N that was asked for.  With N > 1
the gradient all-reduce is captured INTO the step graph (ten steps per graph, as on one GPU); the captured graph is
validated against an eager step + all-reduce first, and any failure falls back to collectives issued after the replay.

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

GRAPH_STEPS = 10   # steps per captured graph (see main): the largest divisor of --steps in 5..25, else this

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
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE of the torchrun environment, else 1)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--knn", type=int, default=8, help="neighbours of the HEALPix k-NN stencil (8 or 20)")
    ap.add_argument("--min-timed-ms", type=float, default=1500.0,
                    help="repeat the timed region (exactly --steps steps each) until this much has been timed; median reported")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="STRONG scaling: this many samples in total, sharded over the ranks with dsw_amd.parallel.shard_batch "
                         "(ragged and empty shards allowed); default: the workload's batch PER GPU (weak scaling)")
    ap.add_argument("--pmc-leg", default=None, choices=["fwd", "adj", "pool", "bwdd", "fwd1l"],
                    help="profiling aid (tools/pmc_traffic.sh): run ONLY that leg of the roofline measurement - the forward "
                         "recurrence, the adjoint recurrence or the pooling products of the workload - and exit; every "
                         "dispatch of the run then belongs to the leg")
    ap.add_argument("--grad-handling", default="auto", choices=["auto", "autograd", "bucket"],
                    help="where the parameter gradients live: 'bucket' = one flat buffer (one memset per step) the weight-gradient "
                         "kernels add into - what every N > 1 run does; 'autograd' = zero_grad(set_to_none) + autograd's tensors; "
                         "'auto' = bucket when N > 1, autograd when N = 1.  `--gpus 1 --grad-handling bucket` runs the N > 1 step "
                         "without the exchange, so that a scaling curve compares like with like")
    ap.add_argument("--kernel-trace-child", type=int, default=0,
                    help="internal (graph_kernel_durations): build, capture, warm up, replay this many steps and exit - the "
                         "process rocprofv3 traces to time the kernels inside the replayed graph")
    ap.add_argument("--detail", default="file", choices=["file", "inline", "none"],
                    help="where the long form of the roofline measurement goes (every role of the step, pooling products, stand-alone "
                         "legs): a side file (gpurun_out/bench_detail_<workload>.json, or --detail-file), the line itself, or nowhere")
    ap.add_argument("--detail-file", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--zero-inputs", action="store_true",
                    help="diagnostics: zero-filled activations / gradients (same launches, least switching power) - how much of the "
                         "step time is the clock the chip sustains under real data; the line says data = zeros")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--plain-resblock", action="store_true",
                    help="unet: residual-block tails as separate passes instead of GEMM epilogues (A/B)")
    ap.add_argument("--resblock-mode", default="", help="unet A/B: fused residual-block tails (off by default): subset of 'ef' "
                    "(e = forward epilogue, f = fork), '-' = backward only")
    ap.add_argument("--torch-cat", action="store_true",
                    help="unet: decoder concatenations through torch.cat instead of in-place buffers (A/B)")
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
        # conservative (overlap-area) weights between the two Voronoi meshes: what the reference builds with xsphere + CDO
        pool_m, unpool_m = sphere.conservative_pool_matrices(fine.coords, coarse.coords)
        torch.manual_seed(10)
        self.conv_fine = ConvCheb(wl["fin"], wl["fout"], wl["K"], laplacian=prepare_torch_laplacian(fine.L, lmax=1.95))
        self.pool = GeneralAvgPool(pool_m)
        self.conv_coarse = ConvCheb(wl["fout"], wl["fout"], wl["K"],
                                    laplacian=prepare_torch_laplacian(coarse.L, lmax=1.95))
        self.unpool = GeneralAvgUnpool(unpool_m)

    def forward(self, x):
        y = self.conv_fine(x)
        if y.requires_grad:   # y has two consumers: its pooling's backward adds the other one's gradient in its epilogue
            y, (z, idx) = self.pool.forward_fork(y)
        else:
            z, idx = self.pool(y)
        return self.unpool.forward_add(self.conv_coarse(z), y)      # y + unpool(...): the add rides in the product's epilogue


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


def _moved(traffic_key, leg, sec):
    """HBM bytes the counters saw for a one-launch leg (profiles/spmm_traffic.json) and the fraction of 8 TB/s they are in `sec`."""
    m = _traffic(traffic_key, leg) if leg else None
    if m is None:
        return {}
    return {"bytes_moved": int(m["hbm_bytes_per_call"]), "frac_counter": round(m["hbm_bytes_per_call"] / sec / 1e9 / HBM_PEAK_GBS, 4)}


def _traffic(traffic_key, leg):
    """Committed PMC measurement of one leg (profiles/spmm_traffic.json, tools/pmc_traffic.sh): HBM bytes per call of the
    leg and the launches it made, or None."""
    tpath = os.path.join(REPO, "profiles", "spmm_traffic.json")
    if not traffic_key or not os.path.exists(tpath):
        return None
    try:
        return json.load(open(tpath)).get(traffic_key, {}).get(leg)
    except Exception:
        return None


def _kernel_base(name):
    """Identifier of a kernel from a rocprofv3 name ("void (anonymous namespace)::spmm1_dma_kernel<false, 3>(Args)") or from a
    launch-site expression of the library ("(spmm1_dma_kernel<BF16, NST>)")."""
    n = name.strip()
    if n.startswith("void "):
        n = n[5:]
    n = n.lstrip("(").strip()
    cut = len(n)
    for ch in "<(":
        i = n.find(ch, 1 if n.startswith("(") else 0)
        if i > 0:
            cut = min(cut, i)
    n = n[:cut]
    while n.startswith("(anonymous namespace)::"):
        n = n[len("(anonymous namespace)::"):]
    return n.split("::")[-1].strip(" )")


def trace_steps(step_fn, n_steps, warm=5, capacity=1 << 15):
    """ORDER, ROLE and eager durations of the kernels of a step (include/dsw_hip.h: dsw_trace_begin / dsw_trace_end): while
    `step_fn` - the very step the timed region replays - runs eagerly, the library launches each of its kernels with a start /
    stop event pair attached to the dispatch and notes which role of which entry-point call the kernel belongs to.
    Returns (roles, sequence): roles = {(role, aux0, aux1, aux2): {calls_per_step, avg_us, ...}} from the EAGER launches
    (kernels launched eagerly carry heavier fences and run ~10 % longer than the same kernels replayed from a graph);
    sequence = the library kernels of ONE step in launch order: [(call index within the step, role key, kernel base name)] -
    what graph_kernel_durations() folds a rocprofv3 trace of the replayed graph with."""
    from dsw_amd import _native

    for _ in range(warm):
        step_fn()
    torch.cuda.synchronize()
    per_step = []
    with _native.LaunchTrace(capacity) as tr:
        for _ in range(n_steps):
            step_fn()
        torch.cuda.synchronize()
    kernels = tr.kernels
    # split into steps: every step launches the same kernel sequence
    L, rem = divmod(len(kernels), n_steps)
    if rem or L == 0:
        raise RuntimeError("launch trace: %d kernels over %d steps" % (len(kernels), n_steps))
    seq0 = None
    agg = {}
    for sidx in range(n_steps):
        chunk = kernels[sidx * L:(sidx + 1) * L]
        c0 = chunk[0][0]
        seq = [(c - c0, (role, a0, a1, a2), _kernel_base(name)) for c, role, a0, a1, a2, _us, name in chunk]
        if seq0 is None:
            seq0 = seq
        elif seq != seq0:
            raise RuntimeError("launch trace: the steps did not launch identical kernel sequences")
        calls = {}
        for (c, key, _b), rec in zip(seq, chunk):
            calls.setdefault((c, key), []).append(rec)
        for (c, key), recs in calls.items():
            e = agg.setdefault(key, {"us": [], "lus": [], "nk": len(recs), "name": max(recs, key=lambda r: r[5])[6]})
            e["us"].append(sum(r[5] for r in recs))
            e["lus"].append(max(r[5] for r in recs))
    out = {}
    for k, e in agg.items():
        v = sorted(e["us"])
        out[k] = {"calls_per_step": len(v) / n_steps, "avg_us": sum(v) / len(v), "median_us": v[len(v) // 2],
                  "us_per_step": sum(v) / n_steps, "kernels": e["nk"], "longest_kernel": e["name"],
                  "longest_us": sum(e["lus"]) / len(e["lus"]),
                  "timing": "EAGER launches (dsw_trace): ~10 % above the replayed graph's kernels"}
    return out, seq0


def graph_kernel_durations(seq, n_steps=40, timeout=150):
    """Durations of the step's kernels INSIDE the replayed HIP graph: this script is run once more as a child under
    `rocprofv3 --kernel-trace` (same arguments + --kernel-trace-child N: build, capture, warm up, replay N steps, exit), the
    trace is sorted by start time, and the library kernels of its last N steps - identified by the launch order `seq` the
    eager trace recorded, verified name by name - are averaged per position of the step.  Returns (roles, info) with
    roles = {(role, aux0, aux1, aux2): {calls_per_step, avg_us, us_per_step, kernels}} or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None:
        return None, "rocprofv3 not found"
    if any(k.startswith("ROCP") or k.startswith("ROCPROFILER") for k in os.environ):
        return None, "already running under a profiler"
    tmp = tempfile.mkdtemp(prefix="dsw_kt_", dir="/tmp")
    try:
        argv = [a for a in sys.argv[1:]]
        cmd = ["timeout", "-k", "10", str(int(timeout)), rocprof, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "kt",
               "--", sys.executable, os.path.abspath(__file__), *argv, "--kernel-trace-child", str(n_steps),
               "--no-roofline", "--no-cpu-baseline"]
        env = dict(os.environ, TMPDIR="/tmp")
        proc = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if not files:
            return None, "rocprofv3 left no kernel trace (exit code %s): %s" % (proc.returncode, (proc.stderr or "")[-300:])
        rows = []
        for f in files:
            with open(f, newline="") as fh:
                rd = csv.DictReader(fh)
                keys = {k.lower(): k for k in (rd.fieldnames or [])}
                kn, ks, ke = keys.get("kernel_name"), keys.get("start_timestamp"), keys.get("end_timestamp")
                if not (kn and ks and ke):
                    return None, "unexpected columns in the kernel trace: %s" % (rd.fieldnames,)
                for r in rd:
                    rows.append((int(r[ks]), int(r[ke]), r[kn]))
        rows.sort()
        bases = {b for _c, _k, b in seq}
        lib = [(i, _kernel_base(nm)) for i, (_s, _e, nm) in enumerate(rows)]
        lib = [(i, b) for i, b in lib if b in bases]
        L = len(seq)
        if len(lib) < n_steps * L:
            return None, "the trace holds %d library kernels, expected at least %d" % (len(lib), n_steps * L)
        tail = lib[-n_steps * L:]
        want = [b for _c, _k, b in seq]
        for st in range(n_steps):
            got = [b for _i, b in tail[st * L:(st + 1) * L]]
            if got != want:
                bad = next(j for j in range(L) if got[j] != want[j])
                return None, "kernel order of the replayed graph differs from the eager trace at position %d (%s vs %s)" % (
                    bad, got[bad], want[bad])
        pos_us = [0.0] * L
        for st in range(n_steps):
            for j in range(L):
                s0, e0, _nm = rows[tail[st * L + j][0]]
                pos_us[j] += (e0 - s0) * 1e-3 / n_steps
        calls = {}
        for j, (c, key, _b) in enumerate(seq):
            calls.setdefault((c, key), []).append(pos_us[j])
        roles = {}
        for (c, key), durs in calls.items():
            e = roles.setdefault(key, {"sum": 0.0, "lsum": 0.0, "n": 0, "nk": len(durs)})
            e["sum"] += sum(durs)
            e["lsum"] += max(durs)
            e["n"] += 1
        out = {k: {"calls_per_step": float(e["n"]), "avg_us": e["sum"] / e["n"], "us_per_step": e["sum"], "kernels": e["nk"],
                   "longest_us": e["lsum"] / e["n"],
                   "timing": "in-graph kernel durations (rocprofv3 kernel trace of the replayed step graph, %d steps)" % n_steps}
               for k, e in roles.items()}
        # everything the graph ran in those steps (torch kernels included), and the wall time they spanned
        first, last = tail[0][0], len(rows) - 1
        all_us = sum((rows[i][1] - rows[i][0]) for i in range(first, last + 1)) * 1e-3 / n_steps
        span_us = (rows[last][1] - rows[first][0]) * 1e-3 / n_steps
        other = {}
        libset = {i for i, _b in tail}
        for i in range(first, last + 1):
            if i not in libset:
                b = _kernel_base(rows[i][2])
                o = other.setdefault(b, [0, 0.0])
                o[0] += 1
                o[1] += (rows[i][1] - rows[i][0]) * 1e-3
        info = {"steps": n_steps, "library_kernels_per_step": L, "all_kernels_us_per_step": round(all_us, 2),
                "span_us_per_step": round(span_us, 2),
                "other_kernels": sorted(({"kernel": b, "calls_per_step": round(c / n_steps, 2), "us_per_step": round(t / n_steps, 2)}
                                         for b, (c, t) in other.items()), key=lambda d: -d["us_per_step"])[:8]}
        return out, info
    except Exception as exc:  # noqa: BLE001 - any failure: the caller keeps the eager durations and says so
        return None, "%s: %s" % (type(exc).__name__, exc)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_leg(layer, x, steps, warmup, traffic_key=None, pmc_leg=None, traced=None, ms_per_step=None):
    """The SpMM recurrence launches of the workload's dominant ConvCheb layer against the HBM roofline.  Durations of
    everything a timed step launches come from `traced` (trace_steps: in-step HIP events); launches the step does NOT make
    (the forward recurrence alone at the north-star shape, the stand-alone channel mix) are timed as isolated legs and
    say so."""
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

    def bwdd():  # the whole backward as the step runs it where it is ONE launch in the dual form (no basis planes, T = NULL)
        Fo = layer.out_channels
        nb = int(lib.dsw_cheb_bwd_workspace_bytes(B, V, C, Fo, K, dcode))
        if not hasattr(bwdd, "t"):
            bwdd.t = (torch.empty(nb, dtype=torch.uint8, device=x.device), torch.randn(B, V, Fo, dtype=x.dtype, device=x.device),
                      torch.empty_like(x), torch.empty_like(layer.weight), torch.empty(Fo, dtype=x.dtype, device=x.device))
        ws, dy, dx, dw, db = bwdd.t
        rc = lib.dsw_cheb_bwd(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, x.data_ptr(), None,
                              layer.weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb,
                              B, C, Fo, K, dcode, st, ppt)
        assert rc == 0, rc

    def fwd1l():  # the forward of such a layer: one launch, no basis stores
        from dsw_amd import functional as F2

        F2._backend_for(x).cheb_fwd(op, x, layer.weight.detach(), None if layer.bias is None else layer.bias.detach(), keep_basis=False)

    if pmc_leg in ("fwd", "adj", "bwdd", "fwd1l"):     # profiling aid: this leg only
        fn = {"fwd": fwd, "adj": adj, "bwdd": bwdd, "fwd1l": fwd1l}[pmc_leg]
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return {"pmc_leg": pmc_leg, "calls": steps}
    for _ in range(warmup):
        fwd()
        adj()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):   # ISOLATED leg: median of three HIP-event regions of `steps` back-to-back calls (seconds per call)
        reps = []
        for _ in range(3):
            t0.record(stream)
            for _ in range(steps):
                fn()
            t1.record(stream)
            torch.cuda.synchronize()
            reps.append(t0.elapsed_time(t1) * 1e-3 / steps)
        return sorted(reps)[1]

    pairs = (K - 1 + 1) // 2                # launches of a pairwise-fused recurrence

    def n_launches(plan):                   # launches of one recurrence: plain hops, staged hops (hops == 1) or fused pairs
        return (K - 1) if (plan is None or plan.hops == 1) else pairs

    n_fwd, n_adj = n_launches(_k1), n_launches(_k2)
    fwd_b, bwd_b = spmm_algorithmic_bytes(E, Lb, K)
    Fout = layer.out_channels
    N = B * V
    mf = bool(lib.dsw_cheb_mix_first(C, Fout, K))
    ppf, _k3 = F_._plan_ptr(op, x, Fout if mf else C) if K > 2 else (None, None)
    fwd_path = int(lib.dsw_cheb_fwd_path(ppf, C, Fout, K, dcode))
    path_names = {0: "plain hops + GEMM", 1: "fused hop pairs + GEMM", 2: "staged hops + GEMM", 3: "one launch (hops + channel mix)",
                  4: "mix-first (recurrence on the output channels)", 5: "staged hop 1, then hop 2 + channel mix in one launch"}
    fwd_rec_in_step = fwd_path in (0, 1, 2)
    traced = traced or {}
    IN = next((v["timing"] for v in traced.values() if "timing" in v), "in-step kernel durations (dsw_trace)")
    ISO = "ISOLATED leg: median of 3 regions of %d back-to-back calls" % steps

    def tr(role, a0, a1, a2):
        return traced.get((role, a0, a1, a2))

    # ---- in-step durations of this layer's roles (None where the step does not run the role / nothing was traced) -------
    t_one = tr("fwd_one_launch", V, C, Fout)
    t_h2m = tr("fwd_hop2_mix", V, C, Fout)             # forward path 5: hop 2 + channel mix in one launch ...
    t_bfwd = tr("basis_fwd", V, C, 2 if t_h2m is not None else K)     # ... behind the staged hop 1 (traced as a K = 2 basis)
    t_mix = tr("mix_fwd", N, K * C, Fout)
    t_zmix, t_clen = tr("zmix", V, C, Fout), tr("clenshaw_fwd", V, Fout, K)
    t_gemm = tr("bwd_gemm_fused", V, C, Fout)
    t_dgrad, t_wgrad = tr("bwd_dgrad", V, C, Fout), tr("bwd_wgrad", V, C, Fout)
    t_adj = tr("basis_adj", V, C, K)
    t_bwdf = None                                      # (the mix-first one-launch dX was retired in round 6)
    t_dual = tr("bwd_dual", V, C, Fout)                # whole backward in one launch in the dual form (dsw_bwd3d.hip)
    have_trace = t_adj is not None or t_bwdf is not None or t_dual is not None or mf

    # forward recurrence alone: in the step where the step launches it, otherwise an isolated leg (north-star gate)
    fwd_s_iso = timed(fwd)
    fwd_s = t_bfwd["avg_us"] * 1e-6 if (fwd_rec_in_step and t_bfwd is not None and t_h2m is None) else fwd_s_iso
    adj_s_iso = None
    if t_adj is not None:
        adj_s = t_adj["avg_us"] * 1e-6
    elif t_bwdf is not None or t_dual is not None:
        adj_s = None                 # the adjoint recurrence lives inside the one-launch backward: no launch of its own
    else:
        adj_s = adj_s_iso = timed(adj)
    out_mfma = mfma_leg(layer, x, T, steps, in_step=t_mix)

    yb_bytes = N * Fout * es
    mm_peak = MFMA_PEAK_TFLOPS["bf16" if x.dtype == torch.bfloat16 else "f32"]
    mix_flops = 2.0 * N * C * K * Fout          # one contraction over (k, f): forward mix, dgrad or wgrad
    hop1_b = spmm_algorithmic_bytes(E, Lb, 2)[0]

    # ---- one entry per role the step launches.  SURVEY 8(d) splits a ConvCheb layer's work in two: the SpMM recurrence is
    # priced in BYTES against HBM (`bytes_8d`: [2 + 3 (K - 2)] E + (K - 1) Lb forward, [3 + 4 (K - 2)] E + (K - 1) Lb adjoint),
    # the channel-mix contractions in FLOPS against the matrix peak of the storage dtype.  A fused launch carries both: its
    # recurrence bytes over its in-graph time is `hbm_frac`, its flops over the same time `mfma_frac` - GEMM operand bytes of
    # an unfused design are never added to the byte count (no entry can exceed 1 by construction of the count).
    # `frac` repeats the fraction of the entry's `bound`.
    def entry(role, kernel, t, bound, rec_b=None, flops=None, compulsory=None, leg=None, gemm_bytes=None, n_launch=None):
        # `us` = everything the role launches per call (a fused launch + its small reduce; the K - 1 hops of a recurrence);
        # fractions are over that time.  `n_launch` = launches that carry the recurrence; `kernel_us` = the longest kernel alone
        sec = t["avg_us"] * 1e-6
        d = {"role": role, "kernel": kernel, "us": round(sec * 1e6, 2), "calls": t["calls_per_step"], "bound": bound,
             "n_launch": int(n_launch if n_launch is not None else t["kernels"])}
        if t["kernels"] > 1 and t.get("longest_us"):
            d["kernel_us"] = round(t["longest_us"], 2)
        if rec_b is not None:
            d["bytes_8d"] = int(rec_b)
            d["hbm_frac"] = round(rec_b / sec / 1e9 / HBM_PEAK_GBS, 4)
        if flops is not None:
            d["flops"] = flops
            d["TFLOPs"] = round(flops / sec / 1e12, 1)
            d["mfma_frac"] = round(flops / sec / 1e12 / mm_peak, 4)
        if gemm_bytes is not None:      # operand + result bytes of a stand-alone GEMM pass (what it streams; context only)
            d["gemm_bytes"] = int(gemm_bytes)
            d["gemm_GBs"] = round(gemm_bytes / sec / 1e9, 1)
        if compulsory is not None:
            d["compulsory_bytes"] = int(compulsory)
            d["frac_compulsory"] = round(compulsory / sec / 1e9 / HBM_PEAK_GBS, 4)
        d.update(_moved(traffic_key, leg, sec))
        d["frac"] = d["hbm_frac"] if bound == "hbm" else d["mfma_frac"]
        return d

    in_step = []
    if t_one is not None:        # (a backward in the dual form needs no basis planes: the forward then stores Y only)
        in_step.append(entry("forward: both hops + channel mix + bias in ONE launch", "cheb3_fwd_fused", t_one, "hbm", fwd_b, mix_flops,
                             E + yb_bytes if t_dual is not None else K * E + yb_bytes, "fwd1l" if t_dual is not None else None, n_launch=1))
    elif t_h2m is not None:
        if t_bfwd is not None:
            in_step.append(entry("forward hop 1 (staged launch, T1 = L X)", "spmm1_dma", t_bfwd, "hbm", hop1_b, None, 2 * E))
        in_step.append(entry("forward hop 2 + channel mix + bias in ONE launch", "cheb3_hop2mix", t_h2m, "hbm", fwd_b - hop1_b,
                             mix_flops, 2 * E + E + yb_bytes, n_launch=1))
    else:
        if t_bfwd is not None:
            in_step.append(entry("forward recurrence (basis launches)", "spmm2_fused / spmm1_dma / spmm_csr", t_bfwd, "hbm", fwd_b, None,
                                 K * E, "fwd"))
        if t_mix is not None:
            in_step.append(entry("forward channel mix (GEMM + bias)", "ts_gemm_x3 / ts_gemm_x3s / ts_gemm", t_mix, "mfma", None, mix_flops,
                                 gemm_bytes=K * E + yb_bytes))
        if t_zmix is not None:
            in_step.append(entry("forward plane GEMM of a mix-first layer", "ts_gemm* / narrow_*", t_zmix, "mfma", None, mix_flops,
                                 gemm_bytes=E + K * yb_bytes))
        if t_clen is not None:
            in_step.append(entry("forward Clenshaw recurrence on the output channels", "spmm* hops", t_clen, "hbm",
                                 spmm_algorithmic_bytes(yb_bytes, Lb, K)[1]))
    if t_gemm is not None:
        in_step.append(entry("backward GEMM pass (dW partials + db + dgrad planes, + reduce)", "cheb_wgrad_x3<FUSE> + cheb_wgrad_reduce",
                             t_gemm, "mfma", None, 2 * mix_flops, gemm_bytes=K * E + yb_bytes + K * E))
    if t_dgrad is not None:
        in_step.append(entry("backward dgrad GEMM (planes G_k = dY W_k^T)", "ts_gemm_x3 / ts_gemm_x3s", t_dgrad, "mfma", None, mix_flops,
                             gemm_bytes=yb_bytes + K * E))
    if t_wgrad is not None:
        in_step.append(entry("backward wgrad (dW, db)", "cheb_wgrad_x3 / cheb_wgrad_bf16 + reduce", t_wgrad, "mfma", None, mix_flops,
                             gemm_bytes=K * E + yb_bytes))
    if t_bwdf is not None:
        in_step.append(entry("backward dgrad + adjoint recurrence in ONE launch (dY -> dX)", "cheb3_bwd_fused", t_bwdf, "hbm", bwd_b,
                             mix_flops, yb_bytes + E, n_launch=1))
    if t_dual is not None:
        in_step.append(entry("whole backward in ONE launch, dual form (X, dY -> dX, dW, db; + the partial reduce)",
                             "cheb3_bwd_dual + cheb_wgrad_reduce", t_dual, "hbm", bwd_b, 2 * mix_flops, E + yb_bytes + E, "bwdd", n_launch=1))
    if t_adj is not None:
        in_step.append(entry("adjoint recurrence (basis_adj launches)", "spmm_csr x%d" % (K - 1) if ppt is None else
                             "spmm1_dma / spmm1_staged x%d" % (K - 1) if _k2.hops == 1 else "spmm2_fused adjoint pair(s)",
                             t_adj, "hbm", bwd_b, None, None, "adj"))
    step_sum_us = sum(d["us"] * d["calls"] for d in in_step)
    for d in in_step:
        d["share"] = round(d["us"] * d["calls"] / max(step_sum_us, 1e-9), 3)

    # ---- headline: the DOMINANT launch among those that carry SpMM recurrence work in the step (SURVEY 8d bytes of its
    # recurrence / its in-graph duration); the whole step's recurrence bytes over the whole step's time beside it
    rec = [d for d in in_step if "bytes_8d" in d]
    if rec:
        dom = max(rec, key=lambda d: d["us"] * d["calls"])
        # a fused launch is ONE kernel (the small partial reduce behind it carries no recurrence work): its own duration, the
        # figure `rocprofv3 --stats` reports for it; a K - 1 hop recurrence is its launches together
        one = dom["n_launch"] == 1 and "kernel_us" in dom
        dom_sec, dom_b, dom_n = (dom["kernel_us"] if one else dom["us"]) * 1e-6, dom["bytes_8d"], dom["n_launch"]
        timing = IN
    else:                        # nothing traced: isolated legs of the two recurrences
        adj_s = adj_s if adj_s is not None else timed(adj)
        dom = {"role": "adjoint recurrence (ISOLATED leg)", "kernel": "basis_adj launches"}
        dom_sec, dom_b, dom_n = adj_s, bwd_b, n_adj
        timing = ISO
    m_dom = dom.get("bytes_moved")
    rec_bytes_step = sum(d["bytes_8d"] * d["calls"] for d in rec)
    flops_step = sum(d.get("flops", 0.0) * d["calls"] for d in in_step)
    out = {
        "bound": "hbm",
        "kernel": "%s [%s]" % (dom["kernel"], dom["role"]),
        "achieved": round(dom_b / dom_sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(dom_b / dom_sec / 1e9 / HBM_PEAK_GBS, 4),
        "traffic": None if m_dom is None else int(m_dom / dom_n),
        "frac_counter": None if m_dom is None else round(m_dom / dom_sec / 1e9 / HBM_PEAK_GBS, 4),
        "bytes_per_launch": int(dom_b / dom_n), "avg_launch_us": round(dom_sec / dom_n * 1e6, 2), "launches": dom_n,
        "launch_timing": timing,
        "definition": "SURVEY 8(d) recurrence bytes of the dominant launch / its in-graph duration / 8 TB/s; GEMM flops -> mfma_*",
    }
    if "flops" in dom:
        out.update({"flops_per_launch": dom["flops"], "mfma_TFLOPs": round(dom["flops"] / dom_sec / 1e12, 1),
                    "mfma_peak_TFLOPs": mm_peak, "mfma_frac": round(dom["flops"] / dom_sec / 1e12 / mm_peak, 4)})
    if "compulsory_bytes" in dom:
        out.update({"compulsory_bytes": dom["compulsory_bytes"],
                    "frac_compulsory": round(dom["compulsory_bytes"] / dom_sec / 1e9 / HBM_PEAK_GBS, 4)})
    if ms_per_step is not None and rec:
        out["step_bytes_8d"] = int(rec_bytes_step)
        out["step_frac"] = round(rec_bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["step_flops"] = flops_step
        out["step_mfma_frac"] = round(flops_step / (ms_per_step * 1e-3) / 1e12 / mm_peak, 4)
    # second launch of the step by time, flat (the driver's parser keeps scalars of this object only)
    others = sorted((d for d in in_step if d is not dom), key=lambda d: -d["us"] * d["calls"])
    if others:
        o = others[0]
        out.update({"second_kernel": o["kernel"], "second_us": o["us"], "second_bound": o["bound"], "second_frac": o["frac"],
                    "second_hbm_frac": o.get("hbm_frac"), "second_mfma_frac": o.get("mfma_frac"),
                    "second_frac_counter": o.get("frac_counter")})
    # north-star gate: the K = 3 forward recurrence alone (in the step where the step launches it, else an isolated leg)
    out.update({"fwd_recurrence_us": round(fwd_s * 1e6, 2), "fwd_recurrence_frac": round(fwd_b / fwd_s / 1e9 / HBM_PEAK_GBS, 4),
                "fwd_recurrence_in_step": bool(fwd_rec_in_step and t_bfwd is not None and t_h2m is None),
                "forward_path": path_names.get(fwd_path),
                "in_step_sum_us": round(step_sum_us, 2)})
    if ms_per_step is not None and in_step:
        out["in_step_sum_vs_ms_per_step"] = round(step_sum_us / (ms_per_step * 1e3), 4)
    out["in_step"] = in_step

    kname = lambda plan, trn: ("spmm_csr_rowsplit x%d" % (K - 1) if plan is None else
                               "spmm1_dma / spmm1_staged x%d (one staged launch per hop)" % (K - 1) if plan.hops == 1 else
                               "spmm2_fused x%d (hops pairwise in one launch)" % pairs) + (" on L^T" if trn else "")

    def rec_entry(leg, plan, trn, n, sec, nbytes, how):
        m = _traffic(traffic_key, leg)
        return {"kernels": kname(plan, trn), "launches": n, "us": round(sec * 1e6, 2), "bytes_8d": int(nbytes),
                "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4), "timing": how,
                "fused": bool(plan is not None and plan.hops != 1),
                "bytes_moved": None if m is None else int(m["hbm_bytes_per_call"]),
                "frac_counter": None if m is None else round(m["hbm_bytes_per_call"] / sec / 1e9 / HBM_PEAK_GBS, 4)}

    # detail (side file / --detail inline): the recurrences as stand-alone launches, the stand-alone channel mix
    detail = {
        "forward_recurrence": dict(rec_entry("fwd", _k1, False, n_fwd, fwd_s, fwd_b,
                                             IN if (fwd_rec_in_step and t_bfwd is not None) else ISO), in_step=fwd_rec_in_step,
                                   note="north-star gate: >= 0.60 of 8 TB/s on the K = 3 forward recurrence"),
        "adjoint_recurrence": (None if adj_s is None else
                               dict(rec_entry("adj", _k2, True, n_adj, adj_s, bwd_b, IN if adj_s_iso is None else ISO),
                                    in_step=t_adj is not None)),
        "mfma": out_mfma,
        "traffic_source": "profiles/spmm_traffic.json: HBM bytes per launch from separate rocprofv3 --pmc passes over exactly these "
                          "launches (2 x FETCH_SIZE + WRITE_SIZE, tools/pmc_traffic.sh); a committed measurement, not read in this run",
        "E_bytes": int(E), "Lb_bytes": int(Lb),
    }
    return out, detail


def pooling_leg(model, run_forward, steps, traffic_key=None, pmc_leg=None, traced=None):
    """The interpolation-pooling products of the workload (RemapBlock: pool / unpool and their transposed backward), each
    with its own shape: bytes = input rows + output rows + operator (int32 column + fp32 value per entry + row pointers),
    time = HIP events around `steps` back-to-back calls.  north_star names this kernel; the U-Net has 4 distinct products
    (x2 directions), workload c5 the cross-sampling pair."""
    from dsw_amd import functional as F_
    from modules.layers import RemapBlock

    seen = []
    hooks = []
    for name, mod in model.named_modules():
        if isinstance(mod, RemapBlock):
            def hook(m, args, out, name=name):
                seen.append((name, m, tuple(args[0].shape), args[0].dtype))
            hooks.append(mod.register_forward_hook(hook))
    with torch.no_grad():
        run_forward()
    for h in hooks:
        h.remove()
    stream = torch.cuda.current_stream()
    entries, calls = [], []
    for name, mod, shape, dt in seen:
        op = F_.get_operator(mod.remap_matrix)
        opt = op.transpose()
        B, Vs, C = shape
        xs = torch.randn(shape, device=mod.remap_matrix.device, dtype=dt)
        dy = torch.randn((B, op.shape[0], C), device=xs.device, dtype=dt)
        es = xs.element_size()
        for what, o, t in (("forward", op, xs), ("backward (transposed matrix)", opt, dy)):
            calls.append((name, what, o, t))
            nbytes = (t.numel() + B * o.shape[0] * C) * es + o.nnz * 8 + 4 * (o.shape[0] + 1)
            entries.append({"layer": name, "product": what, "rows_out": o.shape[0], "rows_in": o.shape[1], "channels": C,
                            "nnz_per_row": round(o.nnz / o.shape[0], 1), "algorithmic_bytes": int(nbytes)})
    if pmc_leg == "pool":
        for _ in range(steps):
            for _n, _w, o, t in calls:
                F_.sparse_remap(o, t)
        torch.cuda.synchronize()
        return {"pmc_leg": "pool", "calls": steps}
    tot_s = tot_b = 0.0
    traced = traced or {}
    for e, (_n, _w, o, t) in zip(entries, calls):
        hit = traced.get(("spmm", o.shape[0], o.shape[1], t.shape[2]))
        if hit is not None:
            # in-step: the product timed inside the running step (products of equal shape - a pooling and the transposed
            # unpooling of the same level - share one entry of the trace: their mean)
            sec = hit["avg_us"] * 1e-6
            e["timing"] = hit.get("timing", "in-step") + "; mean over the step's products of this shape"
        else:
            for _ in range(3):
                F_.sparse_remap(o, t)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(stream)
            for _ in range(steps):
                F_.sparse_remap(o, t)
            t1.record(stream)
            torch.cuda.synchronize()
            sec = t0.elapsed_time(t1) * 1e-3 / steps
            e["timing"] = "ISOLATED leg: %d back-to-back calls" % steps
        e["us"] = round(sec * 1e6, 2)
        e["frac"] = round(e["algorithmic_bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4)
        tot_s += sec
        tot_b += e["algorithmic_bytes"]
    m = _traffic(traffic_key, "pool")
    return {"kernel": "RemapBlock products (dsw_spmm_csr: remap_* kernels for hierarchical / staged cross-sampling matrices, "
                      "spmm_csr_rowsplit otherwise)",
            "bound": "hbm", "products": entries, "us_per_step": round(tot_s * 1e6, 2), "algorithmic_bytes": int(tot_b),
            "frac": round(tot_b / max(tot_s, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
            "bytes_moved": None if m is None else int(m["hbm_bytes_per_call"]),
            "frac_counter": None if m is None else round(m["hbm_bytes_per_call"] / max(tot_s, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "tensors of these products are 2-50 MB: they live in the 256 MB Infinity Cache between the layers, so the "
                    "counter bytes can be far below the algorithmic ones"}


MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # dense matrix peaks of MI355X (MI355X_MICROARCH.md)


def mfma_leg(layer, x, T, steps, in_step=None):
    """The channel-mix GEMM of the forward pass (layers.py:171-178): algorithmic flops 2 N Fin K Fout over the mean
    launch time (HIP events on the launch stream) against the dense MFMA peak of the storage dtype.  fp32 layers run
    on the bf16 matrix pipe with 3-way split operands (6 MFMA terms per product): the utilisation of that pipe is the
    second figure."""
    lib = _native_lib()
    K = layer.kernel_size
    B, V, Fin = x.shape
    Fout = layer.out_channels
    N = B * V
    bf16 = x.dtype == torch.bfloat16
    y = torch.empty((B, V, Fout), dtype=x.dtype, device=x.device)
    st = torch.cuda.current_stream().cuda_stream
    bias = layer.bias

    def mix():
        rc = lib.dsw_cheb_mix_fwd(x.data_ptr(), T.data_ptr() if K > 1 else None, layer.weight.data_ptr(),
                                  None if bias is None else bias.data_ptr(), y.data_ptr(), N, Fin, Fout, K,
                                  1 if bf16 else 0, st)
        assert rc == 0

    for _ in range(3):
        mix()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(stream)
    for _ in range(steps):
        mix()
    t1.record(stream)
    torch.cuda.synchronize()
    avg_s = t0.elapsed_time(t1) * 1e-3 / steps
    how = "ISOLATED leg (the step does not launch the stand-alone channel mix for this layer): %d back-to-back calls" % steps
    if in_step is not None:      # the step launches this GEMM: its in-step duration
        avg_s = in_step["avg_us"] * 1e-6
        how = in_step.get("timing", "in-step kernel durations")
    flops = 2.0 * N * Fin * K * Fout
    peak = MFMA_PEAK_TFLOPS["bf16" if bf16 else "f32"]
    ach = flops / avg_s / 1e12
    hbm = (K * N * Fin + N * Fout) * x.element_size()
    out = {"kernel": "channel mix forward (ts_gemm%s)" % ("" if bf16 else "_x3"), "flops": flops,
           "avg_launch_us": round(avg_s * 1e6, 2), "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
           "frac": round(ach / peak, 4), "hbm_GBs": round(hbm / avg_s / 1e9, 1), "timing": how}
    if not bf16:
        out["bf16_pipe_frac"] = round(6 * ach / MFMA_PEAK_TFLOPS["bf16"], 4)
        out["note"] = ("fp32 product evaluated as 6 bf16 MFMA terms (3-way operand split); frac is against the fp32 "
                       "matrix peak, bf16_pipe_frac = 6 x flops against the 2.5 PFLOP/s pipe the MFMAs issue on; the "
                       "kernel streams K*E + N*Fout*4 bytes (hbm_GBs) and is HBM-side bound before either")
    return out


def cpu_baseline_leg(wl, lap, layer):
    """The oracle's torch restatement of the reference CPU path (torch.sparse.mm + matmul), fp32."""
    from oracle import cheb_oracle as orc

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

    med, n_timed, cores = _pick_threads_and_time(one)
    return {
        "value": B * V * wl["fin"] / med, "unit": "nodes*channels/s", "cores": torch.get_num_threads(),
        "host_cpus": cores,
        "kind": "port",
        "sample": "oracle conv_cheb torch restatement (sparse COO mm + matmul, fp32), full %s shape "
                  "[B=%d,V=%d,%d->%d,K=%d], median of %d fwd+bwd, %.1f ms; best of 8/16/32/64 threads - ATen's sparse "
                  "CPU kernels stop scaling (then regress) far below this host's core count" % (
                      "workload", B, V, wl["fin"], wl["fout"], wl["K"], n_timed, med * 1e3),
    }


def _pick_threads_and_time(one, max_probe_s=6.0, budget_s=20.0, reps=8):
    """ATen's sparse CPU kernels stop scaling (then regress) far below the core count of a GPU host: probe a few
    thread counts once each, keep the fastest, then take the median of up to `reps` runs inside the time budget."""
    cores = os.cpu_count() or 1
    best_thr, best_t = 1, float("inf")
    for thr in sorted({min(cores, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(thr)
        one()
        t = one()
        if t < best_t:
            best_thr, best_t = thr, t
        if t > max_probe_s:
            break
    torch.set_num_threads(best_thr)
    times = []
    budget = time.perf_counter() + budget_s
    while len(times) < reps and (time.perf_counter() < budget or len(times) < 2):
        times.append(one())
    return float(np.median(times)), len(times), cores


def cpu_baseline_unet(model, wl, V):
    """oracle/unet_oracle.py: the reference UNetSpherical's torch op sequence on CPU (sparse COO mm + matmul + relu +
    linear, autograd backward), on a bounded sample of the workload (2 of the 8 spheres of a batch)."""
    from oracle import unet_oracle

    sd = unet_oracle.leaf_state(model.state_dict())
    Bs = 2
    torch.manual_seed(1234)
    x = torch.randn(Bs, 3, V, 6)
    target = torch.randn(Bs, 1, V, 2)

    def one():
        t = time.perf_counter()
        unet_oracle.unet_fwd_bwd(sd, x, target)
        return time.perf_counter() - t

    med, n, cores = _pick_threads_and_time(one)
    return {
        "value": Bs * V * wl["fin"] / med, "unit": "nodes*channels/s", "cores": torch.get_num_threads(), "host_cpus": cores,
        "kind": "port",
        "sample": "oracle UNetSpherical torch restatement (fp32, same parameters and operators), %d of the %d spheres of a "
                  "batch, median of %d fwd+bwd, %.1f ms; ATen sparse kernels stop scaling beyond the thread count chosen "
                  "(best of 8/16/32/64)" % (Bs, wl["batch"], n, med * 1e3),
    }


def cpu_baseline_c5(model, wl, V):
    """The c5 block (ConvCheb on the equiangular graph -> interp pool -> ConvCheb -> unpool, residual add) as the oracle's
    torch op sequence on CPU, 2 of the 8 samples."""
    from oracle import cheb_oracle as orc

    f = lambda t: t.detach().float().cpu()
    lf, lc = f(model.conv_fine.laplacian).coalesce(), f(model.conv_coarse.laplacian).coalesce()
    pm, um = f(model.pool.remap_matrix).coalesce(), f(model.unpool.remap_matrix).coalesce()
    params = [f(p).requires_grad_(True) for p in (model.conv_fine.weight, model.conv_fine.bias,
                                                  model.conv_coarse.weight, model.conv_coarse.bias)]
    Bs = 2
    torch.manual_seed(1234)
    x = torch.randn(Bs, V, wl["fin"], requires_grad=True)
    gy = torch.randn(Bs, V, wl["fout"])

    def one():
        t = time.perf_counter()
        for p in params + [x]:
            p.grad = None
        y = orc.conv_cheb_layer_torch(lf, x, params[0], params[1])
        z = orc.remap_torch(pm, y)
        out = y + orc.remap_torch(um, orc.conv_cheb_layer_torch(lc, z, params[2], params[3]))
        out.backward(gy)
        return time.perf_counter() - t

    med, n, cores = _pick_threads_and_time(one)
    return {
        "value": Bs * V * wl["fin"] / med, "unit": "nodes*channels/s", "cores": torch.get_num_threads(), "host_cpus": cores,
        "kind": "port",
        "sample": "oracle torch restatement of the c5 block (fp32), %d of the %d samples, median of %d fwd+bwd, %.1f ms; "
                  "best of 8/16/32/64 threads (ATen sparse stops scaling)" % (Bs, wl["batch"], n, med * 1e3),
    }


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: run the N ranks under torch.distributed.run and relay rank 0's line.
    Attempts, each under a time limit and in its own process group (a hung attempt is killed as a whole):
    graph-captured collectives -> collectives after the replay -> eager launches."""
    import signal
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n and os.environ.get("DSW_DIST_BACKEND") != "gloo":
        sys.exit("bench.py: --gpus %d, but %d ROCm device(s) are visible (RCCL needs one device per rank; "
                 "DSW_DIST_BACKEND=gloo runs the N > 1 path with several ranks per device for tests)" % (n, have))
    limit = float(os.environ.get("DSW_BENCH_TIMEOUT", "420"))
    attempts = [("graph-captured collectives", {}),
                ("collectives after the graph replay", {"DSW_BENCH_COLLECTIVES": "eager"}),
                ("eager launches", {"DSW_BENCH_COLLECTIVES": "eager", "DSW_BENCH_NO_GRAPH": "1"})]
    if os.environ.get("DSW_BENCH_COLLECTIVES") == "eager":
        attempts = attempts[1:]
    import tempfile
    import time

    for i, (what, extra) in enumerate(attempts):
        for port_try in range(3):
            # the probe socket is closed before torchrun binds the port (unavoidable: torchrun takes a number, not a
            # socket): if somebody else grabs it in between, the rendezvous fails at once and a fresh port is tried
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
            sock.close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
            env = dict(os.environ, **extra)
            env["DSW_BENCH_LAUNCHER"] = "bench.py self-launch, attempt %d: %s" % (i + 1, what)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            t0 = time.time()
            with tempfile.TemporaryFile(mode="w+") as errf:
                proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=errf, text=True, start_new_session=True)
                try:
                    out, _ = proc.communicate(timeout=limit)
                    rc = proc.returncode
                except subprocess.TimeoutExpired:
                    os.killpg(proc.pid, signal.SIGKILL)
                    out, _ = proc.communicate()
                    rc = -9
                errf.seek(0)
                err = errf.read()
            sys.stderr.write(err)
            port_taken = rc != 0 and time.time() - t0 < 60 and ("address already in use" in err.lower() or "EADDRINUSE" in err)
            if not port_taken:
                break
            print("bench.py: port %d was taken before torchrun could bind it; trying another" % port, file=sys.stderr, flush=True)
        lines = [ln for ln in (out or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
        if rc == 0 and lines:
            print(lines[-1], flush=True)
            return 0
        print("bench.py: attempt %d (%s) failed with exit code %s%s" % (
            i + 1, what, rc, "" if i + 1 == len(attempts) else "; retrying: " + attempts[i + 1][0]), file=sys.stderr, flush=True)
    return 1


def roofline_target(args, model, x, B, V, device):
    """The SpMM recurrence the roofline entry is measured on: the workload's dominant ConvCheb layer (whole-model
    workloads: the layer with the most node-channels) with activations of its own shape, and the key of its committed
    counter measurement in profiles/spmm_traffic.json."""
    if args.workload == "unet":
        rl_layer, rl_what = model.conv1.convblock2.conv, "conv1.convblock2.conv (V=%d, 64->128)" % V
        rl_x = torch.randn(B, V, rl_layer.in_channels, device=device)
    elif args.workload == "c5":
        rl_layer, rl_what = model.conv_fine, "conv_fine (equiangular V=%d, 32->32)" % V
        rl_x = x.detach()
    else:
        rl_layer, rl_what, rl_x = model, None, x.detach()
    tkey = (args.workload if (rl_what is not None or args.knn == 8)
            else "ns_k20" if (args.workload == "ns" and args.knn == 20) else None)
    return rl_layer, rl_what, rl_x, tkey


def _next_rung():
    """Environment of the next, more conservative way to run the N > 1 step (the ladder of `launch_ranks`), or None at its end."""
    if os.environ.get("DSW_BENCH_NO_GRAPH") == "1":
        return None
    if os.environ.get("DSW_BENCH_COLLECTIVES") == "eager":
        return {"DSW_BENCH_COLLECTIVES": "eager", "DSW_BENCH_NO_GRAPH": "1"}
    return {"DSW_BENCH_COLLECTIVES": "eager"}


def _arm_watchdog(rank, world, limit=None, what="the run", retry=False):
    """A rank of an N > 1 run that is still alive after DSW_BENCH_TIMEOUT seconds (a collective that never completes)
    exits with an error instead of holding the node: the launcher (ours, or torchrun) then tears the job down.  Also
    used with a short limit around the FIRST replay of a graph that holds collectives."""
    import threading

    if limit is None:
        limit = float(os.environ.get("DSW_BENCH_TIMEOUT", "420")) * 0.9

    def bark():
        nxt = _next_rung()
        if retry and nxt is not None:
            # launched by somebody else's torchrun (the driver's scaling run): nobody will start a second attempt for us,
            # and a rank that exits takes the whole job down.  Every rank sits behind the same guard, so every rank
            # replaces its own process image with the next, more conservative rung (same RANK / WORLD_SIZE / MASTER_*,
            # a new process group behind its own store prefix - dsw_amd.parallel.init_from_env).
            print("bench.py: rank %d/%d: %s still not finished after %.0f s - re-executing with %s" % (
                rank, world, what, limit, nxt), file=sys.stderr, flush=True)
            env = dict(os.environ, **nxt)
            env["DSW_PG_ATTEMPT"] = str(int(os.environ.get("DSW_PG_ATTEMPT", "0") or 0) + 1)
            sys.stdout.flush()
            os.execve(sys.executable, [sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env)
        print("bench.py: rank %d/%d: %s still not finished after %.0f s - giving up" % (rank, world, what, limit),
              file=sys.stderr, flush=True)
        os._exit(124)

    t = threading.Timer(limit, bark)
    t.daemon = True
    t.start()
    return t


def main():
    global GRAPH_STEPS
    args = parse()
    # consecutive graph launches leave the GPU idle for ~40 us: the timed K steps should be whole replays of one graph
    for g in range(25, 4, -1):
        if args.steps % g == 0:
            GRAPH_STEPS = g
            break
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %s rank(s) (WORLD_SIZE)" % (args.gpus, env_world))
    if os.environ.get("DSW_BENCH_NO_GRAPH") == "1":
        args.no_graph = True
    from dsw_amd import _native
    from dsw_amd.parallel import GradBucket, init_from_env

    rank, world, local = init_from_env()
    assert world == args.gpus, (world, args.gpus)
    # ranks started by somebody else's launcher (the driver's torchrun) climb down the ladder in place (see `_arm_watchdog`);
    # under our own self-launch the launcher process does (launch_ranks)
    external = os.environ.get("WORLD_SIZE") is not None and not os.environ.get("DSW_BENCH_LAUNCHER")
    if world > 1:
        _arm_watchdog(rank, world, retry=external)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; a ROCm device is required"
    _native.load()
    local = local % torch.cuda.device_count()   # (gloo smoke runs of the N>1 path put two ranks on one device)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    wl = WORKLOADS[args.workload]
    dtype = torch.bfloat16 if wl["dtype"] == "bf16" else torch.float32
    V = wl["nlat"] * wl["nlon"] if args.workload == "c5" else 12 * wl["nside"] ** 2
    B = wl["batch"]
    if args.global_batch is not None:
        from dsw_amd.parallel import shard_batch

        lo, hi = shard_batch(args.global_batch, rank, world)   # contiguous slices, ragged tail, empty shards when B < world
        B = hi - lo
    torch.manual_seed(1234 + rank)

    if args.workload == "unet":
        if args.torch_cat:
            import modules.my_models_graph as _arch
            _arch.UNetSpherical.concat_in_place = False
        if args.resblock_mode:
            import modules.my_models_graph as _arch
            _arch.ResBlock.fuse_tail = True
        if args.plain_resblock:
            import modules.my_models_graph as _arch
            _arch.ResBlock.fuse_tail = False
        if args.resblock_mode:      # A/B: e = forward epilogue, f = fork (input gradient inside the residual map's dgrad)
            import modules.my_models_graph as _arch
            _arch.ResBlock.fuse_tail_epilogue = "e" in args.resblock_mode
            _arch.ResBlock.fuse_fork = "f" in args.resblock_mode
        model = make_unet(wl, args.knn if args.knn != 8 else 20, device)
        x = torch.randn(B, 3, V, 6, device=device)
        target = torch.randn(B, 1, V, 2, device=device)
        lap = None

        def step():
            zero_grads()
            loss = ((model(x) - target) ** 2).mean()
            loss.backward()
    else:
        if args.workload == "c5":
            model, lap = _C5Block(wl).to(device), None
        else:
            model, lap = make_layer(wl, args.knn, device, dtype)
        x = torch.randn(B, V, wl["fin"], device=device, dtype=dtype).requires_grad_(True)
        gy = torch.randn(B, V, wl["fout"], device=device, dtype=dtype)
        if args.zero_inputs:
            with torch.no_grad():
                x.zero_(); gy.zero_()

        def step():
            zero_grads()
            x.grad = None
            model(x).backward(gy)

    if args.pmc_leg is not None:       # profiling aid (tools/pmc_traffic.sh): one leg of the roofline measurement, no line
        rl_layer, _what, rl_x, _tkey = roofline_target(args, model, x, B, V, device)
        if args.pmc_leg == "pool":
            res = pooling_leg(model, lambda: model(x.detach()), 20, pmc_leg="pool") if args.workload in ("unet", "c5") else {}
        else:
            res = roofline_leg(rl_layer, rl_x, 20, 0, pmc_leg=args.pmc_leg)
        print(json.dumps(res), flush=True)
        return
    # N > 1: the parameter gradients live in one flat bucket (no pack / unpack copies around the collective); RCCL averages
    # it in place.  The exchange is part of the step: captured INTO the step graph when RCCL is the backend, issued
    # right after the replay otherwise (gloo, or DSW_BENCH_COLLECTIVES=eager).
    bucket = GradBucket(model.parameters(), overlap=False, attach=False)
    if args.grad_handling == "autograd" and bucket.active():
        sys.exit("bench.py: --grad-handling autograd has no N > 1 form (the exchange runs on the bucket)")
    use_bucket = bucket.active() or args.grad_handling == "bucket"
    if use_bucket:
        bucket.attach()
        bucket.direct_accumulation()     # the weight-gradient kernels add into the bucket: no autograd `add` per parameter
        zero_grads = bucket.zero
    else:
        def zero_grads():
            model.zero_grad(set_to_none=True)
    sync_grads = bucket.finish
    want_captured = bucket.active() and bucket.graph_capturable and os.environ.get("DSW_BENCH_COLLECTIVES") != "eager"

    def all_ranks_agree(ok):
        """True only if `ok` on every rank (a capture that failed on ONE rank must send everybody down the fallback)."""
        if world == 1 and not dist.is_initialized():
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    # The step is ~8 back-to-back kernels of 80-120 us each; launched eagerly from Python the GPU idles ~5 us between
    # them.  Capture fwd+bwd(+exchange) into a HIP graph and replay it (same kernels, same work, same buffers).  Falls
    # back to eager launches if capture is not possible.
    graph = None          # one step (without the exchange unless `captured_sync`)
    graph_multi = None    # GRAPH_STEPS whole steps
    captured_sync = False
    capture_note = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()          # builds the operator caches / tile plans outside the capture
                    sync_grads()    # (and RCCL's communicator, before anything is recorded)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **({"capture_error_mode": "thread_local"} if dist.is_initialized() else {})):
                step()
            torch.cuda.synchronize()
            graph = g
        except Exception as exc:  # noqa: BLE001 - any capture problem -> eager
            if rank == 0:
                print("bench: HIP graph capture unavailable (%s); running eagerly" % type(exc).__name__, file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
        if bucket.active() and not all_ranks_agree(graph is not None):
            graph = None
        if graph is not None and want_captured:
            # reference result of ONE step + exchange, issued the proven way (replay, then collectives on the stream)
            graph.replay()
            sync_grads()
            torch.cuda.synchronize()
            want = bucket.bucket.clone()
            g1 = gm = None
            try:
                # ("thread_local": RCCL's watchdog thread polls events while this thread records - an error under the default
                # global capture mode, and one that ends the process)
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                    step()
                    sync_grads()
                gm = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gm, capture_error_mode="thread_local"):
                    for _ in range(GRAPH_STEPS):
                        step()
                        sync_grads()
                torch.cuda.synchronize()
                ok = True
            except Exception as exc:  # noqa: BLE001
                capture_note = "capture of the collectives failed on rank %d: %s" % (rank, type(exc).__name__)
                ok = False
                torch.cuda.synchronize()
            if all_ranks_agree(ok):
                # every rank replays the same collectives in the same order: validate what the graph computes.  A replay
                # that never completes must not cost the whole time limit: 90 s, then this rank exits and the launcher
                # moves on to collectives outside the graph
                # (the guard stays armed over the agreement collective as well: a rank whose replay came back while another's
                # did not would otherwise wait there for the whole time limit - every rank must reach the next rung together)
                guard = _arm_watchdog(rank, world, float(os.environ.get("DSW_BENCH_GUARD_S", "90")),
                                      "the first replay of the graph-captured exchange", retry=external)
                if os.environ.get("DSW_BENCH_TEST_HANG") == "replay" and not os.environ.get("DSW_PG_ATTEMPT"):
                    time.sleep(1e6)          # tests: a replay that never returns (first incarnation only)
                g1.replay()
                torch.cuda.synchronize()
                same = torch.equal(bucket.bucket, want) or bool(
                    (bucket.bucket - want).abs().max() <= 1e-5 * want.abs().max().clamp_min(1e-30))
                agreed = all_ranks_agree(same)
                guard.cancel()
                if agreed:
                    graph, graph_multi, captured_sync = g1, (gm if args.steps >= GRAPH_STEPS else None), True
                else:
                    capture_note = "captured exchange did not reproduce the eager result; collectives stay outside the graph"
            if not captured_sync and rank == 0 and capture_note:
                print("bench: " + capture_note, file=sys.stderr)
        elif graph is not None and not bucket.active() and args.steps >= GRAPH_STEPS:
            # single GPU: consecutive graph launches leave the GPU idle for ~40 us (hipGraphLaunch latency), 8 % of
            # this step; a second graph holding GRAPH_STEPS whole steps amortises it
            try:
                gm = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gm):
                    for _ in range(GRAPH_STEPS):
                        step()
                torch.cuda.synchronize()
                graph_multi = gm
            except Exception:  # noqa: BLE001
                graph_multi = None
                torch.cuda.synchronize()

    def full_step():
        if graph is not None:
            graph.replay()
        else:
            step()
        if not captured_sync:
            sync_grads()

    def run_steps(n):
        """exactly n steps"""
        if graph_multi is not None:
            for _ in range(n // GRAPH_STEPS):
                graph_multi.replay()
            n = n % GRAPH_STEPS
        for _ in range(n):
            full_step()

    def timed_region(n):
        """Wall time of EXACTLY n steps, bracketed by barrier + synchronize on both sides, MAX over ranks."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(n)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # warm-up: the W steps asked for, plus one untimed pass of exactly the launch sequence the timed region uses
    # (the first replays of a freshly instantiated multi-step graph are slower than its steady state)
    run_steps(args.warmup)
    if args.kernel_trace_child > 0:      # traced child: exactly these replays are the tail of the kernel trace
        # steady state first (clocks and power management settle over hundreds of milliseconds of load: the first replays of
        # a fresh process run 15-20 % slower), as the timed regions of the parent are
        t_end = time.perf_counter() + args.min_timed_ms * 1e-3
        while time.perf_counter() < t_end:
            run_steps(args.steps)
            torch.cuda.synchronize()
        run_steps(args.kernel_trace_child)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    probe = timed_region(args.steps)
    # The contract times exactly K steps; K steps of this workload are a ~10 ms region at the driver's K = 20, short
    # enough for a single clock-ramp or scheduling hiccup to move the number by 10-20 %.  So the region is repeated
    # back to back - every repetition is exactly K steps inside its own barrier + synchronize bracket - until >= 1.5 s
    # (--min-timed-ms) have been timed, and the MEDIAN region is reported (all regions are summarised in "region_ms").
    # (1.5 s also keeps the GPU visibly busy for a sampling monitor; round 2's 150 ms fell between its samples.)
    n_regions = int(min(1000, max(3, -(-(args.min_timed_ms * 1e-3) // max(probe, 1e-6)))))
    regions = sorted(timed_region(args.steps) for _ in range(n_regions))
    elapsed = regions[len(regions) // 2]

    ms = elapsed / args.steps * 1e3
    # node-channels through the path per step, whole job
    units = (args.global_batch if args.global_batch is not None else B * world) * V * wl["fin"]
    out = {
        "metric": "HEALPix nodes*channels/sec through ConvCheb K=%d (fwd+bwd)" % wl["K"],
        "value": units / (elapsed / args.steps), "unit": "nodes*channels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak" if args.global_batch is None else "strong", "vs_baseline": None,
        "dtype": wl["dtype"], "data": "zeros (diagnostics)" if args.zero_inputs else "synthetic",
        "timing": "median of %d back-to-back timed regions of exactly %d steps each (barrier + synchronize around every region, max over ranks)" % (n_regions, args.steps),
        "region_ms": {"n": n_regions, "min": round(regions[0] * 1e3, 4), "median": round(elapsed * 1e3, 4),
                      "max": round(regions[-1] * 1e3, 4)},
        "config": {
            "workload": {
                "ns": "single ConvCheb layer, HEALPix nside=64 nested (V=49152), K=3, 32->64 ch, B=16/GPU, fp32 (north-star shape)",
                "c3": "single ConvCheb layer, HEALPix nside=64 nested, K=5, 64->128 ch, B=16/GPU, bf16 storage + fp32 accumulate",
                "unet": "UNetSpherical, HEALPix nside=32 nested, K=3, B=8/GPU, fp32, interp pooling (11 ConvCheb + 4 RemapBlock)",
                "c5": "equiangular 200x400 (V=80000, k=20, irregular degree) ConvCheb K=3 32ch + interp pooling to HEALPix nside=32 and back, B=8/GPU, fp32",
            }[args.workload],
            "knn": 20 if args.workload == "c5" else args.knn if args.workload != "unet" else (args.knn if args.knn != 8 else 20),
            "batch_per_gpu": B, "global_batch": args.global_batch if args.global_batch is not None else B * world, "nodes": V,
            "parallelism": "dp%d (batch shards, flat-bucket %s grad all-reduce)" % (
                world, "RCCL" if not dist.is_initialized() or dist.get_backend() == "nccl" else dist.get_backend()),
            # N = 1 and N > 1 do not run byte-identical steps: say which one this line timed
            "grad_handling": ("parameter gradients live in one flat bucket (one memset per step), the weight-gradient kernels "
                              "add into it, all-reduced in place" if bucket.active() else
                              "parameter gradients live in one flat bucket (one memset per step), the weight-gradient kernels "
                              "add into it (the N > 1 step; no exchange in a one-rank world)" if use_bucket else
                              "zero_grad(set_to_none=True) + autograd's gradient tensors (no exchange in a one-rank world)"),
            "launch": (("hip graph replay, %d steps per graph" % GRAPH_STEPS if graph_multi is not None
                        else "hip graph replay of one fwd+bwd" if graph is not None else "eager")
                       + ("" if not bucket.active() else
                          ", gradient all-reduce captured in the graph" if captured_sync else
                          ", gradient all-reduce issued after every replay")),
        },
    }
    if os.environ.get("DSW_BENCH_LAUNCHER"):
        out["config"]["launcher"] = os.environ["DSW_BENCH_LAUNCHER"]
    if os.environ.get("DSW_PG_ATTEMPT"):
        out["config"]["ladder"] = ("rung %d of the in-place fallback ladder (a guard fired in the rung before: the ranks re-executed "
                                   "themselves under the same launcher)" % (int(os.environ["DSW_PG_ATTEMPT"]) + 1))
    if os.environ.get("DSW_HIP_LIB"):
        out["config"]["DSW_HIP_LIB"] = os.environ["DSW_HIP_LIB"]   # an A/B build stands in for the product library
    if bucket.active():
        # the exchange alone (HIP events around bucket.finish() on the launch stream), and proof that it did its job:
        # ranks hold different data (seed 1234 + rank), so their gradients agree only if the all-reduce ran
        full_step()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record()
            sync_grads()
            b.record()
        torch.cuda.synchronize()
        ar = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        full_step()                      # a fresh, once-averaged gradient set for the comparison below
        torch.cuda.synchronize()
        mine = bucket.bucket.detach().clone()
        if dist.get_backend() != "nccl":
            mine = mine.cpu()            # gloo gathers host tensors
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        diff = max(float((t - gathered[0]).abs().max()) for t in gathered)
        out["allreduce_us"] = round(ar[len(ar) // 2], 1)
        out["allreduce"] = {"bytes": int(bucket.bucket.numel() * bucket.bucket.element_size()), "chunks": len(bucket.chunks),
                            "median_us": round(ar[len(ar) // 2], 1), "min_us": round(ar[0], 1),
                            "what": "bucket.finish() alone, eager, HIP events (in the timed steps it is %s)" % (
                                "a node of the step graph" if captured_sync else "issued after each graph replay"),
                            "note": capture_note}
        out["grad_sync"] = {"max_abs_diff_across_ranks": diff, "identical": diff == 0.0,
                            "grad_l2": float(mine.double().norm())}
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
    if rank == 0 and world == 1:
        # roofline: the SpMM recurrence of the workload's dominant ConvCheb layer; cpu_baseline: the oracle's torch
        # restatement of the SAME workload
        rl_layer, rl_what, rl_x, tkey = roofline_target(args, model, x, B, V, device)
        if not args.no_roofline:
            # per-role durations inside running steps: the step of the timed region, launched eagerly back to back with the
            # library's role markers on the launch stream (a graph replay cannot carry them)
            traced = seq = None
            trace_note = None
            try:
                traced, seq = trace_steps(lambda: (step(), sync_grads()), 5 if args.workload == "unet" else 10)
            except Exception as exc:  # noqa: BLE001 - the line then says its legs are isolated
                print("bench: launch trace unavailable (%s: %s); roofline legs are isolated timings" % (type(exc).__name__, exc),
                      file=sys.stderr)
            graph_info = None
            if traced is not None and graph is not None:
                n_child = 10 if args.workload == "unet" else 40
                n_child = max(GRAPH_STEPS, n_child // GRAPH_STEPS * GRAPH_STEPS) if graph_multi is not None else n_child
                in_graph, graph_info = graph_kernel_durations(seq, n_child)
                if in_graph is not None:
                    for k, v in in_graph.items():
                        v["eager_avg_us"] = round(traced[k]["avg_us"], 2) if k in traced else None
                        v["longest_kernel"] = traced[k]["longest_kernel"] if k in traced else None
                    traced = in_graph
                else:
                    trace_note = "in-graph durations unavailable (%s): EAGER kernel durations, ~10 %% above the replayed graph" % graph_info
                    graph_info = None
                    print("bench: " + trace_note, file=sys.stderr)
            out["roofline"], detail = roofline_leg(rl_layer, rl_x, max(10, args.steps), 5, traffic_key=tkey, traced=traced,
                                                   ms_per_step=ms if args.workload in ("ns", "c3") else None)
            if rl_what is not None:
                out["roofline"]["kernel"] += "; layer " + rl_what
            if traced is not None:
                tot = sum(v["us_per_step"] for v in traced.values())
                detail["traced_step"] = {
                    "what": "every role the library's entry points ran in one step, us per step: order and roles from the eager "
                            "launch trace (dsw_trace), durations " + ("from the rocprofv3 kernel trace of the replayed graph"
                                                                      if graph_info is not None else "from the EAGER launches"),
                    "note": trace_note, "graph": graph_info,
                    "sum_us": round(tot, 1), "vs_ms_per_step": round(tot / (ms * 1e3), 4),
                    "roles": sorted(({"role": k[0], "aux": list(k[1:]), "calls_per_step": round(v["calls_per_step"], 2),
                                      "avg_us": round(v["avg_us"], 2), "us_per_step": round(v["us_per_step"], 2),
                                      "kernels": v["kernels"], "longest_kernel": v.get("longest_kernel"),
                                      "eager_avg_us": v.get("eager_avg_us")}
                                     for k, v in traced.items()), key=lambda d: -d["us_per_step"])[:24]}
                out["roofline"]["traced_sum_vs_ms_per_step"] = detail["traced_step"]["vs_ms_per_step"]
                if trace_note:
                    out["roofline"]["trace_note"] = trace_note
            if args.workload in ("unet", "c5"):
                po = pooling_leg(model, lambda: model(x.detach()), max(10, args.steps), traffic_key=tkey, traced=traced)
                detail["pooling"] = po
                out["roofline"].update({"pooling_us_per_step": po["us_per_step"], "pooling_frac": po["frac"],
                                        "pooling_frac_counter": po["frac_counter"]})
            # the long form (every role of the step, the pooling products, the stand-alone recurrence / GEMM legs) goes to a side
            # file - the line itself stays short enough for the driver to keep whole - or inline on request
            if args.detail == "inline":
                out["roofline"]["detail"] = detail
            elif args.detail != "none":
                path = args.detail_file or os.path.join(REPO, "gpurun_out", "bench_detail_%s%s.json" % (
                    args.workload, "_k20" if (args.workload == "ns" and args.knn == 20) else ""))
                try:
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    with open(path, "w") as fh:
                        json.dump(detail, fh)
                    out["roofline"]["detail_file"] = os.path.relpath(path, REPO)
                except OSError as exc:
                    out["roofline"]["detail_file"] = "not written (%s)" % type(exc).__name__
        if not args.no_cpu_baseline:
            if args.workload == "unet":
                out["cpu_baseline"] = cpu_baseline_unet(model, wl, V)
            elif args.workload == "c5":
                out["cpu_baseline"] = cpu_baseline_c5(model, wl, V)
            else:
                out["cpu_baseline"] = cpu_baseline_leg(wl, lap, model)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
