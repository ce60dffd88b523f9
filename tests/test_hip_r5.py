"""Round-5 additions, GPU only: launch tracing (dsw_trace_begin / dsw_trace_end), the N = 1 step with the N > 1 gradient
handling, the refusal of diagnostics builds."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_trace_reports_the_roles_of_a_step():
    """One ConvCheb + one pooling forward / backward under a trace: the roles the entry points ran come back with plausible
    durations, nothing is recorded once the trace is closed, and a second trace can be opened."""
    from dsw_amd import _native, sphere
    from modules.layers import ConvCheb, GeneralAvgPool, prepare_torch_laplacian

    g = sphere.SphereHealpix(16, nest=True, k=8)
    layer = ConvCheb(32, 64, 3, laplacian=prepare_torch_laplacian(g.L, lmax=1.9)).to(DEV)
    pool = GeneralAvgPool(sphere.healpix_pool_matrices(16, nest=True)[0]).to(DEV)
    x = torch.randn(4, 3072, 32, device=DEV, requires_grad=True)

    def step():
        y = layer(x)
        z, _ = pool(y)
        z.sum().backward()

    step()
    torch.cuda.synchronize()
    for _ in range(2):
        with _native.LaunchTrace(1024) as tr:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        roles = {}
        for role, a0, a1, a2, us in tr.intervals:
            roles.setdefault(role, []).append((a0, a1, a2, us))
        assert "spmm" in roles and len(roles["spmm"]) == 6                      # pooling forward + transposed backward
        assert {(r[0], r[1], r[2]) for r in roles["spmm"]} == {(768, 3072, 64), (3072, 768, 64)}
        fwd = roles.get("fwd_one_launch") or roles.get("basis_fwd")
        assert fwd and len(fwd) == 3
        assert "basis_adj" in roles or "bwd_fused" in roles or "bwd_dual" in roles
        for rs in roles.values():
            for _a0, _a1, _a2, us in rs:
                assert 0.5 < us < 5000.0, rs
    lib = _native.load()
    assert lib.dsw_trace_end(None, None, None, None, None, None, None, 0, 0) < 0               # no trace open
    assert lib.dsw_build_flags() == 0 or os.environ.get("DSW_HIP_LIB")


def _bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--min-timed-ms", "1000", "--no-cpu-baseline", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_n1_runs_the_n_gt_1_step_and_agrees():
    """`--gpus 1 --grad-handling bucket` times the step every N > 1 rank runs (gradients in one flat bucket, weight-gradient
    kernels adding into it) without the exchange: it must cost what the default N = 1 step costs (3 %), so that the first
    scaling curve compares like with like; and the in-step role durations of the line add up to the
    step (item 1)."""
    # (two separate processes on a shared box: the bucket step is ~1 % slower by construction and a clock / neighbour hiccup moves
    # either line by a few per cent - one repeat before the comparison counts as failed; round 6: a 3 % gate without a repeat failed
    # once in five suite runs)
    for attempt in range(2):
        a = _bench("--no-roofline")
        b = _bench("--grad-handling", "bucket")
        assert "bucket" in b["config"]["grad_handling"] and "autograd" in a["config"]["grad_handling"]
        r = b["roofline"]
        if abs(a["ms_per_step"] - b["ms_per_step"]) <= 0.05 * a["ms_per_step"] and 0.96 <= r["in_step_sum_vs_ms_per_step"] <= 1.04:
            break
    assert abs(a["ms_per_step"] - b["ms_per_step"]) <= 0.05 * a["ms_per_step"], (a["ms_per_step"], b["ms_per_step"])
    assert r["launch_timing"].startswith("in-graph"), (r["launch_timing"], r.get("trace_note"))
    assert 0.96 <= r["in_step_sum_vs_ms_per_step"] <= 1.04, r["in_step_sum_vs_ms_per_step"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the headline is the SURVEY 8(d) figure of ONE launch: recurrence bytes (7E + 2Lb for the adjoint side, 5E + 2Lb forward) over
    # its duration - never the bytes of launches an unfused design would have made - and no entry exceeds its roof
    E = 16 * 49152 * 32 * 4
    # (Lb = nnz * 8 + 4 (V + 1) of the operator the bench built: 3.74-3.76 MB at k = 8)
    got = r["bytes_per_launch"] * r["launches"]
    assert any(0 <= got - n * E - 2 * 3_700_000 <= 2 * 100_000 for n in (7, 5)), r["bytes_per_launch"]
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_us"] / 1e3) <= 0.01 * r["achieved"]
    for e in r["in_step"]:
        assert 0.0 < e["frac"] <= 1.0 and e.get("hbm_frac", 0) <= 1.0 and e.get("mfma_frac", 0) <= 1.0, e
    assert 0.0 < r["step_frac"] <= 1.0 and r["mfma_frac"] <= 1.0
    # short enough for the driver to keep whole; the long form is in the side file
    assert len(json.dumps(b)) < 6000, len(json.dumps(b))
    det = json.load(open(os.path.join(ROOT, r["detail_file"])))
    assert {"forward_recurrence", "mfma", "traced_step"} <= set(det), sorted(det)


def _op_from_scipy(m):
    from dsw_amd import functional as F_
    from oracle import cheb_oracle as orc

    return F_.CsrOperator.from_sparse_coo(orc.coo_from_scipy(m).float().to(DEV))


@pytest.mark.parametrize("dt,C,B,m", [(torch.float32, 128, 8, 4), (torch.float32, 32, 3, 4), (torch.float32, 256, 1, 4),
                                       (torch.float32, 64, 5, 3), (torch.float32, 16, 2, 7), (torch.bfloat16, 64, 4, 4),
                                       (torch.float32, 20, 2, 4), (torch.float32, 33, 2, 4)])
def test_remap_regular_hierarchy_kernels_equal_the_generic_product(dt, C, B, m):
    """RemapBlock products of a regular hierarchy (m children per parent in consecutive rows: HEALPix nested) through the
    planned entry point (`dsw_remap_csr`: streaming kernels without any CSR walk) - pooling (GROUPS), unpooling (BROADCAST)
    and both transposes, with and without the fork's epilogue operand, dense and as channel slices of wider tensors - are
    BIT-identical to the generic CSR product and match the fp64 oracle.  Arbitrary weights; an unaligned channel count
    (33) takes the generic kernel through the same entry point."""
    import numpy as np
    from scipy import sparse
    from dsw_amd import functional as F_
    from oracle import cheb_oracle as orc

    rng = np.random.default_rng(3)
    vc = 5 * 77
    vf = vc * m
    rows = np.repeat(np.arange(vc), m)
    pool = sparse.csr_matrix((rng.random(vf).astype(np.float32) + 0.1, (rows, np.arange(vf))), shape=(vc, vf))
    unpool = sparse.csr_matrix((rng.random(vf).astype(np.float32) + 0.1, (np.arange(vf), rows)), shape=(vf, vc))
    tol = 2e-6 if dt == torch.float32 else 1e-2
    for mat, kind in ((pool, 1), (unpool, 2)):
        op = _op_from_scipy(mat)
        for o, k in ((op, kind), (op.transpose(), 3 - kind)):
            plan = o.remap_plan()
            assert (plan.kind, plan.m) == (k, m), (plan.kind, plan.m, k, m)
            x = torch.randn(B, o.shape[1], C, device=DEV).to(dt)
            z = torch.randn(B, o.shape[0], C, device=DEV).to(dt)
            y = F_._HIP.remap(o, x)
            assert torch.equal(y, F_._HIP.spmm(o, x))
            yz = F_._HIP.remap(o, x, z=z, beta=1.0)
            assert torch.equal(yz, F_._HIP.spmm(o, x, 1.0, z, 1.0))
            sc = o.values.cpu().double().numpy()
            msc = sparse.csr_matrix((sc, o.colind.cpu().numpy(), o.rowptr.cpu().numpy()), shape=o.shape)
            ref = np.stack([msc @ x[b].float().cpu().double().numpy() for b in range(B)])
            assert orc.max_rel_err(y.float(), ref) <= tol
            if (C * x.element_size()) % 16 == 0:
                wide_x = torch.randn(B, o.shape[1], C + 16, device=DEV).to(dt)
                wide_y = torch.zeros(B, o.shape[0], 2 * C + 8, device=DEV).to(dt)
                pad = 16 // x.element_size()
                F_._HIP.remap(o, wide_x[..., pad:pad + C], out=wide_y[..., C:2 * C])
                assert torch.equal(wide_y[..., C:2 * C], F_._HIP.spmm(o, wide_x[..., pad:pad + C].contiguous()))
                assert float(wide_y[..., :C].abs().max()) == 0.0 and float(wide_y[..., 2 * C:].abs().max()) == 0.0


def test_remap_plan_of_the_healpix_hierarchy_and_of_a_cross_sampling_matrix():
    """The plans the product paths get: HEALPix nested pooling / unpooling = the regular kinds; a conservative matrix between
    two different samplings = generic with its long (polar) rows listed - and the listed-rows launch equals the scanning
    launch to rounding, with and without the epilogue operand, on few samples and many."""
    import numpy as np
    from dsw_amd import functional as F_, sphere
    from oracle import cheb_oracle as orc

    pool_m, unpool_m = sphere.healpix_pool_matrices(8, nest=True)
    po, uo = _op_from_scipy(pool_m), _op_from_scipy(unpool_m)
    assert (po.remap_plan().kind, po.remap_plan().m) == (1, 4) and (uo.remap_plan().kind, uo.remap_plan().m) == (2, 4)
    assert po.transpose().remap_plan().kind == 2 and uo.transpose().remap_plan().kind == 1
    fine = sphere.SphereEquiangular(nlat=40, nlon=80, k=8)
    coarse = sphere.SphereHealpix(4, nest=True, k=8)
    pm, um = sphere.conservative_pool_matrices(fine.coords, coarse.coords)
    for mat in (pm, um):
        op = _op_from_scipy(mat)
        for o in (op, op.transpose()):
            plan = o.remap_plan()
            lens = np.diff(o.rowptr.cpu().numpy())
            assert plan.kind == 0
            assert plan.n_long == int((lens > max(16, int(2.0 * o.nnz / o.shape[0] + 0.5))).sum())
            for B in (1, 3, 8):
                x = torch.randn(B, o.shape[1], 32, device=DEV)
                z = torch.randn(B, o.shape[0], 32, device=DEV)
                # (rows between the plan's threshold and the scan's 64 entries are summed by a wave here, by a lane group
                # there: equal to rounding, not bit for bit)
                assert orc.max_rel_err(F_._HIP.remap(o, x), F_._HIP.spmm(o, x).cpu().numpy()) <= 1e-6
                assert orc.max_rel_err(F_._HIP.remap(o, x, z=z, beta=1.0), F_._HIP.spmm(o, x, 1.0, z, 1.0).cpu().numpy()) <= 1e-6


@pytest.mark.parametrize("dt,C,B", [(torch.float32, 32, 8), (torch.float32, 64, 3), (torch.float32, 128, 2), (torch.bfloat16, 32, 4),
                                     (torch.float32, 16, 1)])
def test_remap_rows_shared_by_lane_groups_and_the_fused_add(dt, C, B):
    """Cross-sampling pooling (rows of 6-30 entries, polar rows listed): the lane-groups-per-row kernel of the planned entry
    point against the fp64 oracle - forward, transposed, with empty rows - and `RemapBlock.forward_add` (the addend in the
    product's epilogue) against `forward(x) + addend`, values and both gradients."""
    import numpy as np
    from scipy import sparse
    from dsw_amd import functional as F_, sphere
    from modules.layers import GeneralAvgUnpool
    from oracle import cheb_oracle as orc

    fine = sphere.SphereEquiangular(nlat=36, nlon=72, k=8)
    coarse = sphere.SphereHealpix(8, nest=True, k=8)
    pm, um = sphere.conservative_pool_matrices(fine.coords, coarse.coords)
    pm = sparse.csr_matrix(pm).astype(np.float32)
    pm = sparse.vstack([pm[:100], sparse.csr_matrix((3, pm.shape[1]), dtype=np.float32), pm[100:]]).tocsr()   # empty rows
    tol = 2e-6 if dt == torch.float32 else 1e-2
    op = _op_from_scipy(pm)
    assert op.remap_plan().kind == 0 and op.remap_plan().parts in (2, 4)
    x = torch.randn(B, pm.shape[1], C, device=DEV).to(dt)
    z = torch.randn(B, pm.shape[0], C, device=DEV).to(dt)
    P64 = pm.astype(np.float64)
    ref = np.stack([P64 @ x[b].float().cpu().double().numpy() for b in range(B)])
    assert orc.max_rel_err(F_._HIP.remap(op, x).float(), ref) <= tol
    assert orc.max_rel_err(F_._HIP.remap(op, x, z=z, beta=1.0).float(), ref + z.float().cpu().double().numpy()) <= tol
    xt = torch.randn(B, pm.shape[0], C, device=DEV).to(dt)
    ref_t = np.stack([P64.T @ xt[b].float().cpu().double().numpy() for b in range(B)])
    assert orc.max_rel_err(F_._HIP.remap(op.transpose(), xt).float(), ref_t) <= tol
    if dt == torch.float32:
        unpool = GeneralAvgUnpool(um).to(DEV)
        xc = torch.randn(B, um.shape[1], C, device=DEV, requires_grad=True)
        add = torch.randn(B, um.shape[0], C, device=DEV, requires_grad=True)
        g = torch.randn(B, um.shape[0], C, device=DEV)
        y1 = unpool.forward_add(xc, add)
        y1.backward(g)
        gx1, ga1 = xc.grad.clone(), add.grad.clone()
        xc.grad = add.grad = None
        y2 = unpool(xc) + add
        y2.backward(g)
        assert orc.max_rel_err(y1, y2.detach().cpu().numpy()) <= 1e-6
        assert torch.equal(gx1, xc.grad) and torch.equal(ga1, add.grad)


def test_north_star_shape_one_launch_paths_full_size():
    """The nside-64 k = 8 plan (fattest tile: 175 / 115 rows) fits the LDS budgets of the one-launch forward (taken by
    dsw_cheb_fwd: a silent fall-back would cost 10 % of the headline step) and of the one-launch dual backward (no basis planes)."""
    from dsw_amd import _native, functional as F_, sphere
    from modules.layers import ConvCheb, prepare_torch_laplacian

    g = sphere.SphereHealpix(64, nest=True, k=8)
    lap = prepare_torch_laplacian(g.L, lmax=1.95)
    torch.manual_seed(10)
    layer = ConvCheb(32, 64, 3, laplacian=lap).to(DEV)
    op = F_.get_operator(layer.laplacian)
    x = torch.randn(1, op.shape[0], 32, device=DEV)
    lib = _native.load()
    pf, _k1 = F_._plan_ptr(op, x)
    assert int(lib.dsw_cheb_fwd_path(pf, 32, 64, 3, 0)) == 3           # DSW_FWD_ONE_LAUNCH
    pt, _k2 = F_._plan_ptr(op.transpose(), x)
    assert int(lib.dsw_cheb_bwd_needs_basis(pt, op.shape[0], 32, 64, 3, 0)) == 0


@pytest.mark.parametrize("sampling,Fout,B,relu", [("healpix16", 64, 5, False), ("healpix16", 32, 2, True), ("healpix8", 64, 1, False),
                                                   ("equiangular", 32, 3, False), ("equiangular", 64, 4, True)])
def test_forward_hop1_then_one_launch_vs_oracle(sampling, Fout, B, relu, monkeypatch):
    """Dense stencils (k = 20 HEALPix, equiangular: one-hop plans), K = 3, 32 input channels, fp32: the forward runs as the staged
    hop 1 + ONE launch for hop 2 and the channel mix (dsw_fwd3.hip: cheb3_hop2mix_kernel) - output, the kept basis (through the
    backward's weight gradients) and all gradients against the fp64 oracle; ragged last tiles, clustered tiles, fused ReLU."""
    from dsw_amd import _native, functional as F_, sphere
    from modules.layers import ConvCheb, prepare_torch_laplacian
    from oracle import cheb_oracle as orc

    monkeypatch.setattr(F_, "MIN_CLUSTERED_TILES", 1)
    if sampling == "equiangular":
        g = sphere.SphereEquiangular(nlat=30, nlon=60, k=20)
    else:
        g = sphere.SphereHealpix(int(sampling[7:]), nest=True, k=20)
    lap = prepare_torch_laplacian(g.L, lmax=1.9)
    torch.manual_seed(B + Fout)
    layer = ConvCheb(32, Fout, 3, laplacian=lap).to(DEV)
    with torch.no_grad():
        layer.bias.normal_(0, 0.1)
    V = g.L.shape[0]
    x = torch.randn(B, V, 32, device=DEV, requires_grad=True)
    gy = torch.randn(B, V, Fout, device=DEV)
    op = F_.get_operator(layer.laplacian)
    pp, _keep = F_._plan_ptr(op, x)
    assert pp is not None and int(_native.load().dsw_cheb_fwd_path(pp, 32, Fout, 3, 0)) == 5      # DSW_FWD_HOP1_THEN_ONE_LAUNCH
    y = layer.forward_activated(x, "relu") if relu else layer(x)
    y.backward(gy)
    rp, ci, va = orc.csr_arrays_from_coo(layer.laplacian.cpu())
    xn, wn, bn = (t.detach().cpu().numpy() for t in (x, layer.weight, layer.bias))
    y64 = orc.cheb_forward_f64(rp, ci, va, xn, wn, bn)
    import numpy as np

    g64 = gy.cpu().double().numpy()
    if relu:
        g64 = g64 * (y64 > 0)
        y64 = np.maximum(y64, 0)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, xn, wn, g64, True)
    assert orc.max_rel_err(y, y64) <= 2e-6
    assert orc.max_rel_err(x.grad, dx64) <= 2e-6
    assert orc.max_rel_err(layer.weight.grad, dw64) <= 4e-6
    assert orc.max_rel_err(layer.bias.grad, db64) <= 4e-6
