"""Parity of the HIP path (through the C ABI) against the golden vectors and the oracle.  GPU only.

Tolerances (errors normalised by max|ref|):
  fp32: <= 1e-5 vs the reference's fp32 output (golden) and <= 2e-6 vs the fp64 closed form;
  bf16 storage (fp32 operator + accumulation): <= 1e-2 vs fp64.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cheb_oracle as orc
import recipes

pytestmark = pytest.mark.gpu

TOL_GOLD = 1e-5
TOL_F64 = 2e-6
TOL_BF16 = 1e-2    # measured 2e-3 .. 5e-3 (SURVEY 8c proposed 3e-2: the reference's OWN bf16 CPU path is 1.1e-2 .. 2.6e-2 off)
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    from dsw_amd import _native

    _native.load()  # fails loudly if libdsw_hip.so is absent


def _layer_from(g, prefix, dtype=torch.float32):
    from modules.layers import ConvCheb

    B, V, Fin, Fout, K, has_bias, _ = [int(v) for v in g[prefix + "meta"]]
    lap = orc.coo_from_csr_arrays(g[prefix + "rowptr"], g[prefix + "colind"], g[prefix + "values"], (V, V))
    layer = ConvCheb(Fin, Fout, K, laplacian=lap, bias=bool(has_bias))
    layer.set_parameters(torch.from_numpy(g[prefix + "w"]), torch.from_numpy(g[prefix + "b"]) if has_bias else None)
    return layer.to(DEV).to(dtype), has_bias


def _run_layer(layer, x, gy):
    x = x.clone().requires_grad_(True)
    y = layer(x)
    y += 0.0  # in-place use of the output must be legal
    y.backward(gy)
    torch.cuda.synchronize()
    return y.detach(), x.grad, layer.weight.grad, None if layer.bias is None else layer.bias.grad


def _check_fixture(g, prefix=""):
    layer, has_bias = _layer_from(g, prefix)
    x = torch.from_numpy(g[prefix + "x"]).to(DEV)
    gy = torch.from_numpy(g[prefix + "gy"]).to(DEV)
    y, dx, dw, db = _run_layer(layer, x, gy)
    assert y.is_contiguous() and y.dtype == torch.float32
    assert orc.max_rel_err(y, g[prefix + "y"]) <= TOL_GOLD
    assert orc.max_rel_err(dx, g[prefix + "dx"]) <= TOL_GOLD
    assert orc.max_rel_err(dw, g[prefix + "dw"]) <= TOL_GOLD
    if has_bias:
        assert orc.max_rel_err(db, g[prefix + "db"]) <= TOL_GOLD
    # and against the fp64 closed form
    rp, ci, va = g[prefix + "rowptr"], g[prefix + "colind"], g[prefix + "values"]
    b = g[prefix + "b"] if has_bias else None
    y64 = orc.cheb_forward_f64(rp, ci, va, g[prefix + "x"], g[prefix + "w"], b)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, g[prefix + "x"], g[prefix + "w"], g[prefix + "gy"], bool(has_bias))
    assert orc.max_rel_err(y, y64) <= TOL_F64
    assert orc.max_rel_err(dx, dx64) <= TOL_F64
    assert orc.max_rel_err(dw, dw64) <= TOL_F64
    if has_bias:
        assert orc.max_rel_err(db, db64) <= TOL_F64
    # and against the plain-C restatement (third independent statement of the arithmetic)
    from oracle import c_oracle

    yc, basis_c = c_oracle.cheb_forward(rp, ci, va, g[prefix + "x"], g[prefix + "w"], b)
    dxc, dwc, _ = c_oracle.cheb_backward(rp, ci, va, basis_c, g[prefix + "w"], g[prefix + "gy"], bool(has_bias))
    assert orc.max_rel_err(y, yc) <= TOL_F64
    assert orc.max_rel_err(dx, dxc) <= TOL_F64
    assert orc.max_rel_err(dw, dwc) <= TOL_F64


@pytest.mark.parametrize("name", ["G1_conv_c1_k8", "G1_conv_c1_k20", "G6_conv_irregular"])
def test_conv_golden(name):
    _check_fixture(load_golden(name))


@pytest.mark.parametrize("kind", ["sym", "ns"])
@pytest.mark.parametrize("K", [1, 2, 3, 5])
def test_conv_golden_K_sweep(kind, K):
    _check_fixture(load_golden("G2_conv_K_sweep"), f"{kind}_K{K}_")


def test_conv_noncontiguous_input_and_functional_api():
    from modules.layers import conv_cheb

    g = load_golden("G2_conv_K_sweep")
    p = "ns_K3_"
    V = int(g[p + "meta"][1])
    lap = orc.coo_from_csr_arrays(g[p + "rowptr"], g[p + "colind"], g[p + "values"], (V, V)).to(DEV)
    x = torch.from_numpy(g[p + "x"]).to(DEV).permute(1, 0, 2).contiguous().permute(1, 0, 2)
    assert not x.is_contiguous()
    w = torch.from_numpy(g[p + "w"]).to(DEV)
    y = conv_cheb(lap, x, w) + torch.from_numpy(g[p + "b"]).to(DEV)
    assert orc.max_rel_err(y, g[p + "y"]) <= TOL_GOLD
    with pytest.raises(ValueError, match="does not match the expected shape"):
        conv_cheb(lap, torch.zeros(1, V, 9, device=DEV), w)
    with pytest.raises(TypeError, match="unsupported dtype"):
        conv_cheb(lap, x.double(), w.double())


@pytest.mark.parametrize("tag", ["hier", "interp"])
def test_remap_golden(tag):
    from modules.layers import GeneralAvgPool, GeneralAvgUnpool
    from scipy import sparse

    g = load_golden("G3_remap")
    pm = sparse.csr_matrix((g[f"{tag}_pool_values"], g[f"{tag}_pool_colind"], g[f"{tag}_pool_rowptr"]), shape=(192, 768))
    um = sparse.csr_matrix((g[f"{tag}_unpool_values"], g[f"{tag}_unpool_colind"], g[f"{tag}_unpool_rowptr"]), shape=(768, 192))
    pool = GeneralAvgPool(sparse.coo_matrix(pm)).to(DEV)
    unpool = GeneralAvgUnpool(sparse.coo_matrix(um)).to(DEV)
    x = torch.from_numpy(g[f"{tag}_x"]).to(DEV).requires_grad_(True)
    y, idx = pool(x)
    assert idx is None
    y.backward(torch.from_numpy(g[f"{tag}_gyp"]).to(DEV))
    assert orc.max_rel_err(y, g[f"{tag}_yp"]) <= TOL_GOLD
    assert orc.max_rel_err(x.grad, g[f"{tag}_dxp"]) <= TOL_GOLD
    # WAIVED on purpose: the reference hands out a permuted view of its internal [V, F, B] product (strides recorded in
    # the fixture, reproduced by the oracle restatement); the HIP path computes in the callers' [B, V, F] layout and
    # returns it contiguous - same shape and values, and every consumer in the reference (ConvCheb.forward: layers.py:158
    # makes its input contiguous; torch.cat) accepts either
    assert tuple(g[f"{tag}_yp_strides"]) != tuple(y.stride()) and y.is_contiguous()
    xu = torch.from_numpy(g[f"{tag}_xu"]).to(DEV).requires_grad_(True)
    yu = unpool(xu, None)
    yu.backward(torch.from_numpy(g[f"{tag}_gyu"]).to(DEV))
    assert orc.max_rel_err(yu, g[f"{tag}_yu"]) <= TOL_GOLD
    assert orc.max_rel_err(xu.grad, g[f"{tag}_dxu"]) <= TOL_GOLD


def test_unet_golden_g5():
    from test_host_logic import build_g5_model, check_g5

    model, g, names = build_g5_model(DEV)
    assert model.conv1.convblock1.conv.laplacian.is_cuda
    # measured on MI355X (round 2): output 5.9e-7, loss <1e-7, gradient fingerprints 1.2e-6 -> bound 5e-6 / 5e-5
    errs = check_g5(model, g, names, device=DEV, tol=5e-6)
    print("G5 on the device:", {k: "%.2e" % v for k, v in errs.items()})
    try:
        import json
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_fullsize.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data["g5_unet_nside8_fp32"] = {k: float("%.3e" % v) for k, v in errs.items()}
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


# ---------------------------------------------------------------------------------------------
# seeded inputs vs the oracle at sizes the oracle finishes in seconds (awkward channel counts,
# ragged batches, empty rows, rectangular operators)
# ---------------------------------------------------------------------------------------------
def _rand_case(V, B, Fin, Fout, K, seed, bias=True, op="healpix"):
    from dsw_amd import sphere

    if op == "healpix":
        g = sphere.SphereHealpix(int(np.sqrt(V // 12)), nest=True, k=8)
        lap = orc.prepare_laplacian_fixed_lmax(g.L, 1.9)
        rp, ci, va = orc.csr_arrays_from_coo(lap)
    else:
        rp, ci, va = recipes.irregular_operator(V, seed=seed, min_deg=0, max_deg=64)
    x = recipes.rand(seed, (B, V, Fin))
    w = recipes.rand(seed + 1, (Fin, K, Fout), np.sqrt(2.0 / (Fin * K)))
    b = recipes.rand(seed + 2, (Fout,), 0.1) if bias else None
    gy = recipes.rand(seed + 3, (B, V, Fout))
    return (rp, ci, va), x, w, b, gy


CASES = [
    # V, B, Fin, Fout, K, bias, operator
    (768, 1, 18, 64, 3, True, "healpix"),     # first UNet layer: Fin=18 (not a multiple of 4)
    (768, 3, 64, 2, 3, True, "healpix"),      # last UNet layer: Fout=2
    (768, 5, 32, 64, 3, False, "healpix"),    # ragged batch (B % 4 != 0)
    (192, 2, 192, 256, 3, True, "healpix"),   # multi column tile, Fin > 128
    (192, 2, 7, 5, 4, True, "healpix"),       # odd everything
    (48, 8, 512, 256, 3, True, "healpix"),    # coarsest UNet level
    (1000, 2, 12, 20, 5, True, "irregular"),  # rows with 0..64 entries, non-symmetric
    (3072, 2, 32, 32, 2, True, "healpix"),
    (130, 1, 1, 1, 3, True, "irregular"),
    # channel-shrinking layers: mix-first evaluation order (dsw_cheb_mix_first)
    (768, 3, 64, 32, 3, True, "healpix"),     # K = 3: one fused Clenshaw pair
    (768, 2, 128, 64, 2, True, "healpix"),    # K = 2: single hop
    (192, 2, 64, 16, 4, False, "healpix"),    # K = 4
    (192, 3, 96, 40, 5, True, "healpix"),     # K = 5, Fout not a multiple of 32
    (1000, 2, 48, 20, 3, True, "irregular"),  # non-symmetric operator: pins L vs L^T in both directions
    (3072, 4, 256, 128, 3, True, "healpix"),  # decoder shape of the UNet
    # shapes of the fused dgrad + wgrad pass (fp32, Fout = 64, 3-4 (k, f) tiles)
    (192, 2, 32, 64, 4, True, "healpix"),
    (768, 2, 64, 64, 2, False, "healpix"),
    (1000, 3, 32, 64, 3, True, "irregular"),  # N % 32 != 0 -> falls back to the separate kernels; same answer
    (1024, 2, 32, 64, 3, True, "irregular"),
    # ... and with 32-column o-tiles (fp32, Fout = 32 mod 64): the equiangular block of bench workload c5
    (768, 3, 32, 32, 3, True, "healpix"),     # fused pass, 3 tiles
    (192, 2, 32, 32, 4, False, "healpix"),    # fused pass, 4 tiles
    (768, 2, 96, 96, 2, True, "healpix"),     # separate wgrad: 6 (k, f) tiles x 3 o-tiles of 32
    (1024, 5, 64, 32, 1, True, "irregular"),  # K = 1
    # narrow outputs (K * Fout <= 16): vector-ALU kernels of dsw_narrow.hip, forward / dgrad / wgrad
    (768, 2, 128, 4, 3, True, "healpix"),
    (192, 3, 32, 5, 2, False, "healpix"),
    (1000, 2, 256, 1, 3, True, "irregular"),  # ragged row count, one output channel
    (768, 5, 16, 3, 4, True, "healpix"),      # 4 lanes per row
]


@pytest.mark.parametrize("V,B,Fin,Fout,K,bias,op", CASES)
def test_conv_vs_oracle_fp32(V, B, Fin, Fout, K, bias, op):
    from modules.layers import ConvCheb

    (rp, ci, va), x, w, b, gy = _rand_case(V, B, Fin, Fout, K, seed=1000 + V + Fin, bias=bias, op=op)
    lap = orc.coo_from_csr_arrays(rp, ci, va, (V, V))
    layer = ConvCheb(Fin, Fout, K, laplacian=lap, bias=bias)
    layer.set_parameters(torch.from_numpy(w), None if b is None else torch.from_numpy(b))
    layer = layer.to(DEV)
    y, dx, dw, db = _run_layer(layer, torch.from_numpy(x).to(DEV), torch.from_numpy(gy).to(DEV))
    y64 = orc.cheb_forward_f64(rp, ci, va, x, w, b)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, x, w, gy, bias)
    assert orc.max_rel_err(y, y64) <= TOL_F64
    assert orc.max_rel_err(dx, dx64) <= TOL_F64
    assert orc.max_rel_err(dw, dw64) <= 2 * TOL_F64  # long fp32 reduction over B*V rows
    if bias:
        assert orc.max_rel_err(db, db64) <= 2 * TOL_F64


@pytest.mark.parametrize("V,B,Fin,Fout,K", [(768, 4, 64, 128, 5), (768, 3, 32, 64, 3), (192, 2, 24, 40, 3), (192, 1, 6, 10, 2),
                                            (768, 4, 128, 64, 3), (192, 2, 64, 32, 5)])   # last two: mix-first order
def test_conv_vs_oracle_bf16(V, B, Fin, Fout, K):
    from modules.layers import ConvCheb

    (rp, ci, va), x, w, b, gy = _rand_case(V, B, Fin, Fout, K, seed=77 + Fin, bias=True)
    # bf16-representable inputs so that only the kernel's arithmetic is compared
    q = lambda a: torch.from_numpy(a).to(torch.bfloat16)
    xq, wq, bq, gyq = q(x), q(w), q(b), q(gy)
    lap = orc.coo_from_csr_arrays(rp, ci, va, (V, V))
    layer = ConvCheb(Fin, Fout, K, laplacian=lap, bias=True)
    layer.set_parameters(wq.float(), bq.float())
    layer = layer.to(DEV).to(torch.bfloat16)
    # the module cast leaves the operator buffer at fp32 (_Fp32OperatorMixin; the reference rounds it to bf16): the
    # oracle reads back exactly the values the kernels use
    va_q = layer.laplacian.coalesce().values().float().cpu().numpy()
    y, dx, dw, db = _run_layer(layer, xq.to(DEV), gyq.to(DEV))
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and dw.dtype == torch.bfloat16
    f = lambda t: t.float().numpy()
    y64 = orc.cheb_forward_f64(rp, ci, va_q, f(xq), f(wq), f(bq))
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va_q, f(xq), f(wq), f(gyq), True)
    assert orc.max_rel_err(y.float(), y64) <= TOL_BF16
    assert orc.max_rel_err(dx.float(), dx64) <= TOL_BF16
    assert orc.max_rel_err(dw.float(), dw64) <= TOL_BF16
    assert orc.max_rel_err(db.float(), db64) <= TOL_BF16


def test_spmm_axpby_and_rectangular():
    from dsw_amd import functional as F_

    rng = np.random.default_rng(5)
    for (vo, vi, C, B) in [(37, 91, 8, 3), (300, 120, 33, 2), (64, 64, 128, 9), (5, 7, 2, 1)]:
        from scipy import sparse

        m = sparse.random(vo, vi, density=0.15, random_state=int(rng.integers(1 << 30)), format="csr", dtype=np.float32)
        m.sort_indices()
        op = F_.CsrOperator.from_sparse_coo(orc.coo_from_scipy(m).float().to(DEV))
        x = torch.from_numpy(recipes.rand(1, (B, vi, C))).to(DEV)
        z = torch.from_numpy(recipes.rand(2, (B, vo, C))).to(DEV)
        z2 = torch.from_numpy(recipes.rand(3, (B, vo, C))).to(DEV)
        ref = 2.0 * orc.remap_f64(m.indptr, m.indices, m.data, (vo, vi), x.cpu().numpy()) \
            - 1.0 * z.cpu().double().numpy() + 0.5 * z2.cpu().double().numpy()
        y = F_._HIP.spmm(op, x, 2.0, z, -1.0, z2, 0.5)
        assert orc.max_rel_err(y, ref) <= TOL_F64
        # in place (Y aliases Z), as used by the adjoint recurrence
        zc = z.clone()
        F_._HIP.spmm(op, x, 2.0, zc, -1.0, z2, 0.5, out=zc)
        assert torch.equal(zc, y)
        # transpose operator == autograd of the forward
        xt = torch.from_numpy(recipes.rand(4, (B, vo, C))).to(DEV)
        yt = F_._HIP.spmm(op.transpose(), xt)
        ref_t = orc.remap_backward_f64(m.indptr, m.indices, m.data, (vo, vi), xt.cpu().numpy())
        assert orc.max_rel_err(yt, ref_t) <= TOL_F64


@pytest.mark.parametrize("dt,C,B", [(torch.float32, 32, 8), (torch.float32, 8, 5), (torch.float32, 128, 1),
                                      (torch.float32, 33, 3), (torch.float32, 256, 2), (torch.bfloat16, 64, 9)])
def test_spmm_rows_with_hundreds_of_entries(dt, C, B):
    """Cross-sampling pooling matrices have rows with hundreds of entries next to rows with a handful (polar cells): rows
    beyond 64 entries are summed by a whole wave (dsw_spmm.hip, spmm_long_rows) - exactly 64 / 65 entries, empty rows,
    the epilogue operands, in-place output and the transpose (every column short) against fp64."""
    from scipy import sparse
    from dsw_amd import functional as F_

    rng = np.random.default_rng(11)
    vo, vi = 333, 4000
    lens = rng.integers(0, 20, vo)
    lens[[0, 7, 100, 101, 332]] = [303, 64, 65, 1200, 97]
    lens[[5, 6]] = 0
    rows = np.repeat(np.arange(vo), lens)
    cols = np.concatenate([rng.choice(vi, n, replace=False) for n in lens])
    m = sparse.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32), (rows, cols)), shape=(vo, vi))
    m.sort_indices()
    op = F_.CsrOperator.from_sparse_coo(orc.coo_from_scipy(m).float().to(DEV))
    tol = TOL_F64 if dt == torch.float32 else TOL_BF16
    x = torch.from_numpy(recipes.rand(1, (B, vi, C))).to(DEV).to(dt)
    z = torch.from_numpy(recipes.rand(2, (B, vo, C))).to(DEV).to(dt)
    z2 = torch.from_numpy(recipes.rand(3, (B, vo, C))).to(DEV).to(dt)
    f64 = lambda t: t.float().cpu().double().numpy()
    ref = 2.0 * orc.remap_f64(m.indptr, m.indices, m.data, (vo, vi), x.float().cpu().numpy()) - f64(z) + 0.5 * f64(z2)
    y = F_._HIP.spmm(op, x, 2.0, z, -1.0, z2, 0.5)
    assert orc.max_rel_err(y.float(), ref) <= tol
    zc = z.clone()
    F_._HIP.spmm(op, x, 2.0, zc, -1.0, z2, 0.5, out=zc)       # in place (Y aliases Z)
    assert torch.equal(zc, y)
    assert torch.equal(F_._HIP.spmm(op, x, 2.0, z, -1.0, z2, 0.5), y)   # deterministic
    # the planned remap entry point lists the long rows instead of scanning for them (other threshold, other grid): same sums
    assert op.remap_plan().kind == 0 and op.remap_plan().n_long >= 4
    assert orc.max_rel_err(F_._HIP.remap(op, x, z=z, beta=-1.0).float(),
                           orc.remap_f64(m.indptr, m.indices, m.data, (vo, vi), x.float().cpu().numpy()) - f64(z)) <= tol
    xt = torch.from_numpy(recipes.rand(4, (B, vo, C))).to(DEV).to(dt)
    yt = F_._HIP.spmm(op.transpose(), xt)
    ref_t = orc.remap_backward_f64(m.indptr, m.indices, m.data, (vo, vi), xt.float().cpu().numpy())
    assert orc.max_rel_err(yt.float(), ref_t) <= tol
    if C % 8 == 0 and dt == torch.float32:                     # channel slices of wider tensors (row strides)
        wide_x = torch.from_numpy(recipes.rand(5, (B, vi, C + 8))).to(DEV)
        wide_y = torch.zeros(B, vo, C + 16, device=DEV)
        F_._HIP.spmm(op, wide_x[..., 8:], out=wide_y[..., 4:4 + C])
        ref_s = orc.remap_f64(m.indptr, m.indices, m.data, (vo, vi), wide_x[..., 8:].cpu().numpy())
        assert orc.max_rel_err(wide_y[..., 4:4 + C], ref_s) <= tol
        assert float(wide_y[..., :4].abs().max()) == 0.0 and float(wide_y[..., 4 + C:].abs().max()) == 0.0


def test_empty_batch_and_determinism():
    from modules.layers import ConvCheb

    (rp, ci, va), x, w, b, gy = _rand_case(768, 4, 32, 64, 3, seed=9)
    lap = orc.coo_from_csr_arrays(rp, ci, va, (768, 768))
    layer = ConvCheb(32, 64, 3, laplacian=lap)
    layer.set_parameters(torch.from_numpy(w), torch.from_numpy(b))
    layer = layer.to(DEV)
    y0 = layer(torch.zeros(0, 768, 32, device=DEV))
    assert y0.shape == (0, 768, 64)
    xs, gys = torch.from_numpy(x).to(DEV), torch.from_numpy(gy).to(DEV)
    a = _run_layer(layer, xs, gys)
    layer.zero_grad(set_to_none=True)
    b2 = _run_layer(layer, xs, gys)
    for t0, t1 in zip(a, b2):
        assert torch.equal(t0, t1)  # bit-identical reruns (deterministic_training in the reference configs)


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE north-star shape: nside=64, 32->64, K=3, B=16) - no oracle run
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ns_layer():
    from dsw_amd import sphere
    from modules.layers import ConvCheb, prepare_torch_laplacian

    g = sphere.SphereHealpix(64, nest=True, k=8)
    lap = prepare_torch_laplacian(g.L, lmax=1.9)
    torch.manual_seed(10)
    return ConvCheb(32, 64, 3, laplacian=lap).to(DEV)


def test_full_size_linearity_and_sample_independence(ns_layer):
    torch.manual_seed(1234)
    B, V = 16, 49152
    x1 = torch.randn(B, V, 32, device=DEV)
    x2 = torch.randn(B, V, 32, device=DEV)
    with torch.no_grad():
        bias = ns_layer.bias.detach()
        y1, y2, y12 = ns_layer(x1) - bias, ns_layer(x2) - bias, ns_layer(x1 + 0.5 * x2) - bias
        assert orc.max_rel_err(y12, (y1 + 0.5 * y2)) <= 1e-5           # linear in x
        y_perm = ns_layer(x1.flip(0)) - bias
        assert torch.equal(y_perm.flip(0), y1)                         # samples are independent
        # one sample of the big batch equals the same sample run alone (bitwise: same kernels)
        y_one = ns_layer(x1[3:4].contiguous()) - bias
        assert orc.max_rel_err(y_one, y1[3:4]) <= 1e-6


def test_full_size_adjoint_identity(ns_layer):
    """<gy, J x> == <J^T gy, x> and dW == d/dW <gy, y>: ties backward to forward at full size."""
    torch.manual_seed(4321)
    B, V = 16, 49152
    x = torch.randn(B, V, 32, device=DEV, requires_grad=True)
    gy = torch.randn(B, V, 64, device=DEV)
    y = ns_layer(x)
    y.backward(gy)
    with torch.no_grad():
        lhs = (gy.double() * (y - ns_layer.bias).double()).sum().item()
        rhs_x = (x.grad.double() * x.double()).sum().item()
        rhs_w = (ns_layer.weight.grad.double() * ns_layer.weight.double()).sum().item()
        scale = gy.double().norm().item() * y.double().norm().item()
        assert abs(lhs - rhs_x) / scale < 1e-6
        assert abs(lhs - rhs_w) / scale < 1e-6
        assert orc.max_rel_err(ns_layer.bias.grad, gy.double().sum(dim=(0, 1))) < 1e-5


def test_full_size_sample_vs_oracle(ns_layer):
    """One sphere of the full-size batch against the fp64 oracle (single sample keeps it in seconds)."""
    torch.manual_seed(7)
    V = 49152
    x = torch.randn(16, V, 32, device=DEV)
    with torch.no_grad():
        y = ns_layer(x)
    rp, ci, va = orc.csr_arrays_from_coo(ns_layer.laplacian.cpu())
    y64 = orc.cheb_forward_f64(rp, ci, va, x[11:12].cpu().numpy(), ns_layer.weight.detach().cpu().numpy(),
                               ns_layer.bias.detach().cpu().numpy())
    assert orc.max_rel_err(y[11:12], y64) <= TOL_F64


# ---------------------------------------------------------------------------------------------
# fused two-hop SpMM (dsw_spmm2_fused) against the plain formula, and basis/adjoint with plans
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opname,C,B,dt", [("nest", 32, 3, torch.float32), ("ring", 8, 2, torch.float32),
                                            ("irregular", 12, 2, torch.float32), ("nest", 64, 2, torch.bfloat16)])
def test_spmm2_fused_vs_formula(opname, C, B, dt):
    import ctypes
    from dsw_amd import functional as F_, _native, sphere
    from scipy import sparse

    if opname == "irregular":
        rp, ci, va = recipes.irregular_operator(300, seed=8, min_deg=0, max_deg=30)  # no locality: S1 = S2 = everything
    else:
        m = sphere.SphereHealpix(8, nest=(opname == "nest"), k=8).L
        rp, ci, va = m.indptr, m.indices, m.data.astype(np.float32)
    V = len(rp) - 1
    op = F_.CsrOperator.from_sparse_coo(orc.coo_from_csr_arrays(rp, ci, va, (V, V)).to(DEV))
    plan = op.hop2_plan(C * (2 if dt == torch.bfloat16 else 4))
    assert plan is not None
    lib = _native.load()
    assert lib.dsw_spmm2_supported(ctypes.addressof(plan._struct), C, 1 if dt == torch.bfloat16 else 0) == 1
    mk = lambda s: torch.from_numpy(recipes.rand(s, (B, V, C))).to(dt).to(DEV)
    U, Z1, Z1b, Z2 = mk(1), mk(2), mk(3), mk(4)
    Y1, Y2 = torch.empty_like(U), torch.empty_like(U)
    a1, b1, d1, a2, b2, c2 = 2.0, 1.0, -1.0, 1.5, -1.0, 0.5
    rc = lib.dsw_spmm2_fused(ctypes.addressof(plan._struct), V, U.data_ptr(), Z1.data_ptr(), Z1b.data_ptr(),
                             Z2.data_ptr(), Y1.data_ptr(), Y2.data_ptr(), B, C, a1, b1, d1, a2, b2, c2,
                             1 if dt == torch.bfloat16 else 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    f = lambda t: t.float().cpu().numpy().astype(np.float64)
    r1 = a1 * orc.remap_f64(rp, ci, va, (V, V), f(U)) + b1 * f(Z1) + d1 * f(Z1b)
    if dt == torch.bfloat16:
        r1_used = torch.from_numpy(r1).to(torch.bfloat16).float().numpy().astype(np.float64)  # Y1 is stored in bf16
    else:
        r1_used = r1
    r2 = a2 * orc.remap_f64(rp, ci, va, (V, V), r1_used) + b2 * f(U) + c2 * f(Z2)
    tol = TOL_BF16 if dt == torch.bfloat16 else TOL_F64
    assert orc.max_rel_err(Y1.float(), r1) <= tol
    assert orc.max_rel_err(Y2.float(), r2) <= tol
    # NULL optionals + in-place Y2 == Z2
    Z2c = Z2.clone()
    rc = lib.dsw_spmm2_fused(ctypes.addressof(plan._struct), V, U.data_ptr(), None, None, Z2c.data_ptr(), None,
                             Z2c.data_ptr(), B, C, 1.0, 0.0, 0.0, 2.0, -1.0, 1.0,
                             1 if dt == torch.bfloat16 else 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    t1 = orc.remap_f64(rp, ci, va, (V, V), f(U))
    if dt == torch.bfloat16:
        t1 = torch.from_numpy(t1).to(torch.bfloat16).float().numpy().astype(np.float64)
    ref = 2.0 * orc.remap_f64(rp, ci, va, (V, V), t1) - f(U) + f(Z2)
    assert orc.max_rel_err(Z2c.float(), ref) <= tol


@pytest.mark.parametrize("opname,C,B,dt", [("nest20", 32, 3, torch.float32), ("ring20", 8, 2, torch.float32),
                                            ("irregular", 12, 2, torch.float32), ("nest20", 64, 2, torch.bfloat16),
                                            ("nest20", 128, 2, torch.float32), ("nest8", 32, 1, torch.float32)])
def test_spmm_staged_vs_formula(opname, C, B, dt, monkeypatch):
    """The staged one-hop kernel (dsw_spmm_staged: dense stencils) against Y = a A U + b Z + c Z2 in fp64, incl. NULL
    optionals, the in-place forms of the adjoint step (Y aliases Z) and wide rows (128-byte channel chunks)."""
    import ctypes
    from dsw_amd import functional as F_, _native, sphere

    monkeypatch.setattr(F_, "HOP_MODE", "staged")
    monkeypatch.setattr(F_, "MIN_CLUSTERED_TILES", 1)
    if opname == "irregular":
        rp, ci, va = recipes.irregular_operator(300, seed=8, min_deg=0, max_deg=30)
    else:
        m = sphere.SphereHealpix(8, nest=opname.startswith("nest"), k=int(opname[4:])).L
        rp, ci, va = m.indptr, m.indices, m.data.astype(np.float32)
    V = len(rp) - 1
    op = F_.CsrOperator.from_sparse_coo(orc.coo_from_csr_arrays(rp, ci, va, (V, V)).to(DEV))
    es = 2 if dt == torch.bfloat16 else 4
    plan = op.hop2_plan(min(C * es, 128))
    assert plan is not None and plan.hops == 1
    lib = _native.load()
    code = 1 if dt == torch.bfloat16 else 0
    pa = ctypes.addressof(plan._struct)
    assert lib.dsw_spmm_staged_supported(pa, C, code) == 1 and lib.dsw_spmm2_supported(pa, C, code) == 0
    mk = lambda s: torch.from_numpy(recipes.rand(s, (B, V, C))).to(dt).to(DEV)
    U, Z, Z2 = mk(1), mk(2), mk(3)
    Y = torch.empty_like(U)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda t: t.float().cpu().numpy().astype(np.float64)
    tol = TOL_BF16 if dt == torch.bfloat16 else TOL_F64
    AU = orc.remap_f64(rp, ci, va, (V, V), f(U))
    for so in (0, 1):
        assert lib.dsw_spmm_staged(pa, V, U.data_ptr(), Z.data_ptr(), Z2.data_ptr(), Y.data_ptr(), B, C, 2.0, -1.0, 0.5, code, st, so) == 0
        assert orc.max_rel_err(Y.float(), 2.0 * AU - f(Z) + 0.5 * f(Z2)) <= tol
    assert lib.dsw_spmm_staged(pa, V, U.data_ptr(), None, None, Y.data_ptr(), B, C, 1.0, 0.0, 0.0, code, st, 0) == 0
    assert orc.max_rel_err(Y.float(), AU) <= tol
    Zc = Z.clone()     # adjoint step, in place on the Z plane
    assert lib.dsw_spmm_staged(pa, V, U.data_ptr(), Zc.data_ptr(), Z2.data_ptr(), Zc.data_ptr(), B, C, 2.0, 1.0, -1.0, code, st, 0) == 0
    assert orc.max_rel_err(Zc.float(), 2.0 * AU + f(Z) - f(Z2)) <= tol
    Zc = Z.clone()     # only the second epilogue operand
    assert lib.dsw_spmm_staged(pa, V, U.data_ptr(), None, Zc.data_ptr(), Zc.data_ptr(), B, C, 1.0, 0.0, -1.0, code, st, 0) == 0
    assert orc.max_rel_err(Zc.float(), AU - f(Z)) <= tol
    assert lib.dsw_spmm_staged(pa, V, U.data_ptr(), None, None, Y.data_ptr(), 0, C, 1.0, 0.0, 0.0, code, st, 0) == 0   # empty batch
    assert lib.dsw_spmm2_fused(pa, V, U.data_ptr(), None, None, None, None, Y.data_ptr(), B, C, 1.0, 0.0, 0.0, 2.0, -1.0, 0.0,
                               code, st) != 0      # a one-hop plan is not a two-hop plan


@pytest.mark.parametrize("K", [2, 3, 4, 5])
def test_staged_recurrences_equal_unfused(K, monkeypatch):
    """Forward basis and adjoint recurrence on a dense (k = 20) non-symmetric operator: one staged launch per hop vs the
    plain one-hop kernel, and the forward basis against the fp64 oracle."""
    from dsw_amd import functional as F_, _native, sphere

    monkeypatch.setattr(F_, "HOP_MODE", "auto")
    m = sphere.SphereHealpix(8, nest=True, k=20).L
    rp, ci, va = m.indptr, m.indices, (m.data * 0.7).astype(np.float32)
    va = (va * np.repeat(0.5 + np.random.default_rng(1).random(768), np.diff(rp))).astype(np.float32)
    op = F_.CsrOperator.from_sparse_coo(orc.coo_from_csr_arrays(rp, ci, va, (768, 768)).to(DEV))
    opt = op.transpose()
    B, V, C = 3, 768, 32
    lib = _native.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(recipes.rand(5, (B, V, C))).to(DEV)
    res = {}
    for staged in (False, True):
        pp, keep = F_._plan_ptr(op, x) if staged else (None, None)
        ppt, keept = F_._plan_ptr(opt, x) if staged else (None, None)
        assert (pp is not None) == staged and (not staged or (keep.hops == 1 and keept.hops == 1))
        T = torch.empty(K - 1, B, V, C, device=DEV)
        assert lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz,
                                      x.data_ptr(), T.data_ptr(), B, C, K, 0, st, pp) == 0
        G0 = torch.from_numpy(recipes.rand(6, (B, V, C))).to(DEV)
        Gr = torch.from_numpy(recipes.rand(7, (K - 1, B, V, C))).to(DEV)
        spare = torch.empty(2, B, V, C, device=DEV)
        assert lib.dsw_cheb_basis_adj(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz,
                                      G0.data_ptr(), Gr.data_ptr(), B, C, K, 0, st, ppt, spare.data_ptr()) == 0
        torch.cuda.synchronize()
        res[staged] = (T.clone(), G0.clone())
    assert orc.max_rel_err(res[True][0], res[False][0].double()) <= TOL_F64
    assert orc.max_rel_err(res[True][1], res[False][1].double()) <= TOL_F64
    L = orc._csr64(rp, ci, va, (V, V))
    Tref = np.stack(orc.cheb_basis_f64(L, x.cpu().numpy(), K)[1:])
    assert orc.max_rel_err(res[True][0], Tref) <= TOL_F64


@pytest.mark.parametrize("K", [3, 4, 5, 6])
def test_fused_recurrences_equal_unfused(K):
    """Forward basis and adjoint recurrence: pairwise-fused launches vs one launch per hop."""
    from dsw_amd import functional as F_, _native, sphere

    m = sphere.SphereHealpix(8, nest=True, k=8).L
    rp, ci, va = m.indptr, m.indices, (m.data * 0.7).astype(np.float32)
    # non-symmetric on purpose: scale rows
    va = (va * np.repeat(0.5 + np.random.default_rng(1).random(768), np.diff(rp))).astype(np.float32)
    op = F_.CsrOperator.from_sparse_coo(orc.coo_from_csr_arrays(rp, ci, va, (768, 768)).to(DEV))
    opt = op.transpose()
    B, V, C = 3, 768, 32
    lib = _native.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(recipes.rand(5, (B, V, C))).to(DEV)
    res = {}
    for fused in (False, True):
        pp = F_._plan_ptr(op, x)[0] if fused else None
        ppt = F_._plan_ptr(opt, x)[0] if fused else None
        assert (pp is not None) == fused
        T = torch.empty(K - 1, B, V, C, device=DEV)
        assert lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz,
                                      x.data_ptr(), T.data_ptr(), B, C, K, 0, st, pp) == 0
        G0 = torch.from_numpy(recipes.rand(6, (B, V, C))).to(DEV)
        Gr = torch.from_numpy(recipes.rand(7, (K - 1, B, V, C))).to(DEV)
        spare = torch.empty(2, B, V, C, device=DEV)
        assert lib.dsw_cheb_basis_adj(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz,
                                      G0.data_ptr(), Gr.data_ptr(), B, C, K, 0, st, ppt, spare.data_ptr()) == 0
        torch.cuda.synchronize()
        res[fused] = (T.clone(), G0.clone())
    assert orc.max_rel_err(res[True][0], res[False][0].double()) <= TOL_F64
    assert orc.max_rel_err(res[True][1], res[False][1].double()) <= TOL_F64
    # and the unfused forward basis against the fp64 oracle
    L = orc._csr64(rp, ci, va, (V, V))
    Tref = np.stack(orc.cheb_basis_f64(L, x.cpu().numpy(), K)[1:])
    assert orc.max_rel_err(res[True][0], Tref) <= TOL_F64


@pytest.mark.parametrize("N,Fin,Fout", [(768 * 3, 256, 128), (1000, 48, 20), (64, 512, 256), (768 * 3, 128, 64), (1024, 96, 64),
                                         (1537, 64, 2), (98304, 64, 2), (999, 128, 16), (40, 32, 7)])   # last four: narrow outputs
def test_dense_mix_vs_f64(N, Fin, Fout):
    """K = 1 channel mix (residual branch of ResBlock) against an fp64 matmul: forward, dX, dW, db."""
    from dsw_amd import functional as F_

    torch.manual_seed(3)
    x = torch.randn(3, N // 3 if N % 3 == 0 else N, Fin, device=DEV)[: (3 if N % 3 == 0 else 1)].contiguous().requires_grad_(True)
    w = (torch.randn(Fout, Fin, device=DEV) / Fin ** 0.5).requires_grad_(True)   # nn.Linear layout
    b = torch.randn(Fout, device=DEV).requires_grad_(True)
    gy = torch.randn(*x.shape[:-1], Fout, device=DEV)
    y = F_.dense_mix(x, w.t(), b)
    y.backward(gy)
    x64, w64, b64, g64 = (t.detach().double().cpu() for t in (x, w, b, gy))
    y64 = x64 @ w64.t() + b64
    assert orc.max_rel_err(y, y64.numpy()) <= TOL_F64
    assert orc.max_rel_err(x.grad, (g64 @ w64).numpy()) <= TOL_F64
    assert orc.max_rel_err(w.grad, (g64.reshape(-1, Fout).t() @ x64.reshape(-1, Fin)).numpy()) <= TOL_F64
    assert orc.max_rel_err(b.grad, g64.reshape(-1, Fout).sum(0).numpy()) <= TOL_F64


@pytest.mark.parametrize("N,Kd,Fout,epi", [(5000, 768, 128, "relu"), (98304, 768, 128, "res"), (24576, 576, 256, "none"),
                                            (6144, 1536, 256, "res"), (4999, 384, 200, "none"), (1300, 1536, 512, "relu")])
def test_streaming_gemm_balanced_decomposition(N, Kd, Fout, epi):
    """Wide fp32 channel mix through dsw_cheb_fwd_ws WITH caller scratch: the streaming-W GEMM cuts its chunk steps into equal
    ranges per workgroup (tiles split between workgroups, pieces parked in the scratch, summed in workgroup order).  Against an
    fp64 matmul, bitwise repeatable, and equal (to rounding) to the same call without scratch (whole tiles per workgroup)."""
    from dsw_amd import _native

    lib = _native.load()
    torch.manual_seed(11)
    x = torch.randn(N, Kd, device=DEV)
    w = torch.randn(Kd, 1, Fout, device=DEV) / Kd ** 0.5
    b = torch.randn(Fout, device=DEV)
    r = torch.randn(N, Fout, device=DEV) if epi == "res" else None
    sc = torch.tensor([0.37], device=DEV) if epi == "res" else None
    st = torch.cuda.current_stream().cuda_stream
    nws = int(lib.dsw_cheb_fwd_workspace_bytes(1, N, Kd, Fout, 1, 0))
    assert nws > 0

    def run(ws):
        y = torch.full((N, Fout), float("nan"), device=DEV)
        rc = lib.dsw_cheb_fwd_ws(None, None, None, N, 0, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), Fout, None, 1, Kd, Fout,
                                 1, 0, st, None, 1 if epi == "relu" else 0, None if sc is None else sc.data_ptr(),
                                 None if r is None else r.data_ptr(), 0 if r is None else Fout,
                                 None if ws is None else ws.data_ptr(), 0 if ws is None else nws)
        assert rc == 0
        torch.cuda.synchronize()
        return y

    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    ws.fill_(0xA5)                       # stale flags / pieces must not matter
    y1 = run(ws)
    x_keep = x.clone()
    x.mul_(-1.7)                         # other pieces in the scratch (and in the L2s) between the two runs compared bitwise
    run(ws)
    x.copy_(x_keep)
    y2 = run(ws)
    ws.fill_(0x5A)
    y3 = run(ws)
    y0 = run(None)
    assert torch.equal(y1, y3)
    ref = x.double().cpu() @ w[:, 0].double().cpu() + b.double().cpu()
    if epi == "res":
        ref = 0.37 * ref + r.double().cpu()
    if epi == "relu":
        ref = ref.clamp_min(0)
    assert torch.equal(y1, y2)
    assert orc.max_rel_err(y1, ref.numpy()) <= TOL_F64
    assert orc.max_rel_err(y0, ref.numpy()) <= TOL_F64
    # stress of the release ordering: a flag that became visible before its piece would show up as a run that
    # differs from the others - 25 back-to-back launches, other data through the L2s in between
    for i in range(25):
        if i % 5 == 0:
            x.mul_(0.5); run(ws); x.copy_(x_keep)
        assert torch.equal(run(ws), y1), i


def test_config_driven_training_driver(tmp_path):
    """scripts_training/train_synthetic_state.py: JSON config -> model -> AR steps with Adam; the loss of a
    fixed synthetic batch must go down and stay finite (whole-path smoke through fwd, bwd, optimizer)."""
    import json
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts_training"))
    import train_synthetic_state as drv

    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      "configs/UNetSpherical/Healpix_400km/InterpPool-Graph_knn.synthetic.json")))
    cfg["model_settings"]["sampling_kwargs"]["subdivisions"] = 8
    cfg["model_settings"]["knn"] = 8
    cfg["training_settings"]["learning_rate"] = 0.002
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    losses = drv.main(["--config_file", str(path), "--steps", "12", "--warmup", "0", "--batch_size", "2",
                       "--ar_iterations", "1"])
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]


def test_equiangular_conv_and_cross_sampling_pooling():
    """BASELINE configs[4] in small: equiangular 24 x 48 k-NN Laplacian (irregular degree at the poles), K = 3,
    32 channels, interpolation pooling to a HEALPix sampling and back - every piece against the fp64 oracle."""
    from dsw_amd import sphere
    from modules.layers import ConvCheb, GeneralAvgPool, GeneralAvgUnpool, prepare_torch_laplacian
    from scipy import sparse

    fine = sphere.SphereEquiangular(nlat=24, nlon=48, k=20)
    coarse = sphere.SphereHealpix(4, nest=True, k=8)
    deg = np.diff(fine.L.tocsr().indptr)
    assert deg.max() > deg.min()                       # the stress: rows of different length
    pool_m, unpool_m = sphere.conservative_pool_matrices(fine.coords, coarse.coords)   # overlap areas (layers.py:529-581)
    lap = prepare_torch_laplacian(fine.L, lmax=1.95)
    torch.manual_seed(5)
    conv = ConvCheb(32, 32, 3, laplacian=lap).to(DEV)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    pool, unpool = GeneralAvgPool(pool_m).to(DEV), GeneralAvgUnpool(unpool_m).to(DEV)
    V = fine.n_vertices
    x = torch.randn(3, V, 32, device=DEV, requires_grad=True)
    y = conv(x)
    z, idx = pool(y)
    out = unpool(z, idx)
    gy = torch.randn_like(out)
    out.backward(gy)
    rp, ci, va = orc.csr_arrays_from_coo(conv.laplacian.cpu())
    xn, wn, bn = (t.detach().cpu().numpy() for t in (x, conv.weight, conv.bias))
    y64 = orc.cheb_forward_f64(rp, ci, va, xn, wn, bn)
    P = sparse.csr_matrix(pool_m).astype(np.float32).astype(np.float64)
    U = sparse.csr_matrix(unpool_m).astype(np.float32).astype(np.float64)
    out64 = np.stack([U @ (P @ y64[b]) for b in range(3)])
    assert orc.max_rel_err(y, y64) <= TOL_F64
    assert orc.max_rel_err(out, out64) <= TOL_F64
    g_y64 = np.stack([P.T @ (U.T @ gy[b].double().cpu().numpy()) for b in range(3)])
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, xn, wn, g_y64, True)
    assert orc.max_rel_err(x.grad, dx64) <= TOL_F64
    assert orc.max_rel_err(conv.weight.grad, dw64) <= 2 * TOL_F64
    assert orc.max_rel_err(conv.bias.grad, db64) <= 2 * TOL_F64


def test_mix_first_equals_basis_first(monkeypatch):
    """The two evaluation orders of a channel-shrinking layer agree to fp32 rounding (same inputs, both through
    the C ABI; the product library has no run-time switch, so the basis-first side is evaluated via the explicit
    basis + mix entry points)."""
    from dsw_amd import _native, functional as F_
    from dsw_amd import sphere
    from modules.layers import ConvCheb, prepare_torch_laplacian

    lib = _native.load()
    assert lib.dsw_cheb_mix_first(64, 32, 3) == 1 and lib.dsw_cheb_mix_first(32, 64, 3) == 0
    assert lib.dsw_cheb_mix_first(64, 32, 1) == 0 and lib.dsw_cheb_mix_first(64, 33, 3) == 0
    g = sphere.SphereHealpix(8, nest=True, k=8)
    lap = prepare_torch_laplacian(g.L, lmax=1.9)
    torch.manual_seed(11)
    layer = ConvCheb(64, 32, 3, laplacian=lap).to(DEV)
    with torch.no_grad():
        layer.bias.normal_(0, 0.2)
    x = torch.randn(2, 768, 64, device=DEV)
    y = layer(x)                                            # mix-first
    op = F_.get_operator(layer.laplacian)
    T = F_.cheb_basis(op, x, 3)                             # basis-first pieces
    yb = torch.empty_like(y)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.dsw_cheb_mix_fwd(x.data_ptr(), T.data_ptr(), layer.weight.data_ptr(), layer.bias.data_ptr(), yb.data_ptr(),
                              2 * 768, 64, 32, 3, 0, st)
    assert rc == 0
    torch.cuda.synchronize()
    assert orc.max_rel_err(y, yb.cpu().numpy()) <= TOL_F64


def test_c_abi_from_a_c_client(tmp_path):
    """The boundary is a C ABI, not a torch extension: tests/cabi/cabi_client.cpp (HIP runtime + include/dsw_hip.h,
    no Python, no torch) runs forward + backward through libdsw_hip.so and checks them against the plain-C oracle."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from oracle import c_oracle

    c_oracle.lib()   # make sure oracle/_build/libcheb_oracle.so exists
    libdir = os.path.join(root, "deepsphere-weather_amd", "dsw_amd")
    odir = os.path.join(root, "oracle", "_build")
    exe = str(tmp_path / "cabi_client")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "cabi", "cabi_client.cpp"), "-L", libdir, "-ldsw_hip", "-L", odir,
                    "-lcheb_oracle", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{odir}", "-o", exe],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "C-ABI CLIENT: PASS" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("shape,dt", [((3, 768, 64), torch.float32), ((2, 193, 7), torch.float32), ((4, 3072, 128), torch.bfloat16),
                                      ((1, 5, 3), torch.bfloat16)])
def test_rezero_residual(shape, dt):
    """y = w * c + r and its gradients against fp64 (odd sizes exercise the scalar tails)."""
    from dsw_amd import functional as F_

    torch.manual_seed(9)
    c = torch.randn(*shape, device=DEV, dtype=dt, requires_grad=True)
    r = torch.randn(*shape, device=DEV, dtype=dt, requires_grad=True)
    w = torch.tensor([0.37], device=DEV, dtype=dt, requires_grad=True)
    g = torch.randn(*shape, device=DEV, dtype=dt)
    y = F_.rezero_residual(c, r, w)
    y.backward(g)
    c64, r64, w64, g64 = (t.detach().double().cpu() for t in (c, r, w, g))
    tol = TOL_F64 if dt == torch.float32 else TOL_BF16
    assert orc.max_rel_err(y, (w64 * c64 + r64).numpy()) <= tol
    assert orc.max_rel_err(c.grad, (w64 * g64).numpy()) <= tol
    assert torch.equal(r.grad, g)
    ref_gw = float((g64 * c64).sum())
    assert abs(float(w.grad) - ref_gw) <= (1e-5 if dt == torch.float32 else 2e-2) * max(1.0, float((g64 * c64).abs().sum()) ** 0.5 * 10)


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the fields the driver and the judge read (short run, default workload)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 16 * 49152 * 32 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference")


@pytest.mark.parametrize("tag", ["hier", "interp"])
def test_max_pooling_golden_g8(tag):
    """GeneralMaxValPool / Unpool (segmented arg-max + gather kernels) and GeneralMaxAreaPool / Unpool on the device
    against the reference's own outputs (fixture G8): selection and values exact, gradients <= 1e-6."""
    from test_host_logic import check_g8

    check_g8(tag, device=DEV)


@pytest.mark.parametrize("dt,C,B", [(torch.float32, 7, 3), (torch.float32, 64, 2), (torch.bfloat16, 40, 2), (torch.bfloat16, 5, 1)])
def test_maxval_pool_vs_oracle_shapes(dt, C, B):
    """Max-value pooling on an irregular rectangular matrix (rows of 1..9 entries, negative weights), odd channel
    counts (scalar lanes) and both storage types, against the numpy statement of the reference's selection."""
    from dsw_amd import functional as F_
    from scipy import sparse

    rng = np.random.default_rng(12)
    vo, vi = 150, 400
    rows = np.repeat(np.arange(vo), rng.integers(1, 10, size=vo))
    cols = rng.integers(0, vi, size=rows.size)
    m = sparse.csr_matrix((rng.standard_normal(rows.size).astype(np.float32), (rows, cols)), shape=(vo, vi))
    m.sum_duplicates(); m.sort_indices()
    op = F_.CsrOperator.from_sparse_coo(orc.coo_from_scipy(m).float().to(DEV))
    x = torch.from_numpy(recipes.rand(21, (B, vi, C))).to(dt)
    xd = x.to(DEV).requires_grad_(True)
    y, sel = F_.maxval_pool(op, xd)
    y_ref, sel_ref = orc.maxval_pool_np(m.indptr, m.indices, m.data, x.float().numpy())
    assert torch.equal(sel.cpu(), torch.from_numpy(sel_ref))
    assert torch.equal(y.float().cpu(), torch.from_numpy(y_ref))
    gy = torch.from_numpy(recipes.rand(22, (B, vo, C))).to(dt)
    y.backward(gy.to(DEV))
    dx_ref = orc.maxval_pool_backward_np(sel_ref, vi, gy.float().numpy())
    assert orc.max_rel_err(xd.grad.float(), dx_ref) <= (TOL_BF16 if dt == torch.bfloat16 else TOL_F64)
    xu = torch.from_numpy(recipes.rand(23, (B, vo, C))).to(dt).to(DEV).requires_grad_(True)
    yu = F_.maxval_unpool(xu, sel, vi)
    assert torch.equal(yu.float().cpu(), torch.from_numpy(orc.maxval_unpool_np(sel_ref, vi, xu.detach().float().cpu().numpy())))
    gyu = torch.from_numpy(recipes.rand(24, (B, vi, C))).to(dt)
    yu.backward(gyu.to(DEV))
    assert torch.equal(xu.grad.float().cpu(), torch.from_numpy(orc.maxval_unpool_backward_np(sel_ref, gyu.float().numpy())))


def test_empty_shard_backward_writes_zero_parameter_gradients():
    """A rank with no samples (B < world size) must contribute exact zeros to the gradient all-reduce - for both
    evaluation orders of the layer (the mix-first branch used to leave dW / db unwritten)."""
    from modules.layers import ConvCheb

    (rp, ci, va), _, _, _, _ = _rand_case(192, 1, 8, 8, 3, seed=3)
    lap = orc.coo_from_csr_arrays(rp, ci, va, (192, 192))
    for fin, fout in ((32, 64), (64, 16)):       # basis-first, mix-first
        layer = ConvCheb(fin, fout, 3, laplacian=lap).to(DEV)
        x = torch.zeros(0, 192, fin, device=DEV, requires_grad=True)
        for _ in range(2):                        # second pass: the caching allocator hands back dirty memory
            junk = torch.full((fin * 3 * fout + fout + 64,), float("nan"), device=DEV)
            del junk
            layer.zero_grad(set_to_none=True)
            y = layer(x)
            y.backward(torch.zeros(0, 192, fout, device=DEV))
            assert torch.count_nonzero(layer.weight.grad) == 0 and torch.count_nonzero(layer.bias.grad) == 0
            assert torch.isfinite(layer.weight.grad).all() and torch.isfinite(layer.bias.grad).all()


def test_rezero_residual_unaligned_views():
    """Contiguous views with a storage offset that is not a multiple of 16 bytes (a batch slice of odd-sized rows)
    take the scalar path instead of failing."""
    from dsw_amd import functional as F_

    torch.manual_seed(2)
    base_c = torch.randn(3, 5, 7, device=DEV)
    base_r = torch.randn(3, 5, 7, device=DEV)
    c, r = base_c[1:], base_r[1:]                  # offset 35 floats = 140 bytes: not 16-byte aligned
    assert c.is_contiguous() and c.data_ptr() % 16 != 0
    c = c.detach().requires_grad_(True)
    w = torch.tensor([0.3], device=DEV, requires_grad=True)
    y = F_.rezero_residual(c, r, w)
    g = torch.randn_like(y)[...]
    y.backward(g)
    assert orc.max_rel_err(y, (0.3 * c.detach().double() + r.double()).cpu().numpy()) <= TOL_F64
    assert orc.max_rel_err(c.grad, (0.3 * g.double()).cpu().numpy()) <= TOL_F64
    assert abs(float(w.grad) - float((g.double() * c.detach().double()).sum())) <= 1e-4


@pytest.mark.parametrize("use_graph,min_tiles", [(False, None), (True, None), (False, 1)])
def test_ar_training_steps_golden_g9(use_graph, min_tiles):
    """Three autoregressive optimisation steps of the training driver (UNetSpherical nside=8, WeightedMSELoss, Adam
    eps=1e-7; two forwards per step) against the reference run of fixture G9 - launched eagerly and as a replayed HIP
    graph of the whole step (zero_grad + forwards + backward + Adam).

    The eager runs compare gradient fingerprints, which on the 48-node level react to a single ReLU mask: the ~400
    pre-activations the reference found within 1e-4 of zero get the reference's recorded mask (`pin_g9_ties`), so the
    comparison does not depend on the order in which a kernel sums a row.  `min_tiles = 1` proves it: the fused two-hop
    kernels (another summation order) then also run the 192- and 48-node levels, which the product keeps on one launch
    per hop for speed - the fixture no longer constrains that choice."""
    from dsw_amd import functional as F_
    from test_host_logic import build_g9_trainer, check_g9, pin_g9_ties

    keep = F_.MIN_CLUSTERED_TILES
    try:
        if min_tiles is not None:
            F_.MIN_CLUSTERED_TILES = min_tiles
            F_.invalidate_operator_caches()
        trainer, g, names = build_g9_trainer(DEV, use_graph=use_graph)
        flipped = None
        if use_graph:
            assert trainer.graph is not None and trainer.launch.startswith("hip graph"), trainer.launch
        else:
            flipped, _handles = pin_g9_ties(trainer.model, g)
        losses = check_g9(trainer, g, names)
        print("G9 losses", losses, "reference", g["losses"], "pinned masks that differed:", flipped)
        if flipped is not None:
            assert flipped["n"] <= 12, flipped       # only elements with |z| of a few 1e-7 may land on the other side
    finally:
        F_.MIN_CLUSTERED_TILES = keep
        if min_tiles is not None:
            F_.invalidate_operator_caches()


@pytest.mark.parametrize("V,B,Fin,Fout,K,dt", [
    (768, 3, 32, 64, 3, torch.float32),     # whole-forward kernel (dsw_fwd3.hip)
    (768, 2, 64, 128, 3, torch.float32),    # x3 GEMM epilogue
    (192, 2, 512, 256, 3, torch.float32),   # mix-first: in-place pass after the Clenshaw recurrence
    (192, 2, 256, 384, 2, torch.float32),   # streaming-W x3s GEMM epilogue
    (192, 3, 18, 7, 3, torch.float32),      # exact-fp32 GEMM (unaligned shapes)
    (768, 2, 64, 128, 3, torch.bfloat16),   # bf16 packed epilogue
    (192, 2, 24, 40, 2, torch.bfloat16),
])
def test_convblock_fused_bias_relu_vs_oracle(V, B, Fin, Fout, K, dt):
    """ConvBlock = conv + bias + ReLU (my_models_graph.py:104-118) with the activation in the kernel epilogue: forward,
    and the backward through the ReLU mask (dX, dW, db), against the fp64 oracle of the same composition."""
    import modules.my_models_graph as arch

    (rp, ci, va), x, w, b, gy = _rand_case(V, B, Fin, Fout, K, seed=4000 + Fin + Fout, bias=True)
    q = lambda a: torch.from_numpy(a).to(dt)
    xq, wq, bq, gyq = q(x), q(w), q(b), q(gy)
    lap = orc.coo_from_csr_arrays(rp, ci, va, (V, V))
    blk = arch.ConvBlock(Fin, Fout, laplacian=lap, kernel_size=K, conv_type="graph", bias=True, activation=True,
                         activation_fun="relu")
    blk.conv.set_parameters(wq.float(), bq.float())
    blk = blk.to(DEV).to(dt)
    xd = xq.to(DEV).requires_grad_(True)
    y = blk(xd)
    assert (y >= 0).all()
    y.backward(gyq.to(DEV))
    f = lambda t: t.float().numpy()
    z64 = orc.cheb_forward_f64(rp, ci, va, f(xq), f(wq), f(bq))
    y64 = np.maximum(z64, 0.0)
    # the mask is taken from the DEVICE output (an element within rounding of 0 may legitimately land on either side)
    mask = (y.detach().float().cpu().numpy() > 0)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, f(xq), f(wq), f(gyq) * mask, True)
    tol = TOL_BF16 if dt == torch.bfloat16 else TOL_F64
    assert orc.max_rel_err(y.float(), y64) <= tol
    assert orc.max_rel_err(xd.grad.float(), dx64) <= tol
    assert orc.max_rel_err(blk.conv.weight.grad.float(), dw64) <= 2 * tol
    assert orc.max_rel_err(blk.conv.bias.grad.float(), db64) <= 2 * tol
    # and the mask itself: only elements whose pre-activation is within rounding of zero may differ
    flips = (mask != (z64 > 0))
    assert np.abs(z64[flips]).max(initial=0.0) <= tol * np.abs(z64).max()


@pytest.mark.parametrize("bn_before_act,pool_method", [(True, "interp"), (False, "maxval"), (False, "maxarea")])
def test_unet_variants_vs_oracle_backed_run(bn_before_act, pool_method):
    """The other branches of the model code the configs can select (my_models_graph.py:104-118, :401-412): batch norm
    on either side of the activation (the conv then carries no bias and the activation is NOT fused when the norm sits
    in between) and the max-value / max-area poolings - the same model object on the device and, with the fp64 oracle
    behind the layers, on the CPU."""
    import modules.my_models_graph as arch
    from dsw_amd import functional
    from _oracle_backend import OracleBackend

    V = 768
    tensor_info = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
    }
    torch.manual_seed(3)
    model = arch.UNetSpherical(tensor_info, sampling="healpix", sampling_kwargs={"subdivisions": 8, "nest": True},
                               kernel_size_conv=3, conv_type="graph", graph_type="knn", knn=8, pool_method=pool_method,
                               batch_norm=True, batch_norm_before_activation=bn_before_act)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("rezero_weight"):
                p.fill_(0.4)
            elif n.endswith("bn.weight"):
                p.fill_(0.8)       # (the last BN of a block starts at 0: give every branch a signal)
    model.train()
    x = torch.from_numpy(recipes.rand(71, (3, 3, V, 6)))
    target = torch.from_numpy(recipes.rand(72, (3, 1, V, 2)))

    def run(m, dev):
        m.zero_grad(set_to_none=True)
        y = m(x.to(dev))
        loss = ((y - target.to(dev)) ** 2).mean()
        loss.backward()
        return y.detach().cpu(), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}

    import copy
    from test_host_logic import pin_relu_masks, record_relu_masks

    ref_model = copy.deepcopy(model)          # BatchNorm running statistics are updated by a forward: separate copies
    functional.set_test_backend(OracleBackend())
    try:
        masks, _h = record_relu_masks(ref_model)
        y_ref, g_ref = run(ref_model, "cpu")
    finally:
        functional.set_test_backend(None)
    # the device run takes the reference's decision for the (rare) pre-activations that land on the other side of zero
    # by ~1e-7: with batch statistics over 144 samples per channel one such element moves whole weight gradients by 1e-2
    flipped, _h = pin_relu_masks(model, masks)
    y_dev, g_dev = run(model.to(DEV), DEV)
    print("ReLU decisions that differed (|z| < 1e-4):", flipped)
    assert flipped["n"] <= max(8, 2e-6 * flipped["total"]), flipped
    assert set(g_dev) == set(g_ref)
    if pool_method == "maxval":
        # an arg-max is discontinuous: where two candidates tie to within fp32 rounding, the fp32 device run and the
        # fp64-backed run may legitimately pick different cells; such flips are rare and local - bound their share
        def close(a, b, tol):
            a, b = a.double().numpy(), b.double().numpy()
            return float(np.mean(np.abs(a - b) > tol * np.abs(b).max()))
        assert close(y_dev, y_ref, 2e-5) <= 2e-3
        for n in g_ref:
            assert close(g_dev[n], g_ref[n], 2e-4) <= 2e-2, n
    else:
        assert orc.max_rel_err(y_dev, y_ref) <= 2e-5
        for n in g_ref:
            assert orc.max_rel_err(g_dev[n], g_ref[n]) <= 2e-4, n


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_concat_in_place_strided_kernels(dt):
    """dsw_spmm_csr_ld / dsw_rezero_residual_fwd_ld behind the decoder's copy-free concatenation."""
    from test_host_logic import check_concat_in_place

    check_concat_in_place(DEV, dt, TOL_F64 if dt == torch.float32 else TOL_BF16)


@pytest.mark.parametrize("dt,C,width", [(torch.float32, 32, 96), (torch.float32, 20, 20), (torch.bfloat16, 64, 192),
                                        (torch.float32, 6, 10)])
def test_remap_fork_adds_the_other_consumers_gradient_in_the_product(dt, C, width):
    """`sparse_remap_fork`: forward = plain remap, backward dX = g_other + M^T dY with g_other a channel slice of a wider
    gradient tensor (strided Z operand of dsw_spmm_csr_ld; C = 6 of 10 floats: a slice the strided entry point does not take, copied first) -
    against fp64."""
    from dsw_amd import functional as F_

    Vs, Vd, B = 768, 192, 3
    from scipy import sparse as sp
    rng = np.random.default_rng(4)
    dense = (rng.random((Vd, Vs)) < 0.02) * rng.random((Vd, Vs))
    dense[np.arange(Vd), rng.integers(0, Vs, Vd)] += 0.5             # no empty rows; a few source nodes stay unused
    pool_mat = sp.csr_matrix(dense.astype(np.float32))
    from modules.layers import convert_to_torch_sparse
    op = F_.get_operator(convert_to_torch_sparse(pool_mat).to(DEV))
    x = torch.from_numpy(recipes.rand(81, (B, Vs, C))).to(DEV).to(dt).requires_grad_(True)
    wide = torch.from_numpy(recipes.rand(82, (B, Vs, width))).to(DEV).to(dt)
    g_other = wide[..., width - C:]                                   # what a concatenation's backward hands out
    gy = torch.from_numpy(recipes.rand(83, (B, Vd, C))).to(DEV).to(dt)
    x_again, y = F_.sparse_remap_fork(op, x)
    assert x_again.data_ptr() == x.data_ptr() and x_again.stride() == x.stride()
    torch.autograd.backward([x_again, y], [g_other, gy])
    M = pool_mat.astype(np.float64)
    x64 = x.detach().float().cpu().numpy().astype(np.float64)
    y64 = np.stack([M @ x64[b] for b in range(B)])
    dx64 = g_other.float().cpu().numpy().astype(np.float64) + np.stack(
        [M.T @ gy[b].float().cpu().numpy().astype(np.float64) for b in range(B)])
    tol = TOL_F64 if dt == torch.float32 else TOL_BF16
    assert orc.max_rel_err(y.float(), y64) <= tol
    assert orc.max_rel_err(x.grad.float(), dx64) <= tol
    # only one of the two consumers sends a gradient
    x.grad = None
    x_again, y = F_.sparse_remap_fork(op, x)
    y.backward(gy)
    assert orc.max_rel_err(x.grad.float(), dx64 - g_other.float().cpu().numpy()) <= tol
    x.grad = None
    x_again, y = F_.sparse_remap_fork(op, x)
    x_again.backward(g_other)
    assert torch.equal(x.grad, g_other)


def test_strided_entry_points_reject_bad_strides():
    from dsw_amd import functional as F_, _native

    lib = _native.load()
    x = torch.randn(2, 8, 8, device=DEV)
    y = torch.empty(2, 8, 8, device=DEV)
    w = torch.ones(1, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dsw_rezero_residual_fwd_ld(x.data_ptr(), x.data_ptr(), w.data_ptr(), y.data_ptr(), 16, 8, 4, 0, st) == -1
    assert lib.dsw_rezero_residual_fwd_ld(x.data_ptr(), x.data_ptr(), w.data_ptr(), y.data_ptr(), 16, 6, 6, 0, st) == -5
    with pytest.raises(ValueError):
        F_.rezero_residual(x, x, w, out=torch.empty(2, 8, 9, device=DEV)[..., 1:])   # misaligned slice


@pytest.mark.parametrize("sampling,knn,Fin,Fout,K,B,dt", [
    ("ring", 8, 32, 64, 3, 3, torch.float32),        # whole-forward kernel + fused backward on clustered tiles
    ("ring", 20, 32, 32, 3, 2, torch.float32),       # k = 20 variants (5 / 3 / 1 slots, two workgroups per CU)
    ("equiangular", 20, 32, 64, 3, 2, torch.float32),
    ("equiangular", 20, 64, 32, 4, 2, torch.float32),  # mix-first, K = 4 (second first-hop operand), 256-byte rows
    ("ring", 20, 64, 128, 5, 2, torch.bfloat16),
    ("scattered", 10, 32, 64, 3, 2, torch.float32),   # random points on the sphere in RANDOM row order: no structure at all
])
def test_conv_on_non_local_row_orders_takes_clustered_tiles(sampling, knn, Fin, Fout, K, B, dt):
    """HEALPix ring order / equiangular row-major: strips of consecutive rows do not fit LDS, the plan's tiles are
    clustered from the graph (explicit row sets) - forward and backward against the fp64 oracle through that path."""
    from dsw_amd import functional as F_, sphere
    from modules.layers import ConvCheb

    if sampling == "scattered":
        rng = np.random.default_rng(5)
        pts = rng.standard_normal((3000, 3))
        pts /= np.linalg.norm(pts, axis=1, keepdims=True)
        graph_l = sphere.knn_graph_laplacian(pts, knn)[1]
    else:
        g = sphere.SphereHealpix(32, nest=False, k=knn) if sampling == "ring" else sphere.SphereEquiangular(nlat=72, nlon=144, k=knn)
        graph_l = g.L
    lap = orc.prepare_laplacian_fixed_lmax(graph_l, 1.9)
    rp, ci, va = orc.csr_arrays_from_coo(lap)
    V = len(rp) - 1
    layer = ConvCheb(Fin, Fout, K, laplacian=lap)
    w = recipes.rand(71, (Fin, K, Fout), np.sqrt(2.0 / (Fin * K)))
    b = recipes.rand(72, (Fout,), 0.1)
    layer.set_parameters(torch.from_numpy(w), torch.from_numpy(b))
    layer = layer.to(DEV).to(dt)
    plan = F_.get_operator(layer.laplacian).hop2_plan(128)
    assert plan is not None and plan.explicit_tiles, "expected tiles clustered from the graph"
    x = recipes.rand(73, (B, V, Fin))
    gy = recipes.rand(74, (B, V, Fout))
    y, dx, dw, db = _run_layer(layer, torch.from_numpy(x).to(DEV).to(dt), torch.from_numpy(gy).to(DEV).to(dt))
    if dt == torch.bfloat16:
        x, w, b, gy = (torch.from_numpy(a).to(dt).float().numpy() for a in (x, w, b, gy))
    y64 = orc.cheb_forward_f64(rp, ci, va, x, w, b)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, x, w, gy, True)
    tol = TOL_F64 if dt == torch.float32 else TOL_BF16
    assert orc.max_rel_err(y.float(), y64) <= tol
    assert orc.max_rel_err(dx.float(), dx64) <= tol
    assert orc.max_rel_err(dw.float(), dw64) <= 2 * tol
    assert orc.max_rel_err(db.float(), db64) <= 2 * tol


@pytest.mark.parametrize("sampling,kwargs", [("icosahedral", {"subdivisions": 16}), ("cubed", {"subdivisions": 24}),
                                             ("gauss", {"nlat": 48, "nlon": "ecmwf-octahedral"})])
def test_conv_on_the_other_samplings_vs_oracle(sampling, kwargs):
    """f2: ConvCheb forward + backward on the k = 20 graphs of the samplings added in round 4 (the sizes of the reference's
    Icosahedral_400km / Cubed_400km / O24 configs) against the fp64 oracle - whatever tiles the plan builder finds for
    their row orders."""
    from modules.layers import ConvCheb
    from modules.utils_models import get_pygsp_graph

    g = get_pygsp_graph(sampling, dict(kwargs), knn=20)
    lap = orc.prepare_laplacian_fixed_lmax(g.L, 1.9)
    rp, ci, va = orc.csr_arrays_from_coo(lap)
    V = len(rp) - 1
    Fin, Fout, K, B = 32, 64, 3, 3
    layer = ConvCheb(Fin, Fout, K, laplacian=lap)
    w = recipes.rand(81, (Fin, K, Fout), np.sqrt(2.0 / (Fin * K)))
    b = recipes.rand(82, (Fout,), 0.1)
    layer.set_parameters(torch.from_numpy(w), torch.from_numpy(b))
    layer = layer.to(DEV)
    x = recipes.rand(83, (B, V, Fin))
    gy = recipes.rand(84, (B, V, Fout))
    y, dx, dw, db = _run_layer(layer, torch.from_numpy(x).to(DEV), torch.from_numpy(gy).to(DEV))
    y64 = orc.cheb_forward_f64(rp, ci, va, x, w, b)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, x, w, gy, True)
    assert orc.max_rel_err(y, y64) <= TOL_F64 and orc.max_rel_err(dx, dx64) <= TOL_F64
    assert orc.max_rel_err(dw, dw64) <= 2 * TOL_F64 and orc.max_rel_err(db, db64) <= 2 * TOL_F64


@pytest.mark.parametrize("pool_method", ["interp", "maxval"])
def test_unet_bf16_storage_tracks_fp32(pool_method):
    """`model.to(torch.bfloat16)` end to end (every kernel's bf16 variant, the copy-free concatenation on 2-byte rows,
    operators kept at fp32): output and parameter gradients stay within bf16 rounding of the fp32 run."""
    import modules.my_models_graph as arch
    from test_host_logic import build_g5_model

    base, g, names = build_g5_model(DEV)
    if pool_method != "interp":
        tensor_info = {
            "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
            "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
            "input_shape_info": {"dynamic": {"node": 768}}, "output_shape_info": {"dynamic": {"node": 768}},
        }
        other = arch.UNetSpherical(tensor_info, sampling="healpix", sampling_kwargs={"subdivisions": 8, "nest": True},
                                   kernel_size_conv=3, conv_type="graph", graph_type="knn", knn=20, pool_method=pool_method)
        sd = {k: v for k, v in base.state_dict().items() if "pool" not in k}
        other.load_state_dict(sd, strict=False)
        base = other.to(DEV)
    x = torch.from_numpy(recipes.rand(501, (2, 3, 768, 6))).to(DEV)
    target = torch.from_numpy(recipes.rand(502, (2, 1, 768, 2))).to(DEV)

    def run(model, dt):
        model = model.to(dt)
        model.zero_grad(set_to_none=True)
        y = model(x.to(dt))
        ((y.float() - target) ** 2).mean().backward()
        return y.detach().float(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}

    y32, g32 = run(base, torch.float32)
    y16, g16 = run(base, torch.bfloat16)
    assert base.conv1.convblock1.conv.laplacian.dtype == torch.float32      # operators are not rounded by the cast
    assert torch.isfinite(y16).all()
    # max-value pooling: bf16 rounding moves the arg-max of near-ties to another fine cell, a discrete change
    tol_y, tol_g = (6e-2, 0.15) if pool_method == "interp" else (0.25, 0.6)
    assert orc.max_rel_err(y16, y32.cpu().numpy()) <= tol_y
    worst = max(float((g16[n] - g32[n]).norm() / (g32[n].norm() + 1e-12)) for n in g32)
    assert worst <= tol_g, worst


def test_bench_two_ranks_self_launched():
    """`python bench.py --gpus 2` (no torchrun around it) starts two ranks itself, and the line says
    so.  One GPU here, so both ranks share it and talk through gloo (DSW_DIST_BACKEND); the launch path, the world-size
    check, the bucket + exchange after every step, the timed region and the max-over-ranks are the RCCL run's."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DSW_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2", "--min-timed-ms", "100"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["config"]["batch_per_gpu"] == 16
    assert "self-launch" in out["config"]["launcher"]
    assert out["grad_sync"]["identical"] and out["grad_sync"]["grad_l2"] > 0, out["grad_sync"]
    assert out["allreduce_us"] > 0 and out["allreduce"]["bytes"] == 4 * (32 * 3 * 64 + 64)
    assert out["value"] == pytest.approx(2 * 16 * 49152 * 32 / (out["ms_per_step"] * 1e-3), rel=1e-6)
    assert "roofline" not in out and "cpu_baseline" not in out          # N = 1 legs only


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


@pytest.mark.parametrize("ranks,global_batch", [(4, 6), (8, 5)])
def test_bench_many_ranks_ragged_and_empty_shards(ranks, global_batch):
    """the N > 1 path with 4 and 8 ranks before the driver's 8-GPU box sees it - one GPU here, so the
    ranks share it over gloo.  `--global-batch` shards with shard_batch: 6 over 4 ranks = 2, 2, 1, 1 (ragged), 5 over 8
    ranks = five ranks with one sample and three EMPTY shards, whose backward must contribute exact zeros to the
    exchange.  Every rank ends with bit-identical, non-zero averaged gradients."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DSW_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--global-batch", str(global_batch),
                        "--steps", "6", "--warmup", "1", "--min-timed-ms", "20"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "strong" and out["config"]["global_batch"] == global_batch
    assert out["grad_sync"]["identical"] and out["grad_sync"]["grad_l2"] > 0, out["grad_sync"]
    assert out["allreduce"]["bytes"] == 4 * (32 * 3 * 64 + 64)
    assert out["value"] == pytest.approx(global_batch * 49152 * 32 / (out["ms_per_step"] * 1e-3), rel=1e-6)


def test_bench_one_gpu_line_is_the_same_under_torchrun():
    """`python bench.py --gpus 1` and the driver-style `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`
    must print the same kind of line (same keys, same config apart from the launcher note, N = 1 legs present in both)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    args = ["--gpus", "1", "--steps", "10", "--warmup", "2", "--min-timed-ms", "50", "--no-cpu-baseline"]
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), *args],
                        env=env, capture_output=True, text=True, timeout=900)
    assert tr.returncode == 0, tr.stderr[-2000:]
    a, b = (json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]) for r in (plain, tr))
    assert sorted(a) == sorted(b), (sorted(a), sorted(b))
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["scaling"] == b["scaling"] == "weak"
    ca, cb = dict(a["config"]), dict(b["config"])
    ca.pop("launcher", None); cb.pop("launcher", None)
    assert ca == cb, (ca, cb)
    assert "roofline" in a and "roofline" in b and sorted(a["roofline"]) == sorted(b["roofline"])
    assert abs(a["ms_per_step"] - b["ms_per_step"]) <= 0.15 * a["ms_per_step"]


def test_bench_one_rank_world_captures_the_exchange():
    """The RCCL exchange recorded INTO the step graph (what N > 1 runs replay): a one-rank RCCL world on this box
    (DSW_FORCE_GRAD_SYNC=1).  The captured graph must reproduce the eager step + all-reduce, and the bench must say which
    of the two launch modes it timed."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSW_DIST_BACKEND")}
    env.update(DSW_FORCE_GRAD_SYNC="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "2", "--no-cpu-baseline",
                        "--no-roofline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(out["config"]["launch"], out.get("allreduce"))
    assert out["n_gpus"] == 1 and "all-reduce" in out["config"]["launch"]
    assert "captured in the graph" in out["config"]["launch"], (out["config"]["launch"], out["allreduce"], r.stderr[-1500:])
    assert "20 steps per graph" in out["config"]["launch"]        # --steps 20: one replay of a 20-step graph


def test_bench_under_foreign_torchrun_walks_the_ladder_in_place():
    """the driver starts the N > 1 bench under ITS torchrun, so nobody re-launches a failed attempt.  A
    first replay of the graph-captured exchange that never returns (simulated: DSW_BENCH_TEST_HANG=replay) must end in ONE
    JSON line from the next rung - the ranks re-execute themselves into a new process group (own store prefix) with the
    collectives issued after the replay - and the remaining rungs (collectives outside the graph, eager launches) must
    each print one complete line as well: `allreduce_us`, `grad_sync.identical`, `config.grad_handling`."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSW_DIST_BACKEND", "DSW_PG_ATTEMPT",
                                                             "DSW_BENCH_LAUNCHER", "DSW_BENCH_COLLECTIVES", "DSW_BENCH_NO_GRAPH")}
    args = ["--gpus", "1", "--steps", "20", "--warmup", "2", "--min-timed-ms", "100", "--no-cpu-baseline", "--no-roofline"]

    def run(extra):
        env = dict(base, DSW_FORCE_GRAD_SYNC="1", **extra)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                            "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), *args],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2500:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        assert len(lines) == 1, (len(lines), r.stderr[-1500:])
        out = json.loads(lines[0])
        assert out["grad_sync"]["identical"] is True and out["allreduce_us"] > 0 and "bucket" in out["config"]["grad_handling"]
        return out, r.stderr

    out, err = run({"DSW_BENCH_TEST_HANG": "replay", "DSW_BENCH_GUARD_S": "6"})
    assert "re-executing" in err, err[-1500:]
    assert "issued after every replay" in out["config"]["launch"] and "rung 2" in out["config"]["ladder"], out["config"]
    out, _ = run({})
    assert "captured in the graph" in out["config"]["launch"] and "ladder" not in out["config"], out["config"]
    out, _ = run({"DSW_BENCH_COLLECTIVES": "eager"})
    assert "issued after every replay" in out["config"]["launch"], out["config"]
    out, _ = run({"DSW_BENCH_COLLECTIVES": "eager", "DSW_BENCH_NO_GRAPH": "1"})
    assert out["config"]["launch"].startswith("eager"), out["config"]


@pytest.mark.parametrize("dt", [torch.float32])
def test_resblock_fused_tail_vs_fp64(dt):
    """ReZero scale + residual add in the epilogue of the block's last convolution
    (dsw_cheb_fwd_res: resident-panel x3 GEMM 64 -> 128, mix-first plane GEMM 128 -> 64 and the 64 -> 2 output layer,
    32 -> 32 with an identity residual, an unaligned 7 -> 12 -> 5 block on the exact-fp32 kernels), the backward on the
    unscaled dY (dsw_cheb_bwd_res + dsw_rezero_param_grads) and the block input's gradient added inside the residual
    map's dgrad GEMM - fused and plain evaluation against the reference's op order in fp64."""
    from test_host_logic import check_resblock_tail

    worst = check_resblock_tail(DEV, dt, TOL_F64 if dt == torch.float32 else TOL_BF16)
    print("ResBlock tail, worst max-rel error", worst)


def test_resblock_fused_tail_wide_layers():
    """The same on the streaming-W GEMM (Kd = 3 * 192 = 576 -> 256 columns: ts_gemm_x3s) and a 512 -> 256 mix-first layer."""
    from test_host_logic import check_resblock_tail

    check_resblock_tail(DEV, torch.float32, TOL_F64, shapes=[(2, 128, (192, 256)), (2, 256, (512, 256)), (2, 512, (256, 128))])


def test_bench_two_ranks_under_torchrun():
    """The driver's other way of starting N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` (here both ranks on the one GPU, gloo).  bench.py must
    notice the torchrun environment (no second launcher level) and report the same things as the self-launched run."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DSW_DIST_BACKEND"] = "gloo"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10",
                        "--warmup", "2", "--min-timed-ms", "100"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and "launcher" not in out["config"]
    assert out["grad_sync"]["identical"] and out["allreduce_us"] > 0


@pytest.mark.parametrize("Fin,Fout,K", [(18, 64, 3), (24, 40, 2), (48, 32, 3), (18, 128, 1)])
def test_unaligned_input_width_runs_zero_padded(Fin, Fout, K):
    """Input widths that are not whole 32-channel chunks (the U-Net's first layer: 18) are evaluated zero-padded on the
    aligned kernels (`functional._padded_width`): forward and every gradient against the fp64 oracle, and against the
    unpadded evaluation of the same layer (exact-fp32 kernels) - the padding must be invisible."""
    from dsw_amd import functional as F_
    from modules.layers import ConvCheb

    assert F_._padded_width(Fin, torch.float32) == (Fin + 31) // 32 * 32 and F_._padded_width(7, torch.float32) == 7
    assert F_._padded_width(Fin, torch.bfloat16) == Fin and F_._padded_width(64, torch.float32) == 64
    (rp, ci, va), x, w, b, gy = _rand_case(768, 3, Fin, Fout, K, seed=300 + Fin, bias=True)
    lap = orc.coo_from_csr_arrays(rp, ci, va, (768, 768))
    layer = ConvCheb(Fin, Fout, K, laplacian=lap, bias=True)
    layer.set_parameters(torch.from_numpy(w), torch.from_numpy(b))
    layer = layer.to(DEV)
    res = {}
    for pad in (True, False):
        F_.PAD_INPUT_CHANNELS = pad
        layer.zero_grad(set_to_none=True)
        try:
            res[pad] = _run_layer(layer, torch.from_numpy(x).to(DEV), torch.from_numpy(gy).to(DEV))
        finally:
            F_.PAD_INPUT_CHANNELS = True
    y64 = orc.cheb_forward_f64(rp, ci, va, x, w, b)
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, x, w, gy, True)
    for pad in (True, False):
        y, dx, dw, db = res[pad]
        assert dx.shape == x.shape and dw.shape == w.shape
        for got, ref in ((y, y64), (dx, dx64), (dw, dw64), (db, db64)):
            assert orc.max_rel_err(got, ref) <= TOL_F64, pad


def test_training_driver_one_rank_world_whole_step_graph(tmp_path):
    """The training driver's N > 1 graph mode in a one-rank RCCL world (DSW_FORCE_GRAD_SYNC=1, --graph): the exchange and
    the Adam update are recorded into the step graph ("whole step incl. the RCCL gradient all-reduce"), and the losses
    are those of the plain single-process run (averaging over one rank changes nothing)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = json.load(open(os.path.join(root, "configs/UNetSpherical/Healpix_400km/InterpPool-Graph_knn.synthetic.json")))
    cfg["model_settings"]["sampling_kwargs"]["subdivisions"] = 8
    cfg["model_settings"]["knn"] = 8
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    base = [sys.executable, os.path.join(root, "scripts_training", "train_synthetic_state.py"), "--config_file", str(path),
            "--steps", "6", "--warmup", "0", "--batch_size", "2", "--ar_iterations", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DSW_DIST_BACKEND")}
    outs = []
    for extra_env, extra_args in (({}, []), ({"DSW_FORCE_GRAD_SYNC": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())},
                                             ["--graph"])):
        r = subprocess.run(base + extra_args, env=dict(env, **extra_env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    plain, ranked = outs
    assert plain["launch"] == "hip graph: whole step"
    assert ranked["launch"] == "hip graph: whole step incl. the RCCL gradient all-reduce", ranked["launch"]
    assert abs(ranked["loss_first"] - plain["loss_first"]) <= 1e-6 * abs(plain["loss_first"])
    assert abs(ranked["loss_last"] - plain["loss_last"]) <= 1e-4 * abs(plain["loss_last"])


def test_direct_gradient_accumulation_matches_autograd():
    """GradBucket.direct_accumulation(): the weight-gradient kernels add into the parameters' gradient buffers
    (dsw_cheb_bwd_res, accumulate_dw) instead of returning tensors for autograd to add - a model whose layers are applied
    twice per backward (an autoregressive window) must end up with the same gradients either way, the first (padded,
    18-channel) layer and the column-major residual maps included."""
    from dsw_amd.parallel import GradBucket
    from test_host_logic import build_g5_model

    model, g, names = build_g5_model(DEV)
    x = torch.from_numpy(recipes.rand(77, (2, 3, 768, 6))).to(DEV)
    t = torch.from_numpy(recipes.rand(78, (2, 1, 768, 2))).to(DEV)
    bucket = GradBucket(model.parameters(), overlap=False)
    grads = {}
    for direct in (False, True):
        bucket.direct_accumulation(direct)
        assert hasattr(model.conv1.convblock2.conv.weight, "_dsw_grad_acc") == direct
        bucket.zero()
        y1 = model(x)
        x2 = torch.cat((x[:, 1:], torch.cat((x[:, -1:, :, :4], y1), dim=3)), dim=1)      # second forward sees the first's output
        loss = ((y1 - t) ** 2).mean() + ((model(x2) - t) ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        assert all(p.grad is bucket.views[p] for p in bucket.params)
        grads[direct] = bucket.bucket.clone()
    ref = grads[False]
    assert float(ref.abs().max()) > 0
    assert float((grads[True] - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
