"""Element-wise parity of the HIP path against the fp64 oracle at the FULL sizes of BASELINE.json's configs.  GPU only.

The small-size tests (test_hip_parity.py) exercise every code path once; these run the kernels in the very geometry
the benchmarks use - 786 432-row slabs of the fused backward GEMM pass, the K >= 4 fused adjoint pairs with their
spare planes, the bf16 raw-copy wgrad, the UNet's wide layers at nside=32, the 80 000-node equiangular operator -
and compare EVERY element of y, dX, dW, db with the oracle (numpy fp64: seconds to a minute per case).

Tolerances (normalised by max|ref|): fp32 <= 2e-6 (y, dX) / <= 1e-5 (dW, db: fp32 sums over 786 432
rows); bf16 storage <= 1e-2.  Measured errors are appended to gpurun_out/parity_fullsize.json for DESIGN.md.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cheb_oracle as orc
import recipes

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    from dsw_amd import _native

    _native.load()


def _record(case, errs):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_fullsize.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[case] = {k: float("%.3e" % v) for k, v in errs.items()}
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print("parity[%s]: %s" % (case, {k: "%.2e" % v for k, v in errs.items()}))


def _healpix_operator(nside, knn, lmax=1.9):
    from dsw_amd import sphere
    from modules.layers import prepare_torch_laplacian

    return prepare_torch_laplacian(sphere.SphereHealpix(nside, nest=True, k=knn).L, lmax=lmax)


def _layer_case(lap, B, Fin, Fout, K, dtype, seed):
    """ConvCheb forward + backward on the device and in fp64 on the host, same (storage-rounded) inputs."""
    from modules.layers import ConvCheb

    V = lap.shape[0]
    x = recipes.rand(seed, (B, V, Fin))
    w = recipes.rand(seed + 1, (Fin, K, Fout), np.sqrt(2.0 / (Fin * K)))
    b = recipes.rand(seed + 2, (Fout,), 0.1)
    gy = recipes.rand(seed + 3, (B, V, Fout))
    q = lambda a: torch.from_numpy(a).to(dtype)     # storage-representable inputs: only the arithmetic is compared
    xq, wq, bq, gyq = q(x), q(w), q(b), q(gy)
    layer = ConvCheb(Fin, Fout, K, laplacian=lap, bias=True)
    layer.set_parameters(wq.float(), bq.float())
    layer = layer.to(DEV).to(dtype)
    xd = xq.to(DEV).requires_grad_(True)
    y = layer(xd)
    y.backward(gyq.to(DEV))
    torch.cuda.synchronize()
    # the oracle sees the operator the kernels see (a module cast to bf16 keeps it fp32: see ConvCheb._apply)
    rp, ci, va = orc.csr_arrays_from_coo(layer.laplacian.cpu())
    f = lambda t: t.float().numpy()
    y64 = orc.cheb_forward_f64(rp, ci, va, f(xq), f(wq), f(bq))
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, f(xq), f(wq), f(gyq), True)
    return {
        "y": orc.max_rel_err(y.float(), y64), "dx": orc.max_rel_err(xd.grad.float(), dx64),
        "dw": orc.max_rel_err(layer.weight.grad.float(), dw64), "db": orc.max_rel_err(layer.bias.grad.float(), db64),
    }


@pytest.mark.parametrize("knn", [8, 20])
def test_ns_full_size_forward_backward_vs_oracle(knn):
    """North-star shape (nside=64, K=3, 32->64, B=16, fp32), both stencils: every element of y, dX, dW, db."""
    errs = _layer_case(_healpix_operator(64, knn), 16, 32, 64, 3, torch.float32, seed=3000 + knn)
    _record("ns_k%d_fp32" % knn, errs)
    assert errs["y"] <= 2e-6 and errs["dx"] <= 2e-6
    assert errs["dw"] <= 1e-5 and errs["db"] <= 1e-5


def test_c3_full_size_forward_backward_vs_oracle():
    """BASELINE configs[2] (nside=64, K=5, 64->128, B=16, bf16 storage): fused pairs + spare planes, bf16 wgrad."""
    errs = _layer_case(_healpix_operator(64, 8), 16, 64, 128, 5, torch.bfloat16, seed=3100)
    _record("c3_k8_bf16", errs)
    assert max(errs.values()) <= 1e-2     # measured 2.4e-3 .. 4.5e-3 (round 2)


def test_c3_shape_fp32_full_size_vs_oracle():
    """The same K=5 64->128 layer in fp32 (wide basis-first layer: K >= 4 adjoint pairs at full size)."""
    errs = _layer_case(_healpix_operator(64, 8), 4, 64, 128, 5, torch.float32, seed=3200)
    _record("c3shape_k8_fp32_B4", errs)
    assert errs["y"] <= 2e-6 and errs["dx"] <= 2e-6
    assert errs["dw"] <= 1e-5 and errs["db"] <= 1e-5


def test_mix_first_full_size_vs_oracle():
    """A channel-shrinking layer (mix-first order, Clenshaw recurrence) at nside=64: 64->32, K=3, B=8."""
    errs = _layer_case(_healpix_operator(64, 20), 8, 64, 32, 3, torch.float32, seed=3300)
    _record("mixfirst_k20_fp32", errs)
    assert errs["y"] <= 2e-6 and errs["dx"] <= 2e-6
    assert errs["dw"] <= 1e-5 and errs["db"] <= 1e-5


def test_c5_equiangular_full_size_vs_oracle():
    """BASELINE configs[4]: equiangular 200 x 400 (V = 80 000, irregular degree), K=3, 32 ch + interp pooling to
    HEALPix nside=32 and back, two samples."""
    from dsw_amd import sphere
    from modules.layers import ConvCheb, GeneralAvgPool, GeneralAvgUnpool, prepare_torch_laplacian
    from scipy import sparse

    fine = sphere.SphereEquiangular(nlat=200, nlon=400, k=20)
    coarse = sphere.SphereHealpix(32, nest=True, k=20)
    pool_m, unpool_m = sphere.conservative_pool_matrices(fine.coords, coarse.coords)   # overlap areas, up to ~300 per row
    lap = prepare_torch_laplacian(fine.L, lmax=1.95)
    torch.manual_seed(5)
    conv = ConvCheb(32, 32, 3, laplacian=lap).to(DEV)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    pool, unpool = GeneralAvgPool(pool_m).to(DEV), GeneralAvgUnpool(unpool_m).to(DEV)
    V, B = fine.n_vertices, 2
    assert V == 80000
    x = torch.from_numpy(recipes.rand(41, (B, V, 32))).to(DEV).requires_grad_(True)
    y = conv(x)
    z, idx = pool(y)
    out = unpool(z, idx)
    gy = torch.from_numpy(recipes.rand(42, (B, V, 32))).to(DEV)
    out.backward(gy)
    torch.cuda.synchronize()
    rp, ci, va = orc.csr_arrays_from_coo(conv.laplacian.cpu())
    xn, wn, bn = (t.detach().cpu().numpy() for t in (x, conv.weight, conv.bias))
    y64 = orc.cheb_forward_f64(rp, ci, va, xn, wn, bn)
    P = sparse.csr_matrix(pool_m).astype(np.float32).astype(np.float64)
    U = sparse.csr_matrix(unpool_m).astype(np.float32).astype(np.float64)
    out64 = np.stack([U @ (P @ y64[b]) for b in range(B)])
    g_y64 = np.stack([P.T @ (U.T @ gy[b].double().cpu().numpy()) for b in range(B)])
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, xn, wn, g_y64, True)
    errs = {"y": orc.max_rel_err(y, y64), "pool_unpool": orc.max_rel_err(out, out64),
            "dx": orc.max_rel_err(x.grad, dx64), "dw": orc.max_rel_err(conv.weight.grad, dw64),
            "db": orc.max_rel_err(conv.bias.grad, db64)}
    _record("c5_equiangular_fp32", errs)
    assert errs["y"] <= 2e-6 and errs["pool_unpool"] <= 2e-6 and errs["dx"] <= 2e-6
    assert errs["dw"] <= 1e-5 and errs["db"] <= 1e-5


def test_unet_nside32_batch8_vs_cpu_restatement():
    """BASELINE configs[1]: UNetSpherical nside=32, K=3, B=8, fp32.  The same model object definition runs once on the
    device (HIP kernels) and once on the CPU with the fp64 oracle behind the layers (tests/_oracle_backend.py); output,
    loss and the gradients of all 38 parameter tensors are compared element-wise."""
    import modules.my_models_graph as arch
    from dsw_amd import functional
    from _oracle_backend import OracleBackend

    V = 12 * 32 ** 2
    tensor_info = {
        "dim_order": {"dynamic": ["sample", "time", "node", "feature"]},
        "input_n_feature": 6, "output_n_feature": 2, "input_n_time": 3, "output_n_time": 1,
        "input_shape_info": {"dynamic": {"node": V}}, "output_shape_info": {"dynamic": {"node": V}},
    }
    torch.manual_seed(10)
    model = arch.UNetSpherical(tensor_info, sampling="healpix", sampling_kwargs={"subdivisions": 32, "nest": True},
                               kernel_size_conv=3, conv_type="graph", graph_type="knn", knn=20, pool_method="interp")
    names = sorted(n for n, _ in model.named_parameters())
    with torch.no_grad():
        for i, n in enumerate(names):
            p = dict(model.named_parameters())[n]
            p.copy_(torch.from_numpy(recipes.unet_param_fill(i, n, tuple(p.shape))))
    x = torch.from_numpy(recipes.rand(601, (8, 3, V, 6)))
    target = torch.from_numpy(recipes.rand(602, (8, 1, V, 2)))

    def run(m, dev):
        m.zero_grad(set_to_none=True)
        y = m(x.to(dev))
        loss = ((y - target.to(dev)) ** 2).mean()
        loss.backward()
        return y.detach().cpu(), loss.item(), {n: p.grad.detach().cpu() for n, p in m.named_parameters()}

    from test_host_logic import pin_relu_masks, record_relu_masks

    functional.set_test_backend(OracleBackend())
    try:
        masks, handles = record_relu_masks(model)
        y_ref, loss_ref, g_ref = run(model, "cpu")
    finally:
        functional.set_test_backend(None)
        for h in handles:
            h.remove()
    # second checker: the torch-CPU restatement of the reference model (oracle/unet_oracle.py, fp32 ATen kernels,
    # autograd backward; pinned on fixture G5).  Its own fp32 reductions are ~1e-4 off fp64 on the scalar / bias
    # gradients, so it gets the looser bound.
    from oracle import unet_oracle

    sd = unet_oracle.leaf_state(model.state_dict())
    y_t, loss_t, g_t = unet_oracle.unet_fwd_bwd(sd, x, target)
    # ReLU ties: of ~4e8 pre-activations a handful (|z| ~ 1e-7) fall on the other side of zero in a second correct fp32
    # evaluation; on the 768-node level one of them moves a weight-gradient element by 1e-5 .. 1e-4.  The device run
    # takes the reference run's decision for exactly those elements (they must be < 1e-4, else the hook raises)
    # first an UN-PINNED device run against the same checker with a loose gate: the pinning below must matter for the tie
    # elements only - without it the forward agrees as tightly as with it, and the gradients to 1e-3 (one flipped decision
    # with |z| ~ 1e-7 on the 768-node level moves a weight-gradient element by up to ~1e-4 of the tensor's maximum)
    model = model.to(DEV)
    y_un, loss_un, g_un = run(model, DEV)
    un = {"y": orc.max_rel_err(y_un, y_ref), "loss": abs(loss_un - loss_ref) / max(1.0, abs(loss_ref)),
          "grads": max(orc.max_rel_err(g_un[n], g_ref[n]) for n in names)}
    print("un-pinned device run vs the fp64-backed run:", un)
    assert un["y"] <= 1e-5 and un["loss"] <= 1e-5 and un["grads"] <= 1e-3, un
    flipped, handles = pin_relu_masks(model, masks)
    y_dev, loss_dev, g_dev = run(model, DEV)
    for h in handles:
        h.remove()
    print("ReLU decisions that differed between the fp64-backed and the device run:", flipped)
    assert flipped["n"] <= max(8, 2e-7 * flipped["total"]), flipped
    # weight tensors [Fin, K, Fout] / [Fout, Fin] vs the 1-D ones (biases, ReZero scalars): the latter are plain sums of
    # ~1e6 signed products whose fp32 evaluation cancels heavily - the reference's own fp32 CPU path is 1.2e-4 off fp64
    # there (recorded as torch32_vs_f64_*), so they get the looser bound
    mats = [n for n in names if g_ref[n].dim() >= 2]
    vecs = [n for n in names if g_ref[n].dim() < 2]
    worst = lambda pool, a, b: max(orc.max_rel_err(a[n], b[n]) for n in pool)
    errs = {"y": orc.max_rel_err(y_dev, y_ref), "loss": abs(loss_dev - loss_ref) / max(1.0, abs(loss_ref)),
            "grad_weights_max": worst(mats, g_dev, g_ref), "grad_bias_rezero_max": worst(vecs, g_dev, g_ref),
            "y_vs_torch32": orc.max_rel_err(y_dev, y_t),
            "grad_weights_max_vs_torch32": worst(mats, g_dev, g_t), "grad_bias_rezero_max_vs_torch32": worst(vecs, g_dev, g_t),
            "torch32_vs_f64_grad_weights_max": worst(mats, g_t, g_ref),
            "torch32_vs_f64_grad_bias_rezero_max": worst(vecs, g_t, g_ref)}
    _record("unet_nside32_B8_fp32", errs)
    per_tensor = sorted(((orc.max_rel_err(g_dev[n], g_ref[n]), orc.max_rel_err(g_t[n], g_ref[n]), n) for n in names),
                        reverse=True)
    worst_name = per_tensor[0][2]
    for e_dev, e_t32, n in per_tensor[:6]:
        print("grad %-40s hip vs f64 %.2e   torch-fp32 vs f64 %.2e   shape %s" % (n, e_dev, e_t32, tuple(g_ref[n].shape)))
    try:
        path = os.path.join(ROOT, "gpurun_out", "parity_fullsize.json")
        data = json.load(open(path))
        data["unet_nside32_B8_fp32"]["worst_tensors"] = [[n, float("%.3e" % a), float("%.3e" % b)] for a, b, n in per_tensor[:6]]
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    assert errs["y"] <= 1e-5 and errs["loss"] <= 1e-5, errs
    # gradients of a 22-layer network against the fp64-backed run of the same model (same ReLU decisions, see above):
    # what is left is fp32 storage rounding of the inter-layer tensors and the fp32 sums of ~1e6 signed products behind
    # the biases / ReZero scalars.  The reference's own fp32 CPU path (torch32_vs_f64_*) is 3e-5 .. 1e-4 off the same run
    # measured with the decisions pinned: weight tensors 1.1e-6, ReZero scalars / biases 1.3e-5 -> gates at ~4-5x
    assert errs["grad_weights_max"] <= 5e-6 and errs["grad_bias_rezero_max"] <= 5e-5, (worst_name, errs)
    assert errs["y_vs_torch32"] <= 1e-5 and errs["grad_weights_max_vs_torch32"] <= 3e-4, errs   # the restatement's own fp32 error


# ----------------------------------------------------------------------------------------------------------------------
# Size-independent properties at the full benchmark sizes (no oracle in the loop): what a linear operator and its
# hand-written adjoint must satisfy whatever their size
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("knn,Fin,Fout,K,dt", [(8, 32, 64, 3, torch.float32), (20, 32, 64, 3, torch.float32),
                                              (8, 64, 128, 5, torch.float32), (20, 128, 64, 3, torch.float32)])
def test_full_size_adjoint_identity_and_linearity(knn, Fin, Fout, K, dt):
    """ConvCheb without bias is linear in x and in W, and its closed-form backward is the adjoint of its forward:
        <conv(x; W), g> = <x, dX(g)> = <W, dW(x, g)>        (fp64 inner products of the fp32 device results)
        conv(a x1 + x2; W) = a conv(x1; W) + conv(x2; W)
    at nside 64, B 16 (786 432 rows): the whole-forward kernel, the fused backward GEMM pass, both two-hop pairs, the
    K = 5 pairs with spare planes, a mix-first layer - each against ITSELF, so a slab / tile / halo bug that an
    element-wise comparison at small size cannot see shows up here as a broken identity."""
    from modules.layers import ConvCheb

    lap = _healpix_operator(64, knn)
    V, B = lap.shape[0], 16
    torch.manual_seed(11)
    layer = ConvCheb(Fin, Fout, K, laplacian=lap, bias=False).to(DEV).to(dt)
    x1 = torch.randn(B, V, Fin, device=DEV, dtype=dt)
    x2 = torch.randn(B, V, Fin, device=DEV, dtype=dt)
    g = torch.randn(B, V, Fout, device=DEV, dtype=dt)
    xa = x1.clone().requires_grad_(True)
    y = layer(xa)
    y.backward(g)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    lhs = dot(y.detach(), g)
    rhs_x, rhs_w = dot(x1, xa.grad), dot(layer.weight.detach(), layer.weight.grad)
    scale = float(y.detach().double().norm() * g.double().norm())
    assert abs(lhs - rhs_x) <= 2e-6 * scale and abs(lhs - rhs_w) <= 2e-6 * scale, (lhs, rhs_x, rhs_w, scale)
    with torch.no_grad():
        y2, y12 = layer(x2), layer(0.5 * x1 + x2)
        err = float((y12.double() - (0.5 * y.detach().double() + y2.double())).abs().max() / y12.double().abs().max())
    assert err <= 2e-6, err
    _record("adjoint_identity_k%d_%dto%d_K%d" % (knn, Fin, Fout, K),
            {"adjoint_x": abs(lhs - rhs_x) / scale, "adjoint_w": abs(lhs - rhs_w) / scale, "linearity": err})


def test_full_size_pool_unpool_adjoint_and_row_sums():
    """Interpolation pooling at nside 64 -> 32 (B 16, 64 channels): M 1 = 1 (rows sum to one: a constant field stays
    constant), and backward is the transposed product: <M x, g> = <x, M^T g>."""
    from dsw_amd import sphere
    from modules.layers import GeneralAvgPool

    pool_m, _ = sphere.healpix_pool_matrices(64, nest=True)
    pool = GeneralAvgPool(pool_m).to(DEV)
    B, C = 16, 64
    x = torch.randn(B, 49152, C, device=DEV).requires_grad_(True)
    g = torch.randn(B, 12288, C, device=DEV)
    y, _ = pool(x)
    y.backward(g)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    lhs, rhs = dot(y.detach(), g), dot(x.detach(), x.grad)
    assert abs(lhs - rhs) <= 2e-6 * float(y.detach().double().norm() * g.double().norm())
    ones, _ = pool(torch.full((1, 49152, 4), 3.25, device=DEV))
    assert float((ones - 3.25).abs().max()) <= 1e-5
