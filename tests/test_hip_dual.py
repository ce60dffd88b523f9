"""Round-5, GPU only: the WHOLE backward of the K = 3, 32 -> 64 fp32 layer in one launch in the dual form (dsw_bwd3d.hip)

    U_k = T_k(L^T) dY (on chip),   dX = sum_k U_k W_k^T,   dW_k = X^T U_k,   db = 1^T dY

against the fp64 closed form of the oracle (the autograd of /root/reference/modules/layers.py:163-178 restated), against the
route that reads the forward's basis planes, and the forward that no longer stores them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(nside, knn=8, seed=0, lap=None, fout=64):
    from dsw_amd import sphere
    from modules.layers import ConvCheb, prepare_torch_laplacian

    if lap is None:
        g = sphere.SphereHealpix(nside, nest=True, k=knn)
        lap = prepare_torch_laplacian(g.L, lmax=1.9)
    torch.manual_seed(seed)
    return ConvCheb(32, fout, 3, laplacian=lap).to(DEV)


def _needs_basis(layer, x):
    from dsw_amd import _native, functional as F_

    op = F_.get_operator(layer.laplacian)
    pt, _keep = F_._plan_ptr(op.transpose(), x, 32)
    return int(_native.load().dsw_cheb_bwd_needs_basis(pt, x.shape[1], 32, layer.out_channels, 3, 0))


def _oracle(layer, x, gy):
    from oracle import cheb_oracle as orc

    rp, ci, va = orc.csr_arrays_from_coo(layer.laplacian.cpu())
    return orc.cheb_backward_f64(rp, ci, va, x.detach().cpu().numpy(), layer.weight.detach().cpu().numpy(), gy.cpu().numpy(), True)


@pytest.mark.parametrize("nside,B,fout", [(8, 1, 64), (8, 3, 64), (16, 5, 64), (16, 16, 64), (8, 3, 32), (16, 7, 32)])
def test_dual_backward_vs_oracle(nside, B, fout, monkeypatch):
    """Module-level forward + backward of the eligible shape: the launch trace shows ONE backward role (bwd_dual: no dgrad
    planes, no adjoint launches), the forward keeps no basis planes, dX / dW / db agree with the fp64 closed form in every
    element (odd sample counts and batch chunks included), reruns are bit-identical.  Fout = 32 (round 6): ONE chunk phase per
    sample, the whole dX reduction in waves 0-3."""
    from dsw_amd import _native, functional as F_
    from oracle import cheb_oracle as orc

    monkeypatch.setattr(F_, "MIN_CLUSTERED_TILES", 1)
    layer = _layer(nside, seed=nside + B, fout=fout)
    V = 12 * nside * nside
    x = torch.randn(B, V, 32, device=DEV, requires_grad=True)
    gy = torch.randn(B, V, fout, device=DEV)
    assert _needs_basis(layer, x) == 0
    layer(x).backward(gy)                      # (plans built, caches warm)
    grads = []
    for _ in range(2):
        x.grad = layer.weight.grad = layer.bias.grad = None
        with _native.LaunchTrace(256) as tr:
            layer(x).backward(gy)
            torch.cuda.synchronize()
        roles = [r[0] for r in tr.intervals]
        assert roles.count("bwd_dual") == 1 and "basis_adj" not in roles and "bwd_gemm_fused" not in roles, roles
        grads.append((x.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone()))
    dx64, dw64, db64 = _oracle(layer, x, gy)
    dx, dw, db = grads[0]
    assert orc.max_rel_err(dx, dx64) <= 2e-6
    assert orc.max_rel_err(dw, dw64) <= 1e-5
    assert orc.max_rel_err(db, db64) <= 1e-5
    assert all(torch.equal(a, b) for a, b in zip(grads[0], grads[1]))


def test_dual_backward_partial_requests_and_accumulation(monkeypatch):
    """dX only (frozen parameters), dW / db only (an input without gradient) and the accumulating form (dW += into the
    parameters' gradient buffers: functional.grad_accumulators) all run the dual launch and agree with the full call."""
    from dsw_amd import functional as F_
    from oracle import cheb_oracle as orc

    monkeypatch.setattr(F_, "MIN_CLUSTERED_TILES", 1)
    layer = _layer(8, seed=5)
    x = torch.randn(3, 768, 32, device=DEV, requires_grad=True)
    gy = torch.randn(3, 768, 64, device=DEV)
    layer(x).backward(gy)
    full = (x.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone())
    # dW / db only
    x2 = x.detach().clone()
    layer.weight.grad = layer.bias.grad = None
    layer(x2).backward(gy)
    assert torch.equal(layer.weight.grad, full[1]) and torch.equal(layer.bias.grad, full[2])
    # dX only
    for p in layer.parameters():
        p.requires_grad_(False)
    x3 = x.detach().clone().requires_grad_(True)
    layer(x3).backward(gy)
    assert torch.equal(x3.grad, full[0])
    for p in layer.parameters():
        p.requires_grad_(True)
    # accumulation through the backend: dW += , db +=
    be = F_._backend_for(x)
    op = F_.get_operator(layer.laplacian)
    aw, ab = torch.ones_like(layer.weight), torch.full_like(layer.bias, 2.0)
    dx, _w, _b = be.cheb_bwd_res(op, x.detach(), None, layer.weight.detach(), gy, True, True, acc_w=aw, acc_b=ab)
    assert torch.equal(dx, full[0])
    assert orc.max_rel_err(aw - 1.0, full[1].cpu().numpy()) <= 1e-6
    assert orc.max_rel_err(ab - 2.0, full[2].cpu().numpy()) <= 1e-6


@pytest.mark.parametrize("fout", [64, 32])
def test_dual_backward_non_symmetric_operator(fout, monkeypatch):
    """The dual form runs its hops with L^T (the plan of the transposed operator) and its weight gradient with X^T T_k(L^T) dY
    = (T_k(L) X)^T dY: a NON-symmetric L (random row scaling of a HEALPix Laplacian) pins both against the oracle."""
    from scipy import sparse
    from dsw_amd import functional as F_, sphere
    from oracle import cheb_oracle as orc

    monkeypatch.setattr(F_, "MIN_CLUSTERED_TILES", 1)
    g = sphere.SphereHealpix(8, nest=True, k=8)
    L = sparse.csr_matrix(g.L).astype(np.float64)
    rng = np.random.default_rng(0)
    L = sparse.diags(rng.uniform(0.3, 1.2, L.shape[0])) @ L * 0.5
    L = sparse.csr_matrix(L).astype(np.float32)
    L.sort_indices()
    layer = _layer(8, seed=3, lap=orc.coo_from_scipy(L).float(), fout=fout)
    x = torch.randn(4, 768, 32, device=DEV, requires_grad=True)
    gy = torch.randn(4, 768, fout, device=DEV)
    assert _needs_basis(layer, x) == 0
    layer(x).backward(gy)
    dx64, dw64, db64 = _oracle(layer, x, gy)
    assert orc.max_rel_err(x.grad, dx64) <= 2e-6
    assert orc.max_rel_err(layer.weight.grad, dw64) <= 1e-5
    assert orc.max_rel_err(layer.bias.grad, db64) <= 1e-5


def test_forward_without_basis_planes_is_the_same_forward(monkeypatch):
    """dsw_cheb_fwd with T = NULL (what the module does where the backward is the dual launch) returns bit for bit the Y of
    the call that stores the basis planes, and a k = 20 graph (one-hop plan: no dual form) still gets its planes."""
    from dsw_amd import functional as F_

    monkeypatch.setattr(F_, "MIN_CLUSTERED_TILES", 1)
    layer = _layer(16, seed=2)
    x = torch.randn(5, 3072, 32, device=DEV)
    be = F_._backend_for(x)
    op = F_.get_operator(layer.laplacian)
    w, b = layer.weight.detach(), layer.bias.detach()
    y_keep, T_keep = be.cheb_fwd(op, x, w, b)
    y_drop, T_drop = be.cheb_fwd(op, x, w, b, keep_basis=False)
    assert T_keep is not None and T_drop is None
    assert torch.equal(y_keep, y_drop)
    y_relu, T_relu = be.cheb_fwd(op, x, w, b, True, keep_basis=False)
    assert T_relu is None and torch.equal(y_relu, torch.relu(y_keep))
    l20 = _layer(8, knn=20, seed=2)
    x20 = torch.randn(2, 768, 32, device=DEV)
    assert _needs_basis(l20, x20) == 1
    _y, T20 = be.cheb_fwd(F_.get_operator(l20.laplacian), x20, l20.weight.detach(), l20.bias.detach(), keep_basis=False)
    assert T20 is not None


@pytest.mark.parametrize("fout", [64, 32])
def test_dual_backward_equals_the_basis_route_at_full_size(fout):
    """North-star shape (nside 64, B 16): the dual launch against dsw_cheb_bwd on the forward's basis planes WITHOUT a plan
    (dgrad GEMM, plain adjoint hops, wgrad from T_k: a route that shares no kernel with it) in every element of dX, dW, db;
    dsw_cheb_bwd with T = NULL and no plan is refused (every other route reads the planes)."""
    from dsw_amd import _native, functional as F_, sphere
    from modules.layers import prepare_torch_laplacian
    from oracle import cheb_oracle as orc

    lib = _native.load()
    g = sphere.SphereHealpix(64, nest=True, k=8)
    op = F_.get_operator(prepare_torch_laplacian(g.L, lmax=1.95).to(DEV))
    opt = op.transpose()
    B, V, fin, K = 16, op.shape[0], 32, 3
    torch.manual_seed(11)
    x = torch.randn(B, V, fin, device=DEV)
    w = torch.randn(fin, K, fout, device=DEV) * 0.1
    dy = torch.randn(B, V, fout, device=DEV)
    be = F_._backend_for(x)
    _y, T = be.cheb_fwd(op, x, w, None)
    nb = int(lib.dsw_cheb_bwd_workspace_bytes(B, V, fin, fout, K, 0))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    pt, _keep = F_._plan_ptr(opt, x)
    assert int(lib.dsw_cheb_bwd_needs_basis(pt, V, fin, fout, K, 0)) == 0
    st = torch.cuda.current_stream().cuda_stream

    def bwd(plan, basis):
        out = [torch.full_like(x, float("nan")), torch.full_like(w, float("nan")), torch.full((fout,), float("nan"), device=DEV)]
        rc = lib.dsw_cheb_bwd(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, x.data_ptr(),
                              T.data_ptr() if basis else None, w.data_ptr(), dy.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                              out[2].data_ptr(), ws.data_ptr(), nb, B, fin, fout, K, 0, st, plan)
        return rc, out

    rc, ref = bwd(None, True)
    assert rc == 0
    rc, new = bwd(pt, False)
    assert rc == 0
    assert bwd(None, False)[0] < 0
    assert orc.max_rel_err(new[0], ref[0].cpu().numpy()) <= 2e-6
    assert orc.max_rel_err(new[1], ref[1].cpu().numpy()) <= 1e-5
    assert orc.max_rel_err(new[2], ref[2].cpu().numpy()) <= 1e-5
    rc, again = bwd(pt, False)
    assert rc == 0 and all(torch.equal(a, b) for a, b in zip(new, again))
