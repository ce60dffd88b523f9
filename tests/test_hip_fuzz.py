"""Randomised sweep over layer shapes: every dispatch branch of the library (aligned / unaligned GEMMs, resident /
streamed W, fused / single-hop SpMM, basis-first / mix-first order, fused backward pass, fp32 / bf16 storage) must
agree with the fp64 oracle.  Seeds are fixed: the cases are the same on every run.  GPU only."""
import numpy as np
import pytest
import torch

from oracle import cheb_oracle as orc
import recipes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

_CH = [1, 2, 3, 5, 8, 12, 16, 18, 24, 32, 40, 48, 64, 96, 128, 160, 192, 256]


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        healpix = rng.random() < 0.6
        V = int(rng.choice([48, 192, 768])) if healpix else int(rng.integers(33, 700))
        B = int(rng.integers(1, 6))
        Fin, Fout = int(rng.choice(_CH)), int(rng.choice(_CH))
        K = int(rng.integers(1, 6))
        bias = bool(rng.random() < 0.7)
        bf16 = bool(rng.random() < 0.25)
        out.append((i, V, B, Fin, Fout, K, bias, healpix, bf16))
    return out


import os


@pytest.mark.parametrize("i,V,B,Fin,Fout,K,bias,healpix,bf16",
                         _cases(int(os.environ.get("DSW_FUZZ_N", "48")), int(os.environ.get("DSW_FUZZ_SEED", "20260928"))))
def test_random_layer(i, V, B, Fin, Fout, K, bias, healpix, bf16):
    from dsw_amd import sphere
    from modules.layers import ConvCheb

    if healpix:
        g = sphere.SphereHealpix(int(np.sqrt(V // 12)), nest=bool(i % 2), k=8)
        rp, ci, va = orc.csr_arrays_from_coo(orc.prepare_laplacian_fixed_lmax(g.L, 1.9))
    else:
        rp, ci, va = recipes.irregular_operator(V, seed=100 + i, min_deg=0, max_deg=30)
    x = recipes.rand(10 * i + 1, (B, V, Fin))
    w = recipes.rand(10 * i + 2, (Fin, K, Fout), np.sqrt(2.0 / (Fin * K)))
    b = recipes.rand(10 * i + 3, (Fout,), 0.1) if bias else None
    gy = recipes.rand(10 * i + 4, (B, V, Fout))
    dt = torch.bfloat16 if bf16 else torch.float32
    q = (lambda a: torch.from_numpy(a).to(dt)) if bf16 else torch.from_numpy
    xq, wq, gyq = q(x), q(w), q(gy)
    bq = None if b is None else q(b)
    lap = orc.coo_from_csr_arrays(rp, ci, va, (V, V))
    layer = ConvCheb(Fin, Fout, K, laplacian=lap, bias=bias)
    layer.set_parameters(wq.float(), None if bq is None else bq.float())
    layer = layer.to(DEV).to(dt)
    xin = xq.to(DEV).requires_grad_(True)
    y = layer(xin)
    y.backward(gyq.to(DEV))
    torch.cuda.synchronize()
    # oracle on exactly the (possibly bf16-rounded) inputs the device saw; operator values stay fp32 on the device
    f64 = lambda t: None if t is None else t.double().numpy()   # noqa: E731
    y64 = orc.cheb_forward_f64(rp, ci, va, f64(xq), f64(wq), f64(bq))
    dx64, dw64, db64 = orc.cheb_backward_f64(rp, ci, va, f64(xq), f64(wq), f64(gyq), bias)
    tol = 1e-2 if bf16 else 2e-6
    assert orc.max_rel_err(y, y64) <= tol
    assert orc.max_rel_err(xin.grad, dx64) <= tol
    assert orc.max_rel_err(layer.weight.grad, dw64) <= 2 * tol
    if bias:
        assert orc.max_rel_err(layer.bias.grad, db64) <= 2 * tol
