#!/usr/bin/env python3
"""Forward / adjoint Chebyshev recurrences (K = 3): fused two-hop launches vs one launch per hop, HIP-event timed.
usage: tools/bench_hops.py [nside,C,B,knn]..."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native, sphere, functional as F_
from modules.layers import prepare_torch_laplacian


def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def run(nside, C, B, knn):
    lib = _native.load()
    g = sphere.SphereHealpix(nside, nest=True, k=knn)
    op = F_.get_operator(prepare_torch_laplacian(g.L, lmax=1.95).to("cuda"))
    opt = op.transpose()
    V = op.shape[0]
    x = torch.randn(B, V, C, device="cuda")
    T = torch.empty(2, B, V, C, device="cuda")
    G0 = torch.randn_like(x); Gr = torch.randn(2, B, V, C, device="cuda"); spare = torch.empty(2, B, V, C, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    pp = F_._plan_ptr(op, x)[0]; ppt = F_._plan_ptr(opt, x)[0]
    res = {}
    for name, p, pt in (("fused", pp, ppt), ("single", None, None)):
        if name == "fused" and p is None:
            res[name] = (float("nan"), float("nan")); continue
        f = timeit(lambda: lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz, x.data_ptr(), T.data_ptr(), B, C, 3, 0, st, p))
        a = timeit(lambda: lib.dsw_cheb_basis_adj(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, G0.data_ptr(), Gr.data_ptr(), B, C, 3, 0, st, pt, spare.data_ptr()))
        res[name] = (f, a)
    E = B * V * C * 4
    print(f"nside={nside} C={C} B={B} knn={knn} E={E/1e6:.0f}MB: fused fwd {res['fused'][0]:7.1f} adj {res['fused'][1]:7.1f} us | single-hop fwd {res['single'][0]:7.1f} adj {res['single'][1]:7.1f} us"
          f" | per hop per (row x 128B): fused {res['fused'][0]/2/(B*V*C/32)*1e3:.3f} ns, single {res['single'][0]/2/(B*V*C/32)*1e3:.3f} ns", flush=True)


if __name__ == "__main__":
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
        (64, 32, 16, 8), (64, 32, 16, 20), (32, 64, 8, 20), (32, 128, 8, 20), (32, 256, 8, 20), (16, 256, 8, 20), (16, 512, 8, 20), (64, 64, 16, 20)]
    for c in cases: run(*c)
