#!/usr/bin/env python3
"""A/B of the K = 3 recurrences under the two staged kernels (fused two-hop pair vs one staged launch per hop):
    tools/bench_hops_ab.py [knn ...]          (default 20 8; nside 64, C 32, B 16, fp32 = the NS shape)
Prints us per recurrence (forward pair T1, T2; adjoint pair) and the fraction of 8 TB/s by SURVEY 8d bytes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native, sphere, functional as F_
from modules.layers import prepare_torch_laplacian

knns = [int(v) for v in sys.argv[1:]] or [20, 8]
nside, C, B = (int(os.environ.get(k, d)) for k, d in (("NSIDE", 64), ("C", 32), ("B", 16)))
lib = _native.load()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for knn in knns:
    g = sphere.SphereHealpix(nside, nest=True, k=knn)
    lap = prepare_torch_laplacian(g.L, lmax=1.95).to("cuda")
    V = lap.shape[0]
    x = torch.randn(B, V, C, device="cuda")
    T = torch.empty(2, B, V, C, device="cuda")
    G0 = torch.randn_like(x); Gr = torch.randn(2, B, V, C, device="cuda"); spare = torch.empty(2, B, V, C, device="cuda")
    E = B * V * C * 4
    for mode, tiles in (("fused", (64,)), ("staged", (64,)), ("staged", (128,))):
        F_.HOP_MODE, F_.STAGED_TILE_ROWS = mode, tiles
        op = F_.get_operator(lap); opt = op.transpose()
        Lb = op.nnz * 8 + 4 * (V + 1)
        pp, keep = F_._plan_ptr(op, x); ppt, keept = F_._plan_ptr(opt, x)
        fwd = lambda: lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz, x.data_ptr(), T.data_ptr(), B, C, 3, 0, st, pp)
        adj = lambda: lib.dsw_cheb_basis_adj(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, G0.data_ptr(), Gr.data_ptr(), B, C, 3, 0, st, ppt, spare.data_ptr())
        tf, ta = timeit(fwd), timeit(adj)
        print("knn=%d %s tiles=%s plan(hops=%d rows=%d n2=%d lds=%d): fwd %.1f us (%.3f)  adj %.1f us (%.3f)" % (
            knn, mode, tiles, keep.hops, keep.tile_rows, keep.max_n2, keep.lds_bytes(128, True),
            tf, (5 * E + 2 * Lb) / tf / 8e6, ta, (7 * E + 2 * Lb) / ta / 8e6), flush=True)
