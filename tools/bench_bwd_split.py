#!/usr/bin/env python3
"""Experiment: backward of the NS layer as ONE call vs per batch part (the dgrad planes of a part - 3 E / parts - then fit the
256 MB Infinity Cache between the GEMM pass that writes them and the adjoint recurrence that reads them).
    tools/bench_bwd_split.py [knn] [parts ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native, sphere, functional as F_
from modules.layers import prepare_torch_laplacian

knn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
parts_list = [int(v) for v in sys.argv[2:]] or [1, 2, 4]
lib = _native.load()
nside, Fin, Fout, K, B = 64, 32, 64, 3, 16
g = sphere.SphereHealpix(nside, nest=True, k=knn)
op = F_.get_operator(prepare_torch_laplacian(g.L, lmax=1.95).to("cuda"))
opt = op.transpose()
V = op.shape[0]
x = torch.randn(B, V, Fin, device="cuda"); w = torch.randn(Fin, K, Fout, device="cuda") * 0.1
dy = torch.randn(B, V, Fout, device="cuda")
T = torch.empty(K - 1, B, V, Fin, device="cuda")
st = torch.cuda.current_stream().cuda_stream
pp = F_._plan_ptr(op, x)[0]; ppt = F_._plan_ptr(opt, x)[0]
assert lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz, x.data_ptr(), T.data_ptr(), B, Fin, K, 0, st, pp) == 0
dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(Fout, device="cuda")


def run(parts):
    b = B // parts
    n = int(lib.dsw_cheb_bwd_workspace_bytes(b, V, Fin, Fout, K, 0))
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    Tp = [torch.stack([T[k][i * b:(i + 1) * b] for k in range(K - 1)]).contiguous() for i in range(parts)] if parts > 1 else [T]

    def f():
        for i in range(parts):
            sl = slice(i * b, (i + 1) * b)
            rc = lib.dsw_cheb_bwd_res(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz,
                                      x[sl].data_ptr(), Tp[i].data_ptr(), w.data_ptr(), dy[sl].data_ptr(), dx[sl].data_ptr(),
                                      dw.data_ptr(), db.data_ptr(), ws.data_ptr(), n, b, Fin, Fout, K, 0, st, ppt, None, None, 0,
                                      1 if i > 0 else 0)
            assert rc == 0
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3 / 30, dx.clone(), dw.clone()


ref = None
for parts in parts_list:
    us, dxv, dwv = run(parts)
    if ref is None: ref = (dxv, dwv)
    print("knn=%d parts=%d: %.1f us  dx diff %.2e dw diff %.2e" % (knn, parts, us, (dxv - ref[0]).abs().max().item() / ref[0].abs().max().item(),
          (dwv - ref[1]).abs().max().item() / ref[1].abs().max().item()), flush=True)
