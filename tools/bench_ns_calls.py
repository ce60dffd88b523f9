#!/usr/bin/env python3
"""The three calls of the north-star step (dsw_cheb_fwd, the backward GEMM pass + adjoint = dsw_cheb_bwd), each replayed from
its own HIP graph of 20 calls and from one graph of the whole step: us per call.  For A/B runs of experiment builds:
    DSW_HIP_LIB=_ab_libs/x.so python tools/bench_ns_calls.py [knn]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native, sphere, functional as F_
from modules.layers import prepare_torch_laplacian

knn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, fin, fout, K = 16, 32, 64, 3
lib = _native.load()
g = sphere.SphereHealpix(64, nest=True, k=knn)
op = F_.get_operator(prepare_torch_laplacian(g.L, lmax=1.95).to("cuda"))
opt = op.transpose()
V = op.shape[0]
torch.manual_seed(0)
x = torch.randn(B, V, fin, device="cuda")
w = torch.randn(fin, K, fout, device="cuda") * 0.1
bias = torch.randn(fout, device="cuda")
dy = torch.randn(B, V, fout, device="cuda")
T = torch.empty(K - 1, B, V, fin, device="cuda")
y = torch.empty(B, V, fout, device="cuda")
dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(bias)
nb = lib.dsw_cheb_bwd_workspace_bytes(B, V, fin, fout, K, 0)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
pf, _k1 = F_._plan_ptr(op, x)
pt, _k2 = F_._plan_ptr(opt, x)


def fwd():
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dsw_cheb_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz, x.data_ptr(), w.data_ptr(),
                            bias.data_ptr(), y.data_ptr(), T.data_ptr(), B, fin, fout, K, 0, st, pf) == 0


def bwd():
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dsw_cheb_bwd(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, x.data_ptr(), T.data_ptr(),
                            w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, fin, fout,
                            K, 0, st, pt) == 0


def graphed(fn, n=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    return gr, n


def time_graph(gr, n, reps=30):
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record()
        for _ in range(reps):
            gr.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / (reps * n))
    return sorted(ts)[2]


gf = graphed(fwd)
gb = graphed(bwd)
gs = graphed(lambda: (fwd(), bwd()))
# spin up clocks
t_end = __import__("time").time() + 1.0
while __import__("time").time() < t_end:
    gs[0].replay()
torch.cuda.synchronize()
tf, tb, tstep = time_graph(*gf), time_graph(*gb), time_graph(*gs)
print("%s knn=%d: fwd %.1f us  bwd %.1f us  step %.1f us" % (os.environ.get("DSW_HIP_LIB", "product").split("/")[-1], knn, tf, tb, tstep), flush=True)
