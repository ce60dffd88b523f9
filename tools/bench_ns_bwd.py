#!/usr/bin/env python3
"""dX of the north-star layer from dY, replayed from HIP graphs: dsw_cheb_dx_one_launch (dsw_bwd3.hip) next to the dX-only
dsw_cheb_bwd (dgrad GEMM + adjoint pair), us per call.    DSW_HIP_LIB=_ab_libs/x.so python tools/bench_ns_bwd.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native, sphere, functional as F_
from modules.layers import prepare_torch_laplacian

B, fin, fout, K = 16, 32, 64, 3
lib = _native.load()
g = sphere.SphereHealpix(64, nest=True, k=8)
op = F_.get_operator(prepare_torch_laplacian(g.L, lmax=1.95).to("cuda"))
opt = op.transpose()
V = op.shape[0]
torch.manual_seed(0)
x = torch.randn(B, V, fin, device="cuda")
w = torch.randn(fin, K, fout, device="cuda") * 0.1
dy = torch.randn(B, V, fout, device="cuda")
T = torch.empty(K - 1, B, V, fin, device="cuda")
dx = torch.empty_like(x)
nb = lib.dsw_cheb_bwd_workspace_bytes(B, V, fin, fout, K, 0)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
pt, _k2 = F_._plan_ptr(opt, x)


nws1 = int(lib.dsw_cheb_dx_one_launch_workspace_bytes())
ws1 = torch.empty(nws1, dtype=torch.uint8, device="cuda")


def one():
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dsw_cheb_dx_one_launch(pt, V, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws1.data_ptr(), nws1, B, fin, fout, K, 0, st) == 0


def bwd():
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dsw_cheb_bwd(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, x.data_ptr(), T.data_ptr(),
                            w.data_ptr(), dy.data_ptr(), dx.data_ptr(), None, None, ws.data_ptr(), nb, B, fin, fout, K, 0, st, pt) == 0


def graphed_us(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            fn()
    t_end = time.time() + 1.0
    while time.time() < t_end:
        gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record()
        for _ in range(20):
            gr.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 400)
    return sorted(ts)[2]


print("%s: dX in one launch %.1f us | dX-only dsw_cheb_bwd (dgrad GEMM + adjoint pair) %.1f us" % (
    os.environ.get("DSW_HIP_LIB", "product").split("/")[-1], graphed_us(one), graphed_us(bwd)), flush=True)
