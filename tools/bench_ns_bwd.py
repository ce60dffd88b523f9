#!/usr/bin/env python3
"""dsw_cheb_bwd with dX only (the one-launch dgrad + adjoint at the north-star shape) replayed from a HIP graph: us per call.
    DSW_HIP_LIB=_ab_libs/x.so python tools/bench_ns_bwd.py"""
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


def bwd():
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dsw_cheb_bwd(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, x.data_ptr(), T.data_ptr(),
                            w.data_ptr(), dy.data_ptr(), dx.data_ptr(), None, None, ws.data_ptr(), nb, B, fin, fout, K, 0, st, pt) == 0


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        bwd()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20):
        bwd()
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
print("%s: dX-only backward %.1f us" % (os.environ.get("DSW_HIP_LIB", "product").split("/")[-1], sorted(ts)[2]), flush=True)
