#!/usr/bin/env python3
"""Time individual C-ABI entry points (HIP events on the current stream) for given layer shapes.
usage: tools/bench_kernels.py [nside K fin fout batch knn dtype]..."""
import ctypes, sys, os
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

def run(nside, K, fin, fout, B, knn, dt):
    dtype = torch.bfloat16 if dt == "bf16" else torch.float32
    dcode = 1 if dt == "bf16" else 0
    lib = _native.load()
    g = sphere.SphereHealpix(nside, nest=True, k=knn)
    op = F_.get_operator(prepare_torch_laplacian(g.L, lmax=1.95).to("cuda"))
    V = op.shape[0]; N = B * V
    x = torch.randn(B, V, fin, device="cuda", dtype=dtype)
    w = torch.randn(fin, K, fout, device="cuda", dtype=dtype) * 0.1
    bias = torch.randn(fout, device="cuda", dtype=dtype)
    dy = torch.randn(B, V, fout, device="cuda", dtype=dtype)
    T = torch.empty(max(K - 1, 1), B, V, fin, device="cuda", dtype=dtype)
    y = torch.empty(B, V, fout, device="cuda", dtype=dtype)
    st = torch.cuda.current_stream().cuda_stream
    es = x.element_size()
    E = N * fin * es
    t_basis = timeit(lambda: lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz, x.data_ptr(), T.data_ptr(), B, fin, K, dcode, st, F_._plan_ptr(op, x)[0])) if K > 1 else 0.0
    t_mix = timeit(lambda: lib.dsw_cheb_mix_fwd(x.data_ptr(), T.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), N, fin, fout, K, dcode, st))
    nb = lib.dsw_cheb_bwd_workspace_bytes(B, V, fin, fout, K, dcode)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(bias)
    opt = op.transpose()
    a = (opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, x.data_ptr(), T.data_ptr(), w.data_ptr(), dy.data_ptr())
    t_bwd_dx = timeit(lambda: lib.dsw_cheb_bwd(*a, dx.data_ptr(), None, None, ws.data_ptr(), nb, B, fin, fout, K, dcode, st, F_._plan_ptr(opt, x)[0]))
    t_bwd_dw = timeit(lambda: lib.dsw_cheb_bwd(*a, None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, fin, fout, K, dcode, st, F_._plan_ptr(opt, x)[0]))
    fl = 2.0 * N * fin * K * fout
    mix_bytes = (K * N * fin + N * fout) * es
    print(f"nside={nside} K={K} {fin}->{fout} B={B} knn={knn} {dt}: basis {t_basis:7.1f} us | mix_fwd {t_mix:7.1f} us ({fl/t_mix/1e6:6.1f} TF, {mix_bytes/t_mix/1e3:6.0f} GB/s) | "
          f"bwd_dx {t_bwd_dx:7.1f} us | bwd_dw {t_bwd_dw:7.1f} us ({fl/t_bwd_dw/1e6:6.1f} TF) | total {t_basis+t_mix+t_bwd_dx+t_bwd_dw:7.1f} us", flush=True)

if __name__ == "__main__":
    args = sys.argv[1:]
    if not args:
        cases = [(64,3,32,64,16,8,"f32"), (64,1,32,64,16,8,"f32"), (64,2,32,64,16,8,"f32"), (64,3,32,32,16,8,"f32"), (64,3,32,128,16,8,"f32"), (64,3,64,64,16,8,"f32")]
    else:
        cases = [tuple(int(v) if v.isdigit() else v for v in a.split(",")) for a in args]
    for c in cases: run(*c)
