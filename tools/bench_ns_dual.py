#!/usr/bin/env python3
"""The one-launch dual backward (dsw_bwd3d.hip) of the north-star layer: parity against the plan-less route of dsw_cheb_bwd
(dgrad GEMM + plain adjoint hops + wgrad from the forward's basis planes) and against an fp64 closed form, then us per call
of both full backwards and of the forward with / without basis stores, replayed from HIP graphs.
    python tools/bench_ns_dual.py [nside] [B]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native, sphere, functional as F_
from modules.layers import prepare_torch_laplacian

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fin, fout, K = 32, (32 if "--fout32" in sys.argv else 64), 3
lib = _native.load()
g = sphere.SphereHealpix(nside, nest=True, k=8)
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
nb = int(lib.dsw_cheb_bwd_workspace_bytes(B, V, fin, fout, K, 0))
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
pf, _k1 = F_._plan_ptr(op, x)
pt, _k2 = F_._plan_ptr(opt, x)
print("V %d B %d needs_basis %d workspace %.1f MB" % (V, B, lib.dsw_cheb_bwd_needs_basis(pt, V, fin, fout, K, 0), nb / 1e6), flush=True)


def fwd(keep=True):
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dsw_cheb_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz, x.data_ptr(), w.data_ptr(),
                            bias.data_ptr(), y.data_ptr(), T.data_ptr() if keep else None, B, fin, fout, K, 0, st, pf) == 0


def bwd(plan, out, basis=True):
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.dsw_cheb_bwd(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, x.data_ptr(),
                          T.data_ptr() if basis else None, w.data_ptr(), dy.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                          out[2].data_ptr(), ws.data_ptr(), nb, B, fin, fout, K, 0, st, plan)
    assert rc == 0, rc


def graphed_us(fn, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
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
        ts.append(a.elapsed_time(b) * 1e3 / (20 * reps))
    return sorted(ts)[2]


def graphed_us_later(fn):
    return graphed_us(fn)


def outs():
    return [torch.full_like(x, float("nan")), torch.full_like(w, float("nan")), torch.full((fout,), float("nan"), device="cuda")]


if "--eager-only" in sys.argv:     # for counter collection (rocprofv3 --pmc does not survive graph replays here)
    o1 = outs()
    fwd()
    for _ in range(20):
        bwd(pt, o1, basis=False)
    torch.cuda.synchronize()
    sys.exit(0)
if "--timeline" in sys.argv:    # -DDSW_D3_TIMELINE build: cycle stamps of waves 0 and 4 of workgroup 0 around the barriers of one sample
    o1 = outs()
    fwd()
    for _ in range(3):
        bwd(pt, o1, basis=False)
    torch.cuda.synchronize()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    base = (-ws.data_ptr()) % 256
    off = base + (2 * cus + 8) * (97 * 64) * 4 + 512 * 8192
    st = ws[off:off + 64 * 8].view(torch.int64).cpu().tolist()
    names = ["->A", "A", "ph1 end", "B", "ph2 end", "C", "mfma end", "D"]
    for wv in (0, 1):
        t = st[wv * 32:(wv + 1) * 32]
        t0 = t[0]
        print("wave %d:" % (4 * wv), " | ".join("c%d %s %d" % (i // 16, names[i % 16], t[i] - t0) for i in list(range(8)) + list(range(16, 24))))
    sys.exit(0)
if "--fwd-only" in sys.argv:
    print("%s: forward with basis stores %.1f us | without %.1f us" % (os.environ.get("DSW_HIP_LIB", "product").split("/")[-1],
                                                                        graphed_us(lambda: fwd(True)), graphed_us(lambda: fwd(False))), flush=True)
    sys.exit(0)
if "--time-only" in sys.argv:
    o1 = outs()
    fwd()
    nb_ = int(lib.dsw_cheb_bwd_needs_basis(pt, V, fin, fout, K, 0)) != 0     # (a diagnostics build with DSW_BWD_DUAL=0: the route on the basis planes)
    print("%s: %s backward %.1f us" % (os.environ.get("DSW_HIP_LIB", "product").split("/")[-1], "basis-route" if nb_ else "dual",
                                        graphed_us_later(lambda: bwd(pt, o1, basis=nb_))), flush=True)
    sys.exit(0)
fwd()
ref, new = outs(), outs()
bwd(None, ref)
bwd(pt, new, basis=False)
torch.cuda.synchronize()
for name, a, b in zip(("dX", "dW", "db"), new, ref):
    err = (a - b).abs().max().item() / b.abs().max().item()
    print("%s: dual vs plan-less route max-rel %.3g (nan %d)" % (name, err, int(torch.isnan(a).sum())), flush=True)
if V * B <= 200000:   # fp64 closed form on the device (dense algebra through torch.sparse)
    Ld = torch.sparse_csr_tensor(op.rowptr.long(), op.colind.long(), op.values.double(), size=(V, V))
    xd, wd, dyd = x.double(), w.double(), dy.double()
    Tk = [xd, torch.stack([Ld @ xd[b] for b in range(B)])]
    Tk.append(2 * torch.stack([Ld @ Tk[1][b] for b in range(B)]) - xd)
    dw64 = torch.stack([torch.einsum("bvf,bvo->fo", Tk[k], dyd) for k in range(K)], 1)
    Lt = Ld.to_dense().t().contiguous()
    Gk = [torch.einsum("bvo,fo->bvf", dyd, wd[:, k]) for k in range(K)]
    H1 = Gk[1] + 2 * torch.einsum("uv,bvf->buf", Lt, Gk[2])
    dx64 = Gk[0] - Gk[2] + torch.einsum("uv,bvf->buf", Lt, H1)
    for name, a, b in (("dX", new[0], dx64), ("dW", new[1], dw64), ("db", new[2], dyd.sum((0, 1)))):
        print("%s: dual vs fp64 closed form max-rel %.3g" % (name, (a.double() - b).abs().max().item() / b.abs().max().item()), flush=True)
# run-to-run determinism
again = outs()
bwd(pt, again, basis=False)
torch.cuda.synchronize()
print("bit-identical rerun:", all(torch.equal(a, b) for a, b in zip(new, again)), flush=True)


o1, o2 = outs(), outs()
if "--no-time" not in sys.argv:
    print("backward: dual one-launch %.1f us | fused wgrad+dgrad pass + adjoint pair %.1f us" % (
        graphed_us(lambda: bwd(pt, o1, basis=False)), float("nan")), flush=True)
    print("forward: with basis stores %.1f us | without %.1f us" % (graphed_us(lambda: fwd(True)), graphed_us(lambda: fwd(False))), flush=True)
    print("step (fwd without basis + dual bwd) %.1f us" % graphed_us(lambda: (fwd(False), bwd(pt, o1, basis=False))), flush=True)
