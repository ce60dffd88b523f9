#!/usr/bin/env python3
"""One configuration of the K = 3 recurrences for profiling (tools/pmc_pair.sh):
    tools/bench_pair_one.py <unused 0|1> <knn> [fwd|adj] [nside C B]
(the first argument selected the row-group kernel of the round-3 experiment, branch exp/quad-gather; ignored on main)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native, sphere, functional as F_
from modules.layers import prepare_torch_laplacian

quads, knn = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "fwd"
nside, C, B = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (64, 32, 16)
if os.environ.get("HOP_MODE"):          # tool-side A/B switch: "fused" | "staged" | "auto"
    F_.HOP_MODE = os.environ["HOP_MODE"]
if os.environ.get("TILES"):
    F_.STAGED_TILE_ROWS = tuple(int(v) for v in os.environ["TILES"].split(","))
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
fwd = lambda: lib.dsw_cheb_basis_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz, x.data_ptr(), T.data_ptr(), B, C, 3, 0, st, pp)
adj = lambda: lib.dsw_cheb_basis_adj(opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz, G0.data_ptr(), Gr.data_ptr(), B, C, 3, 0, st, ppt, spare.data_ptr())
fn = fwd if which == "fwd" else adj
for _ in range(5): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fn()
b.record(); torch.cuda.synchronize()
print("quads=%d knn=%d %s: %.1f us" % (quads, knn, which, a.elapsed_time(b) * 1e3 / 20))
