#!/usr/bin/env python3
"""Cycle timeline of one wave of the streaming GEMM (experiment build: tools/build_variant1.sh tl dsw_gemm_x3s.hip -DDSW_TIMELINE,
DSW_HIP_LIB=_ab_libs/tl.so): where a 32-deep chunk step spends its cycles."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import numpy as np, torch
from dsw_amd import _native
lib = _native.load()
N, Fin, Fout, K = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (98304, 256, 128, 3)))
x = torch.randn(N, Fin, device="cuda"); T = torch.randn(K - 1, N, Fin, device="cuda")
w = torch.randn(Fin, K, Fout, device="cuda") * 0.05; b = torch.randn(Fout, device="cuda"); y = torch.empty(N, Fout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.dsw_cheb_mix_fwd(x.data_ptr(), T.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, Fin, Fout, K, 0, st) == 0
torch.cuda.synchronize()
raw = ctypes.CDLL(_native.LIB_PATH)
out = (ctypes.c_ulonglong * (64 * 8))()
assert raw.dsw_debug_x3s_timeline(out) == 0
t = np.array(out, dtype=np.int64).reshape(64, 8)[:, :7]
names = ["barrier wait", "A frags + split", "MFMAs (48)", "W split + LDS write", "A LDS write", "issue loads", "to next barrier"]
d = np.zeros((63, 7))
for i in range(63):
    d[i, 0] = t[i, 1] - t[i, 0]; d[i, 1] = t[i, 2] - t[i, 1]; d[i, 2] = t[i, 3] - t[i, 2]; d[i, 3] = t[i, 4] - t[i, 3]
    d[i, 4] = t[i, 5] - t[i, 4]; d[i, 5] = t[i, 6] - t[i, 5]; d[i, 6] = t[i + 1, 0] - t[i, 6]
print("chunk step: median %.0f cycles (wave 1 of workgroup 3, chunks 8..70)" % np.median(d.sum(1)))
for j, n in enumerate(names):
    print("  %-22s median %6.0f  p10 %6.0f  p90 %6.0f" % (n, np.median(d[:, j]), np.percentile(d[:, j], 10), np.percentile(d[:, j], 90)))
