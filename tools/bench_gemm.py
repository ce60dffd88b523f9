#!/usr/bin/env python3
"""Channel-mix GEMM alone (dsw_cheb_mix_fwd), HIP-event timed, for layer shapes given as N,Fin,Fout,K ...
Prints us, fp32 TFLOP/s, fraction of the bf16 pipe (6 MFMA terms per product) and the HBM rate of the streamed bytes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
from dsw_amd import _native


def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    lib = _native.load()
    zeros = "--zeros" in sys.argv   # zero-filled operands: same instructions, least switching power (is the launch clock-limited?)
    use_ws = "--ws" in sys.argv   # through dsw_cheb_fwd_ws (K = 1 layer of width K * Fin) with the caller scratch a layer call passes
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if not a.startswith("-")] or [
        # the streaming-GEMM launches of one U-Net step (nside 32, B 8; DSW_X3S_TRACE=1 in a diagnostics build lists them):
        # basis-first forwards as (N, Fin, Fout, 3), mix-first plane GEMMs as (N, Fin, 3 * Fout, 1) ...
        (98304, 64, 128, 3), (98304, 256, 384, 1), (24576, 128, 192, 3), (24576, 192, 256, 3), (24576, 512, 768, 1),
        (24576, 256, 384, 1), (6144, 256, 512, 3), (6144, 512, 768, 1),
        # ... and long reductions (>= 24 chunk steps per tile: what the balanced decomposition is for, with --ws)
        (98304, 256, 128, 3), (24576, 512, 256, 3), (6144, 512, 256, 3)]
    st = torch.cuda.current_stream().cuda_stream
    for N, Fin, Fout, K in shapes:
        x = torch.randn(N, Fin, device="cuda")
        T = torch.randn(max(K - 1, 1), N, Fin, device="cuda")
        w = torch.randn(Fin, K, Fout, device="cuda") * 0.05
        b = torch.randn(Fout, device="cuda")
        y = torch.empty(N, Fout, device="cuda")
        if zeros:
            x.zero_(); T.zero_(); w.zero_(); b.zero_()
        f = lambda: lib.dsw_cheb_mix_fwd(x.data_ptr(), T.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, Fin, Fout, K, 0, st)
        if use_ws:
            xw = torch.cat([x] + [T[k] for k in range(K - 1)], 1).contiguous()
            ww = w.permute(1, 0, 2).reshape(K * Fin, 1, Fout).contiguous()
            nws = int(lib.dsw_cheb_fwd_workspace_bytes(1, N, K * Fin, Fout, 1, 0))
            ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
            f = lambda: lib.dsw_cheb_fwd_ws(None, None, None, N, 0, xw.data_ptr(), ww.data_ptr(), b.data_ptr(), y.data_ptr(), Fout, None,
                                            1, K * Fin, Fout, 1, 0, st, None, 0, None, None, 0, ws.data_ptr(), nws)
        assert f() == 0
        us = timeit(f)
        ref = (torch.cat([x] + [T[k] for k in range(K - 1)], 1).double() @ w.permute(1, 0, 2).reshape(K * Fin, Fout).double() + b.double())
        err = ((y.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        fl = 2.0 * N * Fin * K * Fout
        by = (N * Fin * K + N * Fout) * 4
        print("N=%7d %4d->%4d K=%d  %8.1f us  %6.1f TF/s fp32  bf16-pipe %.3f  HBM %6.0f GB/s  err %.1e" % (
            N, Fin, Fout, K, us, fl / us / 1e6, 6 * fl / us / 1e6 / 2500.0, by / us / 1e3, err))


if __name__ == "__main__":
    main()
