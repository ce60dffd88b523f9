#!/usr/bin/env python3
"""Calibration: achievable HBM streaming rates on this box (torch copy / add kernels) for the
tensor sizes of the north-star workload.  Prints GB/s counting bytes read + written."""
import torch

dev = "cuda:0"
for mb in (100, 200, 400, 1600):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    z = torch.randn(n, device=dev)
    for name, fn, nbytes in (
        ("copy  (1R+1W)", lambda: y.copy_(x), 2 * n * 4),
        ("add   (2R+1W)", lambda: torch.add(x, z, out=y), 3 * n * 4),
        ("scale (1R+1W)", lambda: torch.mul(x, 2.0, out=y), 2 * n * 4),
    ):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            fn()
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / 20
        print(f"{mb:5d} MB tensors  {name}: {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s")
