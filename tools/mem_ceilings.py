#!/usr/bin/env python3
"""Read-only / write-only / mixed streaming ceilings on this box (torch kernels), sizes of the NS workload."""
import torch
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
E = 16 * 49152 * 32
x = torch.randn(E, device=dev); x2 = torch.randn(E, device=dev); x3 = torch.randn(E, device=dev)
y = torch.empty(2 * E, device=dev)
big = torch.randn(8 * E, device=dev)
us = t(lambda: y.zero_());            print(f"write-only 200 MB: {us:7.1f} us {2*E*4/us/1e3:7.0f} GB/s")
us = t(lambda: big.zero_());          print(f"write-only 800 MB: {us:7.1f} us {8*E*4/us/1e3:7.0f} GB/s")
us = t(lambda: x.sum());              print(f"read-only  100 MB: {us:7.1f} us {E*4/us/1e3:7.0f} GB/s")
us = t(lambda: big.sum());            print(f"read-only  800 MB: {us:7.1f} us {8*E*4/us/1e3:7.0f} GB/s")
us = t(lambda: torch.cat((x, x2), out=y)); print(f"cat 2x100 -> 200 MB (200R+200W): {us:7.1f} us {4*E*4/us/1e3:7.0f} GB/s")
z = torch.empty(E, device=dev)
us = t(lambda: torch.add(x, x2, out=z)); print(f"add 2R+1W (300 MB): {us:7.1f} us {3*E*4/us/1e3:7.0f} GB/s")
us = t(lambda: torch.addcmul(x, x2, x3, out=z)); print(f"addcmul 3R+1W (400 MB): {us:7.1f} us {4*E*4/us/1e3:7.0f} GB/s")
