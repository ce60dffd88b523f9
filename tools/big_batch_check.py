import sys, os
sys.path[:0] = ['/root/repo/deepsphere-weather_amd', '/root/repo']
import torch
from dsw_amd import sphere
from modules.layers import ConvCheb, prepare_torch_laplacian
torch.manual_seed(0)
g = sphere.SphereHealpix(64, nest=True, k=8)
lap = prepare_torch_laplacian(g.L, lmax=1.95)
for (fin, fout, K, dt) in [(32, 64, 3, torch.float32), (128, 64, 3, torch.float32), (64, 128, 5, torch.bfloat16)]:
    layer = ConvCheb(fin, fout, K, laplacian=lap).to('cuda').to(dt)
    B = 96
    x = torch.randn(B, 49152, fin, device='cuda', dtype=dt, requires_grad=True)
    gy = torch.randn(B, 49152, fout, device='cuda', dtype=dt)
    y = layer(x); y.backward(gy)
    dw_big = layer.weight.grad.float().clone(); layer.zero_grad()
    errs = []
    dw_acc = torch.zeros_like(dw_big)
    for b in (0, 47, 95):
        xb = x[b:b+1].detach().clone().requires_grad_(True)
        yb = layer(xb); yb.backward(gy[b:b+1])
        errs.append(((yb - y[b:b+1]).abs().max() / y.abs().max()).item())
        errs.append(((xb.grad - x.grad[b:b+1]).abs().max() / x.grad.abs().max()).item())
        layer.zero_grad()
    print(fin, fout, K, dt, "max rel diff big-batch vs single-sample:", max(errs), "elements", x.numel())
