#!/usr/bin/env python3
"""Which torch (non-dsw) kernels run inside one eager UNet training step, and from which op / autograd node.
usage: tools/find_torch_kernels.py [unet|c5]   (GPU)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "deepsphere-weather_amd"), REPO]
import torch
import bench
from torch.profiler import profile, ProfilerActivity

wl = bench.WORKLOADS["unet"]
dev = "cuda:0"
model = bench.make_unet(wl, 20, dev)
V = 12 * wl["nside"] ** 2
x = torch.randn(wl["batch"], 3, V, 6, device=dev)
tgt = torch.randn(wl["batch"], 1, V, 2, device=dev)
params = [p for p in model.parameters() if p.requires_grad]


def step():
    model.zero_grad(set_to_none=True)
    y = model(x)
    loss = ((y - tgt) ** 2).mean()
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.self_device_time_total > 0:
        parent = e.cpu_parent
        chain = []
        while parent is not None and len(chain) < 3:
            chain.append(parent.name)
            parent = parent.cpu_parent
        rows.append((e.self_device_time_total, e.name, str(e.input_shapes)[:90], " < ".join(chain)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("aten ops with device time in one step: %.1f us total" % tot)
for t, n, s, c in rows[:40]:
    print("%8.1f us  %-28s %-90s %s" % (t, n, s, c))
