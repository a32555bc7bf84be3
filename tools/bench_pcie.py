#!/usr/bin/env python3
"""The batch-16 training step of bench.py with the batch handed over as HOST buffers (pinned): input 16 x 1 x 192 x 192 and the float
targets (7 + 14 channels) copied host -> device every step, in front of the step on the same stream (not overlapped) -- the
PCIe-inclusive rate DESIGN.md quotes next to `value` (which has the inputs resident in HBM, as the device-side loader keeps them).

    python tools/bench_pcie.py [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nat.check(nat.lib().dfl_set_math_mode(bench.MATH['bf16s'][0]), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(1234)
net = dfl_amd.UNet(**bench.PAPER).to(dev).train()
x, tseg, theat = bench.synth_batch(16, 4321, dev)
hx, hseg, hheat = (t.cpu().pin_memory() for t in (x, tseg, theat))
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)


def step(host):
    if host:
        x.copy_(hx, non_blocking=True)
        tseg.copy_(hseg, non_blocking=True)
        theat.copy_(hheat, non_blocking=True)
    opt.zero_grad()
    seg, heat = net(x)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
    loss.backward()
    opt.step()


out = {'host_bytes_per_step': sum(t.numel() * t.element_size() for t in (hx, hseg, hheat))}
for host in (False, True, False, True):
    for _ in range(10):
        step(host)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(host)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out.setdefault('host_buffers' if host else 'resident', []).append({'ms_per_step': round(ms, 3), 'images_per_sec': round(16e3 / ms, 1)})
print(json.dumps(out))
