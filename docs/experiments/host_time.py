#!/usr/bin/env python3
"""Host enqueue time of each part of a training step against the GPU time (bf16s, paper batch 16)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd
from dfl_amd import _native as nat
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16s'
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(bench.MATH[mode][0]), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(1234)
net = dfl_amd.UNet(**bench.PAPER).to(dev).train()
x, tseg, theat = bench.synth_batch(16, 4321, dev)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
acc = [0.0] * 7
N = 20
for it in range(N + 3):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    seg, heat = net(x); t.append(time.perf_counter())
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat)); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    l = loss.item(); t.append(time.perf_counter())
    if it >= 3:
        for i in range(6):
            acc[i] += t[i + 1] - t[i]
print('host ms per step: zero_grad %.3f, forward %.3f, loss %.3f, backward %.3f, optimizer %.3f, wait for the GPU (loss.item) %.3f; total %.3f'
      % tuple([a / N * 1e3 for a in acc[:6]] + [sum(acc[:6]) / N * 1e3]))
