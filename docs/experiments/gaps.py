#!/usr/bin/env python3
"""Step time with and without the per-step loss.item() synchronisation, and of the optimizer step alone."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd
from dfl_amd import _native as nat
import bench
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(4), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(1234)
net = dfl_amd.UNet(**bench.PAPER).to(dev).train()
x, tseg, theat = bench.synth_batch(16, 4321, dev)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
def step(item, do_opt=True):
    opt.zero_grad()
    seg, heat = net(x)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
    loss.backward()
    if do_opt:
        opt.step()
    return loss.item() if item else loss
for mode in ('item', 'no item', 'item', 'no item', 'no item, no optimizer'):
    for _ in range(3):
        step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(mode == 'item', 'no optimizer' not in mode)
    torch.cuda.synchronize()
    print('%-24s %.3f ms per step' % (mode, (time.perf_counter() - t0) / 20 * 1e3))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0
for _ in range(10):
    step(True, False)
    torch.cuda.synchronize()
    e0.record(); opt.step(); e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print('optimizer step + weight re-layout alone: %.3f ms (GPU, events)' % (tot / 10))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
acc = [0.0] * 5
for it in range(13):
    opt.zero_grad()
    ev[0].record()
    seg, heat = net(x)
    ev[1].record()
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
    ev[2].record()
    loss.backward()
    ev[3].record()
    opt.step()
    ev[4].record()
    l = loss.item()
    torch.cuda.synchronize()
    if it >= 3:
        for i in range(4):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
print('GPU ms between in-stream events: forward %.3f, loss %.3f, backward %.3f, optimizer+relayout %.3f' % tuple(a / 10 for a in acc[:4]))
