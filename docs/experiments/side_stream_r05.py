#!/usr/bin/env python3
"""Round 5 re-measurement of the side-stream weight gradients (plan.SIDE_STREAM, off since round 1: measured with the fp32 kernels):
a layer's weight gradient next to the NEXT layer's data gradient -- the drain of one kernel over the fill of the other -- with the
bf16 patch kernels.  Training graphs serialise branches, so they are off here.  python docs/experiments/side_stream_r05.py <0|1> [max pixels]"""
import os
import sys
import time

os.environ['DFL_TRAIN_GRAPH'] = '0'
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402
import importlib
plan_mod = importlib.import_module(dfl_amd.UNet.__module__.rsplit('.', 1)[0] + '.plan')

side = int(sys.argv[1]) if len(sys.argv) > 1 else 0
plan_mod.UNetPlan.SIDE_STREAM = bool(side)
if len(sys.argv) > 2:
    plan_mod.UNetPlan.SIDE_MAX_PIXELS = int(sys.argv[2])
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(4), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(1234)
net = dfl_amd.UNet(**bench.PAPER).to(dev).train()
x, tseg, theat = bench.synth_batch(16, 4321, dev)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)


def step():
    opt.zero_grad()
    seg, heat = net(x)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
    loss.backward()
    opt.step()
    return loss


for _ in range(200):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        l = step()
    l.item()
    torch.cuda.synchronize()
    print('side %d: %.3f ms per step' % (side, (time.perf_counter() - t0) / 100 * 1e3))
