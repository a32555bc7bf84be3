#!/usr/bin/env python3
"""A/B timing of the recorded forward + backward programs of the paper network at batch 16 (op by op with hipEvents,
several repetitions, per kernel family) -- run several times in ONE gpurun call under different DFL_* switches: boxes
differ by up to 30 % in clock, only numbers of the same call compare.   python tools/kbench_bf16.py [mode] [reps]"""
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16s'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(bench.MATH[mode][0]), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(1234)
net = dfl_amd.UNet(**bench.PAPER).to(dev).train()
x, tseg, theat = bench.synth_batch(16, 4321, dev)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
import time
def step():
    opt.zero_grad()
    seg, heat = net(x)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
    loss.backward()
    opt.step()
    return loss.item()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
ms_step = (time.perf_counter() - t0) / 10 * 1e3
plan = [p for ps in net._plans.values() for p in ps if p.need_grad][0]
seg, heat = net(x)
hold = (torch.randn_like(seg) * 1e-6, torch.randn_like(heat) * 1e-6)
plan.head_bwd.seg, plan.head_bwd.dseg, plan.head_bwd.dheat = seg.data_ptr(), hold[0].data_ptr(), hold[1].data_ptr()
stream = torch.cuda.current_stream().cuda_stream
tot = defaultdict(float)
cnt = defaultdict(int)
for rep in range(reps):
    g = bench.op_profile(plan, lib, nat, stream)
    for k, v in g.items():
        tot[k] += v[0]
        cnt[k] = v[2]
plan.busy = False
fam = defaultdict(float)
for k, v in tot.items():
    f = 'convp' if k.startswith('convp') else ('wgradp' if k.startswith('wgradp') else k)
    fam[f] += v / reps
print('%s: step %.3f ms (%.0f images/s); op time fwd+bwd %.3f ms; ' % (os.environ.get('TAG', ''), ms_step, 16e3 / ms_step, sum(fam.values())) +
      ' '.join('%s %.3f' % (k.replace('Args', ''), v) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:9]))
if os.environ.get('DETAIL'):
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print('    %-28s %7.3f ms %3d launches' % (k, v / reps, cnt[k]))
if os.environ.get('LAYERS'):
    det = []
    bench.op_profile(plan, lib, nat, stream, detail=det)
    plan.busy = False
    print('\n'.join(det))
