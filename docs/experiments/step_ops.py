#!/usr/bin/env python3
"""Every op of the recorded training step (paper network, batch 16) with its time (hipEvent pair per op, best of 5) and shape:
python docs/experiments/step_ops.py [mode]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16s'
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(bench.MATH[mode][0]), 'mode')
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
    return loss.item()


for _ in range(3):
    step()
torch.cuda.synchronize()
plan = [p for ps in net._plans.values() for p in ps if p.need_grad][0]
seg, heat = net(x)
hold = (torch.randn_like(seg) * 1e-6, torch.randn_like(heat) * 1e-6)
plan.head_bwd.seg, plan.head_bwd.dseg, plan.head_bwd.dheat = seg.data_ptr(), hold[0].data_ptr(), hold[1].data_ptr()
stream = torch.cuda.current_stream().cuda_stream
for name, prog in (('fwd', plan.fwd), ('bwd', plan.bwd)):
    best = None
    for rep in range(5):
        ms = prog.run_timed(stream)
        best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
    tot = 0.0
    for st, t in zip(prog.structs, best):
        tot += t
        if isinstance(st, nat.ConvArgs):
            cfg = lib.dfl_conv_config(C.addressof(st))
            kn = bench.CONV_KERNELS[cfg] if cfg < 16 else 'convp<%s>' % bench.CONVP_TILES[cfg - 16]
            M = st.N * (st.Hin * st.Win if st.scatter2x2 else st.Hout * st.Wout)
            fl = 2.0 * M * st.KH * st.KW * st.Cin * st.Ntot
            by = 2.0 * (st.N * st.Hin * st.Win * st.Cin * (2 if st.x_mode else 1) + M * st.Ntot * (2 if st.add else 1))
            print('%s %-20s %7.1f us %6.0f TF %6.0f GB/s  %dx%d Cin%d -> %dx%d N%d k%d s%d sp%d%s%s%s' % (
                name, kn, t * 1e3, fl / t / 1e9, by / t / 1e6, st.Hin, st.Win, st.Cin, st.Hout, st.Wout, st.Ntot, st.KH, st.stride, st.splits,
                ' brb' if st.x_mode else '', ' xout' if st.x_out else '', ' stats' if (st.stat_totals or st.stat_partials) else ''))
        elif isinstance(st, nat.WgradArgs):
            M = st.N * st.Hout * st.Wout
            fl = 2.0 * M * st.KH * st.KW * st.Cg * st.Cm
            by = 2.0 * (st.N * st.Hin * st.Win * st.Cg + M * st.Cm * (2 if st.d_mode else 1))
            print('%s %-20s %7.1f us %6.0f TF %6.0f GB/s  %dx%d Cg%d Cm%d k%d sp%d%s' % (
                name, 'wgrad', t * 1e3, fl / t / 1e9, by / t / 1e6, st.Hout, st.Wout, st.Cg, st.Cm, st.KH, st.splits, ' dbrb' if st.d_mode else ''))
        else:
            print('%s %-20s %7.1f us' % (name, type(st).__name__, t * 1e3))
    print('%s total %.3f ms' % (name, tot))
