#!/usr/bin/env python3
"""Is a gradient deviation of the HIP path explained by ReLU-mask / max-pool choices that differ from the fp64 oracle's?
Runs one tiny golden problem on the GPU, takes the HIP run's ReLU masks and pooling choices from the plan, and compares
the HIP gradients with (a) the clean fp64 oracle and (b) the fp64 oracle FORCED to the HIP run's masks and choices."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dfl_amd
from dfl_amd import _native as nat
from conftest import TINY_CFGS, load_golden
from oracle import ref_cpu as R
import noise_floor as NF

names = sys.argv[2:] or ['tiny_bd3_nosm']
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
lib = nat.lib()
nat.check(lib.dfl_set_math_mode({'fp32': 0, 'bf16x3': 1, 'bf16': 3, 'bf16s': 4}[mode]), 'mode')
_t = lambda a: torch.from_numpy(np.asarray(a))
for name in names:
    cfg = TINY_CFGS[name]
    g = load_golden(name)
    net = dfl_amd.UNet(**cfg)
    sd = {k[4:]: _t(v) for k, v in g.items() if k.startswith('sd0/')}
    net.load_state_dict(sd)
    net = net.to('cuda').train()
    x = _t(g['x'])
    out = net(x.cuda())
    nl = cfg['num_lands']
    seg = out[0] if nl > 0 else out
    tseg = _t(g['tseg'])
    theat = _t(g['theat']) if nl > 0 else None
    if nl > 0:
        loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(out[1], theat.shape)), (tseg.cuda(), theat.cuda()))
    else:
        loss = dfl_amd.DiceLoss2D(skip_bg=False)(dfl_amd.center_crop(seg, tseg.shape), tseg.cuda())
    loss.backward()
    plan = [p for ps in net._plans.values() for p in ps if p.need_grad][0]
    forced = NF.hip_choices(plan)
    o = R.OracleUNet(**cfg).double()
    o.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    o.train()

    def run(net_):
        oo = net_(x.double())
        s_ = oo[0] if isinstance(oo, tuple) else oo
        if theat is not None:
            l = R.dice_and_heatmap_loss_2d((R.center_crop(s_, tseg.shape), R.center_crop(oo[1], theat.shape)), (tseg.double(), theat.double()), skip_bg=False, heatmap_wgt=0.5)
        else:
            l = R.dice_loss_2d(R.center_crop(s_, tseg.shape), tseg.double(), skip_bg=False)
        return l, s_
    clean = NF.noisy_gradients(o, lambda n_: run(n_)[0], 0.0, 0)
    with NF.forced_choices(o, forced) as info:
        fgrads = NF.noisy_gradients(o, lambda n_: run(n_)[0], 0.0, 0)
    print('== %s %s: ReLU flips %s, pool flips %s' % (name, mode, info['relu_flips'], info['pool_flips']))
    for k, p in net.named_parameters():
        if clean[k] is None:
            continue
        e1 = NF.rel_l2(p.grad.cpu().numpy(), clean[k].numpy())
        e2 = NF.rel_l2(p.grad.cpu().numpy(), fgrads[k].numpy())
        print('  %-40s vs clean %.3e   vs forced %.3e' % (k, e1, e2))
