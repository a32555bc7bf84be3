#!/usr/bin/env python3
"""Calibration run for the gradient bars of tests/test_gpu_unet.py (GPU box): per product arithmetic, the measured
per-convolution error eps, the HIP path's per-tensor gradient error against the fp64 oracle and the oracle's own spread
under noise of size eps (tests/noise_floor.py).  Prints the ratio table the test bars (k x spread) are read from.

    python tools/calib_noise.py [--batch16] > gpurun_out/calib_noise.txt
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import dfl_amd                                     # noqa: E402
from dfl_amd import _native as nat                 # noqa: E402
from conftest import PAPER_CFGS, TINY_CFGS, load_golden   # noqa: E402
from oracle import ref_cpu as R                    # noqa: E402
import noise_floor as NF                           # noqa: E402

MODES = {'fp32': 0, 'bf16x3': 1, 'bf16': 3}
DEV = 'cuda'


def problem(name, B):
    seed, cfg = PAPER_CFGS[name]
    torch.manual_seed(seed)
    onet = R.OracleUNet(**cfg)
    gen = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, 1, 192, 192, generator=gen)
    lab = torch.randint(0, 7, (B, 184, 184), generator=gen)
    tseg = R.one_hot_masks(lab, 7)
    theat = torch.rand(B, 14, 184, 184, generator=gen) * 0.02 if cfg['num_lands'] > 0 else None
    return cfg, onet, x, tseg, theat


def oracle_loss(tseg, theat, x, dt):
    def f(net):
        o = net(x.to(dt))
        if theat is not None:
            return R.dice_and_heatmap_loss_2d((R.center_crop(o[0], tseg.shape), R.center_crop(o[1], theat.shape)),
                                              (tseg.to(dt), theat.to(dt)), skip_bg=False, heatmap_wgt=0.5)
        return R.dice_loss_2d(R.center_crop(o, tseg.shape), tseg.to(dt), skip_bg=False)
    return f


def hip_grads(cfg, sd, x, tseg, theat):
    net = dfl_amd.UNet(**cfg)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    out = net(x.to(DEV))
    if theat is not None:
        loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)(
            (dfl_amd.center_crop(out[0], tseg.shape), dfl_amd.center_crop(out[1], theat.shape)), (tseg.to(DEV), theat.to(DEV)))
    else:
        loss = dfl_amd.DiceLoss2D(skip_bg=False)(dfl_amd.center_crop(out, tseg.shape), tseg.to(DEV))
    loss.backward()
    return {k: (None if p.grad is None else p.grad.detach().cpu()) for k, p in net.named_parameters()}


def main():
    lib = nat.lib()
    B = 16 if '--batch16' in sys.argv else 2
    eps = {}
    for m, code in MODES.items():
        nat.check(lib.dfl_set_math_mode(code), 'mode')
        NF._EPS_CACHE.clear()
        eps[m] = NF.conv_rel_error(m)
    nat.check(lib.dfl_set_math_mode(0), 'mode')
    print('measured per-convolution relative error:', json.dumps(eps))
    for name in PAPER_CFGS:
        cfg, onet, x, tseg, theat = problem(name, B)
        sd = {k: v.clone() for k, v in onet.state_dict().items()}
        o64 = R.OracleUNet(**cfg).double()
        o64.load_state_dict(sd)
        o64.train()
        t0 = time.time()
        floors = {}
        clean = None
        for m in MODES:
            clean, floors[m] = NF.gradient_noise_floor(o64, oracle_loss(tseg, theat, x, torch.float64), eps[m], seeds=(1, 2, 3))
        print('%s batch %d: oracle fp64 noise floors in %.1f s' % (name, B, time.time() - t0))
        for m, code in MODES.items():
            nat.check(lib.dfl_set_math_mode(code), 'mode')
            g = hip_grads(cfg, sd, x, tseg, theat)
            nat.check(lib.dfl_set_math_mode(0), 'mode')
            rows = []
            num = den = 0.0
            for k, ref in clean.items():
                if ref is None:
                    assert g[k] is None, k
                    continue
                e = NF.rel_l2(g[k].numpy(), ref.numpy())
                num += float((g[k].double() - ref).pow(2).sum())
                den += float(ref.pow(2).sum())
                rows.append((e / max(floors[m][k], 1e-30), e, floors[m][k], k))
            rows.sort(reverse=True)
            whole = (num / den) ** 0.5
            print('  %-7s eps %.2e  whole-gradient rel-L2 %.3e (floor %.3e, ratio %.2f)  worst per-tensor ratio %.2f  median ratio %.2f' % (
                m, eps[m], whole, floors[m]['*'], whole / floors[m]['*'], rows[0][0], float(np.median([r[0] for r in rows]))))
            for r in rows[:6]:
                print('      ratio %6.2f  err %.3e  floor %.3e  %s' % r)
    # tiny presets, same table (worst ratio only)
    for name, cfg in TINY_CFGS.items():
        g_ = load_golden(name)
        if not any(k.startswith('grad/') for k in g_):
            continue
        sd = {k[4:]: torch.from_numpy(v) for k, v in g_.items() if k.startswith('sd0/')}
        x = torch.from_numpy(g_['x'])
        tseg = torch.from_numpy(g_['tseg'])
        theat = torch.from_numpy(g_['theat']) if 'theat' in g_ else None
        o64 = R.OracleUNet(**cfg).double()
        o64.load_state_dict(sd)
        o64.train()
        for m, code in MODES.items():
            clean, floor = NF.gradient_noise_floor(o64, oracle_loss(tseg, theat, x, torch.float64), eps[m], seeds=(1, 2, 3, 4))
            nat.check(lib.dfl_set_math_mode(code), 'mode')
            g = hip_grads(cfg, sd, x, tseg, theat)
            nat.check(lib.dfl_set_math_mode(0), 'mode')
            rows = []
            for k, ref in clean.items():
                if ref is None:
                    continue
                e = NF.rel_l2(g[k].numpy(), ref.numpy())
                rows.append((e / max(floor[k], 1e-30), e, floor[k], k))
            rows.sort(reverse=True)
            print('%s %-7s worst ratio %.2f (err %.3e floor %.3e %s), median ratio %.2f' % (
                name, m, rows[0][0], rows[0][1], rows[0][2], rows[0][3], float(np.median([r[0] for r in rows]))))


if __name__ == '__main__':
    main()
