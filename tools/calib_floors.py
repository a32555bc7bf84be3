#!/usr/bin/env python3
"""Sensitivity of every gradient tensor of every test problem (tests/problems.py) to rounding noise -- the numbers the
gradient bars of the GPU tests are made of (tests/noise_floor.py).  Runs in the BUILD container (CPU, fp64 oracle);
the results are committed under tests/golden/floors/<problem>.npz, so that no random draw and no noisy pass decides a
verdict on the GPU box.

    python tools/calib_floors.py                 # every problem of the registry that has no file yet
    python tools/calib_floors.py --force KEY...  # recompute these
    python tools/calib_floors.py --report KEY... # (GPU box) HIP error / modelled spread per tensor and arithmetic mode

Per problem: the oracle's clean gradient g and forward output o on its own activation pattern; then, with the pattern
FROZEN (ReLU masks and pooling choices fixed -- the tests compare on one pattern, see tests/noise_floor.py), `seeds` noisy
passes each for
  * convolution noise: every Conv2d / ConvTranspose2d output, data gradient and weight / bias gradient
    <- + EPS_REF x rms x N(0,1)                                   -> s_conv[t] = rms_seeds(|g_noisy - g| / |g|) / EPS_REF
  * BatchNorm noise at the fp32 rounding level 2^-23, same places -> s_bn[t]   = rms_seeds(|g_noisy - g| / |g|)
and the same for the forward output (fwd_conv per unit of noise, fwd_bn absolute).  '*' = the whole gradient.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import noise_floor as NF          # noqa: E402
import problems as PR             # noqa: E402


def n_seeds(problem):
    work = problem.x.numel() * (2 ** (problem.cfg['wf'] - 2)) ** 2
    return 16 if work < 4e7 else 8


def calibrate(problem, seeds=None):
    net = problem.oracle64()
    seeds = seeds or n_seeds(problem)
    pattern = NF.natural_choices(net, problem.run)
    with NF.forced_choices(net, pattern) as info:
        clean, out = NF.gradients(net, problem.run)
        assert info['relu_flips'] == 0 and info['pool_flips'] == 0
        names = [k for k, v in clean.items() if v is not None]
        den = {k: max(float(clean[k].pow(2).sum()), 1e-300) for k in names}
        den_all = sum(den.values())
        oden = max(float(out.pow(2).sum()), 1e-300)
        res = {}
        for what, ec, eb in (('conv', NF.EPS_REF, 0.0), ('bn', 0.0, NF.BN_EPS)):
            acc = {k: 0.0 for k in names}
            acc['*'] = 0.0
            facc = 0.0
            has_bn = any(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules())
            if what == 'bn' and not has_bn:
                res[what] = ({k: 0.0 for k in acc}, 0.0)
                continue
            for s in range(1, seeds + 1):
                g, o = NF.gradients(net, problem.run, ec, eb, s)
                num_all = 0.0
                for k in names:
                    num = float((g[k] - clean[k]).pow(2).sum())
                    num_all += num
                    acc[k] += num / den[k]
                acc['*'] += num_all / den_all
                facc += float((o - out).pow(2).sum()) / oden
            res[what] = ({k: (a / seeds) ** 0.5 for k, a in acc.items()}, (facc / seeds) ** 0.5)
    names_all = names + ['*']
    gnorm = [den[k] ** 0.5 for k in names] + [den_all ** 0.5]
    return dict(names=np.array(names_all), gnorm=np.array(gnorm), s_conv=np.array([res['conv'][0][k] / NF.EPS_REF for k in names_all]),
                s_bn=np.array([res['bn'][0][k] for k in names_all]), fwd_conv=np.float64(res['conv'][1] / NF.EPS_REF),
                fwd_bn=np.float64(res['bn'][1]), seeds=np.int64(seeds), eps_ref=np.float64(NF.EPS_REF))


def report(keys):
    """GPU box: per problem and arithmetic mode, the HIP gradient error of each tensor against the oracle on the HIP run's
    pattern, divided by the modelled spread (bar / K): the calibration of K."""
    import dfl_amd
    from dfl_amd import _native as nat
    lib = nat.lib()
    for key in keys:
        pr = PR.REGISTRY[key]()
        if pr is None:
            continue
        for mode, mid in (('fp32', 0), ('bf16x3', 1), ('bf16s', 4)):
            if mid == 4 and pr.cfg['wf'] < 4:
                continue
            nat.check(lib.dfl_set_math_mode(mid), 'mode')
            try:
                net = dfl_amd.UNet(**pr.cfg) if 'in_channels' in pr.cfg else dfl_amd.UNet(1, **pr.cfg)
                net.load_state_dict(pr.sd)
                net = net.to('cuda').train()
                out = net(pr.x.cuda())
                seg = out[0] if isinstance(out, tuple) else out
                if pr.theat is not None:
                    loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)(
                        (dfl_amd.center_crop(seg, pr.tseg.shape), dfl_amd.center_crop(out[1], pr.theat.shape)), (pr.tseg.cuda(), pr.theat.cuda()))
                else:
                    loss = dfl_amd.DiceLoss2D(skip_bg=pr.skip_bg)(dfl_amd.center_crop(seg, pr.tseg.shape), pr.tseg.cuda())
                loss.backward()
                gc = NF.cached_check(key, lambda: pr)
                ref, oout, info = gc.reference(NF.train_plan(net))
                d = NF.rel_l2(seg.detach().double().cpu().numpy(), oout.numpy())
                eps_eff, bars = gc.bars(d, NF.conv_rel_error(mode))
                ratios = []
                for k, p in net.named_parameters():
                    if ref[k] is None:
                        continue
                    e = NF.rel_l2(p.grad.cpu().numpy(), ref[k].numpy())
                    ratios.append((e / ((bars[k] - gc.abs_term) / NF.K_TENSOR + gc.abs_term), e, k))
                ratios.sort(reverse=True)
                r = np.array([t[0] for t in ratios])
                print('%-28s %-7s eps_eff %.2e fwd %.2e flips %d/%d margin %.1e | error/spread median %.2f p90 %.2f max %.2f (%s %.2e)' % (
                    key, mode, eps_eff, d, info['relu_flips'], info['pool_flips'], info['max_margin'], np.median(r), np.percentile(r, 90),
                    r[0], ratios[0][2], ratios[0][1]), flush=True)
            finally:
                nat.check(lib.dfl_set_math_mode(0), 'mode')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('keys', nargs='*')
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--report', action='store_true')
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--add-gnorm', action='store_true', help='add the clean gradient norms to files written before they were stored')
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    keys = args.keys or sorted(PR.REGISTRY)
    if args.report:
        return report(keys)
    os.makedirs(NF.FLOOR_DIR, exist_ok=True)
    if args.add_gnorm:
        for key in keys:
            path = os.path.join(NF.FLOOR_DIR, key + '.npz')
            if not os.path.exists(path):
                continue
            d = dict(np.load(path, allow_pickle=False))
            if 'gnorm' in d:
                continue
            pr = PR.REGISTRY[key]()
            clean, _ = NF.gradients(pr.oracle64(), pr.run)
            names = [str(n) for n in d['names']]
            gn = [float(clean[k].pow(2).sum()) ** 0.5 for k in names[:-1]]
            d['gnorm'] = np.array(gn + [sum(g * g for g in gn) ** 0.5])
            np.savez_compressed(path, **d)
            print('%s: gradient norms added' % key, flush=True)
        return
    for key in keys:
        path = os.path.join(NF.FLOOR_DIR, key + '.npz')
        if os.path.exists(path) and not args.force and not args.keys:
            continue
        pr = PR.REGISTRY[key]()
        if pr is None:
            print('%s: rejected by the reference architecture, no file' % key)
            continue
        assert pr.key == key, (pr.key, key)
        t0 = time.time()
        d = calibrate(pr)
        np.savez_compressed(path, **d)
        sc = d['s_conv']
        print('%-30s %2d seeds %6.1f s: s_conv median %.3g max %.3g (%s) whole %.3g | s_bn max %.2e | fwd_conv %.3g fwd_bn %.2e' % (
            key, int(d['seeds']), time.time() - t0, np.median(sc[:-1]), sc[:-1].max(), d['names'][int(sc[:-1].argmax())], sc[-1],
            d['s_bn'][:-1].max(), float(d['fwd_conv']), float(d['fwd_bn'])), flush=True)


if __name__ == '__main__':
    main()
