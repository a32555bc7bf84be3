#!/usr/bin/env python3
"""End-to-end walk through the path with the pieces wired as the reference wires them, on a synthetic data set
(no HDF5 file needed): GPU-resident loader -> U-Net -> Dice + heat-map loss -> SGD with warm restarts
(train.py:380-440) -> validation loss (util.test_dataset) -> 2-net ensemble labels and heat maps
(test_ensemble.py -> util.seg_dataset_ensemble) -> hard Dice (compute_actual_dice_on_test.py) -> landmark locations
(est_lands_csv.py).  Everything the GPU does goes through libdfl_hip.so.

    python examples/train_toy.py [--epochs 40] [--images 32] [--math bf16x3]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat, dataset, util  # noqa: E402


def toy_ellipses(n, H, W, seed, num_lands=14):
    """Noise + 6 axis-aligned ellipses (labels 1..6); landmarks = ellipse centres and top points (+ one fixed, one out of view)."""
    g = torch.Generator().manual_seed(seed)
    projs = 0.1 * torch.randn(n, H, W, generator=g)
    segs = torch.zeros(n, H, W, dtype=torch.uint8)
    lands = torch.zeros(n, 2, num_lands)
    Y, X = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    for i in range(n):
        for c in range(1, 7):
            cx, cy = float(torch.rand(1, generator=g)) * W * 0.6 + W * 0.2, float(torch.rand(1, generator=g)) * H * 0.6 + H * 0.2
            rx, ry = float(torch.rand(1, generator=g)) * W * 0.10 + W * 0.05, float(torch.rand(1, generator=g)) * H * 0.10 + H * 0.05
            m = ((X - cx) / rx) ** 2 + ((Y - cy) / ry) ** 2 <= 1.0
            segs[i][m] = c
            projs[i][m] += 0.3 * c
            lands[i, :, c - 1] = torch.tensor([cx, cy])
            lands[i, :, 6 + c - 1] = torch.tensor([cx, cy - ry])
        lands[i, :, 12] = torch.tensor([W * 0.25, H * 0.25])
        lands[i, :, 13] = torch.tensor([-5.0, H * 0.5])
    return projs, segs, lands


class _MemH5:
    """The two methods of an h5py.File that util.seg_dataset* use, backed by numpy arrays."""

    def __init__(self):
        self.d = {}

    def create_dataset(self, name, shape, dtype='f4', **kw):
        self.d[name] = np.zeros(shape, dtype=dtype)
        return self.d[name]


def train_one(net, train_ds, valid_ds, epochs, batch, lr, heat_coeff=0.5):
    """train.py:287-334, 380-440 without the bookkeeping."""
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=heat_coeff)
    opt = dfl_amd.SGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4, nesterov=True)
    sched = dfl_amd.WarmRestartLR(opt, init_run_period_epochs=2, lr_min=lr * 1e-2)
    n = len(train_ds)
    hist = []
    for epoch in range(epochs):
        net.train()
        done, run = 0, 0.0
        for (proj, mask, lands, heat) in train_ds.batches(batch, shuffle=True):
            heats = heat.view(heat.shape[0], heat.shape[1], heat.shape[3], heat.shape[4])
            opt.zero_grad()
            seg, hm = net(proj)
            loss = crit((dfl_amd.center_crop(seg, mask.shape), dfl_amd.center_crop(hm, heats.shape)), (mask, heats))
            loss.backward()
            opt.step()
            done += proj.shape[0]
            sched.intra_epoch_step(min(done / n, 1.0))
            run += loss.item() * proj.shape[0]
        sched.step()
        vmean, vstd = util.test_dataset(valid_ds, net, num_lands=14)
        hist.append((run / n, float(vmean)))
    return hist


def main(epochs=40, images=32, math='bf16x3', size=46, quiet=False):
    nat.check(nat.lib().dfl_set_math_mode({'fp32': 0, 'bf16x3': 1, 'bf16x6': 2, 'bf16': 3}[math]), 'dfl_set_math_mode')
    try:
        dev = util.get_device()
        projs, segs, lands = toy_ellipses(images + 8, size, size, seed=5)
        mk = lambda sl: dataset.DeviceDataSet(projs[sl].unsqueeze(1), segs[sl], lands[sl], proj_pad_dim=48, num_classes=7, device=dev)
        train_ds, valid_ds = mk(slice(0, images)), mk(slice(images, images + 8))
        for ds in (train_ds, valid_ds):
            ds.rob_orig_img_shape = (size, size)
        nets, hists = [], []
        for k in range(2):
            torch.manual_seed(100 + k)
            net = dfl_amd.UNet(1, n_classes=7, depth=3, wf=3, padding=True, batch_norm=True, max_pool=False, num_lands=14).to(dev)
            hists.append(train_one(net, train_ds, valid_ds, epochs, batch=8, lr=0.05))
            nets.append(net)
        out = _MemH5()
        times = []
        util.seg_dataset_ensemble(valid_ds, nets, out, dev=dev, num_lands=14, times=times)
        est = torch.from_numpy(out.d['nn-segs']).to(dev)
        dice = util.hard_dice(est, segs[images:images + 8].to(dev), 7)
        rc = util.est_lands(torch.from_numpy(out.d['nn-heats']).to(dev), est, [1, 2, 3, 4, 5, 6] * 2 + [None, None])
        res = {'math': math, 'epochs': epochs, 'train_loss_first_last': [round(hists[0][0][0], 4), round(hists[0][-1][0], 4)],
               'valid_loss_last': round(hists[0][-1][1], 4), 'ensemble_mean_hard_dice': round(float(dice.mean()), 4),
               'landmarks_found': int((rc[..., 0] >= 0).sum()), 'landmarks_total': int(rc.shape[0] * rc.shape[1]),
               'ensemble_ms_per_image': round(1e3 * float(np.mean(times)), 3)}
        if not quiet:
            print(json.dumps(res))
        return res
    finally:
        nat.check(nat.lib().dfl_set_math_mode(0), 'dfl_set_math_mode')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=40)
    ap.add_argument('--images', type=int, default=32)
    ap.add_argument('--math', default='bf16x3', choices=['fp32', 'bf16x3', 'bf16x6', 'bf16'])
    a = ap.parse_args()
    main(a.epochs, a.images, a.math)
