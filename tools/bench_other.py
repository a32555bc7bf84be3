#!/usr/bin/env python3
"""Timings of the BASELINE configurations that are not the bench.py headline (one MI355X, fp32):
  configs[3]  2x-downsampled 736x736 (padded 768), paper U-Net dual head, batch 8: training images/s
  configs[4]  full-resolution 1436x1436 (padded 1440), 5-net ensemble, eval forward + ensemble reduction: ms per image
Same step bodies as bench.py / util.seg_dataset_ensemble; prints one JSON line per configuration."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import dfl_amd  # noqa: E402
from dfl_amd import util  # noqa: E402

PAPER = dict(n_classes=7, depth=6, wf=5, batch_norm=True, padding=True, max_pool=False, num_lands=14, do_res=True,
             block_depth=2)


def train_cfg3(steps=6, warmup=2, B=8, H=736, P=768):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    net = dfl_amd.UNet(**PAPER).to(dev)
    opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 1, P, P, generator=g).to(dev)
    lab = torch.randint(0, 7, (B, H, H), generator=g)
    tseg = torch.stack([(lab == c) for c in range(7)], 1).float().to(dev)
    theat = (torch.rand(B, 14, H, H, generator=g) * 0.02).to(dev)
    net.train()

    def step():
        opt.zero_grad()
        seg, heat = net(x)
        loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
        loss.backward()
        opt.step()
        return loss.item()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop = 54.5e9 * (P * P) / (192 * 192) * B
    print(json.dumps({'config': 'configs[3] 736x736 (padded 768) train, batch %d' % B, 'math': os.environ.get('DFL_MATH', 'fp32'), 'images_per_sec': round(B * steps / dt, 2),
                      'ms_per_step': round(dt / steps * 1e3, 2), 'tflops': round(flop / (dt / steps) / 1e12, 1),
                      'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


def infer_cfg4(reps=5, H=1436, P=1440, nnets=5):
    dev = torch.device('cuda:0')
    nets = []
    for i in range(nnets):
        torch.manual_seed(10 + i)
        nets.append(dfl_amd.UNet(**PAPER).to(dev).eval())
    x = torch.randn(1, 1, P, P, generator=torch.Generator().manual_seed(3)).to(dev)

    def one():
        with torch.no_grad():
            outs = [n(x) for n in nets]
            return util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (H, H))
    one()
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = 18.17e9 * (P * P) / (192 * 192) * nnets
    print(json.dumps({'config': 'configs[4] 1436x1436 (padded 1440) %d-net ensemble inference' % nnets, 'math': os.environ.get('DFL_MATH', 'fp32'),
                      'ms_per_image': round(dt * 1e3, 2), 'ms_per_net': round(dt * 1e3 / nnets, 2),
                      'tflops': round(flop / dt / 1e12, 1), 'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', '3'):
        train_cfg3()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    if which in ('all', '4'):
        infer_cfg4()
