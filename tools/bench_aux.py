#!/usr/bin/env python3
"""Timings of the kernels either side of the network at the sizes of BASELINE's configurations:
the loader arithmetic (dfl_prep_batch), landmark extraction (dfl_est_lands) and hard Dice (dfl_hard_dice).  One JSON line each:
milliseconds per call (hipEvent pair over `reps` calls, inputs resident in HBM) and the bytes the call has to touch.
    python tools/bench_aux.py            # on the GPU box"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import dataset as D, util  # noqa: E402

dev = torch.device('cuda:0')


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, nbytes, **kw):
    print(json.dumps(dict(kernel=name, ms_per_call=round(ms, 4), mb_touched=round(nbytes / 1e6, 1), gb_per_s=round(nbytes / ms / 1e6, 1), **kw)))


g = torch.Generator().manual_seed(3)
for (tag, n, hw, pad_to, batch) in (('configs[1] 8x-downsampled', 64, 184, 192, 16), ('configs[3] 2x-downsampled', 16, 736, 768, 8),
                                      ('configs[4] full resolution', 4, 1436, 1440, 1)):
    projs = torch.rand(n, 1, hw, hw, generator=g)
    segs = torch.randint(0, 7, (n, hw, hw), generator=g)
    lands = torch.rand(n, 2, 14, generator=g) * hw
    ds = D.DeviceDataSet(projs, segs, lands, proj_pad_dim=pad_to, num_classes=7, device=dev)
    idx = list(range(batch))
    ms = timed(lambda: ds._prepare(idx))
    out_bytes = batch * (pad_to * pad_to * 4 + 7 * hw * hw * 4 + 14 * hw * hw * 4)
    in_bytes = batch * (hw * hw * 4 + hw * hw)
    report('dfl_prep_batch (+ index_select, allocation)', ms, in_bytes + out_bytes, size=tag, batch=batch)
    heats = torch.rand(batch, 14, hw, hw, device=dev)
    lab = torch.randint(0, 7, (batch, hw, hw), device=dev, dtype=torch.uint8)
    ms = timed(lambda: util.est_lands(heats, lab, [1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 1, 2]))
    report('dfl_est_lands', ms, heats.numel() * 4 + lab.numel(), size=tag, batch=batch)
    lab2 = torch.randint(0, 7, (batch, hw, hw), device=dev, dtype=torch.uint8)
    ms = timed(lambda: util.hard_dice(lab, lab2, 7))
    report('dfl_hard_dice', ms, 2 * lab.numel(), size=tag, batch=batch)
