#!/usr/bin/env python3
"""dfl_reduce_batch alone: time of summing `count` slices of n floats for slice strides n, n + pad (HBM channel aliasing probe)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402,F401
from dfl_amd import _native as nat  # noqa: E402

lib = nat.lib()
DEV = 'cuda'


def run(n, count, pad, reps=20):
    stride = n + pad
    src = torch.randn(count * stride, device=DEV)
    dst = torch.empty(n, device=DEV)
    job = (nat.ReduceJob * 1)()
    job[0].src, job[0].dst, job[0].n, job[0].stride, job[0].count, job[0].first_block, job[0].T = src.data_ptr(), dst.data_ptr(), n, stride, count, 0, 1
    blocks = lib.dfl_reduce_job_blocks(n, count)
    dev = torch.from_numpy(np.frombuffer(bytes(job), dtype=np.uint8).copy()).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        nat.check(lib.dfl_reduce_batch(dev.data_ptr(), 1, blocks, st), 'reduce')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        nat.check(lib.dfl_reduce_batch(dev.data_ptr(), 1, blocks, st), 'reduce')
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ref = src.view(count, stride)[:, :n].double().sum(0).float()
    err = float((dst - ref).abs().max())
    print('n %8d count %4d pad %5d: %7.1f us  %6.2f TB/s  (%d blocks, err %.1e)' % (n, count, pad, us, 4.0 * n * count / us / 1e6, blocks, err))


for n, count in ((36864, 256), (9216, 512), (147456, 64), (36864, 64), (589824, 8), (2359296, 4)):
    for pad in (0, 32, 64, 256, 1056):
        run(n, count, pad)
