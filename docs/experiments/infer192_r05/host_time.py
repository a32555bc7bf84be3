#!/usr/bin/env python3
"""Host time of one eval forward call (batch 1, 192x192) against its GPU time: 300 calls enqueued, clock read before and after
the final synchronize.  python docs/experiments/infer192_r05/host_time.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402

lib = nat.lib()
nat.check(lib.dfl_set_math_mode(4), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(7)
net = dfl_amd.UNet(**bench.PAPER).to(dev).eval()
x = torch.randn(1, 1, 192, 192, device=dev)
with torch.no_grad():
    for _ in range(20):
        net(x)
    torch.cuda.synchronize()
    for n in (20, 300):
        t0 = time.perf_counter()
        for _ in range(n):
            net(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('%d calls: enqueued in %.3f ms per call, finished in %.3f ms per call' % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
