#!/usr/bin/env python3
"""Eval-mode forward of the paper network, batch 1 at 192x192, replayed N times (for rocprofv3 --kernel-trace):
python docs/experiments/infer192_r05/infer_loop.py [replays] [mode]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16s'
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(bench.MATH[mode][0]), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(7)
net = dfl_amd.UNet(**bench.PAPER).to(dev).eval()
x = torch.randn(1, 1, 192, 192, device=dev)
with torch.no_grad():
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
print('%.4f ms per forward' % ((time.perf_counter() - t0) / n * 1e3))
