#!/usr/bin/env python3
"""Five nets on one 192x192 image (the ensemble loop of util.py:318-356 at the 8x-downsampled size): one after the other on one stream
against one stream per net -- their launch chains are latency-bound and independent.  python docs/experiments/infer192_r05/ensemble_streams.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat, util  # noqa: E402
import bench  # noqa: E402

lib = nat.lib()
nat.check(lib.dfl_set_math_mode(4), 'mode')
dev = torch.device('cuda:0')
nets = []
for i in range(5):
    torch.manual_seed(10 + i)
    nets.append(dfl_amd.UNet(**bench.PAPER).to(dev).eval())
x = torch.randn(1, 1, 192, 192, device=dev)
streams = [torch.cuda.Stream() for _ in nets]


def serial():
    outs = [n(x) for n in nets]
    return util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (184, 184))


def parallel():
    cur = torch.cuda.current_stream()
    outs = []
    for n, s in zip(nets, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(n(x))
    for s in streams:
        cur.wait_stream(s)
    for o in outs:
        for t in o:
            t.record_stream(cur)
    return util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (184, 184))


with torch.no_grad():
    ref = serial()
    got = parallel()
    torch.cuda.synchronize()
    print('labels equal:', bool(torch.equal(ref[0], got[0])), ' heats equal:', bool(torch.equal(ref[1], got[1])))
    for name, fn in (('serial', serial), ('one stream per net', parallel), ('serial', serial), ('one stream per net', parallel)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print('%-20s %.3f ms per image (5 nets + reduction); host enqueue %.3f ms per image' % (name, (time.perf_counter() - t0) / 100 * 1e3, (t1 - t0) / 100 * 1e3))
