#!/usr/bin/env python3
"""Phase clocks of the patch-resident convolution kernel (diagnosis build: csrc/convp_bf16.hip with -DDFL_CONVP_TRACE,
see docs/experiments/build_trace.sh): per workgroup the shader-clock time in staging, in the k-loop and in the epilogue, for
the 3x3 layer shapes of the paper network at batch 16.   DFL_LIB_OVERRIDE=docs/experiments/bin/libdfl_trace.so python docs/experiments/convp_trace.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dfl_amd  # noqa: E402,F401
from dfl_amd import _native as nat  # noqa: E402

DEV = 'cuda'
BF = torch.bfloat16
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(4), 'mode')
os.environ['DFL_CONVP_DEBUG'] = '1'


def run(B, Cin, Cout, H, KH=3):
    import test_gpu_bf16 as T
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, KH, KH, generator=g) * 0.05
    wp = T.pack16(w, 1)
    xd = torch.randn(B, H, H, Cin, generator=g).to(DEV).to(BF)
    yd = torch.empty(B, H, H, Cout, device=DEV, dtype=BF)
    a = nat.ConvArgs()
    a.x, a.w, a.y = xd.data_ptr(), wp.data_ptr(), yd.data_ptr()
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = B, H, H, Cin, Cin
    a.KH, a.KW, a.stride, a.pad = KH, KH, 1, KH // 2
    a.Hout, a.Wout, a.Ntot, a.ldy = H, H, Cout, Cout
    a.splits = 1
    trace = torch.zeros(12 << 16, dtype=torch.int64, device=DEV)
    a.partial = trace.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        nat.check(lib.dfl_conv2d(C.addressof(a), st), 'conv')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    trace.zero_()
    e0.record()
    nat.check(lib.dfl_conv2d(C.addressof(a), st), 'conv')
    e1.record()
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 12)
    t = t[t[:, 7] != 0]
    start, stage, kloop, tk, tend, r0, r1 = (t[:, i] for i in range(7))
    us = 1e-2                                              # s_memrealtime: 100 MHz
    tick = ((r1 - r0).sum() * us) / max((tend - start).sum(), 1)     # us per s_memtime tick, from the two clocks of the same intervals
    tot = (tend - start) * tick
    print('B%d %4d->%4d %3dx%-3d: %4d WGs, kernel %.1f us (event), span %.1f us; tick %.2f ns; per WG us: total %.1f (min %.1f max %.1f) = '
          'staging %.1f + k-loop %.1f + epilogue %.1f (barrier %.2f, LDS image %.2f, rows+stores %.2f); WGs started after the first one finished: %d'
          % (B, Cin, Cout, H, H, len(t), e0.elapsed_time(e1) * 1e3, (r1.max() - r0.min()) * us, tick * 1e3, tot.mean(), tot.min(), tot.max(),
             stage.mean() * tick, kloop.mean() * tick, (tend - tk).mean() * tick,
             t[:, 8].mean() * tick, t[:, 9].mean() * tick, t[:, 10].mean() * tick, int((r0 > r1.min()).sum())))


for (Cin, Cout, H) in ((32, 32, 192), (32, 64, 96), (64, 64, 96), (64, 128, 48), (128, 128, 48), (256, 128, 48), (128, 256, 24), (256, 256, 24),
                       (512, 256, 24), (256, 512, 12), (512, 512, 12), (512, 1024, 6), (1024, 1024, 6), (64, 32, 192), (128, 64, 96)):
    run(16, Cin, Cout, H)
