#!/usr/bin/env python3
"""Phase clocks of the unrolled 3x3 convolution (csrc/convq_bf16.hip built with -DDFL_CONVQ_TRACE -> docs/experiments/bin/libdfl_qtrace.so):
per workgroup the time until the first image's loads and the tables are out, until the image is staged, in the k loop, until the
accumulators are in LDS, in the rows.   DFL_LIB_OVERRIDE=docs/experiments/bin/libdfl_qtrace.so python docs/experiments/convq_trace.py"""
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


def run(B, Cin, Cout, H, tile, aff):
    import test_gpu_bf16 as T
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    wp = T.pack16(w, 1)
    xd = torch.randn(B, H, H, Cin, generator=g).to(DEV).to(BF)
    yd = torch.empty(B, H, H, Cout, device=DEV, dtype=BF)
    bias = torch.randn(Cout, device=DEV)
    sc, sh = torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV)
    a = nat.ConvArgs()
    a.x, a.w, a.y = xd.data_ptr(), wp.data_ptr(), yd.data_ptr()
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = B, H, H, Cin, Cin
    a.KH, a.KW, a.stride, a.pad = 3, 3, 1, 1
    a.Hout, a.Wout, a.Ntot, a.ldy = H, H, Cout, Cout
    a.bias, a.relu = bias.data_ptr(), 1
    if aff:
        a.in_scale, a.in_shift = sc.data_ptr(), sh.data_ptr()
    a.splits = 1
    trace = torch.zeros(24 << 14, dtype=torch.int64, device=DEV)
    a.partial = trace.data_ptr()
    gv = (C.c_int32 * 5)(tile, 1, 16 if tile == 42 else 8, 12, 1)
    nat.check(lib.dfl_conv_force_geometry(C.addressof(gv)), 'force')
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
    lib.dfl_conv_force_geometry(None)
    tt = trace.cpu().numpy().reshape(-1, 2, 12)
    t1 = tt[:, 1][tt[:, 0, 7] != 0]
    t = tt[:, 0][tt[:, 0, 7] != 0]
    r0, r1 = t[:, 6], t[:, 7]
    us = 1e-2                                              # s_memrealtime: 100 MHz
    tick = ((r1 - r0).sum() * us) / max((t[:, 5] - t[:, 0]).sum(), 1)
    d = [(t[:, i + 1] - t[:, i]).mean() * tick for i in range(5)]
    print('tile %d aff%d B%d %4d->%4d %3dx%-3d: %4d WGs, kernel %.1f us (event), span %.1f us; tick %.2f ns; per WG us: total %.1f = loads+tables %.2f + '
          'staging %.2f + k-loop %.2f + acc->LDS,constants %.2f + rows %.2f; first start -> last start %.2f us'
          % (tile, aff, B, Cin, Cout, H, H, len(t), e0.elapsed_time(e1) * 1e3, (r1.max() - r0.min()) * us, tick * 1e3, sum(d), d[0], d[1], d[2], d[3], d[4],
             (r0.max() - r0.min()) * us))
    ph = lambda q: ' '.join('%.2f' % ((q[:, j] - q[:, i]).mean() * tick) for i, j in ((3, 8), (8, 9), (9, 10), (10, 4)))
    print('      wave 0: k-loop end -> barrier, acc -> LDS, constants, barrier: %s' % ph(t))
    if t1[:, 7].any():
        print('      wave 4: k-loop %.2f us (wave 0 %.2f), its k-loop ends %.2f us after wave 0s; %s' % (
            (t1[:, 3] - t1[:, 2]).mean() * tick, (t[:, 3] - t[:, 2]).mean() * tick, (t1[:, 3] - t[:, 3]).mean() * tick, ph(t1)))


for (Cin, Cout, H) in ((128, 128, 48), (256, 256, 24), (512, 512, 12), (256, 128, 48), (64, 128, 48)):
    for tile in (40, 41, 42):
        for aff in (1,):
            run(16, Cin, Cout, H, tile, aff)
