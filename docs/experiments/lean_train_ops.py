#!/usr/bin/env python3
"""Would the latency form serve the SMALL ops of a batch-16 training step (1x1 / 2x2 convolutions of the deep levels: 13 - 25 us each in the
patch kernels for 0.1 - 2.4 GFLOP)?  Times dfl_conv2d with and without the hint on those shapes (bf16 tensors), launches back to back.
python docs/experiments/lean_train_ops.py"""
import ctypes as C
import os
import sys

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
import test_gpu_bf16 as T  # noqa: E402


def run(N, Cin, Cout, H, K, stride, scatter, hint):
    g = torch.Generator().manual_seed(1)
    if scatter:
        w = torch.randn(Cin, Cout, 2, 2, generator=g) * 0.05
        wp = T.pack16(w, 3)
        Ntot, Ho = 4 * Cout, 2 * H
        KH = 1
    else:
        w = torch.randn(Cout, Cin, K, K, generator=g) * 0.05
        wp = T.pack16(w, 1)
        Ntot, Ho = Cout, (H - K) // stride + 1
        KH = K
    xd = torch.randn(N, H, H, Cin, generator=g).to(DEV).to(BF)
    yd = torch.empty(N, Ho, Ho, Cout, device=DEV, dtype=BF)
    a = nat.ConvArgs()
    a.x, a.w, a.y = xd.data_ptr(), wp.data_ptr(), yd.data_ptr()
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = N, H, H, Cin, Cin
    a.KH, a.KW, a.stride, a.pad = KH, KH, stride, 0
    a.Hout, a.Wout, a.Ntot, a.ldy = Ho, Ho, Ntot, Cout
    a.scatter2x2 = scatter
    a.latency_form = hint
    sp = nat.check(lib.dfl_conv_suggest_splits(C.addressof(a)), 'suggest')
    if sp > 1:
        M = N * (H * H if scatter else Ho * Ho)
        part = torch.empty(sp * M * Ntot, device=DEV)
        a.splits, a.partial = sp, part.data_ptr()
    cfg = lib.dfl_conv_config(C.addressof(a))
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        nat.check(lib.dfl_conv2d(C.addressof(a), st), 'conv')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        nat.check(lib.dfl_conv2d(C.addressof(a), st), 'conv')
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3, cfg, sp


# batch 16: (Cin, Cout, H, K, stride, scatter)
for (Cin, Cout, H, K, stride, sc, what) in ((64, 128, 48, 1, 1, 0, 'res 1x1 L2'), (128, 256, 24, 1, 1, 0, 'res 1x1 L3'), (256, 512, 12, 1, 1, 0, 'res 1x1 L4'),
                                            (512, 1024, 6, 1, 1, 0, 'res 1x1 L5'), (1024, 512, 12, 1, 1, 0, 'res 1x1 D4'), (512, 256, 24, 1, 1, 0, 'res 1x1 D3'),
                                            (128, 128, 48, 2, 2, 0, 'down L2'), (256, 256, 24, 2, 2, 0, 'down L3'), (512, 512, 12, 2, 2, 0, 'down L4'),
                                            (1024, 512, 6, 1, 1, 1, 'up 5->4'), (512, 256, 12, 1, 1, 1, 'up 4->3'), (256, 128, 24, 1, 1, 1, 'up 3->2')):
    t0, c0, s0 = run(16, Cin, Cout, H, K, stride, sc, 0)
    t1, c1, s1 = run(16, Cin, Cout, H, K, stride, sc, 1)
    print('%-12s Cin%-4d Cout%-4d %2dx%-2d: patch kernels %5.1f us (cfg %d, %d slices)   hint %5.1f us (cfg %d, %d slices)' % (what, Cin, Cout, H, H, t0, c0, s0, t1, c1, s1))
