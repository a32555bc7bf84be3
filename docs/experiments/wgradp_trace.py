#!/usr/bin/env python3
"""Phase clocks of the patch-resident weight-gradient kernel (diagnosis build: docs/experiments/build_trace.sh).
DFL_LIB_OVERRIDE=docs/experiments/bin/libdfl_trace.so python docs/experiments/wgradp_trace.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402,F401
from dfl_amd import _native as nat  # noqa: E402

DEV = 'cuda'
BF = torch.bfloat16
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(4), 'mode')
trace = torch.zeros(12 << 16, dtype=torch.int64, device=DEV)
os.environ['DFL_WGP_TRACE_PTR'] = hex(trace.data_ptr())


def run(B, Cg, Cm, H, K, stride=1, fused=False):
    Ho = H if stride == 1 else H // 2
    pad = K // 2 if stride == 1 else 0
    g = torch.Generator().manual_seed(1)
    gd = torch.randn(B, H, H, Cg, generator=g).to(DEV).to(BF)
    dd = torch.randn(B, Ho, Ho, Cm, generator=g).to(DEV).to(BF)
    dw = torch.empty(Cm, Cg, K, K, device=DEV)
    a = nat.WgradArgs()
    a.g, a.d, a.dw = gd.data_ptr(), dd.data_ptr(), dw.data_ptr()
    a.g_bf16, a.d_bf16 = 1, 1
    a.N, a.Hin, a.Win, a.Cg, a.ldg = B, H, H, Cg, Cg
    a.KH, a.KW, a.stride, a.pad = K, K, stride, pad
    a.Hout, a.Wout, a.Cm, a.ldd = Ho, Ho, Cm, Cm
    keep = []
    if fused:            # the operand of the training step: BatchNorm + ReLU backward of (dy, r) formed in the staging path, affine on g
        r = torch.randn(B, Ho, Ho, Cm, generator=g).to(DEV).to(BF)
        coef = torch.randn(3 * Cm, generator=g).to(DEV)
        bias = torch.empty(4096 * Cm, device=DEV)
        sc, sh = torch.rand(Cg, generator=g).to(DEV) + 0.5, torch.randn(Cg, generator=g).to(DEV)
        a.d2, a.ldd2, a.d_mode, a.coef, a.bias_partial = r.data_ptr(), Cm, 1, coef.data_ptr(), bias.data_ptr()
        a.in_scale, a.in_shift = sc.data_ptr(), sh.data_ptr()
        keep += [r, coef, bias, sc, sh]
    a.splits = 1
    s = nat.check(lib.dfl_wgrad_suggest_splits(C.addressof(a)), 'suggest')
    a.splits = s
    part = torch.empty(max(s, 2) * Cm * Cg * K * K, device=DEV)
    a.partial = part.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)
    for _ in range(3):
        nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), st), 'wgrad')
    torch.cuda.synchronize()
    os.dup2(saved, 2)
    trace.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), st), 'wgrad')
    e1.record()
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 12)
    t = t[t[:, 9] != 0]
    us = 1e-2
    tick = ((t[:, 8] - t[:, 7]).sum() * us) / max((t[:, 6] - t[:, 0]).sum(), 1)
    tot = (t[:, 6] - t[:, 0]) * tick
    byts = 2.0 * B * (H * H * Cg + Ho * Ho * Cm)
    print(('fused ' if fused else 'plain ') + 'B%d g %dx%dx%d d %dx%dx%d k%d: %d WGs, kernel %.1f us (%.2f TB/s compulsory); per WG us: total %.1f (max %.1f) = barrier %.1f + LDS commit (incl. load wait) %.1f '
          '+ issue %.1f + k-steps %.1f + tail %.1f; before the first patch %.1f, output %.1f, first start -> last end %.1f us'
          % (B, H, H, Cg, Ho, Ho, Cm, K, len(t), e0.elapsed_time(e1) * 1e3, byts / (e0.elapsed_time(e1) * 1e-3) / 1e12, tot.mean(), tot.max(),
             t[:, 1].mean() * tick, t[:, 2].mean() * tick, t[:, 3].mean() * tick, t[:, 4].mean() * tick, (t[:, 6] - t[:, 5]).mean() * tick,
             (t[:, 0] - t[:, 10]).mean() * tick, (t[:, 11] - t[:, 6]).mean() * tick, (t[:, 8].max() - t[:, 7].min()) * us), flush=True)


for (Cg, Cm, H, K, s) in ((32, 32, 192, 3, 1), (64, 32, 192, 3, 1), (64, 32, 192, 1, 1), (64, 64, 96, 3, 1), (128, 64, 96, 3, 1), (128, 64, 96, 1, 1), (128, 128, 48, 3, 1),
                          (256, 128, 48, 3, 1), (256, 256, 24, 3, 1), (512, 256, 24, 3, 1), (512, 512, 12, 3, 1), (1024, 1024, 6, 3, 1), (32, 64, 192, 2, 2),
                          (256, 512, 24, 2, 2)):
    run(16, Cg, Cm, H, K, s)
    if K == 3:
        run(16, Cg, Cm, H, K, s, fused=True)
