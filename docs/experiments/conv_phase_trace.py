#!/usr/bin/env python3
"""Reads the per-workgroup phase times the -DDFL_CONV_TRACE build of conv_gemm_kernel leaves (see conv_phase_trace.sh)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from dfl_amd import _native as nat  # noqa: E402

nat.LIB_PATH = os.environ['DFL_LIB_OVERRIDE']        # the instrumented build, not the product library
lib = nat.lib()


def run(N, H, Cin, Cout, mode):
    lib.dfl_set_math_mode(mode)
    dev = 'cuda'
    x = torch.randn(N, H, H, Cin, device=dev)
    w = torch.randn(9 * Cin * Cout, device=dev) / (9 * Cin) ** 0.5
    y = torch.empty(N, H, H, Cout, device=dev)
    a = nat.ConvArgs(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), N=N, Hin=H, Win=H, Cin=Cin, ldx=Cin, KH=3, KW=3, stride=1,
                     pad=1, Hout=H, Wout=H, Ntot=Cout, ldy=Cout, relu=1)
    sp = lib.dfl_conv_suggest_splits(C.addressof(a))
    keep = []
    if sp > 1:
        part = torch.empty(sp * N * H * H * Cout, device=dev)
        a.splits, a.partial = sp, part.data_ptr()
        keep.append(part)
    nblk = ((N * H * H + 63) // 64) * ((Cout + 63) // 64) * max(sp, 1)
    sink = torch.zeros(max(nblk * 6, N * H * H * Cout // 2 + 64), dtype=torch.int64, device=dev)
    a.add, a.ldadd = sink.data_ptr(), Cout
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        nat.check(lib.dfl_conv2d(C.addressof(a), st))
    torch.cuda.synchronize()
    t = sink[:nblk * 6].view(nblk, 6).cpu().double()
    its = t[:, 4].clamp(min=1)
    per = t[:, :3] / its[:, None]
    print('N%d %dx%d %d->%d mode %d cfg %d splits %d: per trip (first half-iteration), shader clocks: compute+loads %.0f | '
          'split+LDS write %.0f | barrier %.0f ; whole kernel %.0f clocks for %d trips (loop = %.0f%%)' % (
              N, H, H, Cin, Cout, mode, lib.dfl_conv_config(C.addressof(a)), sp, per[:, 0].mean(), per[:, 1].mean(),
              per[:, 2].mean(), t[:, 3].mean(), int(its.mean()), 100 * float((2 * t[:, :3].sum(1) / t[:, 3]).mean())))


for mode in (0, 1):
    run(16, 48, 128, 128, mode)
    run(16, 24, 256, 256, mode)
