#!/usr/bin/env python3
"""Single-layer micro-benchmark of dfl_conv2d / dfl_conv2d_wgrad (tuning aid; prints TFLOP/s per shape).

    python tools/kbench.py conv  N H W Cin Cout K [stride] [reps]
    python tools/kbench.py wgrad N H W Cin Cout K [stride] [reps]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402


def main():
    kind = sys.argv[1]
    N, H, W, Cin, Cout, K = [int(v) for v in sys.argv[2:8]]
    stride = int(sys.argv[8]) if len(sys.argv) > 8 else 1
    reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
    flags = sys.argv[10] if len(sys.argv) > 10 else 'sabr'      # s: statistics, a: input affine, b: bias, r: relu, W / X: split weights / input
    lib = nat.lib()
    dev = 'cuda'
    pad = 1 if K == 3 else 0
    Ho = (H + 2 * pad - K) // stride + 1
    Wo = (W + 2 * pad - K) // stride + 1
    x = torch.randn(N, H, W, Cin, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    if kind == 'conv':
        w = torch.randn(K * K * Cin, Cout, device=dev) / (K * K * Cin) ** 0.5
        y = torch.empty(N, Ho, Wo, Cout, device=dev)
        sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
        b = torch.randn(Cout, device=dev)
        a = nat.ConvArgs(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=b.data_ptr(), in_scale=sc.data_ptr(),
                         in_shift=sh.data_ptr(), N=N, Hin=H, Win=W, Cin=Cin, ldx=Cin, KH=K, KW=K, stride=stride, pad=pad,
                         Hout=Ho, Wout=Wo, Ntot=Cout, ldy=Cout, relu=1)
        sp = lib.dfl_conv_suggest_splits(C.addressof(a))
        if sp > 1:
            part = torch.empty(sp * N * Ho * Wo * Cout, device=dev)
            a.splits, a.partial = sp, part.data_ptr()
        if 'a' not in flags:
            a.in_scale = a.in_shift = None
        if 'b' not in flags:
            a.bias = None
        a.relu = int('r' in flags)
        a.w_split = int('W' in flags)         # operands declared pre-split (timing only: the bits are whatever randn left)
        a.x_split = int('X' in flags)
        if 's' in flags:
            gm = lib.dfl_conv_grid_m(C.addressof(a))
            stats = torch.empty(gm * 2 * Cout, device=dev)
            a.stat_partials = stats.data_ptr()
        fn, desc = lib.dfl_conv2d, 'cfg %d splits %d flags %s' % (lib.dfl_conv_config(C.addressof(a)), sp, flags)
        flops = 2.0 * N * Ho * Wo * K * K * Cin * Cout
    else:
        d = torch.randn(N, Ho, Wo, Cout, device=dev)
        dw = torch.empty(Cout, Cin, K, K, device=dev)
        a = nat.WgradArgs(g=x.data_ptr(), d=d.data_ptr(), dw=dw.data_ptr(), N=N, Hin=H, Win=W, Cg=Cin, ldg=Cin, KH=K, KW=K,
                          stride=stride, pad=pad, Hout=Ho, Wout=Wo, Cm=Cout, ldd=Cout, splits=1)
        sp = lib.dfl_wgrad_suggest_splits(C.addressof(a))
        a.splits = sp
        if sp > 1:
            part = torch.empty(sp * Cout * Cin * K * K, device=dev)
            a.partial = part.data_ptr()
        fn, desc = lib.dfl_conv2d_wgrad, 'cfg %d splits %d' % (lib.dfl_wgrad_config(C.addressof(a)), sp)
        flops = 2.0 * N * Ho * Wo * K * K * Cin * Cout
    for _ in range(3):
        nat.check(fn(C.addressof(a), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(C.addressof(a), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('%s N%d %dx%d Cin%d Cout%d k%d s%d  %s: %.4f ms  %.1f TFLOP/s' % (kind, N, H, W, Cin, Cout, K, stride, desc, ms,
                                                                           flops / ms / 1e9))


if __name__ == '__main__':
    main()
