#!/usr/bin/env python3
"""Does running a layer's weight-gradient and data-gradient kernels on two streams beat back-to-back launches?"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import dfl_amd  # noqa
from dfl_amd import _native as nat


def mk(N, H, W, Cin, Cout, K):
    lib = nat.lib()
    dev = 'cuda'
    pad = 1
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(K * K * Cin * Cout, device=dev)
    y = torch.empty(N, H, W, Cout, device=dev)
    a = nat.ConvArgs(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), N=N, Hin=H, Win=W, Cin=Cin, ldx=Cin, KH=K, KW=K, stride=1,
                     pad=pad, Hout=H, Wout=W, Ntot=Cout, ldy=Cout, relu=0)
    sp = lib.dfl_conv_suggest_splits(C.addressof(a))
    keep = [x, w, y]
    if sp > 1:
        part = torch.empty(sp * N * H * W * Cout, device=dev)
        a.splits, a.partial = sp, part.data_ptr()
        keep.append(part)
    d = torch.randn(N, H, W, Cout, device=dev)
    dw = torch.empty(Cout, Cin, K, K, device=dev)
    b = nat.WgradArgs(g=x.data_ptr(), d=d.data_ptr(), dw=dw.data_ptr(), N=N, Hin=H, Win=W, Cg=Cin, ldg=Cin, KH=K, KW=K,
                      stride=1, pad=pad, Hout=H, Wout=W, Cm=Cout, ldd=Cout, splits=1)
    s = lib.dfl_wgrad_suggest_splits(C.addressof(b))
    b.splits = s
    if s > 1:
        p2 = torch.empty(s * Cout * Cin * K * K, device=dev)
        b.partial = p2.data_ptr()
        keep.append(p2)
    keep += [d, dw]
    return a, b, keep


def main():
    lib = nat.lib()
    shapes = [(16, 48, 48, 128, 128, 3), (16, 24, 24, 256, 256, 3), (16, 96, 96, 64, 64, 3), (16, 12, 12, 512, 512, 3), (16, 192, 192, 32, 32, 3)]
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    for sh in shapes:
        a, b, keep = mk(*sh)
        reps = 30

        def run(two):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s0)
            for _ in range(reps):
                lib.dfl_conv2d(C.addressof(a), s0.cuda_stream)
                if two:
                    ev = torch.cuda.Event()
                    ev.record(s0)
                    s1.wait_event(ev)
                    lib.dfl_conv2d_wgrad(C.addressof(b), s1.cuda_stream)
                else:
                    lib.dfl_conv2d_wgrad(C.addressof(b), s0.cuda_stream)
            if two:
                ev = torch.cuda.Event()
                ev.record(s1)
                s0.wait_event(ev)
            e1.record(s0)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        run(False); run(True)
        print(sh, 'serial %.4f ms   two streams %.4f ms' % (run(False), run(True)))


if __name__ == '__main__':
    main()
