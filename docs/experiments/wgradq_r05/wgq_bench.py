#!/usr/bin/env python3
"""Per-layer timing (and, with the diagnosis build, phase clocks) of the LDS-DMA 3x3 weight gradient (csrc/wgradq_bf16.hip).
   python docs/experiments/wgq_bench.py                      # the 3x3 layers of the paper network at batch 16
   DFL_LIB_OVERRIDE=docs/experiments/bin/libdfl_wgqtrace.so python docs/experiments/wgq_bench.py   # + phase clocks
Environment: DFL_WGQ=0 (old kernel), DFL_WGQ_PATCH=ipp,ph,pw,nbuf, DFL_WGQ_WGS, AFF=0/1, ONLY=index list."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402,F401
from dfl_amd import _native as nat  # noqa: E402

DEV, BF = 'cuda', torch.bfloat16
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(4), 'mode')
trace = torch.zeros(16 << 15, dtype=torch.int64, device=DEV)
os.environ['DFL_WGQ_TRACE_PTR'] = hex(trace.data_ptr())
AFF = os.environ.get('AFF', '1') != '0'
REPS = int(os.environ.get('REPS', '20'))
flush = torch.empty(64 << 20, dtype=torch.float32, device=DEV)      # 256 MB: evicts L2 + MALL between launches


def run(B, Cg, Cm, H):
    g = torch.Generator().manual_seed(1)
    gd = torch.randn(B, H, H, Cg, generator=g).to(DEV).to(BF)
    dd = torch.randn(B, H, H, Cm, generator=g).to(DEV).to(BF)
    dw = torch.empty(Cm, Cg, 3, 3, device=DEV)
    a = nat.WgradArgs()
    a.g, a.d, a.dw = gd.data_ptr(), dd.data_ptr(), dw.data_ptr()
    a.g_bf16, a.d_bf16 = 1, 1
    a.N, a.Hin, a.Win, a.Cg, a.ldg = B, H, H, Cg, Cg
    a.KH, a.KW, a.stride, a.pad = 3, 3, 1, 1
    a.Hout, a.Wout, a.Cm, a.ldd = H, H, Cm, Cm
    sc, sh = torch.rand(Cg, generator=g).to(DEV) + 0.5, torch.randn(Cg, generator=g).to(DEV)
    if AFF:
        a.in_scale, a.in_shift = sc.data_ptr(), sh.data_ptr()
    a.splits = 1
    s = nat.check(lib.dfl_wgrad_suggest_splits(C.addressof(a)), 'suggest')
    a.splits = s
    bias = torch.empty(s * Cm, device=DEV)
    a.bias_partial = bias.data_ptr()
    part = torch.empty(max(s, 2) * Cm * Cg * 9, device=DEV)
    a.partial = part.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), st), 'wgrad')
    torch.cuda.synchronize()
    # warm: launches back to back; cold: caches flushed before each launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), st), 'wgrad')
    e1.record()
    torch.cuda.synchronize()
    warm = e0.elapsed_time(e1) * 1e3 / REPS
    cold = 0.0
    for _ in range(5):
        flush.zero_()
        e0.record()
        nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), st), 'wgrad')
        e1.record()
        torch.cuda.synchronize()
        cold += e0.elapsed_time(e1) * 1e3 / 5
    trace.zero_()
    flush.zero_()
    nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), st), 'wgrad')
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 12] != 0]
    gf = 2.0 * B * H * H * Cg * Cm * 9
    byts = 2.0 * B * H * H * (Cg + Cm)
    msg = 'B%d %3dx%-3d Cg%-4d Cm%-4d slices %3d: warm %5.1f us (%4.0f TF, %.2f TB/s compulsory)  cold %5.1f us' % (
        B, H, H, Cg, Cm, s, warm, gf / warm / 1e6, byts / warm / 1e6, cold)
    if len(t):
        tick = ((t[:, 11] - t[:, 10]).sum() * 1e-2) / max((t[:, 9] - t[:, 0]).sum(), 1)      # us per shader clock
        f = lambda v: v.mean() * tick
        allrec = trace.cpu().numpy().reshape(-1, 16)
        c = allrec[0::2]
        c = c[c[:, 12] != 0]                # matrix wave 0 of every workgroup
        l = allrec[1::2]
        l = l[l[:, 12] != 0]                # loader wave 0
        msg += ' | %d WGs, us: total %.1f (max %.1f); matrix wave: setup %.1f + [wait for patches %.1f + k-steps %.1f] + rest of loop %.1f + sums/reduce %.1f + output %.1f' % (
            len(c), f(c[:, 9] - c[:, 0]), ((c[:, 9] - c[:, 0]) * tick).max(), f(c[:, 1] - c[:, 0]), f(c[:, 2]), f(c[:, 6]),
            f(c[:, 7] - c[:, 1] - c[:, 2] - c[:, 6]), f(c[:, 8] - c[:, 7]), f(c[:, 9] - c[:, 8]))
        if len(l):
            msg += '; loader wave: [wait for free pair %.1f + issue %.1f + vmcnt %.1f] of %.1f; first start -> last end %.1f' % (
                f(l[:, 2]), f(l[:, 4]), f(l[:, 3]), f(l[:, 7] - l[:, 1]), (t[:, 11].max() - t[:, 10].min()) * 1e-2)
    print(msg, flush=True)


LAYERS = [(32, 32, 192), (64, 32, 192), (32, 64, 96), (64, 64, 96), (128, 64, 96), (64, 128, 48), (128, 128, 48), (256, 128, 48), (128, 256, 24),
          (256, 256, 24), (512, 256, 24), (256, 512, 12), (512, 512, 12), (1024, 512, 12), (512, 1024, 6), (1024, 1024, 6)]
only = os.environ.get('ONLY')
# EXPS="A=1,B=2|A=3": the layer list once per setting, all in this process (a fresh box pays ~20 s per Python start)
for exp in os.environ.get('EXPS', '').split('|'):
    sets = dict(kv.split('=', 1) for kv in exp.split(';') if '=' in kv)
    for k, v in sets.items():
        os.environ[k] = v
    if sets:
        print('== ' + exp, flush=True)
    for i, (Cg, Cm, H) in enumerate(LAYERS):
        if only is None or str(i) in only.split(','):
            run(16, Cg, Cm, H)
    for k in sets:
        del os.environ[k]
