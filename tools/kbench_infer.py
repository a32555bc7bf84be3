#!/usr/bin/env python3
"""Per-layer times of ONE eval-mode forward of the paper network (op by op with hipEvents): python tools/kbench_infer.py [size] [batch] [mode]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else 'bf16s'
lib = nat.lib()
nat.check(lib.dfl_set_math_mode(bench.MATH[mode][0]), 'mode')
dev = torch.device('cuda:0')
torch.manual_seed(7)
net = dfl_amd.UNet(**bench.PAPER).to(dev).eval()
x = torch.randn(B, 1, size, size, device=dev)
with torch.no_grad():
    for _ in range(3):
        out = net(x)
torch.cuda.synchronize()
plan = [p for ps in net._plans.values() for p in ps if not p.need_grad][0]
stream = torch.cuda.current_stream().cuda_stream
seg, heat = plan.new_outputs()
plan.bind_outputs(seg, heat)
best = None
for rep in range(5):
    ms = plan.fwd.run_timed(stream)
    best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
plan.busy = False
tot = 0.0
for st, t in zip(plan.fwd.structs, best):
    tot += t
    if isinstance(st, nat.ConvArgs):
        cfg = lib.dfl_conv_config(__import__('ctypes').addressof(st))
        name = bench.CONV_KERNELS[cfg] if cfg < 16 else 'convp<%s>' % bench.CONVP_TILES[cfg - 16]
        M = st.N * (st.Hin * st.Win if st.scatter2x2 else st.Hout * st.Wout)
        fl = 2.0 * M * st.KH * st.KW * st.Cin * st.Ntot
        by = 2.0 * (st.N * st.Hin * st.Win * st.Cin + st.KH * st.KW * st.Cin * st.Ntot + M * st.Ntot)
        print('%-22s %8.3f ms %7.1f TF %7.0f GB/s  N%d %dx%d Cin%d -> %dx%d Ntot%d k%d s%d splits%d' % (
            name, t, fl / t / 1e9, by / t / 1e6, st.N, st.Hin, st.Win, st.Cin, st.Hout, st.Wout, st.Ntot, st.KH, st.stride, st.splits))
    else:
        print('%-22s %8.3f ms' % (type(st).__name__, t))
print('total %.3f ms' % tot)
