#!/usr/bin/env python3
"""Per-layer A/B of the unrolled 3x3 form (csrc/convq_bf16.hip, tile configurations 40 / 41) against what the library picks
(tuning table / cost model) on the convolution ops of a recorded training plan: same tensors, same epilogues, launches back
to back, best of several rounds.  Only numbers of one process compare.

    python tools/convq_probe.py [--batch 16] [--size 192] [--eval]
"""
import argparse
import ctypes as C
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DFL_PLAN_LATENCY_FORM'] = '0'
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=192)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--eval', action='store_true')
    ap.add_argument('--narrow', action='store_true', help='only the layers with fewer than 128 columns')
    ap.add_argument('--convn', action='store_true', help='time the narrow 3x3 form (tiles 58 ... 63, csrc/convn_bf16.hip) instead of the unrolled one')
    args = ap.parse_args()
    lib = nat.lib()
    nat.check(lib.dfl_set_math_mode(4), 'mode')
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    net = dfl_amd.UNet(**bench.PAPER).to(dev)
    x = torch.randn(args.batch, 1, args.size, args.size, device=dev)
    if args.eval:
        net.eval()
        with torch.no_grad():
            net(x)
        plan = [p for ps in net._plans.values() for p in ps][0]
        structs = list(plan.fwd.structs)
    else:
        net.train()
        seg, heat = net(x)
        (seg.float().mean() + heat.float().mean()).backward()
        plan = [p for ps in net._plans.values() for p in ps if p.need_grad][0]
        structs = list(plan.fwd.structs) + list(plan.bwd.structs)
    torch.cuda.synchronize()
    ops = []
    for st in structs:
        if isinstance(st, nat.ConvArgs) and st.x_bf16 and st.KH == 3 and st.stride == 1 and st.Cin % 32 == 0 and not st.scatter2x2 and (not args.narrow or st.Ntot < 128):
            ops.append(st)
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(1 << 27, device=dev)
    stat_scratch = torch.empty(1 << 24, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def time_geom(st, geom):
        a = nat.ConvArgs.from_buffer_copy(bytes(st))
        if geom is not None:
            g = (C.c_int32 * 5)(*geom)
            rc = lib.dfl_conv_force_geometry(C.addressof(g))
            a.splits = geom[4]
        try:
            if geom is None:
                a.splits = 0
                a.splits = nat.check(lib.dfl_conv_suggest_splits(C.addressof(a)), 'suggest')
            M = a.N * a.Hout * a.Wout
            if a.splits > 1:
                if a.splits * M * a.Ntot > scratch.numel():
                    return None
                a.partial = scratch.data_ptr()
            if lib.dfl_conv_config(C.addressof(a)) < 0:
                return None
            if a.stat_partials:
                gm = nat.check(lib.dfl_conv_grid_m(C.addressof(a)), 'grid_m')
                if gm * 2 * a.Ntot > stat_scratch.numel():
                    return None
                a.stat_partials = stat_scratch.data_ptr()
            best = None
            for r in range(args.rounds + 1):
                e0.record()
                for _ in range(args.reps if r else 2):
                    rc = lib.dfl_conv2d(C.addressof(a), stream)
                    if rc != 0:
                        return None
                e1.record()
                torch.cuda.synchronize()
                if r:
                    t = e0.elapsed_time(e1) * 1e3 / args.reps
                    best = t if best is None else min(best, t)
            return best
        finally:
            lib.dfl_conv_force_geometry(None)

    tot = {'lib': 0.0, 'best': 0.0}
    for st in ops:
        aff = 2 if st.x_mode else (1 if (st.in_scale or st.in_tot) else 0)
        name = 'N%d %dx%d %d->%d aff%d%s%s' % (st.N, st.Hin, st.Win, st.Cin, st.Ntot, aff, ' add' if st.add else '', ' xo' if st.x_out else '')
        t_lib = time_geom(st, None)
        row = []
        best = t_lib
        lay = ((40, 1), (41, 1), (42, 2), (43, 2), (44, 4), (45, 2), (46, 4), (47, 8), (48, 4))
        if args.convn:
            for tile, (ph, pw) in {58: (8, 64), 59: (12, 64), 60: (16, 32), 61: (24, 32), 62: (6, 64), 63: (12, 32), 64: (6, 64), 65: (12, 32)}.items():
                t = time_geom(st, (tile, 1, ph, pw, 1))
                if t is not None:
                    row.append('%d %.1f' % (tile, t))
                    best = min(best, t)
        for tile, wm in (() if args.convn else lay + tuple((t + 9, w) for t, w in lay)):
            for sp in (1, 2, 4, 8):
                t = time_geom(st, (tile, 1, 8 * wm, 12, sp))
                if t is not None:
                    row.append('%d/s%d %.1f' % (tile, sp, t))
                    best = min(best, t)
        tot['lib'] += t_lib
        tot['best'] += best
        print('%-36s lib %6.1f us | %s' % (name, t_lib, '  '.join(row)), flush=True)
    print('sum: library %.1f us, best with the unrolled form %.1f us' % (tot['lib'], tot['best']))


if __name__ == '__main__':
    main()
