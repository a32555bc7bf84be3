#!/usr/bin/env python3
"""Measure the geometry candidates of the bf16 convolution (csrc/convp_bf16.hip) for every layer of a network on THIS
GPU and write the table the library loads at start-up (dfl_amd/tune/gfx950_convp.txt; include/dfl_hip.h:
dfl_conv_tune_add).  The cost model stays the fallback for layers that are not listed.

    python tools/tune_convp.py [--batch 16] [--size 192] [--out PATH] [--append]

Every convolution op of the recorded forward and backward programs (same tensors, same epilogues) is timed under each
candidate with hipEvents, `--reps` launches back to back, best of `--rounds` rounds; ops that share a table key (same
shape, different epilogue) are summed.  Only numbers of one process compare (boxes differ in clock)."""
import argparse
import ctypes as C
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DFL_TUNE'] = '0'                      # measure against the cost model, not an older table
os.environ['DFL_PLAN_LATENCY_FORM'] = '0'          # the table is about the patch-resident kernels: inference plans without the latency form
import dfl_amd  # noqa: E402
from dfl_amd import _native as nat  # noqa: E402
import bench  # noqa: E402


def key_of(a):
    return (a.N, a.Hin, a.Win, a.Cin, a.Ntot, a.KH, a.KW, a.stride, a.pad, 1 if a.scatter2x2 else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=192)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--eval', action='store_true', help='tune the inference program (eval-mode forward) instead of the training step')
    ap.add_argument('--out', default=os.path.join(ROOT, 'deepfluorolabeling-ipcai2020_amd', 'tune', 'gfx950_convp.txt'))
    ap.add_argument('--append', action='store_true', help='keep the entries already in --out (other shapes)')
    ap.add_argument('--narrow', action='store_true', help='only the 3x3 layers with 32 / 64 output columns (the shapes csrc/convn_bf16.hip takes); use with --append')
    args = ap.parse_args()
    lib = nat.lib()
    nat.check(lib.dfl_set_math_mode(4), 'mode')
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    net = dfl_amd.UNet(**bench.PAPER).to(dev)
    x = torch.randn(args.batch, 1, args.size, args.size, device=dev)
    if args.eval:                                               # the inference program only (its operand forms differ from the training step's)
        net.eval()
        with torch.no_grad():
            net(x)
        torch.cuda.synchronize()
        plan = [p for ps in net._plans.values() for p in ps][0]
        structs = list(plan.fwd.structs)
    else:
        net.train()
        seg, heat = net(x)
        (seg.float().mean() + heat.float().mean()).backward()
        torch.cuda.synchronize()
        plan = [p for ps in net._plans.values() for p in ps if p.need_grad][0]
        structs = list(plan.fwd.structs) + list(plan.bwd.structs)
    ops = OrderedDict()
    for st in structs:
        if isinstance(st, nat.ConvArgs) and st.x_bf16 and lib.dfl_conv_config(C.addressof(st)) >= 16:
            if args.narrow and not (st.KH == 3 and st.stride == 1 and st.Ntot in (32, 64) and st.Cin % 32 == 0 and not st.scatter2x2):
                continue
            ops.setdefault(key_of(st), []).append(st)
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(1 << 28, device=dev)                    # 1 GiB: K-slice partial sums
    stat_scratch = torch.empty(1 << 24, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def time_geom(group, geom):
        """Sum over the ops of `group` of the mean launch time (us) under geometry `geom` (None: the cost model's)."""
        total = 0.0
        for st in group:
            a = nat.ConvArgs.from_buffer_copy(bytes(st))
            if geom is not None:
                g = (C.c_int32 * 5)(*geom)
                nat.check(lib.dfl_conv_force_geometry(C.addressof(g)), 'force')
                a.splits = geom[4]
            try:
                if lib.dfl_conv_config(C.addressof(a)) < 0:      # a candidate of the group's first op may not exist for another operand form:
                    if geom is None:                             # the library then plans that op by the cost model, and so does this measurement
                        return None
                    lib.dfl_conv_force_geometry(None)
                    a.splits = 0
                    a.splits = nat.check(lib.dfl_conv_suggest_splits(C.addressof(a)), 'suggest')
                    if lib.dfl_conv_config(C.addressof(a)) < 0:
                        return None
                M = a.N * (a.Hin * a.Win if a.scatter2x2 else a.Hout * a.Wout)
                if a.splits > 1:
                    if a.splits * M * a.Ntot > scratch.numel():
                        return None
                    a.partial = scratch.data_ptr()
                if a.stat_partials:
                    gm = nat.check(lib.dfl_conv_grid_m(C.addressof(a)), 'grid_m')
                    if gm * 2 * a.Ntot > stat_scratch.numel():
                        return None
                    a.stat_partials = stat_scratch.data_ptr()
                best = None
                for r in range(args.rounds + 1):
                    e0.record()
                    for _ in range(args.reps if r else 2):
                        nat.check(lib.dfl_conv2d(C.addressof(a), stream), 'conv')
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        t = e0.elapsed_time(e1) * 1e3 / args.reps
                        best = t if best is None else min(best, t)
                total += best
            finally:
                lib.dfl_conv_force_geometry(None)
        return total

    table = OrderedDict()
    if args.append and os.path.exists(args.out):
        for line in open(args.out):
            v = line.split('#')[0].split()
            if len(v) == 15:
                table[tuple(int(t) for t in v[:10])] = (tuple(int(t) for t in v[10:]), line.split('#', 1)[1].strip() if '#' in line else '')
    sum_model = sum_best = 0.0
    cand = (C.c_int32 * (5 * 4096))()
    for key, group in ops.items():
        n = nat.check(lib.dfl_conv_candidates(C.addressof(group[0]), C.addressof(cand), 4096), 'candidates')
        n = min(n, 4096)
        t_model = time_geom(group, None)
        best_t, best_g = t_model, None
        for i in range(n):
            g = tuple(cand[5 * i + j] for j in range(5))
            t = time_geom(group, g)
            if t is not None and t < best_t:
                best_t, best_g = t, g
        if best_g is not None:
            t2 = time_geom(group, best_g)                        # confirm: a candidate must win twice
            t_model2 = time_geom(group, None)
            if t2 is None or t2 >= t_model2 * 0.97:
                best_g, best_t = None, min(t_model, t_model2)
            else:
                best_t, t_model = t2, t_model2
        sum_model += t_model
        sum_best += best_t
        note = '%d ops, %d candidates: model %.1f us -> %.1f us' % (len(group), n, t_model, best_t)
        print('N%d %dx%d Cin%d Ntot%d k%d s%d scat%d: %s  %s' % (key[0], key[1], key[2], key[3], key[4], key[5], key[7], key[9], note,
                                                                 'tile %d patch %dx%dx%d slices %d' % best_g if best_g else 'model kept'), flush=True)
        if best_g is not None:
            table[key] = (best_g, note)
        else:
            table.pop(key, None)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        f.write('# bf16 convolution geometries measured by tools/tune_convp.py on %s (batch %d, %dx%d)\n' % (torch.cuda.get_device_name(0), args.batch, args.size, args.size))
        f.write('# N Hin Win Cin Ntot KH KW stride pad scatter   tile images_per_patch patch_h patch_w k_slices\n')
        for key, (g, note) in table.items():
            f.write(' '.join(str(v) for v in key) + '   ' + ' '.join(str(v) for v in g) + '   # ' + note + '\n')
    print('convolution time per step: cost model %.3f ms -> tuned %.3f ms (%d layers listed)' % (sum_model / 1e3, sum_best / 1e3, len(table)))


if __name__ == '__main__':
    main()
