#!/usr/bin/env python3
"""Benchmark of the U-Net hot path on MI355X: training images/sec (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the paper architecture (depth 6, 32..1024 channels, BatchNorm, zero padding,
strided-conv down-sampling, residual blocks) with the seg + 14-landmark heat-map heads, batch 16 PER GPU of synthetic
1x192x192 images (184x184 reflect-padded size), Dice + NCC loss, SGD(nesterov 0.9, wd 1e-4): one step =
zero_grad -> forward -> crop -> loss -> backward -> optimizer step -> loss.item(), exactly train.py:405-430.
Inputs and targets are resident in HBM before the timed region.  Arithmetic: --math (default bf16s = BASELINE configs[1] as named:
bf16 tensors, fp32 accumulation / statistics / master weights; the parity-holding fp32 and bf16x3 figures ride along in the same
line as `fp32_products` / `bf16x3_products`); no step of the loop is skipped.  Rank 0 prints ONE JSON line, which also carries
`host_enqueue_ms_per_step` (host time to queue one step, GPU idle at its start) and `gpu_idle_frac` (1 - GPU kernel time / step time).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAPER = dict(n_classes=7, depth=6, wf=5, batch_norm=True, padding=True, max_pool=False, num_lands=14, do_res=True,
             block_depth=2)
F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0        # same guide: dense bf16 MFMA (the 5 PF marketing figure is 2:1 sparse)
HBM_PEAK_GBS = 8000.0                 # same guide: HBM3E, 8 TB/s
# product arithmetic of the GEMM kernels (include/dfl_hip.h): name -> (mode, bf16 MFMA products per fp32 product)
MATH = {'fp32': (0, 0), 'bf16x3': (1, 3), 'bf16x6': (2, 6), 'bf16': (3, 1), 'bf16s': (4, 1)}
TRAFFIC_FILES = ['r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json']   # per-kernel HBM bytes from committed rocprofv3 --pmc passes, newest first


KERNEL_SOURCES = {'wgradp_kernel': ['wgradp_bf16.hip'], 'convp_kernel': ['convp_bf16.hip', 'convp.h'], 'convq_kernel': ['convq_bf16.hip', 'convp.h'], 'convn_kernel': ['convn_bf16.hip', 'convp.h'],
                  'convp_finish_kernel': ['convp_bf16.hip', 'convp.h'], 'conv_gemm_kernel': ['conv_gemm.hip', 'conv_epilogue.h'],
                  'conv_rows_kernel': ['conv_rows.hip', 'conv_rows.h', 'conv_epilogue.h'], 'wgrad_kernel': ['wgrad_gemm.hip'],
                  'reduce_batch_kernel': ['bn_elem.hip'], 'sgd_pack_tiles_kernel': ['bn_elem.hip']}


def csrc_sha16(kernel=None):
    """Hash of the kernel sources: a committed traffic figure is quoted as current only when the kernel it was measured on is the
    kernel of this tree (the line must not quote stale bytes silently).  Per kernel: the source file(s)
    of THAT kernel + common.h -- an edit elsewhere in csrc/ does not invalidate its figure; kernel None: all of csrc/."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'deepfluorolabeling-ipcai2020_amd', 'csrc')
    files = sorted(f for f in os.listdir(d) if f.endswith(('.hip', '.h', '.inc')))
    if kernel is not None:
        base = kernel.split('<')[0]
        files = sorted(set(KERNEL_SOURCES.get(base, files) + ['common.h'])) if base in KERNEL_SOURCES else files
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


CONV_KERNELS = ['conv_gemm_kernel<2,2,2,2>', 'conv_gemm_kernel<2,2,2,1>', 'conv_gemm_kernel<4,1,2,1>',
                'conv_gemm_kernel<2,2,1,1>', 'conv_gemm_kernel<1,2,1,1>', 'direct_conv_kernel',
                'conv_rows_kernel<3,1,2,1>', 'conv_rows_kernel<3,1,1,2>', 'conv_rows_kernel<3,1,1,1>']
CONVP_TILES = ['4,1,2,1', '4,1,1,1', '2,2,4,1', '2,2,3,1', '2,2,2,1', '1,4,2,1', '1,4,3,1', '1,4,4,1', '1,4,6,1', '1,4,9,1', '2,2,1,1', '1,4,1,1',
               '4,1,3,1', '4,1,4,1', '2,2,6,1', '2,2,2,2', '2,2,3,2', '2,2,4,2', '4,1,2,2', '4,1,3,2', '1,4,2,2', '1,4,3,2',
               '4,1,1,1', '4,1,2,1', '2,2,1,1', '2,2,2,1', '1,4,1,1', '1,4,2,1', '4,1,1,2', '2,2,1,2',   # (22 ...: the streamed 1x1 forms)
               '1,4,3,1,k2', '1,4,2,1,k2', '1,4,4,1,k2', '2,2,3,1,k2', '2,2,2,1,k2', '1,4,2,2,k2', '2,2,2,2,k2', '4,1,3,1,k2', '4,1,2,1,k2',   # (30 ...: two k-groups, 512 threads)
               'latency form',      # csrc/convp_bf16.hip kTiles: waves M x N, tiles M x N per wave
               # (40 ...: the unrolled 3x3 form, csrc/convq_bf16.hip: row waves, column waves, k-groups; 49 ...: the same, persistent)
               'q1,4,1', 'q1,4,2', 'q2,4,1', 'q2,2,1', 'q4,2,1', 'q2,2,2', 'q4,1,1', 'q8,1,1', 'q4,1,2',
               'q1,4,1,p', 'q1,4,2,p', 'q2,4,1,p', 'q2,2,1,p', 'q4,2,1,p', 'q2,2,2,p', 'q4,1,1,p', 'q8,1,1,p', 'q4,1,2,p',
               # (58 ...: the narrow 3x3 form, csrc/convn_bf16.hip: waves side by side, rows per wave; named convn_kernel<column tiles, ...>)
               'n2,4', 'n2,6', 'n1,4', 'n1,6', 'n2,3', 'n1,3', 'n2,3,p', 'n1,3,p']      # (64, 65: persistent)
WGRAD_KERNELS = ['wgrad_kernel<2,2,2,2,1>', 'wgrad_kernel<2,2,1,1,1>', 'wgrad_kernel<1,1,1,1,3>',
                 'wgrad_kernel<1,1,1,1,2>', 'wgrad_kernel<1,1,1,1,1>', 'direct_wgrad_kernel', 'wgrad_kernel<2,2,1,1,3>']


def synth_batch(B, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, 192, 192, generator=g)
    lab = torch.randint(0, 7, (B, 184, 184), generator=g)
    tseg = torch.stack([(lab == c) for c in range(7)], 1).float()
    theat = torch.rand(B, 14, 184, 184, generator=g) * 0.02
    return x.to(dev), tseg.to(dev), theat.to(dev)


def op_profile(plan, lib, nat, stream, detail=None):
    """Per-kernel-family time (hipEvents around every op of one forward+backward replay) and algorithmic FLOPs."""
    groups = {}

    def account(prog, ms):
        for st, t in zip(prog.structs, ms):
            if isinstance(st, nat.ConvArgs):
                cfg = lib.dfl_conv_config(C.addressof(st))
                name = CONV_KERNELS[cfg] if cfg < 16 else ('convn_kernel<%d,%s>' % (st.Ntot // 32, CONVP_TILES[cfg - 16][1:]) if cfg >= 74 else
                                                           'convq_kernel<%s>' % CONVP_TILES[cfg - 16][1:] if cfg >= 56 else 'convp_kernel<%s>' % CONVP_TILES[cfg - 16])
                if st.scatter2x2:
                    M = st.N * st.Hin * st.Win
                else:
                    M = st.N * st.Hout * st.Wout
                fl = 2.0 * M * st.KH * st.KW * st.Cin * st.Ntot
                esz = 2.0 if st.x_bf16 else 4.0
                by = esz * (st.N * st.Hin * st.Win * st.Cin + st.KH * st.KW * st.Cin * st.Ntot + M * st.Ntot)
                if st.x_mode:               # the operand is formed from TWO tensors (dy and the saved ReLU output)
                    by += esz * st.N * st.Hin * st.Win * st.Cin
                if st.x_out:                # ... and written once for the layer's weight gradient (dfl_conv_args.x_out)
                    by += esz * st.N * st.Hin * st.Win * st.Cin
            elif isinstance(st, nat.WgradArgs):
                cfg = lib.dfl_wgrad_config(C.addressof(st))
                name = WGRAD_KERNELS[cfg] if cfg < 16 else 'wgradp_kernel<%s>' % ('3,3', '2,2', '1,1')[cfg - 16]
                fl = 2.0 * st.N * st.Hout * st.Wout * st.Cm * st.Cg * st.KH * st.KW
                by = (2.0 if st.g_bf16 else 4.0) * st.N * st.Hin * st.Win * st.Cg + (2.0 if st.d_bf16 else 4.0) * st.N * st.Hout * st.Wout * st.Cm \
                    + 4.0 * st.Cm * st.Cg * st.KH * st.KW
                if st.d_mode:               # the dense operand is formed from TWO tensors (dy and the saved ReLU output)
                    by += (2.0 if st.d_bf16 else 4.0) * st.N * st.Hout * st.Wout * st.Cm
            else:
                name, fl, by = type(st).__name__, 0.0, 0.0
            if detail is not None:
                if isinstance(st, nat.ConvArgs):
                    desc = 'N%d %dx%d Cin%d -> %dx%d Ntot%d k%d s%d splits%d' % (st.N, st.Hin, st.Win, st.Cin, st.Hout, st.Wout,
                                                                                  st.Ntot, st.KH, st.stride, st.splits)
                elif isinstance(st, nat.WgradArgs):
                    desc = 'N%d %dx%d Cg%d Cm%d k%d splits%d' % (st.N, st.Hin, st.Win, st.Cg, st.Cm, st.KH, st.splits)
                elif isinstance(st, nat.SumPartialsArgs):
                    desc = 'n%d splits%d' % (st.n, st.splits)
                else:
                    desc = ''
                detail.append('%-28s %8.3f ms %7.1f TF  %s' % (name, t, fl / (t * 1e-3) / 1e12 if t > 0 else 0, desc))
            gr = groups.setdefault(name, [0.0, 0.0, 0, 0.0])
            gr[0] += t
            gr[1] += fl
            gr[2] += 1
            gr[3] += by            # compulsory bytes: each operand once (input, weights, output)
    account(plan.fwd, plan.fwd.run_timed(stream))
    if detail is not None:
        detail.append('---- backward ----')
    account(plan.bwd, plan.bwd.run_timed(stream))
    return groups


def roofline_of(groups, math_name):
    """`roofline` object of the dominant kernel of an op profile (op_profile): achieved = ALGORITHMIC flops (or compulsory bytes)
    per launch / its average launch time (hipEvent pairs on the launch stream), against the binding roof."""
    tot_ms = sum(v[0] for v in groups.values())
    name, (ms, fl, n, by) = max(groups.items(), key=lambda kv: kv[1][0])
    achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    nprod = MATH[math_name][1]
    # peak = the guide's hardware peak of the matrix instruction this kernel issues (MI355X_MICROARCH.md): dense bf16
    # MFMA for every bf16-product mode, fp32 MFMA for fp32 products.  `achieved` counts ALGORITHMIC flops (2 * MACs of
    # the layer), so emulation overhead (3 / 6 bf16 products per fp32 product) shows as a lower fraction, not as a
    # lower roof; `mfma_issue_frac` = achieved * products-per-product / peak is the share of the pipe's issue slots.
    peak = F32_MFMA_PEAK_TFLOPS if nprod == 0 else BF16_MFMA_PEAK_TFLOPS
    # Which roof binds this kernel: the one its algorithmic work needs longer under -- compulsory bytes at the HBM peak or
    # algorithmic flops at the matrix peak.  (With the BatchNorm + ReLU backward formed inside the weight gradient the
    # kernel reads three activation tensors per layer: its byte floor can exceed its flop floor.)
    gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    mfma_frac, hbm_frac = achieved / peak, gbs / HBM_PEAK_GBS
    if hbm_frac > mfma_frac:
        head = {'bound': 'hbm', 'kernel': name, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(hbm_frac, 4)}
    else:
        head = {'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(mfma_frac, 4)}
    return dict(head, **{'traffic': None, 'mfma_tflops': round(achieved, 2), 'mfma_frac': round(mfma_frac, 4),
                'hbm_gbs_compulsory': round(gbs, 1), 'hbm_frac_compulsory': round(hbm_frac, 4),
                'mfma_issue_frac': round(achieved * max(nprod, 1) / peak, 4),
                'peak_note': ('fp32 MFMA peak (v_mfma_f32_32x32x2_f32)' if nprod == 0 else
                              'dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16); this mode issues %d bf16 product(s) per '
                              'algorithmic product' % nprod),
                'launches_per_step': n, 'avg_launch_ms': round(ms / n, 4),
                'share_of_kernel_time': round(ms / tot_ms, 3),
                'algorithmic_flop_per_launch': round(fl / n), 'compulsory_bytes_per_launch': round(by / n)})


def profile_plan(net, x, lib, nat, detail=None):
    """op_profile of the training plan `net` last ran on (one more forward + backward, replayed op by op)."""
    plan = net._last_train_plan()
    seg, heat = net(x)
    hold = (torch.randn_like(seg) * 1e-6, torch.randn_like(heat) * 1e-6)
    plan.bind_grads(seg, hold[0], hold[1])
    stream = torch.cuda.current_stream().cuda_stream
    plan.bwd.run(stream)
    torch.cuda.synchronize()
    groups = op_profile(plan, lib, nat, stream, detail)
    plan.busy = False
    return plan, groups


def configs3(B=8, H=736, P=768, steps=8, warmup=3):
    """BASELINE configs[3]: 2x-downsampled 736x736 padded to 768, paper U-Net with both heads, batch 8, one MI355X: the same
    step body (train.py:405-430), every loss read; images/s and ms/step in the arithmetic that is current."""
    import dfl_amd
    from dfl_amd.util import LateScalars
    dev = torch.device('cuda', torch.cuda.current_device())
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    torch.manual_seed(1)
    net = dfl_amd.UNet(**PAPER).to(dev)
    opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 1, P, P, generator=g).to(dev)
    lab = torch.randint(0, 7, (B, H, H), generator=g)
    tseg = torch.stack([(lab == c) for c in range(7)], 1).float().to(dev)
    theat = (torch.rand(B, 14, H, H, generator=g) * 0.02).to(dev)
    net.train()
    late = LateScalars(depth=1)

    def step():
        opt.zero_grad()
        seg, heat = net(x)
        loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
        loss.backward()
        opt.step()
        return late.push(loss)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(warmup):
        step()
    late.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    late.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop = 54.5e9 * (P * P) / (192 * 192) * B
    out = {'workload': 'BASELINE configs[3]: 2x-downsampled 736x736 (padded 768), paper U-Net dual head, batch %d, 1 MI355X' % B,
           'images_per_sec': round(B * steps / dt, 2), 'ms_per_step': round(dt / steps * 1e3, 2), 'steps': steps, 'warmup': warmup,
           'tflops': round(flop / (dt / steps) / 1e12, 1), 'peak_mem_gb': round((torch.cuda.max_memory_allocated() - base) / 2 ** 30, 2)}
    del net, opt, x, tseg, theat
    torch.cuda.empty_cache()
    return out


def _cpu_steps(net, opt, x, tseg, theat, warm, steps, R):
    for _ in range(warm):
        R.train_step(net, opt, x, tseg, theat, 0.5)
    t0 = time.perf_counter()
    for _ in range(steps):
        R.train_step(net, opt, x, tseg, theat, 0.5)
    return time.perf_counter() - t0


def cpu_baseline(B):
    """The oracle (CPU restatement of the reference, checked against it in tests/test_oracle_golden.py) on this box's
    host cores: the same step body (train.py:405-430), PyTorch CPU fp32.  Thread count: swept over 8 / 16 / 32 / 64
    threads on short runs of BASELINE configs[0] (batch 4, segmentation head only), the best one is used for 5 timed steps
    of configs[0] and for a bounded sample of the batch-`B` dual-head workload the GPU line is quoted on."""
    from oracle import ref_cpu as R
    prev = torch.get_num_threads()
    ncpu = os.cpu_count() or prev
    cfg0 = dict(PAPER, num_lands=0)
    torch.manual_seed(1234)
    net0 = R.OracleUNet(**cfg0)
    opt0 = torch.optim.SGD(net0.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    x0, t0seg, _ = synth_batch(4, 99, 'cpu')
    net0.train()
    sweep = {}
    t_begin = time.perf_counter()
    for th in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu}):      # (more threads than 64 only ever lost: 0.05 images/s at 256)
        if sweep and time.perf_counter() - t_begin > 20.0:
            break
        torch.set_num_threads(th)
        sweep[th] = round(4 * 2 / _cpu_steps(net0, opt0, x0, t0seg, None, 1, 2, R), 2)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n0 = 5
    d0 = _cpu_steps(net0, opt0, x0, t0seg, None, 0, n0, R)
    torch.manual_seed(1234)
    net = R.OracleUNet(**PAPER)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    x, tseg, theat = synth_batch(B, 4321, 'cpu')
    net.train()
    dw = _cpu_steps(net, opt, x, tseg, theat, 0, 1, R)                    # warm-up, also sizes the sample
    steps = max(2, min(8, int(12.0 / max(dw, 1e-3))))                     # the whole CPU leg stays under about a minute
    d = _cpu_steps(net, opt, x, tseg, theat, 0, steps, R)
    torch.set_num_threads(prev)
    return {'value': round(B * steps / d, 2), 'unit': 'images/sec', 'cores': best, 'kind': 'port',
            'sample': '%d training steps (after 1 warm-up) of the same batch-%d paper dual-head workload, oracle/ref_cpu.py '
                      '(PyTorch CPU fp32), torch.set_num_threads(%d) = the best of the sweep' % (steps, B, best),
            'host_cpus': ncpu, 'thread_sweep_images_per_sec_configs0': sweep,
            'configs0': {'value': round(4 * n0 / d0, 2), 'unit': 'images/sec', 'steps': n0, 'threads': best,
                         'workload': 'BASELINE configs[0]: batch 4, 7-class segmentation head only, Dice loss, SGD nesterov'}}


def fwd_ms_per_img(lib, nat, dev, math_name=''):
    """Second half of BASELINE.json's metric: eval-mode forward time per image -- batch 1 at 192x192 (the 8x-downsampled
    size) and the 5-net full-resolution ensemble of configs[4] (1436x1436 padded to 1440, one hipGraph replay per net plus
    the ensemble reduction of util.py:318-373)."""
    import dfl_amd
    from dfl_amd import util
    out = {}
    torch.manual_seed(7)
    net = dfl_amd.UNet(**PAPER).to(dev).eval()
    x = torch.randn(1, 1, 192, 192, device=dev)
    with torch.no_grad():
        for _ in range(20):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):                 # (a 0.3 ms forward: the pipeline's fill is 1 % of 50 replays)
            net(x)
        torch.cuda.synchronize()
    out['192x192_batch1'] = round((time.perf_counter() - t0) / 200 * 1e3, 4)      # pipelined: replays back to back, one sync at the end
    # ... and as util.py:321,363-366 times an image: the image comes from the host, forward, reduction to labels,
    # a device synchronisation, the labels go back to the host -- one image at a time, nothing overlaps the next one
    x_host = torch.randn(1, 1, 192, 192)
    with torch.no_grad():
        lat = []
        for i in range(120):
            t0 = time.perf_counter()
            seg, heat = net(x_host.to(dev))
            labels, heats, _ = util.ensemble_reduce([seg], [heat], (184, 184))
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            labels.cpu()
            t2 = time.perf_counter()
            if i >= 20:
                lat.append((t1 - t0, t2 - t0))
    lat.sort()
    out['192x192_batch1_latency'] = {'ms_per_img': round(lat[len(lat) // 2][0] * 1e3, 4), 'ms_per_img_with_labels_on_host': round(sorted(v[1] for v in lat)[len(lat) // 2] * 1e3, 4),
                                     'what': 'median of 100 images, one at a time: H2D copy of the image, forward (one hipGraph replay), dfl_ensemble_reduce to '
                                             'labels / heat maps, torch.cuda.synchronize -- the timed region of util.py:321,363-366 -- and with the uint8 labels copied back'}
    # the same forward in the arithmetics that hold north_star's 1e-4 bar (fp32 tensors: fp32 matrix instructions / split-bf16
    # products): the patch-free latency form exists for bf16 tensors only, these replay the fp32-tensor kernels
    prev = lib.dfl_get_math_mode()
    par = {}
    for name in ('fp32', 'bf16x3'):
        if MATH[name][0] == prev:
            continue
        nat.check(lib.dfl_set_math_mode(MATH[name][0]), 'dfl_set_math_mode')
        with torch.no_grad():
            for _ in range(3):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                net(x)
            torch.cuda.synchronize()
        par[name] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
    nat.check(lib.dfl_set_math_mode(prev), 'dfl_set_math_mode')
    out['192x192_batch1_parity_modes'] = par
    del net
    # the 5-net ensemble at the same size (util.py:318-356 on 8x-downsampled images): forwards on a stream per net + the reduction
    nets = []
    for i in range(5):
        torch.manual_seed(10 + i)
        nets.append(dfl_amd.UNet(**PAPER).to(dev).eval())

    def small():
        with torch.no_grad():
            outs = util.forward_nets(nets, x, 14)
            return util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (184, 184))
    for _ in range(10):
        small()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        small()
    torch.cuda.synchronize()
    out['192x192_5net_ensemble'] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
    del nets
    nets = []
    for i in range(5):
        torch.manual_seed(10 + i)
        nets.append(dfl_amd.UNet(**PAPER).to(dev).eval())
    x = torch.randn(1, 1, 1440, 1440, device=dev)

    def one():
        with torch.no_grad():
            outs = [n(x) for n in nets]
            return util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (1436, 1436))
    one()
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    out['1440x1440_5net_ensemble'] = round(dt * 1e3, 3)
    out['1440x1440_per_net'] = round(dt * 1e3 / 5, 3)
    out['unit'] = 'ms per image'
    out['math'] = math_name
    out['note'] = 'eval forward, inputs resident in HBM, hipGraph replay per net; ensemble figure includes dfl_ensemble_reduce'
    del nets
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU')
    ap.add_argument('--prewarm-seconds', type=float, default=2.0,
                    help='untimed steps run for this long BEFORE the --warmup steps: a fresh box (cold GPU clocks, cold host caches) '
                         'runs its first few hundred milliseconds of steps up to 50 %% slower than steady state (measured: first process on a box '
                         '7.9 ms per step, every later one 5.1; training runs for hours).  0 = off')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-fwd', action='store_true', help='skip the eval-forward timings (fwd_ms_per_img)')
    ap.add_argument('--no-configs3', action='store_true', help='skip the 768x768 batch-8 training timing (configs3)')
    ap.add_argument('--detail', action='store_true', help='per-op table on stderr')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' for a self-test)")
    ap.add_argument('--optimizer', default='dfl', choices=['dfl', 'torch'],
                    help="dfl = dfl_amd.SGD (one dfl_sgd_step launch per contiguous run); torch = torch.optim.SGD")
    ap.add_argument('--math', default='bf16s', choices=sorted(MATH),
                    help='arithmetic of the conv / weight-gradient GEMMs.  bf16s (default) = BASELINE configs[1] as named: bf16 '
                         'activations, activation gradients and GEMM weight copies in HBM, bf16 MFMA, fp32 accumulation / BatchNorm '
                         'statistics / losses / weight gradients / master weights.  bf16x3 = fp32 tensors, values split into hi+lo '
                         'bf16, 3 bf16 MFMA products (2^-16 per product: forward within the 1e-4 parity bar); fp32 = fp32 MFMA '
                         '(the parity gate); bf16 = fp32 tensors, one bf16 product.  The other modes are timed beside the chosen one.')
    ap.add_argument('--no-fp32-reference', action='store_true', help='skip the extra fp32-product timing (profiling runs)')
    ap.add_argument('--sync-loss', action='store_true', help='loss.item() right after every optimizer step (the reference loop verbatim) '
                    'instead of reading each loss one step late')
    ap.add_argument('--no-overlap', action='store_true', help='all-reduce after backward instead of overlapped buckets')
    ap.add_argument('--grad-compress', default=None, choices=['bf16'], help='all-reduce the gradient buckets as bf16 (default: fp32)')
    ap.add_argument('--force-dp', action='store_true',
                    help='keep the collective path on in a one-rank group (self-test of the RCCL path on a 1-GPU box)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        # `python bench.py --gpus N` as typed: become the launcher the contract names (one rank per GPU, rendezvous on
        # 127.0.0.1); rank 0 of the relaunched job prints the one JSON line on this process's stdout
        import socket
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import dfl_amd
    from dfl_amd import _native as nat
    from dfl_amd.parallel import DataParallel, init_process_group_from_env
    import torch.distributed as dist

    rank, world, local = init_process_group_from_env(args.backend, force=args.force_dp)
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit('--gpus %d inside a one-rank group (WORLD_SIZE / RANK are set: launched by hand?)' % args.gpus)
    dev = torch.device('cuda', local % torch.cuda.device_count())   # (gloo self-test: several ranks may share a GPU)
    torch.cuda.set_device(dev)
    lib = nat.lib()
    mode_before = lib.dfl_get_math_mode()
    nat.check(lib.dfl_set_math_mode(MATH[args.math][0]), 'dfl_set_math_mode')

    torch.manual_seed(1234)
    net = dfl_amd.UNet(**PAPER).to(dev)
    dp = DataParallel(net, overlap=not args.no_overlap, force_collectives=args.force_dp, compress=args.grad_compress) if (world > 1 or args.force_dp) else None
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
    SGD = dfl_amd.SGD if args.optimizer == 'dfl' else torch.optim.SGD
    opt = SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    B = args.batch
    x, tseg, theat = synth_batch(B, 4321 + rank, dev)
    net.train()

    from dfl_amd.util import LateScalars
    late = LateScalars(depth=0 if args.sync_loss else 1)   # as train.py's loop: every loss is read, one step late

    def step():
        opt.zero_grad()
        seg, heat = net(x)
        loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
        loss.backward()
        opt.step()
        return late.push(loss)      # train.py:430 reads the loss of every step; so do we (all of them, see the flush below)

    prewarm_steps = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        step()
        prewarm_steps += 1
    for _ in range(args.warmup):
        step()
    late.flush()
    multi = dist.is_initialized()          # (also a one-rank group under --force-dp)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = 0.0
    for _ in range(args.steps):
        last = step()
    last = (late.flush() or [last])[-1]     # inside the timed region: every loss value has reached the host
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    roofline = None
    extra = {'rccl_ranks': world if (multi and args.backend == 'nccl') else 0, 'prewarm_steps': prewarm_steps,
             'prewarm_seconds': args.prewarm_seconds}
    # host time to queue one step when the GPU is idle at its start (nothing to wait for): what a slow host adds to a step once
    # it exceeds the GPU time (8.4 ms per step observed by the driver where the kernels take 5.3)
    nh = 10
    host = 0.0
    late.flush()
    for _ in range(nh):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        host += time.perf_counter() - t1
    late.flush()
    torch.cuda.synchronize()
    extra['host_enqueue_ms_per_step'] = round(host / nh * 1e3, 3)
    if rank == 0 and not args.no_profile:
        # one more forward/backward, replayed op by op with hipEvents on the launch stream
        detail = [] if args.detail else None
        plan, groups = profile_plan(net, x, lib, nat, detail)
        if detail:
            sys.stderr.write('\n'.join(detail) + '\n')
        tot_ms = sum(v[0] for v in groups.values())
        roofline = roofline_of(groups, args.math)
        name, (ms, fl, n, by) = max(groups.items(), key=lambda kv: kv[1][0])
        # HBM bytes per launch of that kernel come from the separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
        # tools/profile_round.sh), which cannot run inside this process; the committed summary is quoted when present
        for tname in TRAFFIC_FILES:
            tfile = os.path.join(ROOT, 'profiles', tname)
            if not os.path.exists(tfile):
                continue
            try:
                tj = json.load(open(tfile))
                rec = tj['kernels'].get(name)
                if rec:
                    current = rec.get('src_sha16', tj.get('csrc_sha16')) == (csrc_sha16(name) if 'src_sha16' in rec else csrc_sha16())
                    roofline['traffic'] = rec['hbm_bytes_per_launch']
                    roofline['hbm_frac_traffic'] = round(rec['hbm_bytes_per_launch'] / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    roofline['traffic_unit'] = 'HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB, rocprofv3 PMC passes of ' \
                                               'the same command (profiles/%s)' % tname
                    roofline['traffic_measured_in'] = tj.get('round', tname.split('_')[0])
                    roofline['traffic_kernels_unchanged_since'] = bool(current)   # False: csrc/ changed after the PMC passes
                    break
            except (ValueError, KeyError):
                pass
        fl_all = sum(v[1] for v in groups.values())
        # GPU time of a whole step = steps queued back to back with NO host read in between (the queue never runs dry: the host
        # queues a step in `host_enqueue_ms_per_step`, a fraction of what the GPU needs for it), losses read at the very end
        from dfl_amd.util import LateScalars as _LS
        nq = 10
        deep = _LS(depth=nq + 1)
        torch.cuda.synchronize()
        tq = time.perf_counter()
        for _ in range(nq):
            opt.zero_grad()
            seg, heat = net(x)
            loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
            loss.backward()
            opt.step()
            deep.push(loss)
        deep.flush()
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - tq) / nq * 1e3
        extra['kernel_time_ms_per_step'] = round(tot_ms, 3)
        extra['gpu_time_ms_per_step'] = round(gpu_ms, 3)
        extra['gpu_time_note'] = 'kernel_time = hipEvent pairs around every op of the forward and backward programs; gpu_time = %d whole ' \
                                 'steps (loss, optimizer, weight re-layout included) queued back to back without a host read' % nq
        extra['gpu_idle_frac'] = round(max(0.0, 1.0 - gpu_ms / ms_per_step), 4)
        extra['program_ops_per_step'] = sum(v[2] for v in groups.values())
        extra['partial_sum_mb_per_step'] = round(plan.partial_sum_bytes / 1e6, 1)   # fp32 slices the batched reductions read
        extra['whole_step_tflops'] = round(fl_all / (ms_per_step * 1e-3) / 1e12, 2)
        extra['kernels'] = {k: {'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 and v[1] > 0 else None,
                                'launches': v[2]} for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0])}

    if rank == 0 and world == 1 and not args.no_profile and not args.no_fp32_reference:
        # the same step in the other arithmetic modes, for reference: fp32 MFMA products (the mode the 1e-4 forward bar is
        # written for), fp32 tensors with 3 bf16 products per product (bf16x3, inside that bar), fp32 tensors with plain bf16
        # products, bf16 storage (BASELINE configs[1] as named)
        n32 = 10
        for mode, key in (('fp32', 'fp32_products'), ('bf16x3', 'bf16x3_products'), ('bf16', 'bf16_products'), ('bf16s', 'bf16_storage')):
            if mode == args.math:
                continue
            nat.check(lib.dfl_set_math_mode(MATH[mode][0]), 'dfl_set_math_mode')
            for _ in range(3):
                step()
            late.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n32):
                step()
            late.flush()
            torch.cuda.synchronize()
            dside = time.perf_counter() - t0
            extra[key] = {'value': round(B * n32 / dside, 2), 'ms_per_step': round(dside / n32 * 1e3, 3), 'steps': n32}
            if mode == 'fp32':
                # the parity gate's own roofline: dominant kernel of the fp32-MFMA mode, same fields as `roofline`
                late.flush()
                _, g32 = profile_plan(net, x, lib, nat)
                extra['roofline_fp32'] = roofline_of(g32, 'fp32')
                extra['roofline_fp32']['kernel_time_ms_per_step'] = round(sum(v[0] for v in g32.values()), 3)
        nat.check(lib.dfl_set_math_mode(MATH[args.math][0]), 'dfl_set_math_mode')

    if rank == 0 and world == 1 and not args.no_profile and not args.no_fwd:
        extra['fwd_ms_per_img'] = fwd_ms_per_img(lib, nat, dev, args.math)
    if rank == 0 and world == 1 and not args.no_profile and not args.no_configs3:
        del net, opt, x, tseg, theat
        torch.cuda.empty_cache()
        extra['configs3'] = dict(configs3(), math=args.math)
    nat.check(lib.dfl_set_math_mode(mode_before), 'dfl_set_math_mode')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(B)

    if rank == 0:
        out = {'metric': 'train images/sec, paper U-Net (depth 6, wf 5, BN, strided-conv down, seg + 14-landmark heads), '
                         '1x192x192 (8x-downsampled 184x184 padded)',
               'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': ('f32' if args.math == 'fp32' else
                         'bf16 (bf16 activations, activation gradients and GEMM weight copies in HBM; bf16 MFMA, fp32 accumulate, '
                         'BatchNorm statistics, losses, weight gradients and master weights)' if args.math == 'bf16s' else
                         '%s (fp32 tensors; GEMM operands split into bf16 parts, bf16 MFMA, fp32 accumulate)' % args.math),
               'data': 'synthetic',
               'config': {'workload': 'BASELINE configs[1]: 8x-downsampled, seg + 14-landmark heat-map dual head, '
                                      'batch %d per GPU, Dice+NCC loss, SGD nesterov' % B,
                          'global_batch': B * world, 'image': '1x192x192', 'parallelism': 'dp%d' % world, 'math': args.math,
                          'optimizer': '%s(momentum 0.9, nesterov, wd 1e-4)' % ('dfl_amd.SGD' if args.optimizer == 'dfl' else 'torch.optim.SGD'), 'last_loss': round(last, 6)},
               'roofline': roofline, 'cpu_baseline': cpu}
        out.update(extra)
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
