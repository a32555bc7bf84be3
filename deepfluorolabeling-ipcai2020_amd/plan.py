"""Execution plans for the U-Net: the graph builder that turns one (architecture, input shape, mode) into recorded
lists of libdfl_hip.so calls (forward, backward, weight re-layout) plus the HBM buffers they run on.

Reference being restated: UNet.forward / UNetConvBlock.forward / UNetUpBlock.forward
(train_test_code/unet.py:161-193, 226-233, 254-260) and their autograd.  Data layout: every internal activation
is NHWC fp32; torch.cat([up, bridge], 1) (unet.py:257) is free because both producers write straight into the two
channel halves of one buffer (pixel stride 2C); BatchNorm is never applied as a separate pass -- its scale/shift
ride on the next consumer's loads (see include/dfl_hip.h).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _native as nat
from ._native import (ConvArgs, WgradArgs, PackJob, BnFinalizeArgs, ColstatsArgs, BnBwdFinalizeArgs, BnReluBwdArgs,
                      AffineCopyArgs, PoolArgs, HeadFwdArgs, HeadBwdArgs, SumPartialsArgs, PackArgs, BnEvalArgs, UpsampleArgs, BnLiveJob, BnLiveArgs, BnBwdLiveJob, BnBwdLiveArgs,
                      ReducePartialsArgs, MemsetArgs, ReduceJob, ReduceBatchArgs, Program)

BN_EPS = 1.0e-5
BN_MOMENTUM = 0.1


class Act:
    """An NHWC activation window: channels [0, C) of rows with pixel stride ld (in elements) starting at ptr; esz = bytes
    per element (4: fp32; 2: bf16 tensors of math mode 4)."""
    __slots__ = ('t', 'ptr', 'ld', 'N', 'H', 'W', 'C', 'esz')

    def __init__(self, t, ptr, ld, N, H, W, C, esz=4):
        self.t, self.ptr, self.ld, self.N, self.H, self.W, self.C, self.esz = t, ptr, ld, N, H, W, C, esz

    @property
    def M(self):
        return self.N * self.H * self.W

    @property
    def bf16(self):
        return int(self.esz == 2)

    def chan_slice(self, c0, c):
        return Act(self.t, self.ptr + self.esz * c0, self.ld, self.N, self.H, self.W, c, self.esz)


class PlanError(RuntimeError):
    pass


# Diagnosis (tests/test_gpu_bf16_stepwise.py): plans built while this is True give every activation gradient its own buffer
# instead of the three shared scratch tensors, so that all of them can be read after a backward pass (UNetPlan.dbg names them).
KEEP_GRADS = False


class UNetPlan:
    def __init__(self, cfg, params, buffers, N, H, W, training, need_grad, device, input_grad=False):
        """cfg: dict of the UNet constructor flags; params / buffers: name -> tensor (module state)."""
        self.cfg = cfg
        self.P = params
        self.Bf = buffers
        self.N, self.H, self.W = N, H, W
        self.training = training
        self.need_grad = need_grad
        self.input_grad = bool(input_grad and need_grad)    # d(loss)/d(input) as well (nn.Module semantics; no reference script asks)
        self.dx_in = None
        self.dev = device
        self.lib = nat.lib()
        self._keep = []
        self.fwd = Program()
        self.bwd = Program() if need_grad else None
        self._prep = Program()          # eval-mode BatchNorm scale / shift of an inference plan: part of its pack program
        self._pack_jobs = []
        self._pack_dsts = []
        self.busy = False
        self.graph = None               # hipGraph of the forward program (inference plans, captured on first use)
        self.generation = 0
        self._scratch = {}
        self._red_pending = []          # deferred sums (src, dst, n, stride, count), see _defer_sum
        self._red_bytes = 0
        self._red_flushes = []          # (index of the batch op in bwd, [dst pointers])
        self.partial_sum_bytes = 0      # bytes of fp32 partial sums the batched reductions of one backward pass read
        self._side = []                 # weight gradients running on the side stream: dict(done=event, op=index, waited=index|None)
        self.math = self.lib.dfl_get_math_mode()   # product arithmetic the plan is recorded for (split operand formats)
        # math mode 4, "bf16 storage": internal activations, their gradients and the GEMM copies of the weights are bf16
        # tensors (BASELINE configs[1] as named); statistics, losses, master weights and weight gradients stay fp32
        self.bf16 = self.math == 4
        self.adt = torch.bfloat16 if self.bf16 else torch.float32
        self.aesz = 2 if self.bf16 else 4
        self._packed_split = {}         # packed-weight address -> stored as split quads
        self.relu_out = {}              # nn.ReLU module name -> Act of its output (saved for backward; introspection for tests)
        self._tot_fwd, self._tot_bwd, self._live_jobs, self._live_bwd_pending = [None, 0], [None, 0], [], []
        self.dbg = {}                   # name -> Act / tensors of intermediate results (introspection for tests, see KEEP_GRADS)
        self.keep_grads = bool(KEEP_GRADS)
        self.pool_in = {}               # level -> Act the max-pool of that level reads
        self._build()

    # ------------------------------------------------------------------------------------------ memory
    def _new(self, nelem, dtype=torch.float32):
        t = torch.empty(max(int(nelem), 1), dtype=dtype, device=self.dev)
        self._keep.append(t)
        return t

    def _const(self, n, value):
        """A plan-owned constant vector (shared per (length, value))."""
        key = ('const', int(n), float(value))
        t = self._scratch.get(key)
        if t is None:
            t = torch.full((int(n),), float(value), dtype=torch.float32, device=self.dev)
            self._keep.append(t)
            self._scratch[key] = t
        return t

    def _act(self, N, H, W, C):
        t = self._new(N * H * W * C, self.adt)
        return Act(t, t.data_ptr(), C, N, H, W, C, self.aesz)

    def _scratch_act(self, key, N, H, W, C):
        """Backward scratch shared by all blocks (stream order makes reuse safe)."""
        need = N * H * W * C
        t = self._scratch.get(key)
        if t is None or t.numel() < need:
            raise PlanError('scratch %s not sized' % key)
        if self.keep_grads:
            return self._act(N, H, W, C)
        return Act(t, t.data_ptr(), C, N, H, W, C, self.aesz)

    # ------------------------------------------------------------------------------------------ weights
    def _pack(self, w, kind, flip=0):
        """Register a re-layout job of parameter w = [A][B][KH][KW] into the quad-packed GEMM operand dfl_conv2d reads
        (include/dfl_hip.h, dfl_pack_job); returns the destination tensor."""
        A, B, KH, KW = w.shape
        Cc = KH * KW
        K = {1: Cc * B, 2: Cc * A, 3: A}[kind]
        N = {1: A, 2: B, 3: Cc * B}[kind]
        cin = B if kind == 1 else A
        if self.bf16 and cin % 16 == 0:
            # bf16 chunk layout [K/16][N][16] of the patch-resident kernels (every layer but the 1-channel first one)
            dst = self._new((K + 15) // 16 * N * 16, torch.bfloat16)
            self._packed_split[dst.data_ptr()] = 2
            self._pack_jobs.append((w, dst, A, B, Cc, kind, flip, 2))
            return dst
        dst = self._new((K + 3) // 4 * N * 4)
        # split quads (hi4 | lo4 bf16) when the consuming conv will take the split-bf16 fast path: its input channels
        # (B for a forward operand, A for the transposed ones) must be a multiple of 16 and K large enough for the GEMM
        split = int(self.math in (1, 3) and cin % 16 == 0 and self.WSPLIT)
        self._packed_split[dst.data_ptr()] = split
        self._pack_jobs.append((w, dst, A, B, Cc, kind, flip, split))
        return dst

    def _pack_conv_fwd(self, w):      # Conv2d [Co][Ci][T]: k = (tap, ci), n = co
        return self._pack(w, 1)

    def _pack_conv_dgrad(self, w):    # stride-1 Conv2d data gradient: k = (flipped tap, co), n = ci
        return self._pack(w, 2, flip=1)

    def _pack_down_dgrad(self, w):    # Conv2d(k2,s2) data gradient, scatter form: k = co, n = (ab, ci)
        return self._pack(w, 3)

    def _pack_convT_fwd(self, w):     # ConvTranspose2d [Ci][Co][ab], scatter form: k = ci, n = (ab, co)
        return self._pack(w, 3)

    def _pack_convT_dgrad(self, w):   # ConvTranspose2d data gradient = conv 2x2 s2 over dy: k = (ab, co), n = ci
        return self._pack(w, 1)

    def _finish_pack(self):
        n = len(self._pack_jobs)
        self.pack = Program()
        self.pack.extend(self._prep)
        if n == 0:
            return
        if not self.PACK_OVERLAP and self.PACK_TILED:
            self._finish_pack_tiled()
            return
        arr = (PackJob * n)()
        mx = 0
        for i, (src, dst, A, B, Cc, kind, flip, split) in enumerate(self._pack_jobs):
            arr[i].src, arr[i].dst = src.data_ptr(), dst.data_ptr()
            arr[i].A, arr[i].B, arr[i].C, arr[i].kind, arr[i].flip, arr[i].split = A, B, Cc, kind, flip, split
            mx = max(mx, A * B * Cc)
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self._jobs_dev = torch.from_numpy(raw).to(self.dev)
        self._keep.append(self._jobs_dev)
        nf = getattr(self, '_n_fwd_pack', n)
        if not self.PACK_OVERLAP or self.bwd is None or nf in (0, n):
            self.pack.add(PackArgs(jobs_dev=self._jobs_dev.data_ptr(), max_elems=mx, njobs=n))
            return
        # Only what the FIRST forward layers read is packed on the main stream.  The side stream -- forked behind the
        # optimizer step -- packs the large forward layouts of the deep levels (9/10 of the bytes: the forward program waits
        # for them in front of its first deep layer, about a millisecond of shallow-level kernels later) and then the layouts
        # only the backward pass reads (flipped / transposed data-gradient operands; the backward program starts by waiting
        # for those): an HBM-bound kernel next to latency-bound convolutions.
        def mx_of(jobs):
            return max(A * B * Cc for (_, _, A, B, Cc, _, _, _) in jobs)
        jobs = self._pack_jobs
        nd = self._n_deep_pack
        ns = nf - nd                         # (_order_pack_jobs: shallow forward jobs, deep forward jobs, backward jobs)
        sz = C.sizeof(PackJob)
        base = self._jobs_dev.data_ptr()
        self.pack.record(self.EV_PACK_FORK, stream=0)
        self.pack.wait(self.EV_PACK_FORK, stream=1)
        if ns > 0:
            self.pack.add(PackArgs(jobs_dev=base, max_elems=mx_of(jobs[:ns]), njobs=ns))
        if nd > 0:
            self.pack.add(PackArgs(jobs_dev=base + ns * sz, max_elems=mx_of(jobs[ns:nf]), njobs=nd), stream=1)
            self.pack.record(self.EV_PACK_DEEP, stream=1)
        self.pack.add(PackArgs(jobs_dev=base + nf * sz, max_elems=mx_of(jobs[nf:]), njobs=n - nf), stream=1)
        self.pack.record(self.EV_PACK_DONE, stream=1)
        self._bwd_needs_pack_wait = True

    PACK_TILED = os.environ.get('DFL_PACK_TILED', '1') != '0'

    def _finish_pack_tiled(self):
        """One-stream pack (round 4; round 5: every arithmetic, not only bf16 storage): the layouts of parameters that tile into 32 x 32 x C blocks go through
        dfl_pack_weights_tiled -- a flat list of tiles over all jobs, BOTH layouts of a parameter (forward operand and data-gradient
        operand) from one read of the fp32 master --, everything else (the first layer, heads, odd sizes) through dfl_pack_weights."""
        tiled, rest, by_src = [], [], {}
        for job in self._pack_jobs:
            src, dst, A, B, Cc, kind, flip, split = job
            if A % 32 == 0 and B % 32 == 0 and Cc <= 9:
                k = src.data_ptr()
                if k in by_src and len(by_src[k]) == 1:
                    by_src[k].append(job)
                else:
                    by_src[k] = [job]
                    tiled.append(by_src[k])
            else:
                rest.append(job)
        def to_dev(arr):
            t = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.dev)
            self._keep.append(t)
            return t
        self._tiled_host, self._tiled_index = None, None      # what sgd.SGD needs to update the weights inside this launch
        if tiled:
            arr = (PackJob * len(tiled))()
            tiles = 0
            for i, group in enumerate(tiled):
                src, dst, A, B, Cc, kind, flip, split = group[0]
                a = arr[i]
                a.src, a.dst, a.A, a.B, a.C, a.kind, a.flip, a.split = src.data_ptr(), dst.data_ptr(), A, B, Cc, kind, flip, split
                if len(group) > 1:
                    a.dst2, a.kind2, a.flip2, a.split2 = group[1][1].data_ptr(), group[1][5], group[1][6], group[1][7]
                a.first_tile = tiles
                tiles += (A // 32) * (B // 32)
            self._tiled_index = len(self.pack)
            self.pack.add(PackArgs(jobs_dev=to_dev(arr).data_ptr(), max_elems=tiles, njobs=len(tiled), tiled=1))
            srcs = [g[0][0].data_ptr() for g in tiled]
            if len(set(srcs)) == len(srcs):                   # (a parameter with a third layout: its second record would read
                self._tiled_host = (bytes(arr), len(tiled), tiles, {g[0][0].data_ptr(): g[0][0].numel() for g in tiled})   # what the first one writes)
        if rest:
            arr = (PackJob * len(rest))()
            mx = 0
            for i, (src, dst, A, B, Cc, kind, flip, split) in enumerate(rest):
                arr[i].src, arr[i].dst = src.data_ptr(), dst.data_ptr()
                arr[i].A, arr[i].B, arr[i].C, arr[i].kind, arr[i].flip, arr[i].split = A, B, Cc, kind, flip, split
                mx = max(mx, A * B * Cc)
            self.pack.add(PackArgs(jobs_dev=to_dev(arr).data_ptr(), max_elems=mx, njobs=len(rest)))

    DEEP_PACK_ELEMS = 256 * 1024   # forward layouts from this size on are packed on the side stream (0: none)

    def _order_pack_jobs(self):
        """Forward jobs: small ones first, then -- from the first large one in forward order on -- the rest ("deep": packed on
        the side stream); a wait for them goes in front of the first forward op that reads one."""
        nf = self._n_fwd_pack
        self._n_deep_pack = 0
        if not self.PACK_OVERLAP or self.bwd is None or self.DEEP_PACK_ELEMS <= 0 or nf == 0:
            return
        fwd_jobs = self._pack_jobs[:nf]
        first = next((i for i, j in enumerate(fwd_jobs) if j[2] * j[3] * j[4] >= self.DEEP_PACK_ELEMS), None)
        if first is None or first == 0:
            return
        deep = {fwd_jobs[i][1].data_ptr() for i in range(first, nf) if fwd_jobs[i][2] * fwd_jobs[i][3] * fwd_jobs[i][4] >= self.DEEP_PACK_ELEMS}
        # small weights that come later in forward order (decoder levels, heads) stay on the main stream
        shallow = [j for j in fwd_jobs if j[1].data_ptr() not in deep]
        deepj = [j for j in fwd_jobs if j[1].data_ptr() in deep]
        self._pack_jobs[:nf] = shallow + deepj
        self._n_deep_pack = len(deepj)
        for idx, st in enumerate(self.fwd.structs):
            if isinstance(st, ConvArgs) and st.w in deep:
                self.fwd.insert(idx, nat.SyncArgs(event=self.EV_PACK_DEEP), nat.OP_WAIT, 0)
                break

    PACK_OVERLAP = False     # (round 4: off -- with the BatchNorm launches gone the one-stream form is 1.2 % faster)
    WSPLIT = True      # split-bf16 modes: weights split once by the pack kernel
    DSPLIT = True      # ... and the BatchNorm/ReLU backward output split once by its producer
    EV_PACK_FORK, EV_PACK_DONE, EV_PACK_DEEP = 60000, 60001, 60002

    # ------------------------------------------------------------------------------------------ op helpers
    def _conv(self, prog, x, w, y, KH, KW, stride, pad, Ntot, bias=None, in_aff=None, relu=0, add=None,
              add_aff=None, accumulate=0, scatter=0, stats=False, stat_other=None, Hout=None, Wout=None, x_split=0, brb=None,
              in_live=None, add_live=None, stat_totals=None, x_out=None, out_aff=None):
        """in_live / add_live: (totals, gamma, beta, count) of a live BatchNorm instead of the (scale, shift) vectors of in_aff /
        add_aff; stat_totals: the producer adds its statistics there instead of leaving partial rows (include/dfl_hip.h)."""
        a = ConvArgs()
        a.x, a.w, a.y = x.ptr, w.data_ptr(), y.ptr
        a.w_split = self._packed_split.get(w.data_ptr(), 0)
        a.x_split = x_split
        a.x_bf16, a.y_bf16 = x.bf16, y.bf16
        if (add is not None and add.bf16 != y.bf16) or (stat_other is not None and stat_other.bf16 != y.bf16):
            raise PlanError('internal: mixed element types in a convolution epilogue')
        a.bias = nat.ptr(bias)
        if in_aff is not None:
            a.in_scale, a.in_shift = in_aff[0].data_ptr(), in_aff[1].data_ptr()
        if in_live is not None:
            a.in_tot, a.in_gamma, a.in_beta, a.in_count = in_live[0], in_live[1].data_ptr(), in_live[2].data_ptr(), float(in_live[3])
            a.bn_eps = BN_EPS
        if add_live is not None:
            a.add_tot, a.add_gamma, a.add_beta, a.add_count = add_live[0], add_live[1].data_ptr(), add_live[2].data_ptr(), float(add_live[3])
            a.bn_eps = BN_EPS
        if brb is not None:
            # the operand is the BatchNorm + ReLU backward of (x = dy, r) formed while the patch is staged (dfl_conv_args.x_mode)
            r_act, coef = brb
            a.x_mode, a.x2, a.ldx2 = 1, r_act.ptr, r_act.ld
            if isinstance(coef, dict):               # live statistics: the kernel derives A, B, C from the totals itself
                a.in_tot, a.in_gamma, a.in_mean, a.in_invstd = coef['tot'], coef['gamma'].data_ptr(), coef['mean'].data_ptr(), coef['invstd'].data_ptr()
                a.in_count = float(coef['count'])
            else:
                a.in_scale = nat.ptr(coef)
            if x_out is not None:
                a.x_out, a.ldxo = x_out.ptr, x_out.ld
        if add is not None:
            a.add, a.ldadd = add.ptr, add.ld
            if add_aff is not None:
                a.add_scale, a.add_shift = add_aff[0].data_ptr(), add_aff[1].data_ptr()
        a.N, a.Hin, a.Win, a.Cin, a.ldx = x.N, x.H, x.W, x.C, x.ld
        a.KH, a.KW, a.stride, a.pad = KH, KW, stride, pad
        a.Hout, a.Wout = (y.H, y.W) if Hout is None else (Hout, Wout)
        a.Ntot, a.ldy = Ntot, y.ld
        a.relu, a.accumulate, a.scatter2x2 = relu, accumulate, scatter
        # inference plans ask for the latency form of the bf16 convolution (include/dfl_hip.h: a hint the library honours for the
        # small problems of a batch-1 forward -- the per-image loops of util.py -- and ignores for everything else)
        a.latency_form = 1 if (not self.training and not self.need_grad and os.environ.get('DFL_PLAN_LATENCY_FORM', '1') != '0') else 0
        self._out_aff_taken = False
        if out_aff is not None:
            # the consumer's eval-mode BatchNorm applied by this (producing) kernel: only the latency form does that
            a.out_scale, a.out_shift = out_aff[0].data_ptr(), out_aff[1].data_ptr()
            if self.lib.dfl_conv_config(C.addressof(a)) == 16 + nat.CONV_CFG_LATENCY:
                self._out_aff_taken = True
            else:
                a.out_scale, a.out_shift = None, None
        sp = nat.check(self.lib.dfl_conv_suggest_splits(C.addressof(a)), 'dfl_conv_suggest_splits')
        if sp > 1:
            M = x.N * (x.H * x.W if scatter else a.Hout * a.Wout)
            a.splits = sp
            a.partial = self._shared_scratch('conv_partial', sp * M * Ntot).data_ptr()
        partials = None
        if stat_totals is not None:
            a.stat_totals = stat_totals
            if stat_other is not None:
                a.stat_other, a.ldso = stat_other.ptr, stat_other.ld
            partials = ('live', stat_totals)
        elif stats:
            gm = nat.check(self.lib.dfl_conv_grid_m(C.addressof(a)), 'dfl_conv_grid_m')
            partials = self._new(gm * 2 * Ntot)
            a.stat_partials = partials.data_ptr()
            if stat_other is not None:
                a.stat_other, a.ldso = stat_other.ptr, stat_other.ld
            partials = (partials, 4 * gm if scatter else gm)    # scatter2x2: rows = (row block, 2x2 position), Ntot / 4 columns
        prog.add(a)
        return partials

    # A layer's weight gradient and data gradient both read dpre and are independent: for all but the largest layers
    # neither fills the 256 CUs on its own at batch 16, so the weight gradient goes to the library's side stream
    # (fork after dpre is written; joined before dpre's buffer is rewritten two layers later, before a batched sum
    # reads its slices, and at the end of backward).  Measured: +10 % on an isolated pair for 24x24..96x96 layers, -8 % at
    # 192x192 -- but nothing inside the real backward pass (both kernels just run slower side by side), so it is off by default.
    OUT_AFF = True          # inference: a block's inner BatchNorm applied by the producing kernel (measured: affine on load +3 us per 3x3 layer)
    PAIRS = True            # inference: a block's last 3x3 convolution and its 1x1 as one launch (measured: -4.5 us per block end)
    SIDE_STREAM = False     # off: inside the whole backward pass the pair runs no faster (r01)
    SIDE_MAX_PIXELS = 16 * 96 * 96

    def _side_join(self, prog, buf=None):
        """Main stream waits for the side-stream weight gradients emitted so far (all, or those reading buffer `buf`)."""
        todo = [e for e in self._side if e['waited'] is None and (buf is None or e['buf'] == buf)]
        for e in todo:
            e['waited'] = len(prog.structs)
            prog.wait(e['done'], stream=0)

    def _wgrad(self, prog, g, d, dw, KH, KW, stride, pad, Hout, Wout, in_aff=None, side=False, side_buf=None, d_split=0, brb=None,
               bias_out=None, bias_plain=False):
        """bias_plain: d is the operand itself (materialised through dfl_conv_args.x_out) and the kernel still leaves its column
        sums -- the bias gradient -- per pixel slice."""
        a = WgradArgs()
        a.d_split = d_split
        a.g_bf16, a.d_bf16 = g.bf16, d.bf16
        a.g, a.d, a.dw = g.ptr, d.ptr, dw.data_ptr()
        if in_aff is not None:
            a.in_scale, a.in_shift = in_aff[0].data_ptr(), in_aff[1].data_ptr()
        a.N, a.Hin, a.Win, a.Cg, a.ldg = g.N, g.H, g.W, g.C, g.ld
        a.KH, a.KW, a.stride, a.pad = KH, KW, stride, pad
        a.Hout, a.Wout, a.Cm, a.ldd = Hout, Wout, d.C, d.ld
        if brb is not None:
            r_act, coef = brb                       # d = dy; the operand is formed from (dy, r) while the patch is staged (d_mode)
            a.d_mode, a.d2, a.ldd2 = 1, r_act.ptr, r_act.ld
            if isinstance(coef, dict):               # live statistics (include/dfl_hip.h)
                a.coef_tot, a.bn_gamma, a.bn_mean, a.bn_invstd = coef['tot'], coef['gamma'].data_ptr(), coef['mean'].data_ptr(), coef['invstd'].data_ptr()
                a.bn_count = float(coef['count'])
            else:
                a.coef = nat.ptr(coef)
        a.splits = 1
        s = nat.check(self.lib.dfl_wgrad_suggest_splits(C.addressof(a)), 'dfl_wgrad_suggest_splits')
        a.splits = s
        bias_job = None
        if (brb is not None or bias_plain) and bias_out is not None:
            # the column sums of that operand = the layer's bias gradient, one row per pixel slice
            bpart = self._new(s * d.C)
            a.bias_partial = bpart.data_ptr()
            bias_job = (bpart.data_ptr(), bias_out.data_ptr(), d.C, d.C, s, 1)    # queued BEHIND the launch that writes the rows (below): a
                                                                                  # batched sum flushed from here would read them unwritten
        n = d.C * g.C * KH * KW
        if s > 1:
            big = 4 * n >= self.FLUSH_BYTES
            part = self._wg_partial(s * n) if big else self._new(s * n)   # deferred sums keep their own slices
            a.partial = part.data_ptr()
        if side:
            k = len(self._side)
            prog.record(2 * k, stream=0)            # dpre is complete on the main stream
            prog.wait(2 * k, stream=1)
            self._side.append(dict(done=2 * k + 1, op=len(prog.structs), waited=None, buf=side_buf))
            prog.add(a, stream=1)
            prog.record(2 * k + 1, stream=1)
        else:
            prog.add(a)
        if bias_job is not None:
            self._red_pending.append(bias_job)
            self._red_bytes += 4 * d.C
        if s > 1:
            # the caller flushes (self._maybe_flush) once the main-stream work that may overlap has been emitted
            self._red_pending.append((part.data_ptr(), dw.data_ptr(), n, n, s, KH * KW))
            self._red_bytes += 4 * n
            if not side:
                self._maybe_flush(prog)

    # Small sums (bias gradients, pixel-slice partials of narrow layers) are not launched one by one: they queue up
    # and one dfl_reduce_batch finishes the queue once FLUSH_BYTES of gradient are waiting (and at the end of backward).
    FLUSH_BYTES = int(float(os.environ.get('DFL_FLUSH_MB', '16')) * (1 << 20))   # (4 / 16 / 64 MB measured: 0.414 / 0.373 / 0.367 ms of sums per step)
    FUSE_BWD_STATS = True
    FUSE_DOWN_STATS = os.environ.get('DFL_FUSE_DOWN_STATS', '1') != '0'   # ... and the strided-conv data gradient's scatter (bf16)
    FUSE_BRB = os.environ.get('DFL_FUSE_BRB', '1') != '0'      # BatchNorm + ReLU backward inside the weight-/data-gradient staging (bf16 storage)
    # the data-gradient kernel also WRITES the operand it forms (dfl_conv_args.x_out) for layers whose tensor is at most this large:
    # the weight gradient then reads one plain tensor instead of forming the operand again in each of its (cm, cg) tiles
    DPRE_OUT_BYTES = int(float(os.environ.get('DFL_DPRE_OUT_MB', '12')) * (1 << 20))
    RES_DGRAD_LAST = True   # residual 1x1 data gradient accumulates onto the 3x3 one (not the reverse)
    FUSE_COLSUMS = True   # sums across block boundaries (see the backward program)

    def _defer_sum(self, prog, src, dst, n, stride, count, T=1):
        self._red_pending.append((src, dst, n, stride, count, T))
        self._red_bytes += 4 * n
        self._maybe_flush(prog)

    def _maybe_flush(self, prog):
        if self._red_bytes >= self.FLUSH_BYTES:
            self._flush_sums(prog)

    def _flush_sums(self, prog):
        self._flush_live_bwd(prog)
        jobs = self._red_pending
        if not jobs:
            return
        self._side_join(prog)       # the slices of side-stream weight gradients must be complete
        arr = (ReduceJob * len(jobs))()
        blocks = 0
        for i, (src, dst, n, stride, count, T) in enumerate(jobs):
            arr[i].src, arr[i].dst, arr[i].n, arr[i].stride, arr[i].count, arr[i].first_block = src, dst, n, stride, count, blocks
            arr[i].T = T
            blocks += nat.check(self.lib.dfl_reduce_job_blocks(n, count), 'dfl_reduce_job_blocks')
        dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.dev)
        self._keep.append(dev)
        self._red_flushes.append((len(prog.structs), [j[1] for j in jobs]))
        self.partial_sum_bytes += sum(4 * n * count for (_, _, n, _, count, _) in jobs)
        prog.add(ReduceBatchArgs(jobs_dev=dev.data_ptr(), njobs=len(jobs), total_blocks=blocks))
        self._red_pending, self._red_bytes = [], 0

    def _dz_for(self, g, N, H, W, Cc, fused):
        """Scratch for the data gradient a layer hands to the layer below.  With the BatchNorm + ReLU backward folded into the
        data-gradient kernel that kernel READS the incoming gradient g while it writes: never the same buffer -- 'dz' and 'dpre0'
        (idle in that mode) alternate."""
        key = 'dz'
        if fused and g.t is self._scratch.get('dz'):
            key = 'dpre0'
        if fused:
            self._side_join(self.bwd, buf=key)      # side-stream weight gradients that still read this scratch
        return self._scratch_act(key, N, H, W, Cc)

    LIVE_BN = os.environ.get('DFL_LIVE_BN', '1') != '0'     # BatchNorm statistics completed by their consumers (bf16 patch kernels)
    LIVE_HEAD = os.environ.get('DFL_LIVE_HEAD', '1') != '0'   # ... and the head's backward kernel adds the sums of its dx itself (no colstats pass)
    BN_R = 8                                                # include/dfl_hip.h: DFL_BN_R

    def _bn_totals(self, Cc, bwd=False):
        """Address of [BN_R][2][C] doubles inside the plan's totals arena of the forward (backward) pass, zeroed by one memset at
        the start of that program."""
        need = self.BN_R * 2 * Cc
        ar = self._tot_bwd if bwd else self._tot_fwd
        if ar[0] is None:
            cfg = self.cfg
            chans = [2 ** (cfg['wf'] + i) for i in range(cfg['depth'])]
            ar[0] = self._new(sum(chans) * cfg['block_depth'] * 2 * self.BN_R * 2, torch.float64)
        assert ar[1] + need <= ar[0].numel()
        ptr = ar[0].data_ptr() + 8 * ar[1]
        ar[1] += need
        return ptr

    def _flush_live_bwd(self, prog):
        """One batched launch for the BatchNorm parameter gradients (and residual bias gradients) of the live layers queued so far."""
        jobs = self._live_bwd_pending
        if not jobs:
            return
        arr = (BnBwdLiveJob * len(jobs))()
        for i, j in enumerate(jobs):
            a = arr[i]
            a.totals, a.save_mean, a.save_invstd = j['tot'], j['mean'].data_ptr(), j['invstd'].data_ptr()
            a.dgamma, a.dbeta, a.sum_out, a.C = j['dgamma'].data_ptr(), j['dbeta'].data_ptr(), nat.ptr(j['sum_out']), j['C']
        dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.dev)
        self._keep.append(dev)
        dsts = [j[k].data_ptr() for j in jobs for k in ('dgamma', 'dbeta', 'sum_out') if j[k] is not None]
        self._red_flushes.append((len(prog.structs), dsts))
        prog.add(BnBwdLiveArgs(jobs_dev=dev.data_ptr(), njobs=len(jobs), max_C=max(j['C'] for j in jobs)))
        self._live_bwd_pending = []

    def _finish_live_bn(self, fwd, index=None):
        """The zero fill in front of the forward program and the one batched finalize behind its last BatchNorm layer."""
        if not self._live_jobs:
            return
        arr = (BnLiveJob * len(self._live_jobs))()
        for i, j in enumerate(self._live_jobs):
            a = arr[i]
            a.totals, a.gamma, a.beta = j['totals'], j['gamma'].data_ptr(), j['beta'].data_ptr()
            a.running_mean = self.Bf[j['bname'] + '.running_mean'].data_ptr()
            a.running_var = self.Bf[j['bname'] + '.running_var'].data_ptr()
            a.num_batches_tracked = self.Bf[j['bname'] + '.num_batches_tracked'].data_ptr()
            a.scale, a.shift, a.save_mean, a.save_invstd = (j[k].data_ptr() for k in ('scale', 'shift', 'mean', 'invstd'))
            a.count, a.C, a.eps, a.momentum = j['count'], j['C'], BN_EPS, BN_MOMENTUM
        dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.dev)
        self._keep.append(dev)
        fwd.add(BnLiveArgs(jobs_dev=dev.data_ptr(), njobs=len(self._live_jobs), max_C=max(j['C'] for j in self._live_jobs)))
        fwd.insert(0, MemsetArgs(ptr=self._tot_fwd[0].data_ptr(), bytes=8 * self._tot_fwd[1]))

    def _shared_scratch(self, key, nelem):
        t = self._scratch.get(key)
        if t is None or t.numel() < nelem:
            t = self._new(nelem)
            self._scratch[key] = t
        return t

    def _wg_partial(self, nelem):
        t = self._scratch.get('wg_partial')
        if t is None or t.numel() < nelem:
            t = self._new(nelem)
            self._scratch['wg_partial'] = t
        return t

    def _colsum(self, prog, a, out):
        """out[c] = sum over pixels of a[:, c]."""
        nb = self.lib.dfl_rowblock_count(a.M, a.C)
        part = self._new(nb * 2 * a.C)
        prog.add(ColstatsArgs(a=a.ptr, b=None, partials=part.data_ptr(), M=a.M, C=a.C, lda=a.ld, ldb=0, nblocks=nb,
                              bf16=a.bf16))
        self._defer_sum(prog, part.data_ptr(), out.data_ptr(), a.C, 2 * a.C, nb)

    # ------------------------------------------------------------------------------------------ circular padding
    # pad_mode='circular' (unet.py:211-212: nn.Conv2d(..., padding=1, padding_mode='circular')).  No reference CLI selects
    # it, so it is built from what exists rather than as a third addressing mode of every kernel: the input is copied into a
    # tensor with a one-pixel wrapped frame (9 window copies) and the convolution runs UNPADDED on that; backward, the data
    # gradient comes out on the framed grid (a "full" correlation) and the frame is folded back onto the pixels it copies
    # (9 window copies, 8 of them accumulating).  BatchNorm-on-load keeps working: every framed pixel is a real pixel.
    _WRAP = ((-1, 0, 1), (0, 1, None), (0, None, 1))      # (source start or -1 = last, destination start or None = last, extent or None = all)

    def _wrap_regions(self, H, W):
        out = []
        for sy, dy, h in ((H - 1, 0, 1), (0, 1, H), (0, H + 1, 1)):
            for sx, dx, w in ((W - 1, 0, 1), (0, 1, W), (0, W + 1, 1)):
                out.append((sy, sx, dy, dx, h, w))
        return out

    def _wrap_pad(self, prog, x):
        """Act [N, H+2, W+2, C] holding x with a wrapped one-pixel frame (raw values: an affine-on-load still applies)."""
        n = x.N * (x.H + 2) * (x.W + 2) * x.C
        t = self._new(n, torch.bfloat16 if x.esz == 2 else torch.float32)
        xp = Act(t, t.data_ptr(), x.C, x.N, x.H + 2, x.W + 2, x.C, x.esz)
        for sy, sx, dy, dx, h, w in self._wrap_regions(x.H, x.W):
            prog.add(AffineCopyArgs(x=x.ptr, y=xp.ptr, N=x.N, H=h, W=w, C=x.C, ldx=x.ld, xH=x.H, xW=x.W, xoy=sy, xox=sx,
                                    ldy=xp.ld, yH=xp.H, yW=xp.W, yoy=dy, yox=dx, bf16=x.bf16))
        return xp

    def _wrap_fold(self, prog, dxp, dx, accumulate=0):
        """dx (+)= the framed gradient dxp folded back: the adjoint of _wrap_pad."""
        regs = self._wrap_regions(dx.H, dx.W)
        regs.sort(key=lambda r: 0 if (r[4] == dx.H and r[5] == dx.W) else 1)     # the interior first: it may overwrite
        for i, (sy, sx, dy, dx_, h, w) in enumerate(regs):
            prog.add(AffineCopyArgs(x=dxp.ptr, y=dx.ptr, N=dx.N, H=h, W=w, C=dx.C, ldx=dxp.ld, xH=dxp.H, xW=dxp.W, xoy=dy, xox=dx_,
                                    ldy=dx.ld, yH=dx.H, yW=dx.W, yoy=sy, yox=sx, accumulate=(accumulate if i == 0 else 1), bf16=dx.bf16))

    def _framed_grad(self, like):
        """Scratch Act for a data gradient on the framed grid of `like` ([N, H+2, W+2, C])."""
        n = like.N * (like.H + 2) * (like.W + 2) * like.C
        t = self._new(n, self.adt)
        return Act(t, t.data_ptr(), like.C, like.N, like.H + 2, like.W + 2, like.C, self.aesz)

    # ------------------------------------------------------------------------------------------ build
    def _build(self):
        cfg = self.cfg
        depth, wf = cfg['depth'], cfg['wf']
        pad = 1 if cfg['padding'] else 0
        bd = cfg['block_depth']
        bn = cfg['batch_norm']
        do_res = cfg['do_res']
        circ = bool(pad) and cfg.get('pad_mode', 'zeros') == 'circular'
        upsample = cfg.get('up_mode', 'upconv') == 'upsample'
        self.circular = circ
        N = self.N
        chans = [2 ** (wf + i) for i in range(depth)]
        for c in chans:
            if c % 4 != 0:
                raise PlanError('channel counts must be multiples of 4 (wf >= 2)')
            if self.bf16 and c % 16 != 0:
                raise PlanError('math mode 4 (bf16 storage) needs channel counts that are multiples of 16 (wf >= 4); this '
                                'network has a %d-channel level' % c)
        shrink = 0 if pad else 2 * bd
        # spatial size of every level (input / output of the block), down then up
        hin, win = [self.H], [self.W]
        hout, wout = [], []
        for i in range(depth):
            ho, wo = hin[i] - shrink, win[i] - shrink
            if ho < 1 or wo < 1:
                raise PlanError('input %dx%d too small for this architecture' % (self.H, self.W))
            hout.append(ho)
            wout.append(wo)
            if i != depth - 1:
                if ho < 2 or wo < 2:
                    raise PlanError('input %dx%d too small for this architecture' % (self.H, self.W))
                hin.append(ho // 2)
                win.append(wo // 2)
        if do_res and shrink != 0:
            raise PlanError('do_res=True needs padding=True: the 1x1 residual and the unpadded 3x3 stack differ in '
                            'size (the reference fails at unet.py:231 for the same reason)')

        # backward scratch: sized for the largest [M][C] any block produces
        if self.need_grad:
            mx = 0
            h, w = self.H, self.W
            sizes = []
            for i in range(depth):
                sizes.append((N * hin[i] * win[i], chans[i]))
            uh, uw = hout[depth - 1], wout[depth - 1]
            for i in reversed(range(depth - 1)):
                uh, uw = 2 * uh, 2 * uw
                sizes.append((N * uh * uw, chans[i]))
                uh, uw = uh - shrink, uw - shrink
            mx = max(m * c for m, c in sizes)
            self._scratch['dpre0'] = self._new(mx, self.adt)
            self._scratch['dpre1'] = self._new(mx, self.adt)
            self._dpre_turn = 0
            self._scratch['dz'] = self._new(mx, self.adt)
            if self.bf16 and self.FUSE_BRB and self.DPRE_OUT_BYTES > 0:
                self._scratch['dmat'] = self._new(mx, self.adt)   # operand materialised by the data-gradient kernel (x_out)

        fwd, bwd = self.fwd, self.bwd
        Cin0 = cfg['in_channels']
        if self.bf16 and Cin0 > 4:
            # multi-channel inputs enter the patch-resident kernels as bf16 (the 1..4-channel case keeps the fp32 image:
            # the direct first-layer kernels read it as is)
            if Cin0 % 16 != 0:
                raise PlanError('math mode 4 (bf16 storage) takes 1..4 or a multiple of 16 input channels')
            self.x_in = self._new(N * self.H * self.W * Cin0, torch.bfloat16)
            x0 = Act(self.x_in, self.x_in.data_ptr(), Cin0, N, self.H, self.W, Cin0, 2)
        else:
            self.x_in = self._new(N * self.H * self.W * Cin0)
            x0 = Act(self.x_in, self.x_in.data_ptr(), Cin0, N, self.H, self.W, Cin0)

        # cat buffers for the up path are allocated up front so that down blocks can write their bridge half directly
        up_hw = []           # size of the up-sampled tensor at level i (i = depth-2 .. 0)
        uh, uw = hout[depth - 1], wout[depth - 1]
        for i in reversed(range(depth - 1)):
            uh, uw = 2 * uh, 2 * uw
            up_hw.append((uh, uw))
            uh, uw = uh - shrink, uw - shrink
        self.last_hw = (uh, uw)
        cat = {}
        dcat = {}
        direct_bridge = {}
        for j, i in enumerate(reversed(range(depth - 1))):
            ch, cw = up_hw[j]
            cat[i] = self._act(N, ch, cw, 2 * chans[i])
            direct_bridge[i] = (ch == hout[i] and cw == wout[i])
            if self.need_grad:
                dcat[i] = self._act(N, ch, cw, 2 * chans[i])

        bwd_stages = []      # closures appended in forward order, executed reversed

        # ------------------------------------------------------------------ one residual conv block
        def block(prefix, xin, out, first=False):
            """Forward ops of UNetConvBlock `prefix` reading xin, writing out; returns a backward emitter."""
            P = self.P
            Cout = out.C
            Hb, Wb = xin.H - shrink, xin.W - shrink
            convs = []
            cur, cur_aff, cur_live = xin, None, None
            step = 3 if bn else 2
            patch_in = lambda t: bool(t.bf16) and t.C % 16 == 0            # a convolution reading t runs the bf16 patch kernels
            for d in range(bd):
                wname = '%s.block.%d' % (prefix, d * step)
                w, b = P[wname + '.weight'], P[wname + '.bias']
                wp = self._pack_conv_fwd(w)
                Ho, Wo = cur.H - (0 if pad else 2), cur.W - (0 if pad else 2)
                r = self._act(N, Ho, Wo, Cout)
                gin = self._wrap_pad(fwd, cur) if circ else cur          # what the convolution (and its weight gradient) gathers from
                # Live statistics (round 4): this layer's BatchNorm is not finalised by a launch between producer and consumer --
                # the producing patch kernel adds its sums to the layer's totals, the consuming patch kernel (next 3x3, or the
                # residual 1x1 whose epilogue adds BN(r)) derives scale / shift itself; ONE batched launch at the end of the forward
                # pass leaves the vectors the backward pass and the module state need.  Both kernels must be patch kernels.
                # (the network's first block: its 1-channel layers run the direct kernels, which take part as well -- the 3x3 row
                # form as a producer, the 1x1 form as the consumer of "+ BN(r)", the 3x3 weight gradient as a consumer of A, B, C)
                one_ch = lambda t: (not t.bf16) and t.C == 1 and Cout in (8, 16, 32, 64) and pad == 1
                live = (self.LIVE_BN and bn and self.training and self.bf16 and not circ and (patch_in(gin) or one_ch(gin))
                        and (d < bd - 1 or (do_res and (patch_in(xin) or one_ch(xin)))))
                # (round 5) fp32 tensors: the GEMM kernels take part as well -- producer: any of them (statistics tail, K-slice finish
                # kernel); consumers: the fast gather of the next 3x3 (its scale / shift table in LDS) and the residual 1x1's "+ BN(r)"
                # (the consumer's table form needs the FAST gather, which needs tensors below 2 GiB -- a larger activation keeps
                # the finalize launch and the generic kernel instead of failing in dfl_conv2d)
                small = lambda t: t.N * t.H * t.W * t.ld * 4 < (1 << 31) - 4096
                gemm_in = lambda t: (not t.bf16) and t.C % 16 == 0 and small(t)
                if (self.LIVE_BN and bn and self.training and not self.bf16 and not circ and gemm_in(gin) and Cout % 16 == 0
                        and gin.N * gin.H * gin.W * 2 * Cout * 4 < (1 << 31) - 4096 and (d < bd - 1 or do_res)):
                    live = True
                tot = self._bn_totals(Cout) if live else None
                live_bwd = (self.LIVE_BN and bn and self.training and self.bf16 and self.FUSE_BRB and self.FUSE_BWD_STATS and not circ
                            and self.need_grad and (patch_in(cur) or (one_ch(cur) and d == 0 and first and not self.input_grad)))
                # (this layer's backward can take live (sum dy, sum dy*r): its consumers derive the coefficients)
                # inference: the BatchNorm between this convolution and the block's next one is applied by THIS kernel where it runs in
                # latency form (dfl_conv_args.out_scale: the consumer's two roundings, bit for bit) -- the consumer then reads its operand
                # plain instead of converting 8 channels per lane and k-step on its way into the matrix instruction
                pre_aff = None
                if (bn and self.OUT_AFF and not self.training and not self.need_grad and not circ and pad == 1 and d < bd - 1):
                    pre_aff = (self._new(Cout), self._new(Cout))
                part = self._conv(fwd, gin, wp, r, 3, 3, 1, 0 if circ else pad, Cout, bias=b,
                                   in_aff=None if cur_live is not None else cur_aff, in_live=cur_live, relu=1,
                                   stats=bn and self.training, stat_totals=tot, out_aff=pre_aff)
                out_aff_taken = pre_aff is not None and self._out_aff_taken
                aff = None
                bnrec = None
                if bn:
                    bname = '%s.block.%d' % (prefix, d * step + 2)
                    gamma, beta = P[bname + '.weight'], P[bname + '.bias']
                    scale, shift = pre_aff if out_aff_taken else (self._new(Cout), self._new(Cout))
                    mean, invstd = self._new(Cout), self._new(Cout)
                    if self.training and live:
                        self._live_jobs.append(dict(totals=tot, gamma=gamma, beta=beta, bname=bname, scale=scale, shift=shift, mean=mean,
                                                    invstd=invstd, count=N * Ho * Wo, C=Cout))
                    elif self.training:
                        fa = BnFinalizeArgs(partials=part[0].data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(),
                                            running_mean=self.Bf[bname + '.running_mean'].data_ptr(),
                                            running_var=self.Bf[bname + '.running_var'].data_ptr(),
                                            num_batches_tracked=self.Bf[bname + '.num_batches_tracked'].data_ptr(),
                                            scale=scale.data_ptr(), shift=shift.data_ptr(),
                                            save_mean=mean.data_ptr(), save_invstd=invstd.data_ptr(),
                                            count=N * Ho * Wo, nblocks=part[1], C=Cout, eps=BN_EPS, momentum=BN_MOMENTUM)
                        fwd.add(fa)
                    else:
                        # eval mode: scale / shift depend on parameters and running statistics only -- constants between two
                        # forwards of an inference loop (util.test_dataset, seg_dataset_ensemble: one image after the other).
                        # Inference plans compute them in the PACK program (re-run when a parameter, a buffer or a training
                        # forward has touched them, UNet._ensure_packed) instead of 22 launches in every forward (round 4).
                        (fwd if self.need_grad else self._prep).add(
                            BnEvalArgs(gamma=gamma.data_ptr(), beta=beta.data_ptr(),
                                       running_mean=self.Bf[bname + '.running_mean'].data_ptr(),
                                       running_var=self.Bf[bname + '.running_var'].data_ptr(),
                                       scale=scale.data_ptr(), shift=shift.data_ptr(), C=Cout, eps=BN_EPS))
                        if self.need_grad:
                            # gradients through eval-mode BatchNorm (nn.Module semantics of unet.py:161-193): the statistics
                            # are the running ones -- mean is the buffer itself, 1/sqrt(var + eps) is the scale of (gamma = 1)
                            ones, zeros = self._const(Cout, 1.0), self._const(Cout, 0.0)
                            mean = self.Bf[bname + '.running_mean']
                            fwd.add(BnEvalArgs(gamma=ones.data_ptr(), beta=zeros.data_ptr(), running_mean=mean.data_ptr(),
                                               running_var=self.Bf[bname + '.running_var'].data_ptr(),
                                               scale=invstd.data_ptr(), shift=self._new(Cout).data_ptr(), C=Cout, eps=BN_EPS))
                    aff = (scale, shift)
                    bnrec = (gamma, mean, invstd, bname)
                    self.dbg['bn:' + bname] = (scale, shift, mean, invstd)
                convs.append(dict(w=w, wname=wname, inp=cur, gin=gin, inp_aff=cur_aff, r=r, bn=bnrec, live_bwd=live_bwd))
                self.relu_out['%s.block.%d' % (prefix, d * step + 1)] = r      # (module name of the nn.ReLU: tests read its mask)
                cur, cur_aff = r, (None if out_aff_taken else aff)       # (out_aff_taken: r holds BN(ReLU(.)) already)
                cur_live = (tot, gamma, beta, N * Ho * Wo) if (bn and self.training and live) else None
            assert cur.H == Hb and cur.W == Wb
            if do_res:
                rw, rb = P[prefix + '.res_conv1x1.weight'], P[prefix + '.res_conv1x1.bias']
                rwp = self._pack_conv_fwd(rw)
                self._conv(fwd, xin, rwp, out, 1, 1, 1, 0, Cout, bias=rb, add=cur, add_aff=None if cur_live is not None else cur_aff,
                           add_live=cur_live)
                # inference: the block's last 3x3 convolution and this 1x1 run as ONE launch where the library can (dfl_conv2d_pair:
                # the latency form of a batch-1 forward -- 11 launches less in its chain of 44)
                pk = 0
                if self.PAIRS and not self.training and not self.need_grad and len(fwd) >= 2 and isinstance(fwd.structs[-2], ConvArgs):
                    pk = self.lib.dfl_conv_pair_ok(C.addressof(fwd.structs[-2]), C.addressof(fwd.structs[-1]))
                if pk in (1, 2):
                    b_ = fwd.pop()
                    a_ = fwd.pop()
                    if pk == 2:        # K-sliced: the partial sums of BOTH products go through a buffer of twice the size
                        Mp = a_.N * a_.Hout * a_.Wout
                        a_.partial = self._shared_scratch('pair_partial', 2 * a_.splits * Mp * a_.Ntot).data_ptr()
                    fwd.keep += [a_, b_]
                    fwd.add(nat.ConvPairArgs(a=C.addressof(a_), b=C.addressof(b_)))
            else:
                a = AffineCopyArgs(x=cur.ptr, y=out.ptr, N=N, H=Hb, W=Wb, C=Cout, ldx=cur.ld, xH=Hb, xW=Wb,
                                   ldy=out.ld, yH=out.H, yW=out.W, bf16=cur.bf16)
                if cur_aff is not None:
                    a.scale, a.shift = cur_aff[0].data_ptr(), cur_aff[1].data_ptr()
                fwd.add(a)

            self.dbg['out:' + prefix] = out
            self.dbg['xin:' + prefix] = xin

            def backward(dout, dxin, fused_in=None, dxin_stats=False):
                """dout: Act with d(loss)/d(block output); dxin: Act to receive d/d(xin) (None for the net input).
                fused_in: (partials, rows) with sum(dout), sum(dout * r_last) per channel when the kernel that wrote dout
                already left them (saves this block's first statistics pass).  dxin_stats: let the LAST kernel that
                writes dxin leave its column sums; returns them as (partials, rows) -- the caller's bias gradient."""
                G = self.G
                self.dbg['dout:' + prefix] = dout
                if dxin is not None:
                    self.dbg['dxin:' + prefix] = dxin
                dxin_part = None
                if do_res:
                    self._wgrad(bwd, xin, dout, G[prefix + '.res_conv1x1.weight'], 1, 1, 1, 0, xin.H, xin.W)
                # dxin = (3x3 data gradient of the first conv) + (1x1 data gradient of the residual branch).  The 3x3 one
                # writes, the cheap 1x1 one accumulates on top: the big kernel keeps its simple epilogue (and, on the wide
                # levels, the row-tiled form), the column sums of the finished dxin come from the 1x1 kernel.
                res_dgrad_last = do_res and dxin is not None and self.RES_DGRAD_LAST
                if do_res and dxin is not None and not res_dgrad_last:
                    self._conv(bwd, dout, self._pack_conv_dgrad(rw), dxin, 1, 1, 1, 0, xin.C)
                g = dout
                fused = fused_in if self.FUSE_COLSUMS else None   # (partials, rows): BN-backward sums of g already left by g's producer
                for d in reversed(range(bd)):
                    cv = convs[d]
                    r = cv['r']
                    side = self.SIDE_STREAM and r.M <= self.SIDE_MAX_PIXELS
                    # two dpre buffers alternate, so a side-stream weight gradient may still read one while the next
                    # layer fills the other; the one about to be rewritten must be free
                    self._dpre_turn ^= 1
                    if g.t is self._scratch.get('dpre%d' % self._dpre_turn):
                        self._dpre_turn ^= 1              # never the buffer the incoming gradient lives in (mixed fused / plain layers)
                    self._side_join(bwd, buf='dpre%d' % self._dpre_turn)
                    dpre = self._scratch_act('dpre%d' % self._dpre_turn, N, r.H, r.W, Cout)
                    nb = self.lib.dfl_rowblock_count(r.M, Cout)
                    coef = None
                    self.dbg['g:%s.block.%d' % (prefix, d * step + 1)] = g
                    if cv['bn'] is not None:
                        gamma, mean, invstd, bname = cv['bn']
                        if fused is not None and fused[0] == 'live':
                            # live statistics: the producer of g added (sum g, sum g*r) to this layer's totals; the two consumers
                            # below derive A, B, C themselves, the parameter gradients leave with the next batched launch
                            assert cv['live_bwd']
                            coef = dict(tot=fused[1], gamma=gamma, mean=mean, invstd=invstd, count=r.M)
                            self._live_bwd_pending.append(dict(tot=fused[1], mean=mean, invstd=invstd, dgamma=G[bname + '.weight'],
                                                               dbeta=G[bname + '.bias'], C=Cout,
                                                               sum_out=G[prefix + '.res_conv1x1.bias'] if (do_res and d == bd - 1) else None))
                        else:
                            if fused is not None:
                                part, prow = fused
                            else:
                                part, prow = self._new(nb * 2 * Cout), nb
                                bwd.add(ColstatsArgs(a=g.ptr, b=r.ptr, partials=part.data_ptr(), M=r.M, C=Cout, lda=g.ld,
                                                     ldb=r.ld, nblocks=nb, bf16=r.bf16))
                            coef = self._new(3 * Cout)
                            self.dbg['coef:%s.block.%d' % (prefix, d * step + 1)] = coef
                            bwd.add(BnBwdFinalizeArgs(partials=part.data_ptr(), gamma=gamma.data_ptr(),
                                                      save_mean=mean.data_ptr(), save_invstd=invstd.data_ptr(),
                                                      dgamma=G[bname + '.weight'].data_ptr(),
                                                      dbeta=G[bname + '.bias'].data_ptr(), coef=coef.data_ptr(),
                                                      count=r.M if self.training else 0,      # 0: fixed (running) statistics
                                                      nblocks=prow, C=Cout))
                            if do_res and d == bd - 1:
                                # residual bias gradient = column sums of dout, already in the same partials
                                self._defer_sum(bwd, part.data_ptr(), G[prefix + '.res_conv1x1.bias'].data_ptr(), Cout,
                                                2 * Cout, prow)
                    elif do_res and d == bd - 1:
                        self._colsum(bwd, g, G[prefix + '.res_conv1x1.bias'])
                    inp = cv['inp']
                    # BatchNorm + ReLU backward folded into its two consumers (bf16 patch kernels; round 3): the weight-gradient
                    # and data-gradient kernels form [r > 0] * (A dy + B r + C) from (dy, r) while they stage their patches --
                    # dpre is never written (one tensor pass and one launch per layer less), the bias-gradient sums come out
                    # of the weight-gradient kernel.  The 1-channel first layer (direct kernels) keeps the materialised form.
                    fuse_brb = self.FUSE_BRB and bool(r.bf16) and bool(inp.bf16) and inp.C % 16 == 0 and g.bf16 == r.bf16
                    # ... and the network's first layer (1-channel fp32 input, no data gradient): the row form of the direct
                    # weight-gradient kernel does the same while it reads its d rows
                    if (self.FUSE_BRB and bool(r.bf16) and g.bf16 == r.bf16 and not inp.bf16 and inp.C == 1
                            and d == 0 and dxin is None and Cout in (8, 16, 32, 64)):
                        fuse_brb = True
                    brb = (r, coef) if fuse_brb else None
                    dsplit = 0
                    if fuse_brb:
                        dpre = g                          # what the consumers are given as their "d" / "x": dy itself
                    else:
                        bpart = self._new(nb * Cout)
                        # dpre feeds exactly two GEMMs (weight gradient: dense operand; data gradient: gathered operand);
                        # with split-bf16 products it is written split once here instead of being split by every tile of both
                        dsplit = int(self.math in (1, 3) and Cout % 16 == 0 and inp.C % 4 == 0 and inp.C * 9 > 12 and self.DSPLIT)
                        bwd.add(BnReluBwdArgs(dy=g.ptr, r=r.ptr, coef=nat.ptr(coef), dpre=dpre.ptr,
                                              partials=bpart.data_ptr(), M=r.M, C=Cout, lddy=g.ld, ldr=r.ld, ldo=dpre.ld,
                                              nblocks=nb, split_out=dsplit, bf16=r.bf16))
                        self._defer_sum(bwd, bpart.data_ptr(), G[cv['wname'] + '.bias'].data_ptr(), Cout, Cout, nb)
                    side_buf = 'dpre%d' % self._dpre_turn          # (one key space: scratch names, see _side_join / _dz_for)
                    if fuse_brb:
                        # (a side-stream weight gradient reads dy itself: the scratch it lives in must not be rewritten --
                        # by the data gradient one layer further down -- before it is done: _dz_for joins on this key)
                        side_buf = next((kk for kk in ('dz', 'dpre0') if g.t is self._scratch.get(kk)), None)
                    # Small tensors (deep levels: many (cm, cg) tiles, each of which would form the operand again): the data-gradient
                    # kernel, which forms it anyway, writes it once (x_out) and the weight gradient -- emitted BEHIND it -- reads it
                    dmat = None
                    if (fuse_brb and bool(inp.bf16) and not circ and pad == 1 and not side and (d > 0 or dxin is not None)
                            and 'dmat' in self._scratch and r.M * Cout * 2 <= self.DPRE_OUT_BYTES):
                        dmat = self._scratch_act('dmat', N, r.H, r.W, Cout)
                        self.dbg['dmat:%s.block.%d' % (prefix, d * step + 1)] = dmat

                    def emit_wgrad():
                        if dmat is not None:
                            self._wgrad(bwd, cv['gin'], dmat, G[cv['wname'] + '.weight'], 3, 3, 1, pad, r.H, r.W, in_aff=cv['inp_aff'],
                                        bias_out=G[cv['wname'] + '.bias'], bias_plain=True)
                        else:
                            self._wgrad(bwd, cv['gin'], dpre, G[cv['wname'] + '.weight'], 3, 3, 1, 0 if circ else pad, r.H, r.W,
                                        in_aff=cv['inp_aff'], side=side, side_buf=side_buf, d_split=dsplit, brb=brb,
                                        bias_out=G[cv['wname'] + '.bias'])
                    if dmat is None:
                        emit_wgrad()
                    if circ and (d > 0 or dxin is not None):
                        # data gradient on the framed grid, folded back onto the pixels the frame copies (see _wrap_pad)
                        wd = self._pack_conv_dgrad(cv['w'])
                        dzp = self._framed_grad(inp)
                        self._conv(bwd, dpre, wd, dzp, 3, 3, 1, 2, inp.C, x_split=dsplit, brb=brb)
                        if d > 0:
                            dz = self._dz_for(g, N, inp.H, inp.W, Cout, fuse_brb)
                            self._wrap_fold(bwd, dzp, dz)
                            fused = None
                            g = dz
                        else:
                            self._wrap_fold(bwd, dzp, dxin, accumulate=1 if (do_res and not res_dgrad_last) else 0)
                            if res_dgrad_last:
                                dxin_part = self._conv(bwd, dout, self._pack_conv_dgrad(rw), dxin, 1, 1, 1, 0, xin.C,
                                                       accumulate=1, stats=dxin_stats and self.FUSE_COLSUMS)
                    elif d > 0:
                        wd = self._pack_conv_dgrad(cv['w'])
                        dz = self._dz_for(g, N, inp.H, inp.W, Cout, fuse_brb)
                        prev = convs[d - 1]
                        if prev['bn'] is not None and self.FUSE_BWD_STATS:
                            # the data-gradient conv leaves sum(dz), sum(dz*r) per channel for the next BN backward
                            fused = self._conv(bwd, dpre, wd, dz, 3, 3, 1, 2 - pad, Cout, stats=True, stat_other=prev['r'],
                                               x_split=dsplit, brb=brb, x_out=dmat,
                                               stat_totals=self._bn_totals(Cout, bwd=True) if prev['live_bwd'] else None)
                        else:
                            fused = None
                            self._conv(bwd, dpre, wd, dz, 3, 3, 1, 2 - pad, Cout, x_split=dsplit, brb=brb, x_out=dmat)
                        if dmat is not None:
                            emit_wgrad()
                        g = dz
                    elif dxin is not None:
                        wd = self._pack_conv_dgrad(cv['w'])
                        want = dxin_stats and self.FUSE_COLSUMS
                        dxin_part = self._conv(bwd, dpre, wd, dxin, 3, 3, 1, 2 - pad, xin.C,
                                               accumulate=1 if (do_res and not res_dgrad_last) else 0,
                                               x_split=dsplit, stats=want and not res_dgrad_last, brb=brb, x_out=dmat)
                        if res_dgrad_last:
                            dxin_part = self._conv(bwd, dout, self._pack_conv_dgrad(rw), dxin, 1, 1, 1, 0, xin.C,
                                                   accumulate=1, stats=want)
                        if dmat is not None:
                            emit_wgrad()
                    self._maybe_flush(bwd)
                return dxin_part
            # pre-BatchNorm output of the block's last conv: what a producer of this block's dout needs for fused_in
            backward.last_r = convs[-1]['r'] if convs[-1]['bn'] is not None else None
            backward.last_live = bool(convs[-1]['bn'] is not None and convs[-1]['live_bwd'])    # ... and may leave them as live totals
            return backward

        # ------------------------------------------------------------------ down path
        x = x0
        dx_in = None        # gradient Act of the current block input (None for the network input)
        pending = []        # (kind, data) records consumed when emitting backward
        down_out = []
        for i in range(depth):
            Ci = chans[i]
            if i != depth - 1 and direct_bridge[i]:
                out = cat[i].chan_slice(Ci, Ci)
            else:
                out = self._act(N, hout[i], wout[i], Ci)
            bw = block('down_path.%d' % i, x, out, first=(i == 0))
            down_out.append(out)
            rec = dict(block_bw=bw, out=out, xin=x, level=i)
            if i != depth - 1:
                if not direct_bridge[i]:
                    ch, cw = up_hw[depth - 2 - i]
                    oy, ox = (hout[i] - ch) // 2, (wout[i] - cw) // 2
                    dst = cat[i].chan_slice(Ci, Ci)
                    fwd.add(AffineCopyArgs(x=out.ptr, y=dst.ptr, N=N, H=ch, W=cw, C=Ci, ldx=out.ld, xH=out.H, xW=out.W,
                                           xoy=oy, xox=ox, ldy=dst.ld, yH=ch, yW=cw, bf16=out.bf16))
                    rec['crop'] = (oy, ox, ch, cw)
                nxt = self._act(N, hin[i + 1], win[i + 1], Ci)
                if cfg['max_pool']:
                    self.pool_in[i] = out
                    fwd.add_pool(PoolArgs(x=out.ptr, y=nxt.ptr, N=N, H=out.H, W=out.W, C=Ci, ldx=out.ld, ldy=nxt.ld,
                                          bf16=out.bf16), backward=False)
                else:
                    dw_, db_ = self.P['downsample_convs.%d.weight' % i], self.P['downsample_convs.%d.bias' % i]
                    wp = self._pack_conv_fwd(dw_)
                    self._conv(fwd, out, wp, nxt, 2, 2, 2, 0, Ci, bias=db_)
                rec['nxt'] = nxt
                self.dbg['nxt:%d' % i] = nxt
                x = nxt
            pending.append(rec)

        # ------------------------------------------------------------------ up path
        up_recs = []
        u = down_out[depth - 1]
        for j, i in enumerate(reversed(range(depth - 1))):
            Ci = chans[i]
            name = 'up_path.%d' % j
            ch, cw = up_hw[j]
            up_half = cat[i].chan_slice(0, Ci)
            if upsample:
                # up_mode='upsample' (unet.py:242-244): bilinear x2, then a 1x1 convolution.  Both are linear, the convolution is
                # pointwise and the interpolation weights sum to one: conv(up(x)) = up(conv(x)) with the bias added once -- the 1x1
                # runs on the small grid, dfl_upsample2x_fwd spreads its result into the up half of the concat buffer
                uw_, ub_ = self.P[name + '.up.1.weight'], self.P[name + '.up.1.bias']
                ylow = self._act(N, u.H, u.W, Ci)
                self._conv(fwd, u, self._pack_conv_fwd(uw_), ylow, 1, 1, 1, 0, Ci, bias=ub_)
                fwd.add(UpsampleArgs(x=ylow.ptr, y=up_half.ptr, N=N, H=u.H, W=u.W, C=Ci, ldx=ylow.ld, ldy=up_half.ld, bf16=ylow.bf16),
                        nat.OP_UPSAMPLE_FWD)
            else:
                uw_, ub_ = self.P[name + '.up.weight'], self.P[name + '.up.bias']
                wp = self._pack_convT_fwd(uw_)
                self._conv(fwd, u, wp, up_half, 1, 1, 1, 0, 4 * Ci, bias=ub_, scatter=1, Hout=ch, Wout=cw)
            out = self._act(N, ch - shrink, cw - shrink, Ci)
            self.dbg['cat:%d' % j] = cat[i]
            bw = block(name + '.conv_block', cat[i], out)
            up_recs.append(dict(block_bw=bw, out=out, u=u, level=i, name=name, w=uw_))
            u = out
        self.feat = u
        self._finish_live_bn(fwd)

        # ------------------------------------------------------------------ heads
        NC, L = cfg['n_classes'], cfg['num_lands']
        F = u.C
        self.NC, self.L = NC, L
        w_seg = self.P['seg_conv.weight']
        w_l1 = self.P['lands_1x1.0.weight'] if L > 0 else None
        w_l2 = self.P.get('lands_1x1.1.weight') if L > 0 else None
        NM = w_l1.shape[0] if L > 0 else 0
        # lands_num_1x1 > 2 (unet.py:146-156): the 1x1 convolutions behind the first one are bias-free and have no
        # non-linearity between them, i.e. ONE linear map  W_k ... W_2 W_1  (L x NM).  The head kernels run with that product
        # (fold_tail, re-made whenever weights change) and the gradient they return for it is taken apart afterwards
        # (unfold_tail_grads):  dW_j = (W_k ... W_{j+1})^T  G  (W_{j-1} ... W_1)^T.
        self.tail_names = []
        j = 1
        while L > 0 and ('lands_1x1.%d.weight' % j) in self.P:
            self.tail_names.append('lands_1x1.%d.weight' % j)
            j += 1
        self.w_l2_eff = self.g_l2_eff = None
        if len(self.tail_names) > 1:
            self.w_l2_eff = self._new(L * NM).view(L, NM, 1, 1)
            self.g_l2_eff = self._new(L * NM).view(L, NM, 1, 1)
            w_l2 = self.w_l2_eff
        # lands_block_depth > 0 (unet.py:108-137,185-187): bias-only 3x3 convolutions F -> F/2 (-> F/2 ...) in front of the
        # landmark 1x1.  Their output lb is written next to a copy of the features into ONE wider tensor [u | lb], and the
        # head kernels run on that with widened weight matrices -- w_seg' = [w_seg | 0], w_l1' = [0 | W1[:, :F/2] | W1[:, F/2:]]
        # (fold_tail) -- so seg sees u, the landmark branch sees cat(lb, logits), and no kernel changes.  Their gradients come
        # back as columns of the widened matrices and of dx (unfold_tail_grads; the lb columns of dx feed the convolutions'
        # backward, the u columns go on into the decoder).
        lbd = int(cfg.get('lands_block_depth', 0)) if L > 0 else 0
        self.lb_layers = []
        self.eff_heads = None
        head_x, Fh = u, F
        # Unpadded (padding=False) the landmark block's valid 3x3 convolutions shrink lb against the features by lbd pixels per
        # side; the reference crops the logits to lb (unet.py:185-187), so the heat maps come out SMALLER than the
        # segmentation.  The head kernels then run twice: on u with the segmentation matrix alone (the full-size seg), and on
        # [crop(u) | lb] with the widened matrices (heat maps; its soft-max output, a crop of seg, stays in a plan buffer).
        self.split_heads = lbd > 0 and not cfg['padding']
        hl, wl = u.H, u.W
        if lbd > 0:
            F2 = F // 2
            if F2 % (16 if self.bf16 else 4) != 0:
                raise PlanError('lands_block_depth > 0 needs F/2 = %d channels to be a multiple of %d in this arithmetic' % (F2, 16 if self.bf16 else 4))
            lpad = 1 if cfg['padding'] else 0
            if not lpad:
                hl, wl = u.H - 2 * lbd, u.W - 2 * lbd
                if hl < 1 or wl < 1:
                    raise PlanError('input %dx%d too small for lands_block_depth=%d without padding' % (self.H, self.W, lbd))
            Fh = F + F2
            uw = self._act(N, hl, wl, Fh)
            fwd.add(AffineCopyArgs(x=u.ptr, y=uw.ptr, N=N, H=hl, W=wl, C=F, ldx=u.ld, xH=u.H, xW=u.W, xoy=(u.H - hl) // 2, xox=(u.W - wl) // 2,
                                   ldy=uw.ld, yH=hl, yW=wl, bf16=u.bf16))
            src = u
            for j in range(lbd):
                wname, bname = 'lands_block.%d.weight' % j, 'lands_block.%d.bias' % j
                ho_, wo_ = src.H - (0 if lpad else 2), src.W - (0 if lpad else 2)
                dst = uw.chan_slice(F, F2) if j == lbd - 1 else self._act(N, ho_, wo_, F2)
                gin = self._wrap_pad(fwd, src) if circ else src
                self._conv(fwd, gin, self._pack_conv_fwd(self.P[wname]), dst, 3, 3, 1, 0 if (circ or not lpad) else 1, F2, bias=self.P[bname],
                           Hout=ho_, Wout=wo_)
                self.lb_layers.append((src, dst, wname, bname, gin, ho_, wo_))
                src = dst
            self.eff_heads = dict(F=F, F2=F2, w_seg=self._new(NC * Fh).view(NC, Fh, 1, 1).zero_(),
                                  w_l1=self._new(NM * (Fh + NC)).view(NM, Fh + NC, 1, 1).zero_(),
                                  g_seg=self._new(NC * Fh).view(NC, Fh, 1, 1), g_l1=self._new(NM * (Fh + NC)).view(NM, Fh + NC, 1, 1))
            head_x = uw
        sm = 1 if cfg['do_soft_max'] else 0
        self.head_fwd_seg = None
        if self.split_heads:
            self.head_fwd_seg = HeadFwdArgs(x=u.ptr, w_seg=self.P['seg_conv.weight'].data_ptr(), N=N, H=u.H, W=u.W, F=F, ldx=u.ld,
                                            NC=NC, NM=0, L=0, softmax=sm, x_bf16=u.bf16)
            fwd.add(self.head_fwd_seg, volatile=True)
            self.seg_crop = self._new(N * NC * hl * wl)         # soft-max of the landmark call: what its backward recomputes from
        if self.eff_heads is not None:
            w_seg, w_l1 = self.eff_heads['w_seg'], self.eff_heads['w_l1']
        self.head_fwd = HeadFwdArgs(x=head_x.ptr, w_seg=w_seg.data_ptr(), w_l1=nat.ptr(w_l1), w_l2=nat.ptr(w_l2),
                                    N=N, H=hl, W=wl, F=Fh, ldx=head_x.ld, NC=NC, NM=NM, L=L, softmax=sm, x_bf16=u.bf16)
        fwd.add(self.head_fwd, volatile=True)       # writes the caller-owned outputs: addresses change per call
        self.out_hw = (u.H, u.W)
        self.heat_hw = (hl, wl)

        self._n_fwd_pack = len(self._pack_jobs)
        if not self.need_grad:
            self._finish_pack()
            return

        # ================================================================== backward program
        # flat gradient arena in parameters() order
        self.grad_names = [k for k in self.P.keys()]
        offs, tot = {}, 0
        for k in self.grad_names:
            offs[k] = tot
            tot += (self.P[k].numel() + 3) // 4 * 4          # keep every slice 16-byte aligned
        self.grad_flat = self._new(tot).zero_()
        self.G = {k: self.grad_flat[offs[k]:offs[k] + self.P[k].numel()].view(self.P[k].shape) for k in self.grad_names}
        self.grad_offsets = offs
        self.dead_params = set()
        if not cfg['max_pool']:
            self.dead_params.add('downsample_convs.%d.weight' % (depth - 1))    # never used in forward
            self.dead_params.add('downsample_convs.%d.bias' % (depth - 1))

        # heads
        Fdec = F                                  # channels of the decoder output (Fh: what the landmark head call sees)
        g_seg = self.eff_heads['g_seg'] if self.eff_heads is not None else self.G['seg_conv.weight']
        g_l1 = (self.eff_heads['g_l1'] if self.eff_heads is not None else self.G['lands_1x1.0.weight']) if L > 0 else None
        g_l2 = None
        if L > 0 and w_l2 is not None:
            g_l2 = self.g_l2_eff if self.g_l2_eff is not None else self.G['lands_1x1.1.weight']
        if self.PACK_OVERLAP:
            bwd.wait(self.EV_PACK_DONE, stream=0)      # data-gradient weight layouts are packed on the side stream

        def head_backward(x_act, Fx, H_, W_, ws, w1, w2, nm, nl, gs, g1, g2, live_r=None):
            """HeadBwdArgs + its weight gradients on x_act [N,H_,W_,Fx]; returns (args, dx Act).  live_r: the saved ReLU output behind
            the BatchNorm dx enters -- the matrix-core kernel then adds (sum dx, sum dx*r) to that layer's live totals itself
            (self._head_live: ('live', totals) or None)."""
            sld = self.lib.dfl_head_scratch_ld_for(Fx, NC, nm, nl)
            M_ = N * H_ * W_
            dx = self._act(N, H_, W_, Fx)
            fused_head = bool(x_act.bf16) and Fx == 32 and NC <= 8 and nm <= 24 and nl <= 16 and os.environ.get('DFL_HEAD_FUSED', '1') != '0'
            hb = HeadBwdArgs(x=x_act.ptr, w_seg=ws.data_ptr(), w_l1=nat.ptr(w1), w_l2=nat.ptr(w2), dx=dx.ptr, N=N, H=H_, W=W_, F=Fx,
                             ldx=x_act.ld, lddx=dx.ld, NC=NC, NM=nm, L=nl, softmax=1 if cfg['do_soft_max'] else 0, scratch_ld=sld,
                             x_bf16=x_act.bf16)
            # bf16 features of the paper's width: the head kernel takes its three weight gradients itself (include/dfl_hip.h);
            # otherwise it leaves a per-pixel scratch row and three 1x1 weight-gradient launches follow
            self._head_live = None
            mfma_head = fused_head and nm <= 24 and (nl == 0 or w2 is not None) and x_act.ld % 8 == 0 and os.environ.get('DFL_HEAD_MFMA', '1') != '0'
            if mfma_head and live_r is not None and Fx == 32:
                tot = self._bn_totals(Fx, bwd=True)
                hb.stat_other, hb.ldso, hb.stat_totals = live_r.ptr, live_r.ld, tot
                self._head_live = ('live', tot)
            if fused_head:
                part = self._new(4096 * nat.check(self.lib.dfl_head_wgrad_blocks(M_), 'dfl_head_wgrad_blocks'))
                hb.wg_partial = part.data_ptr()
                hb.dw_seg = gs.data_ptr()
                if nl > 0:
                    hb.dw_l1 = g1.data_ptr()
                    if w2 is not None:
                        hb.dw_l2 = g2.data_ptr()
            else:
                scratch = self._new(M_ * sld)
                hb.scratch = scratch.data_ptr()
            bwd.add(hb, volatile=True)                 # reads the caller-owned seg / dseg / dheat
            if not fused_head:
                off = [self.lib.dfl_head_scratch_off_for(Fx, NC, nm, nl, k) for k in range(5)]

                def sact(o, c):
                    return Act(scratch, scratch.data_ptr() + 4 * o, sld, N, H_, W_, c)
                self._wgrad(bwd, sact(off[0], Fx), sact(off[1], NC), gs, 1, 1, 1, 0, H_, W_)
                if nl > 0:
                    self._wgrad(bwd, sact(off[0], Fx + NC), sact(off[2], nm), g1, 1, 1, 1, 0, H_, W_)
                    if w2 is not None:
                        self._wgrad(bwd, sact(off[3], nm), sact(off[4], nl), g2, 1, 1, 1, 0, H_, W_)
            return hb, dx

        hl, wl = self.heat_hw
        self.head_bwd_seg = None
        if self.split_heads:
            self.g_seg_full = self._new(NC * Fdec).view(NC, Fdec, 1, 1)
            self.head_bwd_seg, dfeat_a = head_backward(u, Fdec, u.H, u.W, self.P['seg_conv.weight'], None, None, 0, 0, self.g_seg_full, None, None)
            self.dseg_zero = self._new(N * NC * hl * wl).zero_()
        last_bw = up_recs[-1]['block_bw'] if up_recs else pending[depth - 1]['block_bw']
        head_live_r = (last_bw.last_r if (self.LIVE_BN and self.LIVE_HEAD and self.FUSE_COLSUMS and not self.lb_layers and not self.split_heads
                                          and last_bw.last_live) else None)
        self.head_bwd, dfeat = head_backward(head_x, Fh, hl, wl, w_seg, w_l1, w_l2, NM, L, g_seg, g_l1, g_l2, live_r=head_live_r)
        head_sums = self._head_live
        self.dbg['dfeat'] = dfeat

        # up path, last block first
        # Column sums ride on the kernels that produce the tensors: the conv that completes dcat leaves sum(dy) (the
        # transposed conv's bias gradient), the conv that produces du leaves the BatchNorm-backward sums of the block
        # that consumes du, the conv that completes a block-input gradient leaves the down-sampling bias gradient.
        dout = dfeat
        if self.lb_layers:
            # backward of the landmark block: the lb columns of dx through the 3x3 convolutions (bias sums, weight gradients,
            # data gradients), the first one's data gradient added onto the u columns, which then enter the decoder
            lpad = 1 if cfg['padding'] else 0
            if self.split_heads:
                # the u columns of the landmark call's dx belong to the centre window of the full-size feature gradient
                dout = dfeat_a
                src_u = dfeat.chan_slice(0, Fdec)
                bwd.add(AffineCopyArgs(x=src_u.ptr, y=dout.ptr, N=N, H=hl, W=wl, C=Fdec, ldx=src_u.ld, xH=hl, xW=wl, ldy=dout.ld,
                                       yH=u.H, yW=u.W, yoy=(u.H - hl) // 2, yox=(u.W - wl) // 2, accumulate=1, bf16=dout.bf16))
            else:
                dout = dfeat.chan_slice(0, Fdec)
            d = dfeat.chan_slice(Fdec, Fh - Fdec)
            for j in reversed(range(len(self.lb_layers))):
                src, dst, wname, bname, gin, ho_, wo_ = self.lb_layers[j]
                self._colsum(bwd, d, self.G[bname])
                self._wgrad(bwd, gin, d, self.G[wname], 3, 3, 1, 0 if (circ or not lpad) else 1, ho_, wo_)
                wd = self._pack_conv_dgrad(self.P[wname])
                tgt = dout if j == 0 else self._act(N, src.H, src.W, Fh - Fdec)
                if circ:
                    dzp = self._framed_grad(src)
                    self._conv(bwd, d, wd, dzp, 3, 3, 1, 2, src.C)
                    self._wrap_fold(bwd, dzp, tgt, accumulate=1 if j == 0 else 0)
                else:
                    self._conv(bwd, d, wd, tgt, 3, 3, 1, 1 if lpad else 2, src.C, accumulate=1 if j == 0 else 0)
                if j > 0:
                    d = tgt
        F = Fdec
        dout_sums = head_sums if not self.lb_layers else None
        for j in reversed(range(len(up_recs))):
            rec = up_recs[j]
            i = rec['level']
            Ci = chans[i]
            st = rec['block_bw'](dout, dcat[i], fused_in=dout_sums, dxin_stats=True)
            dy = dcat[i].chan_slice(0, Ci)                    # gradient of the transposed-conv output
            uin = rec['u']
            bias_name = rec['name'] + ('.up.1.bias' if upsample else '.up.bias')
            if st is not None:                                # rows of (sum v, sum v^2) over all 2*Ci columns of dcat
                self._defer_sum(bwd, st[0].data_ptr(), self.G[bias_name].data_ptr(), Ci, 4 * Ci, st[1])
            else:
                self._colsum(bwd, dy, self.G[bias_name])          # (upsample: the adjoint preserves column sums)
            du = self._act(N, uin.H, uin.W, uin.C)
            self.dbg['du:%d' % j] = du
            consumer = up_recs[j - 1]['block_bw'] if j > 0 else pending[depth - 1]['block_bw']
            r_last = consumer.last_r if self.FUSE_COLSUMS else None
            if upsample:
                # gradient of the small-grid 1x1 result = adjoint of the interpolation applied to dy; then an ordinary 1x1 layer
                dylow = self._act(N, uin.H, uin.W, Ci)
                bwd.add(UpsampleArgs(x=dylow.ptr, y=dy.ptr, N=N, H=uin.H, W=uin.W, C=Ci, ldx=dylow.ld, ldy=dy.ld, bf16=dy.bf16),
                        nat.OP_UPSAMPLE_BWD)
                self._wgrad(bwd, uin, dylow, self.G[rec['name'] + '.up.1.weight'], 1, 1, 1, 0, uin.H, uin.W)
                dout_sums = self._conv(bwd, dylow, self._pack_conv_dgrad(rec['w']), du, 1, 1, 1, 0, uin.C,
                                       stats=r_last is not None, stat_other=r_last,
                                       stat_totals=self._bn_totals(uin.C, bwd=True) if (r_last is not None and consumer.last_live) else None)
            else:
                # dW[ci][co][ab] = sum x[i,j][ci] * dy[2i+a,2j+b][co]
                self._wgrad(bwd, dy, uin, self.G[rec['name'] + '.up.weight'], 2, 2, 2, 0, uin.H, uin.W)
                wd = self._pack_convT_dgrad(rec['w'])
                dout_sums = self._conv(bwd, dy, wd, du, 2, 2, 2, 0, uin.C, stats=r_last is not None, stat_other=r_last,
                                       stat_totals=self._bn_totals(uin.C, bwd=True) if (r_last is not None and consumer.last_live) else None)
            dout = du
        # down path, deepest block first
        for i in reversed(range(depth)):
            rec = pending[i]
            out = rec['out']
            Ci = chans[i]
            down_sums = None
            if i != depth - 1:
                # dout = bridge gradient (+ crop padding) + down-sampling gradient
                if direct_bridge[i]:
                    dout = dcat[i].chan_slice(Ci, Ci)
                else:
                    dout = self._act(N, out.H, out.W, Ci)
                    oy, ox, ch, cw = rec['crop']
                    src = dcat[i].chan_slice(Ci, Ci)
                    bwd.add(MemsetArgs(ptr=dout.ptr, bytes=dout.esz * dout.M * Ci))
                    bwd.add(AffineCopyArgs(x=src.ptr, y=dout.ptr, N=N, H=ch, W=cw, C=Ci, ldx=src.ld, xH=ch, xW=cw,
                                           ldy=dout.ld, yH=out.H, yW=out.W, yoy=oy, yox=ox, bf16=src.bf16))
                dnxt = rec['dnxt']
                nxt = rec['nxt']
                if cfg['max_pool']:
                    bwd.add_pool(PoolArgs(x=out.ptr, y=dnxt.ptr, dx=dout.ptr, N=N, H=out.H, W=out.W, C=Ci, ldx=out.ld,
                                          ldy=dnxt.ld, lddx=dout.ld, bf16=out.bf16), backward=True)
                else:
                    wname = 'downsample_convs.%d' % i
                    st = rec.get('dnxt_sums')
                    if st is not None:
                        self._defer_sum(bwd, st[0].data_ptr(), self.G[wname + '.bias'].data_ptr(), Ci, 2 * Ci, st[1])
                    else:
                        self._colsum(bwd, dnxt, self.G[wname + '.bias'])
                    self._wgrad(bwd, out, dnxt, self.G[wname + '.weight'], 2, 2, 2, 0, nxt.H, nxt.W)
                    wd = self._pack_down_dgrad(self.P[wname + '.weight'])
                    # the last kernel that writes dout leaves sum(dout), sum(dout * r_last) for the block's first BN backward
                    r_last = rec['block_bw'].last_r
                    # (the scatter visits the 2*H' x 2*W' pixels it writes: with an odd out.H / out.W the last row / column of
                    # dout holds bridge gradient only and would be missing from the sums -- then the separate pass takes them)
                    want = (self.FUSE_DOWN_STATS and r_last is not None and bool(dout.bf16) and bool(r_last.bf16)
                            and out.H == 2 * nxt.H and out.W == 2 * nxt.W)
                    down_sums = self._conv(bwd, dnxt, wd, dout, 1, 1, 1, 0, 4 * Ci, accumulate=1, scatter=1,
                                           Hout=out.H, Wout=out.W, stats=want, stat_other=r_last if want else None,
                                           stat_totals=self._bn_totals(Ci, bwd=True) if (want and rec['block_bw'].last_live) else None)
            if i > 0:
                dxin = self._act(N, rec['xin'].H, rec['xin'].W, rec['xin'].C)
                pending[i - 1]['dnxt'] = dxin
            elif self.input_grad:
                if self.bf16:
                    raise PlanError('gradients with respect to the network input are implemented for the fp32-tensor arithmetic '
                                    'modes only (DFL_MATH=fp32 / bf16x3 / bf16x6 / bf16), not for bf16 storage')
                dxin = self._act(N, rec['xin'].H, rec['xin'].W, rec['xin'].C)
                self.dx_in = dxin
            else:
                dxin = None
            st = rec['block_bw'](dout, dxin, fused_in=dout_sums if i == depth - 1 else down_sums,
                                 dxin_stats=dxin is not None and i > 0 and not cfg['max_pool'])
            if i > 0:
                pending[i - 1]['dnxt_sums'] = st
        self._flush_sums(bwd)
        self._side_join(bwd)
        if self._tot_bwd[0] is not None:
            bwd.insert(0, MemsetArgs(ptr=self._tot_bwd[0].data_ptr(), bytes=8 * self._tot_bwd[1]))
            # every op index recorded while the program was emitted moved up by one (the flushes' indices
            # stayed behind and data-parallel buckets were cut one op early)
            self._red_flushes = [(idx + 1, dsts) for idx, dsts in self._red_flushes]
            for e in self._side:
                e['op'] += 1
                if e['waited'] is not None:
                    e['waited'] += 1
        self._order_pack_jobs()
        self._finish_pack()
        # index of the last backward op that writes each parameter gradient (data-parallel bucket scheduling)
        by_ptr = {self.G[k].data_ptr(): k for k in self.grad_names}
        self.grad_ready_op = {}
        joined = {e['op']: e['waited'] for e in self._side}     # side-stream op -> main-stream wait that covers it
        for idx, st in enumerate(bwd.structs):
            for field in ('dw', 'dst', 'out', 'dgamma', 'dbeta', 'dw_seg', 'dw_l1', 'dw_l2'):   # (dw_*: head_bwd's own weight gradients)
                name = by_ptr.get(getattr(st, field, None))
                if name is not None:
                    self.grad_ready_op[name] = joined.get(idx, idx)
        for idx, dsts in self._red_flushes:
            for d in dsts:
                name = by_ptr.get(d)
                if name is not None:
                    self.grad_ready_op[name] = max(idx, self.grad_ready_op.get(name, -1))
        if self.g_l2_eff is not None:            # taken apart after the last op (unfold_tail_grads)
            for name in self.tail_names:
                self.grad_ready_op[name] = len(bwd.structs) - 1
        if self.eff_heads is not None:
            for name in ('seg_conv.weight', 'lands_1x1.0.weight'):
                self.grad_ready_op[name] = len(bwd.structs) - 1

    # ------------------------------------------------------------------------------------------ folded 1x1 tail
    def fold_tail(self):
        """w_l2_eff = W_k ... W_1 of the landmark 1x1 convolutions behind the first (tiny torch matmuls on the current stream)."""
        if self.eff_heads is not None:           # lands_block_depth > 0: widened head matrices (see the heads section)
            e = self.eff_heads
            F, F2 = e['F'], e['F2']
            e['w_seg'][:, :F].copy_(self.P['seg_conv.weight'])
            w1 = self.P['lands_1x1.0.weight']
            e['w_l1'][:, F:F + F2].copy_(w1[:, :F2])
            e['w_l1'][:, F + F2:].copy_(w1[:, F2:])
        if self.w_l2_eff is None:
            return
        mats = [self.P[n].view(self.P[n].shape[0], self.P[n].shape[1]) for n in self.tail_names]
        eff = mats[0]
        for m in mats[1:]:
            eff = m @ eff
        self.w_l2_eff.view(eff.shape).copy_(eff)

    def unfold_tail_grads(self):
        """Gradients of the folded 1x1 convolutions from the gradient of their product (see the heads section)."""
        if self.eff_heads is not None and self.need_grad:
            e = self.eff_heads
            F, F2 = e['F'], e['F2']
            if self.split_heads:                   # full-size segmentation call + the logits' path into the landmark call
                torch.add(self.g_seg_full, e['g_seg'][:, :F], out=self.G['seg_conv.weight'])
            else:
                self.G['seg_conv.weight'].copy_(e['g_seg'][:, :F])
            g1 = self.G['lands_1x1.0.weight']
            g1[:, :F2].copy_(e['g_l1'][:, F:F + F2])
            g1[:, F2:].copy_(e['g_l1'][:, F + F2:])
        if self.g_l2_eff is None:
            return
        mats = [self.P[n].view(self.P[n].shape[0], self.P[n].shape[1]) for n in self.tail_names]
        G = self.g_l2_eff.view(mats[-1].shape[0], mats[0].shape[1])
        k = len(mats)
        below = [None] * k                       # below[j] = W_{j-1} ... W_1 (None: identity)
        for j in range(1, k):
            below[j] = mats[j - 1] if below[j - 1] is None else mats[j - 1] @ below[j - 1]
        above = None                             # W_k ... W_{j+1}
        for j in reversed(range(k)):
            g = G if above is None else above.t() @ G
            if below[j] is not None:
                g = g @ below[j].t()
            self.G[self.tail_names[j]].view(g.shape).copy_(g)
            above = mats[j] if above is None else above @ mats[j]

    # ------------------------------------------------------------------------------------------ run
    def run_pack(self, stream, forward_only=False):
        self.pack.run(stream)

    def run_pack_rest(self, stream):
        """The pack program without its tiled launch: what is left to do after dfl_sgd_pack_tiled (sgd.SGD.step) has written the
        tiled layouts together with the update."""
        i, n = self._tiled_index, len(self.pack)
        if i > 0:
            self.pack.run(stream, 0, i)
        if n - i - 1 > 0:
            self.pack.run(stream, i + 1, n - i - 1)

    def new_outputs(self):
        N = self.N
        h, w = self.out_hw
        hl, wl = self.heat_hw
        seg = torch.empty((N, self.NC, h, w), dtype=torch.float32, device=self.dev)
        heat = torch.empty((N, self.L, hl, wl), dtype=torch.float32, device=self.dev) if self.L > 0 else None
        return seg, heat

    def bind_outputs(self, seg, heat):
        """Point the head op(s) of the forward program at the caller-owned output tensors."""
        if self.split_heads:
            self.head_fwd_seg.seg = seg.data_ptr()
            self.head_fwd.seg = self.seg_crop.data_ptr()
        else:
            self.head_fwd.seg = seg.data_ptr()
        self.head_fwd.heat = nat.ptr(heat)

    def bind_grads(self, seg, dseg, dheat):
        """Point the head op(s) of the backward program at the forward's seg and the incoming gradients (dheat may be None)."""
        if self.split_heads:
            self.head_bwd_seg.seg, self.head_bwd_seg.dseg = seg.data_ptr(), dseg.data_ptr()
            self.head_bwd.seg, self.head_bwd.dseg = self.seg_crop.data_ptr(), self.dseg_zero.data_ptr()
        else:
            self.head_bwd.seg, self.head_bwd.dseg = seg.data_ptr(), dseg.data_ptr()
        self.head_bwd.dheat = nat.ptr(dheat)

    def grads(self):
        return [None if k in self.dead_params else self.G[k] for k in self.grad_names]

    def act_nchw(self, act):
        """Copy of an internal NHWC activation window as an fp32 NCHW tensor (introspection for tests and diagnosis)."""
        esz = act.esz
        base = act.t.view(-1)
        off = (act.ptr - base.data_ptr()) // esz
        v = base.as_strided((act.N, act.H, act.W, act.C), (act.H * act.W * act.ld, act.W * act.ld, act.ld, 1), off)
        return v.permute(0, 3, 1, 2).float().contiguous()
