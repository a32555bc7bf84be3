"""fp64 emulation of the bf16 STORAGE arithmetic (math mode 4, `DFL_MATH=bf16s`) -- TEST INFRASTRUCTURE ONLY.

The clean fp64 oracle (oracle/ref_cpu.py) is 3e-3 away from anything computed with bf16 tensors, so comparing the headline
mode with it needs bars wide enough to absorb bf16 rounding ("what's weak" #1).  This module restates the SAME
network (train_test_code/unet.py:161-260 forward, torch autograd backward as driven by train.py:405-424) in fp64 with a
round-to-bf16 at exactly the places where the product stores or stages a bf16 value:

  forward   * the GEMM copy of every weight whose layer has >= 16 input channels (the 1-channel first layer and its
              residual 1x1 keep fp32 weights: direct kernels);
            * the operand of a convolution behind a BatchNorm: bf16(scale * r + shift), formed while the patch is staged,
              zero padding AFTER the affine (unet.py:211-222 order conv -> ReLU -> BN);
            * every activation written to HBM: r = bf16(ReLU(conv + bias)), block output bf16(conv1x1 + bias + BN(r)),
              down-sampling / transposed convolutions bf16(conv + bias);
            * BatchNorm statistics are taken from the STORED r; scale / shift / mean / invstd are fp32 vectors.
  backward  * dpre = bf16([r > 0] * (A dy + B r + C)) with the fp32 coefficients A, B, C of bn_bwd_finalize, whose sums
              sum(dy), sum(dy * r) run over stored values; the bias gradient is the column sum of the ROUNDED dpre;
            * every activation gradient written to HBM is bf16; accumulating epilogues (residual 1x1 data gradient on
              top of the 3x3 one, down-sampling gradient on top of the bridge gradient) read the stored bf16 value, add in
              fp32 and round again;
            * the matrix-core head (32 bf16 features): exact products in the chain, but its three weight gradients
              contract bf16 copies of [dlogits | dmid | dheat] and [x | logits | mid]; dx is stored as bf16.
  fp64 here = fp32 there: accumulations, statistics, losses, weight gradients.

Against THIS reference, on the ReLU / pooling pattern of the HIP run, what is left of a gradient's error is fp32
accumulation order plus rare one-ulp bf16 flips (an fp32 sum that lands on the other side of a rounding boundary): the
level of the bf16x3 arithmetic, not 3e-3.

Supported: the architectures the reference's command lines select (zero padding or none, BatchNorm on / off, residual on /
off, max-pool or strided convolution, any block depth, one or two heads with <= 2 landmark 1x1 layers).  `up_mode='upsample'`,
`pad_mode='circular'`, `lands_block_depth > 0` raise NotImplementedError: those keep the clean-fp64 comparison.

TWO WAYS TO USE IT.  (1) Free running: the emulation computes everything from the network input; this is what the CPU tests
pin to the oracle (rounding switched off) -- but a free-running comparison with a HIP run CANNOT be tight: rounding to bf16
is discontinuous, a difference of one fp32 ulp in a sum flips the stored bf16 value with probability (fp32 error / bf16
ulp), every flipped element perturbs the 9 x Cout sums it enters, and after five or six layers two valid implementations
that differ in summation order only are a full bf16 rounding (3e-3) apart (measured, docs/experiments/emu_layers.py: 1 element of
72816 differs after the first layer, 14 % after the sixth).  (2) Teacher forced (`teacher=`): every stored tensor and every
statistics vector the emulation is about to use is replaced by the one the HIP run holds, after the two were compared --
each step of the HIP pass is then checked against its exact definition applied to the HIP run's OWN inputs: a stored bf16
tensor must be the correctly rounded result up to rare one-ulp flips, an fp32 result must agree to fp32 accumulation
accuracy.  The network is the composition of the steps, so this pins the headline arithmetic op by op, with nothing
calibrated on the run under test (tests/test_gpu_bf16_stepwise.py).

Only tests/ import this.  Citations: the rounding places are csrc/convp_bf16.hip (staging :344-366, epilogue :569-610),
csrc/wgradp_bf16.hip (:236-275), csrc/bn_elem.hip (bn_finalize_kernel, bn_bwd_finalize_kernel), csrc/head_mfma.inc.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1.0e-5


def rb(t):
    """Round to bf16 (through fp32, as the kernels do), back in fp64."""
    return t.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def f32(t):
    return t.to(torch.float32).to(torch.float64)


def _rms(t):
    return float(t.detach().double().pow(2).mean().sqrt())


def compare_stored(e, h):
    """A stored bf16 tensor of the HIP run (h) against the emulation's (e, both fp64 holding bf16 values): relative L2 distance,
    fraction of differing elements, and the largest difference in units of the bf16 ulp of the larger magnitude -- elements whose
    difference is below 1e-6 of the tensor's rms are not counted (a ReLU input within fp32 noise of zero)."""
    d = (e - h).abs()
    rms = max(_rms(h), 1e-300)
    big = torch.maximum(e.abs(), h.abs())
    ulp = torch.exp2(torch.floor(torch.log2(torch.clamp(big, min=1e-300))) - 7.0)
    live = d > 1e-6 * rms
    ulps = float((d / ulp)[live].max()) if bool(live.any()) else 0.0
    # what is left of the largest difference after ONE bf16 ulp of the element is taken off, in units of the tensor's rms: the
    # fp32 accumulation error of a sum that cancels (its bf16 ulp is small, the error of its terms is not)
    excess = float(torch.clamp(d - ulp, min=0.0).max()) / rms
    return {'kind': 'bf16', 'rel_l2': float(d.pow(2).sum().sqrt() / max(float(h.pow(2).sum().sqrt()), 1e-300)),
            'frac': float(live.double().mean()), 'max_ulps': ulps, 'excess': excess, 'n': h.numel()}


def compare_fp32(e, h):
    """An fp32 result of the HIP run against the emulation's fp64 value: relative L2 and the largest error over the largest value."""
    e, h = e.double().reshape(-1), h.double().reshape(-1)
    den = max(float(e.pow(2).sum().sqrt()), 1e-300)
    return {'kind': 'fp32', 'rel_l2': float((e - h).pow(2).sum().sqrt()) / den, 'max_rel': float((e - h).abs().max()) / max(float(e.abs().max()), 1e-300),
            'norm': den, 'n': h.numel()}


def _conv_bwd(gout, inp, w, stride, padding, transposed=False, want_input=True):
    """(grad_input or None, grad_weight) of a (transposed) convolution in fp64."""
    gi, gw, _ = torch.ops.aten.convolution_backward(gout, inp, w, None, [stride, stride], [padding, padding], [1, 1], transposed,
                                                    [0, 0], 1, [want_input, True, False])
    return gi, gw


class Bf16Emulation:
    """One forward + backward of `onet` (oracle/ref_cpu.OracleUNet in fp64, training mode) in the emulated arithmetic.

    cfg: constructor flags; choices: {'relu': {module name: bool mask}, 'pool': {level: flat indices}} (noise_floor.hip_choices)
    or None for the emulation's own pattern.  run(x, loss_fn) -> dict(grads, seg, heat, loss, info)."""

    def __init__(self, onet, cfg, choices=None, teacher=None):
        if cfg.get('up_mode', 'upconv') != 'upconv' or cfg.get('pad_mode', 'zeros') != 'zeros' or cfg.get('lands_block_depth', 0) > 0:
            raise NotImplementedError('bf16 emulation: upsample / circular / landmark-block architectures are not restated')
        if cfg.get('num_lands', 0) > 0 and cfg.get('lands_num_1x1', 2) > 2:
            raise NotImplementedError('bf16 emulation: more than two landmark 1x1 layers are not restated')
        self.net, self.cfg, self.choices = onet, cfg, choices
        self.P = dict(onet.named_parameters())
        self.info = {'relu_flips': 0, 'pool_flips': 0, 'relu_total': 0, 'max_margin': 0.0}
        self.acts = {}                  # nn.ReLU module name -> stored r (diagnosis: compared with plan.relu_out)
        self.teacher = teacher          # callable(name) -> the HIP run's tensor(s) of that name (fp64, NCHW) or None
        self.report = {}                # teacher-forced mode: name -> compare_stored / compare_fp32 result
        self.bn = bool(cfg.get('batch_norm', False))
        self.pad = 1 if cfg.get('padding', False) else 0
        self.bd = int(cfg.get('block_depth', 2))
        self.do_res = bool(cfg.get('do_res', True))
        self.step = 3 if self.bn else 2

    # ---- rounding policy of the weights: bf16 GEMM copies from 16 input channels on (plan._pack), fp32 below (direct kernels)
    @staticmethod
    def _wq(w, cin):
        return rb(w) if cin % 16 == 0 else w.detach()

    # ---- teacher forcing ---------------------------------------------------------------------------------------------------
    def _st(self, name, value, chans=None):
        """A stored bf16 tensor: compared with the HIP run's and replaced by it (teacher-forced mode); chans: compare / take
        only this channel range (the rest stays the emulation's own)."""
        t = self.teacher(name) if self.teacher is not None else None
        if t is None:
            return value
        if chans is not None:
            lo, hi = chans
            self.report[name] = compare_stored(value[:, lo:hi], t[:, lo:hi])
            out = value.clone()
            out[:, lo:hi] = t[:, lo:hi]
            return out
        assert tuple(t.shape) == tuple(value.shape), (name, tuple(t.shape), tuple(value.shape))
        self.report[name] = compare_stored(value, t)
        return t

    def _vec(self, name, values):
        """fp32 statistic vectors (tuple): compared with the HIP run's and replaced by them."""
        t = self.teacher(name) if self.teacher is not None else None
        if t is None:
            return values
        assert len(t) == len(values), name
        for i, (e, h) in enumerate(zip(values, t)):
            self.report['%s[%d]' % (name, i)] = compare_fp32(e, h)
        return tuple(h.double() for h in t)

    # ---- forward -------------------------------------------------------------------------------------------------------
    def _relu(self, name, v):
        nat = v > 0
        mask = nat
        if self.choices is not None and name in self.choices['relu']:
            mask = self.choices['relu'][name]
            assert tuple(mask.shape) == tuple(v.shape), (name, tuple(mask.shape), tuple(v.shape))
            flip = nat != mask
            nf = int(flip.sum())
            self.info['relu_flips'] += nf
            if nf:
                self.info['max_margin'] = max(self.info['max_margin'], float(v[flip].abs().max()) / max(_rms(v), 1e-300))
        self.info['relu_total'] += v.numel()
        return rb(torch.clamp(v, min=0.0)) * mask.to(v.dtype), mask

    def _block_fwd(self, prefix, xin):
        P = self.P
        convs = []
        cur, aff = xin, None
        for d in range(self.bd):
            wname = '%s.block.%d' % (prefix, d * self.step)
            w, b = P[wname + '.weight'].detach(), P[wname + '.bias'].detach()
            op = rb(cur * aff[0].view(1, -1, 1, 1) + aff[1].view(1, -1, 1, 1)) if aff is not None else cur
            wq = self._wq(w, w.shape[1])
            v = F.conv2d(op, wq, b, padding=self.pad)
            rname = '%s.block.%d' % (prefix, d * self.step + 1)
            r, mask = self._relu(rname, v)
            if self.teacher is not None:
                r = self._st('r:' + rname, r)
                mask = r > 0                              # what the product's backward tests
            rec = dict(wname=wname, op=op, wq=wq, r=r, mask=mask, bn=None, rname=rname)
            self.acts['%s.block.%d' % (prefix, d * self.step + 1)] = r
            naff = None
            if self.bn:
                bname = '%s.block.%d' % (prefix, d * self.step + 2)
                gamma, beta = P[bname + '.weight'].detach(), P[bname + '.bias'].detach()
                mean = r.mean(dim=(0, 2, 3))
                var = torch.clamp((r * r).mean(dim=(0, 2, 3)) - mean * mean, min=0.0)
                invstd = 1.0 / torch.sqrt(var + BN_EPS)
                sc, sh, mean_, invstd_ = self._vec('bn:' + bname, (f32(gamma * invstd), f32(beta - mean * gamma * invstd), f32(mean), f32(invstd)))
                naff = (sc, sh)
                rec['bn'] = dict(name=bname, gamma=gamma, mean=mean_, invstd=invstd_, count=r.numel() // r.shape[1])
            convs.append(rec)
            cur, aff = r, naff
        last = cur * aff[0].view(1, -1, 1, 1) + aff[1].view(1, -1, 1, 1) if aff is not None else cur
        res = None
        if self.do_res:
            rw, rbias = P[prefix + '.res_conv1x1.weight'].detach(), P[prefix + '.res_conv1x1.bias'].detach()
            rwq = self._wq(rw, rw.shape[1])
            out = rb(F.conv2d(xin, rwq, rbias) + last)
            res = dict(wq=rwq)
        else:
            out = rb(last)
        out = self._st('out:' + prefix, out)
        return out, dict(prefix=prefix, xin=xin, convs=convs, res=res)

    def _pool(self, level, x):
        top, nat = F.max_pool2d(x, 2, return_indices=True)
        idx = nat
        if self.choices is not None and level in self.choices['pool']:
            idx = self.choices['pool'][level]
            forced = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
            flip = nat != idx
            nf = int(flip.sum())
            self.info['pool_flips'] += nf
            if nf:
                self.info['max_margin'] = max(self.info['max_margin'], float((top - forced)[flip].max()) / max(_rms(x), 1e-300))
            top = forced
        return top, idx

    def run(self, x, loss_fn):
        """x: the network input (fp32 values); loss_fn(seg, heat or None) -> scalar (fp64 autograd).  Returns a dict with
        grads {parameter name: tensor or None}, seg, heat, loss, info."""
        cfg, P = self.cfg, self.P
        depth = cfg['depth']
        x = x.double()
        # ---------------------------------------------------------------- down path
        downs = []
        cur = x
        for i in range(depth):
            out, rec = self._block_fwd('down_path.%d' % i, cur)
            drec = dict(block=rec, out=out)
            if i != depth - 1:
                if cfg.get('max_pool', True):
                    cur, idx = self._pool(i, out)
                    drec['pool_idx'] = idx
                else:
                    dw, db = P['downsample_convs.%d.weight' % i].detach(), P['downsample_convs.%d.bias' % i].detach()
                    dwq = self._wq(dw, dw.shape[1])
                    cur = self._st('nxt:%d' % i, rb(F.conv2d(out, dwq, db, stride=2)))
                    drec['dwq'] = dwq
                drec['nxt_shape'] = cur.shape
            downs.append(drec)
        # ---------------------------------------------------------------- up path
        ups = []
        u = downs[-1]['out']
        for j, i in enumerate(reversed(range(depth - 1))):
            name = 'up_path.%d' % j
            uw, ub = P[name + '.up.weight'].detach(), P[name + '.up.bias'].detach()
            uwq = self._wq(uw, uw.shape[0])
            up = self._st('up:%d' % j, rb(F.conv_transpose2d(u, uwq, ub, stride=2)))
            bridge = downs[i]['out']
            th, tw = up.shape[2], up.shape[3]
            oy, ox = (bridge.shape[2] - th) // 2, (bridge.shape[3] - tw) // 2          # unet.py:248-252
            cat = torch.cat([up, bridge[:, :, oy:oy + th, ox:ox + tw]], 1)
            out, rec = self._block_fwd(name + '.conv_block', cat)
            ups.append(dict(name=name, block=rec, u=u, uwq=uwq, crop=(oy, ox, th, tw), level=i, Ci=up.shape[1]))
            u = out
        feat = u
        # ---------------------------------------------------------------- heads (autograd on this small piece)
        NC, L = cfg['n_classes'], cfg.get('num_lands', 0)
        Fc = feat.shape[1]
        wseg = P['seg_conv.weight'].detach().clone().requires_grad_(True)
        w1 = P['lands_1x1.0.weight'].detach().clone().requires_grad_(True) if L > 0 else None
        w2 = P['lands_1x1.1.weight'].detach().clone().requires_grad_(True) if (L > 0 and 'lands_1x1.1.weight' in P) else None
        fx = feat.detach().clone().requires_grad_(True)
        logits = F.conv2d(fx, wseg)
        seg = torch.softmax(logits, dim=1) if cfg.get('do_soft_max', True) else logits
        heat = mid = None
        if L > 0:
            mid = F.conv2d(torch.cat((fx, logits), dim=1), w1)
            heat = F.conv2d(mid, w2) if w2 is not None else mid
        outs = [seg] + ([heat] if L > 0 else [])
        # the loss and its gradient with respect to the network outputs -- teacher forced: at the HIP run's own outputs
        t_out = self.teacher('outputs') if self.teacher is not None else None
        if t_out is not None:
            self.report['seg'] = compare_fp32(seg.detach(), t_out[0])
            if L > 0:
                self.report['heat'] = compare_fp32(heat.detach(), t_out[1])
            leaf = [t.detach().clone().requires_grad_(True) for t in t_out[:len(outs)]]
        else:
            leaf = [o.detach().clone().requires_grad_(True) for o in outs]
        loss = loss_fn(leaf[0], leaf[1] if L > 0 else None)
        douts = list(torch.autograd.grad(loss, leaf))
        t_dout = self.teacher('doutputs') if self.teacher is not None else None
        if t_dout is not None:
            for nm, e, h in zip(('dseg', 'dheat'), douts, t_dout):
                self.report[nm] = compare_fp32(e, h)
            douts = [h.double() for h in t_dout[:len(outs)]]
        G = {k: None for k in P}
        # the matrix-core head kernels (head.hip: head_mfma_ok -- bf16 features, F = 32, two landmark layers or none, small head)
        mfma_head = Fc == 32 and NC <= 8 and (L == 0 or (w2 is not None and w1.shape[0] <= 24 and L <= 16))
        if mfma_head:
            wanted = [fx, logits] + ([mid, heat] if L > 0 else [])
            got = torch.autograd.grad(outs, wanted, grad_outputs=douts)
            dfeat, dlog = got[0], got[1]
            G['seg_conv.weight'] = torch.einsum('nchw,nfhw->cf', rb(dlog), feat).view_as(wseg)
            if L > 0:
                dmid, dheat = got[2], got[3]
                xu = torch.cat((feat, rb(logits.detach())), dim=1)
                G['lands_1x1.0.weight'] = torch.einsum('nchw,nfhw->cf', rb(dmid), xu).view_as(w1)
                G['lands_1x1.1.weight'] = torch.einsum('nchw,nfhw->cf', rb(dheat), rb(mid.detach())).view_as(w2)
        else:
            wl = [wseg] + ([w1] if w1 is not None else []) + ([w2] if w2 is not None else [])
            got = torch.autograd.grad(outs, [fx] + wl, grad_outputs=douts)
            dfeat = got[0]
            G['seg_conv.weight'] = got[1]
            if w1 is not None:
                G['lands_1x1.0.weight'] = got[2]
            if w2 is not None:
                G['lands_1x1.1.weight'] = got[3]
        dout = self._st('dfeat', rb(dfeat))
        # ---------------------------------------------------------------- backward: up path, last block first
        dbridge = {}
        for j in reversed(range(len(ups))):
            rec = ups[j]
            Ci = rec['Ci']
            # (the bridge half of dcat is accumulated onto later -- the down-sampling gradient -- so the HIP run's value of it
            # is checked where it is final: 'dout:down_path.i' below; until then the emulation's own value stands in)
            dcat = self._block_bwd(rec['block'], dout, G, need_dxin=True, dxin_chans=(0, Ci))
            dy = dcat[:, :Ci]
            dbridge[rec['level']] = (dcat[:, Ci:], rec['crop'])
            name = rec['name']
            G[name + '.up.bias'] = dy.sum(dim=(0, 2, 3))
            du, gw = _conv_bwd(dy.contiguous(), rec['u'], rec['uwq'], 2, 0, transposed=True)
            G[name + '.up.weight'] = gw
            dout = self._st('du:%d' % j, rb(du))
        # ---------------------------------------------------------------- backward: down path, deepest block first
        dnxt = None
        for i in reversed(range(depth)):
            drec = downs[i]
            out = drec['out']
            if i != depth - 1:
                db_, (oy, ox, th, tw) = dbridge[i]
                dout = torch.zeros_like(out)
                dout[:, :, oy:oy + th, ox:ox + tw] = db_
                if cfg.get('max_pool', True):
                    idx = drec['pool_idx']
                    add = torch.zeros_like(out).flatten(2).scatter_(2, idx.flatten(2), dnxt.flatten(2)).view_as(out)
                    dout = self._st('dout:down_path.%d' % i, rb(dout + add))
                else:
                    wname = 'downsample_convs.%d' % i
                    G[wname + '.bias'] = dnxt.sum(dim=(0, 2, 3))
                    di, gw = _conv_bwd(dnxt, out, drec['dwq'], 2, 0)
                    G[wname + '.weight'] = gw
                    dout = self._st('dout:down_path.%d' % i, rb(dout + di))
            dnxt = self._block_bwd(drec['block'], dout, G, need_dxin=i > 0)
        return dict(grads=G, seg=seg.detach(), heat=None if heat is None else heat.detach(), loss=float(loss.detach()), info=self.info,
                    report=self.report)

    # ---- backward of one block (plan.py: block.backward) -------------------------------------------------------------------
    def _block_bwd(self, rec, dout, G, need_dxin, dxin_chans=None):
        prefix, xin, convs = rec['prefix'], rec['xin'], rec['convs']
        if self.do_res:
            _, gw = _conv_bwd(dout, xin, rec['res']['wq'], 1, 0, want_input=False)
            G[prefix + '.res_conv1x1.weight'] = gw
            G[prefix + '.res_conv1x1.bias'] = dout.sum(dim=(0, 2, 3))
        g = dout
        dxin = None
        for d in reversed(range(self.bd)):
            cv = convs[d]
            r, mask = cv['r'], cv['mask'].to(torch.float64)
            if cv['bn'] is not None:
                bnr = cv['bn']
                cnt = float(bnr['count'])
                sdy, sdyr = g.sum(dim=(0, 2, 3)), (g * r).sum(dim=(0, 2, 3))
                mean, invstd, gamma = bnr['mean'], bnr['invstd'], bnr['gamma']
                sdyx = invstd * (sdyr - mean * sdy)
                G[bnr['name'] + '.weight'] = sdyx
                G[bnr['name'] + '.bias'] = sdy
                s = gamma * invstd
                c1, c2 = sdy / cnt, sdyx / cnt
                A, B, Cc = self._vec('coef:' + cv['rname'], (f32(s), f32(-s * c2 * invstd), f32(-s * c1 + s * c2 * invstd * mean)))
                v = lambda t: t.view(1, -1, 1, 1)
                dpre = rb(mask * f32(v(A) * g + f32(v(B) * r + v(Cc))))       # two fp32 fused multiply-adds, then the bf16 rounding
            else:
                dpre = mask * g
            # (where the data-gradient kernel writes this operand for the weight gradient -- dfl_conv_args.x_out -- it is a stored tensor)
            dpre = self._st('dmat:' + cv['rname'], dpre)
            G[cv['wname'] + '.bias'] = dpre.sum(dim=(0, 2, 3))
            want_in = d > 0 or need_dxin
            di, gw = _conv_bwd(dpre, cv['op'], cv['wq'], 1, self.pad, want_input=want_in)
            G[cv['wname'] + '.weight'] = gw
            if d > 0:
                g = self._st('g:' + convs[d - 1]['rname'], rb(di))
            elif need_dxin:
                dxin = rb(di)
                if self.do_res:
                    dr, _ = _conv_bwd(dout, xin, rec['res']['wq'], 1, 0)
                    dxin = rb(dxin + dr)
                dxin = self._st('dxin:' + prefix, dxin, chans=dxin_chans)
        return dxin
