"""Gradient bars of the GPU tests: how accurate a gradient of the HIP path has to be.  Test infrastructure (imports oracle/).

Two facts shape the comparison (both measured, docs/experiments/flip_probe.py and tools/calib_floors.py):

1. ACTIVATION PATTERN.  A ReLU whose pre-activation lies within the arithmetic's rounding noise of zero, or a 2x2 pooling
   window whose two largest values lie that close together, may resolve differently in the HIP run and in the fp64 oracle.
   Both are valid roundings of the same network, but the two gradients then differ by a whole element of an upstream
   gradient, whatever the noise level: ONE flipped ReLU in the 2x32x32 fixture `tiny_bd3_nosm` moves its 8-element
   `down_path.0.res_conv1x1.weight` gradient by 2.1e-3 and gradients of `tiny_sc_l14` by up to 6e-2 (bf16x3 products,
   forward deviation 5e-6) -- with the oracle run ON THE HIP RUN'S PATTERN the same tensors agree to 4e-6 ... 2e-5.  So the
   reference gradient is the fp64 oracle forced to the masks and pooling choices of the HIP run under test (hip_choices /
   forced_choices), and every flipped decision is checked to be one the noise can flip (its margin in the oracle is tiny).

2. SENSITIVITY.  With the pattern fixed the network is a smooth function and rounding noise of relative size eps in every
   convolution result moves a gradient tensor by  eps x S[tensor]  -- S is large for tensors behind BatchNorm cancellations
   (a small difference of large sums).  S is a property of the PROBLEM, not of the kernels: tools/calib_floors.py measures
   it in the build container (fp64 oracle, Gaussian noise of relative size eps injected into every convolution's output,
   data gradient and weight / bias gradient; 16 seeds; pattern frozen) and commits it as tests/golden/floors/<problem>.npz,
   together with the same for BatchNorm results at the fp32 rounding level (BatchNorm is fp32 in every mode).
   A tensor's bar is   K x sqrt((eps_eff x S_conv)^2 + S_bn^2) + abs   with
     eps_eff = max(the arithmetic's measured per-convolution error (conv_rel_error: one HIP convolution against fp64),
                   the noise level at which the oracle's forward output moves as far as the HIP forward is off it);
     K = 8 per tensor, 4 for the whole gradient (the injected noise is a model; measured error / (bar / K): see the
     calibration table printed by tools/calib_floors.py --report and the ratios the tests print);
     abs = 2e-6 + 4 x 2^-24 x sqrt(pixels): fp32 rounding of the result and of its accumulation over all pixels.
   Nothing random decides a verdict at test time: no seeds are drawn and no noisy passes run on the GPU box; the only
   oracle work there is one forward + backward in fp64 per (problem, arithmetic) on the forced pattern.
"""
import contextlib
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 2.0 ** -23          # fp32 rounding of the BatchNorm results
EPS_REF = 1.0e-6             # convolution noise level the committed sensitivities were measured at
K_TENSOR, K_WHOLE, ABS = 8.0, 4.0, 2.0e-6
FLOOR_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'floors')


def _rms(t):
    return float(t.detach().double().pow(2).mean().sqrt())


def rel_l2(actual, ref):
    a = np.asarray(actual, dtype=np.float64)
    r = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-300))


# ---- the activation pattern of a HIP run, and the oracle forced to a pattern -------------------------------------------
def train_plan(net):
    """The plan the last training forward of `net` ran on (its activations are what backward read)."""
    plan = net._last_train_plan() if net._last_train_plan is not None else None
    assert plan is not None, 'no training forward has run'
    return plan


def hip_choices(plan):
    """{'relu': {module name: bool mask [N,C,H,W]}, 'pool': {level: int64 flat indices [N,C,H/2,W/2]}} of the forward pass
    whose activations the plan holds (plan.relu_out / plan.pool_in: the tensors its backward pass reads)."""
    relu = {k: (plan.act_nchw(a) > 0).cpu() for k, a in plan.relu_out.items()}
    pool = {}
    for lvl, a in plan.pool_in.items():
        pool[lvl] = F.max_pool2d(plan.act_nchw(a), 2, return_indices=True)[1].cpu()
    return {'relu': relu, 'pool': pool}


def natural_choices(onet, run):
    """The oracle's own pattern on a problem (run(net) -> (loss, out)): what tools/calib_floors.py freezes."""
    relu, pool, handles = {}, {}, []
    for name, m in onet.named_modules():
        if isinstance(m, nn.ReLU):
            handles.append(m.register_forward_hook(lambda mod, inp, out, name=name: relu.__setitem__(name, (inp[0].detach() > 0))))
    prev = onet.pool_override

    def rec(level, x):
        y, idx = F.max_pool2d(x, 2, return_indices=True)
        pool[level] = idx
        return y
    onet.pool_override = rec
    try:
        with torch.no_grad():
            run(onet)
    finally:
        onet.pool_override = prev
        for h in handles:
            h.remove()
    return {'relu': relu, 'pool': pool}


@contextlib.contextmanager
def forced_choices(onet, choices):
    """Run `onet` (oracle/ref_cpu.OracleUNet) with the ReLU masks and pooling choices of `choices`.  Yields a dict that
    receives: relu_flips / pool_flips -- decisions whose natural outcome differs from the forced one; relu_total;
    max_margin -- the largest |pre-activation| / rms(pre-activation of that layer) over the flipped ReLUs and the largest
    (winner - forced element) / rms(pool input) over the flipped windows: how far from undecided the oracle was there."""
    info = {'relu_flips': 0, 'pool_flips': 0, 'relu_total': 0, 'max_margin': 0.0}
    handles = []
    mods = dict(onet.named_modules())
    for name, mask in choices['relu'].items():
        m = mods[name]
        assert isinstance(m, nn.ReLU), name

        def hook(mod, inp, out, mask=mask):
            x = inp[0]
            assert tuple(x.shape) == tuple(mask.shape), (tuple(x.shape), tuple(mask.shape))
            xd = x.detach()
            flip = (xd > 0) != mask
            nf = int(flip.sum())
            info['relu_flips'] += nf
            info['relu_total'] += mask.numel()
            if nf:
                info['max_margin'] = max(info['max_margin'], float(xd[flip].abs().max()) / max(_rms(xd), 1e-300))
            return x * mask.to(x.dtype)
        handles.append(m.register_forward_hook(hook))
    prev = onet.pool_override
    if choices['pool']:
        def pool(level, x):
            idx = choices['pool'][level]
            xd = x.detach()
            top, nat_idx = F.max_pool2d(xd, 2, return_indices=True)
            forced = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
            flip = nat_idx != idx
            nf = int(flip.sum())
            info['pool_flips'] += nf
            if nf:
                info['max_margin'] = max(info['max_margin'], float((top - forced.detach())[flip].max()) / max(_rms(xd), 1e-300))
            return forced
        onet.pool_override = pool
    try:
        yield info
    finally:
        onet.pool_override = prev
        for h in handles:
            h.remove()


# ---- noise injection (used by tools/calib_floors.py only: nothing noisy runs inside the tests) --------------------------
class _Noise:
    def __init__(self, eps, seed):
        self.eps = eps
        self.gen = torch.Generator().manual_seed(seed)

    def __call__(self, t):
        if t is None or self.eps == 0.0:
            return t
        return t + (self.eps * _rms(t)) * torch.randn(t.shape, generator=self.gen, dtype=t.dtype)


def gradients(onet, run, eps_conv=0.0, eps_bn=0.0, seed=0):
    """({name: gradient or None}, forward output) of run(onet) = (loss, out), with Gaussian noise of relative size eps_conv
    on every convolution's output / data gradient / weight and bias gradient and of size eps_bn on every BatchNorm's."""
    handles = []

    def attach(mods, noise):
        for m in mods:
            handles.append(m.register_forward_hook(lambda mod, inp, out: noise(out)))
            handles.append(m.register_full_backward_hook(lambda mod, gin, gout: tuple(noise(g) for g in gin)))
            if getattr(m, 'weight', None) is not None:
                handles.append(m.weight.register_hook(lambda g: noise(g)))
            if getattr(m, 'bias', None) is not None:
                handles.append(m.bias.register_hook(lambda g: noise(g)))
    if eps_conv > 0.0:
        attach([m for m in onet.modules() if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))], _Noise(eps_conv, seed))
    if eps_bn > 0.0:
        attach([m for m in onet.modules() if isinstance(m, nn.BatchNorm2d)], _Noise(eps_bn, seed + 7919))
    try:
        onet.zero_grad()
        loss, out = run(onet)
        loss.backward()
        grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in onet.named_parameters()}
    finally:
        for h in handles:
            h.remove()
        onet.zero_grad()
    return grads, out.detach().double().clone()


# ---- the check -----------------------------------------------------------------------------------------------------------
def load_floor(key):
    path = os.path.join(FLOOR_DIR, key + '.npz')
    if not os.path.exists(path):
        raise FileNotFoundError('%s is missing: run tools/calib_floors.py %s in the build container and commit the file' % (path, key))
    f = np.load(path, allow_pickle=False)
    names = [str(n) for n in f['names']]
    return {'s_conv': dict(zip(names, f['s_conv'].tolist())), 's_bn': dict(zip(names, f['s_bn'].tolist())),
            'gnorm': dict(zip(names, f['gnorm'].tolist())),
            'fwd_conv': float(f['fwd_conv']), 'fwd_bn': float(f['fwd_bn']), 'seeds': int(f['seeds'])}


def margin_bar(eps_eff, cap=0.3):
    """How far from undecided (in units of the layer's rms) a flipped ReLU / pooling decision may be in the oracle: noise of
    relative size eps per convolution accumulates over the up to ~25 layers in front of a decision and is not Gaussian
    in its tails -- 100 x eps (measured: 30 x eps in the bf16x3 and bf16 storage arithmetics at batch 16), at least 1e-4,
    never more than 0.3 of the layer's rms."""
    return min(max(1.0e-4, 100.0 * eps_eff), cap)


EPS_CAP = 4.0            # eps_eff <= EPS_CAP x the arithmetic's own measured convolution error: the bars do not grow with the run's deviation
FWD_CAP = 8.0            # forward output (relative L2, same pattern) <= max(1e-4, FWD_CAP x that error)
FLIP_CAP = 4.0           # fraction of decisions that differ from the oracle's <= max(1e-5, FLIP_CAP x that error)


class GradientCheck:
    """One problem (tests/problems.py): the fp64 oracle, its clean forward output, and the committed sensitivities."""

    def __init__(self, problem):
        self.problem = problem
        self.floor = load_floor(problem.key)
        self.net = problem.oracle64()
        with torch.no_grad():
            loss, out = problem.run(self.net)                                 # natural-pattern forward: label masks, 1e-4 bars
        self.out, self.loss = out.detach().double().clone(), float(loss)
        self.heat = None if problem.last_heat is None else problem.last_heat.double().clone()
        pixels = self.out.numel() / max(self.out.shape[1], 1)
        # fp32 rounding of the result itself + the random walk of an fp32 accumulation over the P pixels a weight gradient of
        # the widest level sums (2^-24 sqrt(P); x4: sums of mixed sign cancel, the walk is relative to sum |x|)
        self.abs_term = ABS + 4.0 * 2.0 ** -24 * pixels ** 0.5

    def reference(self, plan):
        """(gradients, forward output, info) of the oracle on the pattern of the HIP run that `plan` holds."""
        with forced_choices(self.net, hip_choices(plan)) as info:
            grads, out = gradients(self.net, self.problem.run)
        return grads, out, info

    def bars(self, d_fwd, eps_conv):
        """(eps_eff, {name: bar, '*': whole-gradient bar}) for a run whose forward output is d_fwd (relative L2) off the
        oracle's on the same pattern and whose arithmetic has the measured per-convolution error eps_conv."""
        fl = self.floor
        d_conv = max(d_fwd ** 2 - fl['fwd_bn'] ** 2, 0.0) ** 0.5
        eps_eff = min(max(eps_conv, d_conv / max(fl['fwd_conv'], 1e-300)), EPS_CAP * eps_conv)
        bars = {}
        for k, s in fl['s_conv'].items():
            kk = K_WHOLE if k == '*' else K_TENSOR
            bars[k] = kk * ((eps_eff * s) ** 2 + fl['s_bn'][k] ** 2) ** 0.5 + self.abs_term
        return eps_eff, bars

    def whole_error(self, net, hip_out):
        """Free-running distance of a run from the CLEAN fp64 oracle on the run's own pattern: relative L2 of the forward output and
        of the whole gradient, and the pattern statistics -- a diagnostic for arithmetics whose parity gate is elsewhere (bf16
        storage: tests/test_gpu_bf16_stepwise.py)."""
        ref, out, info = self.reference(train_plan(net))
        num = den = 0.0
        for k, p in net.named_parameters():
            if ref[k] is None:
                continue
            num += float((p.grad.detach().double().cpu() - ref[k].double()).pow(2).sum())
            den += float(ref[k].double().pow(2).sum())
        return {'whole': (num / max(den, 1e-300)) ** 0.5, 'd_fwd': rel_l2(hip_out.detach().double().cpu().numpy(), out.numpy()), 'info': info}

    def check(self, net, hip_out, eps_conv, what='', margin_cap=0.3):
        """Assert every parameter gradient of `net` (a dfl_amd.UNet after backward) inside its bar.  Returns a dict: worst
        (error / bar), whole (relative L2 of the whole gradient), eps_eff, bars, ref (the reference gradients), info."""
        plan = train_plan(net)
        ref, out, info = self.reference(plan)
        d_fwd = rel_l2(hip_out.detach().double().cpu().numpy(), out.numpy())
        eps_eff, bars = self.bars(d_fwd, eps_conv)
        # fixed per-arithmetic bars on the forward deviation and on the number of differing decisions (nothing below
        # may loosen itself on the deviation of the run under test)
        assert d_fwd <= max(1.0e-4, FWD_CAP * eps_conv), '%sforward output %.3e (relative L2, same pattern) off the oracle: more than %g x the ' \
            'arithmetic\'s convolution error %.2e' % (what, d_fwd, FWD_CAP, eps_conv)
        nflip, ndec = info['relu_flips'] + info['pool_flips'], max(info['relu_total'], 1)
        assert nflip <= max(3, max(1.0e-5, FLIP_CAP * eps_conv) * ndec), '%s%d of %d ReLU / pooling decisions differ from the oracle (convolution error ' \
            '%.2e)' % (what, nflip, ndec, eps_conv)
        assert info['max_margin'] <= margin_bar(eps_eff, margin_cap), \
            '%sa decision differs from the oracle where the oracle is not undecided: margin %.3e of the layer rms (bar %.3e at ' \
            'conv noise %.2e; %d ReLU + %d pooling decisions differ)' % (what, info['max_margin'], margin_bar(eps_eff, margin_cap), eps_eff,
                                                                          info['relu_flips'], info['pool_flips'])
        worst, num_all, den_all = 0.0, 0.0, 0.0
        for k, p in net.named_parameters():
            r = ref[k]
            if r is None:
                assert p.grad is None, '%s%s: gradient where the reference has none' % (what, k)
                continue
            assert p.grad is not None, '%s%s: no gradient' % (what, k)
            got = p.grad.detach().double().cpu()
            num = float((got - r.double()).pow(2).sum())
            den = max(float(r.double().pow(2).sum()), 1e-300)
            gn, gn_all = self.floor['gnorm'][k], self.floor['gnorm']['*']
            if den ** 0.5 < 1e-6 * gn_all and gn < 1e-6 * gn_all:
                # A tensor whose EXACT gradient is zero (the bias of lands_block.0: a constant offset of the heat maps, which the
                # NCC loss ignores): the oracle leaves 1e-17 there, fp32 arithmetic 1e-9 -- a relative error means nothing.  The
                # committed relative sensitivity (huge) x the committed norm (tiny) is the ABSOLUTE sensitivity: the bar is made of it.
                assert num ** 0.5 <= bars[k] * gn + 1e-12, \
                    '%s%s: the exact gradient is zero; got norm %.3e, absolute bar %.3e' % (what, k, num ** 0.5, bars[k] * gn)
                num_all += num
                continue
            e = (num / den) ** 0.5
            num_all += num
            den_all += den
            worst = max(worst, e / bars[k])
            assert e <= bars[k], '%s%s: gradient relative L2 error %.3e > bar %.3e (conv noise %.2e, sensitivity %.3g, BatchNorm ' \
                                 'term %.2e, abs %.1e; %d ReLU / %d pooling decisions forced)' % (
                                     what, k, e, bars[k], eps_eff, self.floor['s_conv'][k], self.floor['s_bn'][k], self.abs_term,
                                     info['relu_flips'], info['pool_flips'])
        whole = (num_all / den_all) ** 0.5
        assert whole <= bars['*'], '%swhole gradient: relative L2 error %.3e > bar %.3e (conv noise %.2e)' % (what, whole, bars['*'], eps_eff)
        return {'worst': worst, 'whole': whole, 'eps_eff': eps_eff, 'bars': bars, 'ref': ref, 'info': info, 'd_fwd': d_fwd}


# ---- the bf16 STORAGE arithmetic against its own fp64 emulation (oracle/bf16_emu.py) ----------------------------------------
# The emulation rounds to bf16 wherever the product stores or stages a bf16 value, so what separates a HIP gradient from it
# (on the HIP run's activation pattern) is fp32 accumulation order plus rare one-ulp bf16 flips -- measured on MI355X at the
# level of the bf16x3 arithmetic.  Fixed bars, nothing calibrated on the run under test: the forward output has
# to be within BF16S_FWD_BAR of the emulation's, a flipped ReLU / pooling decision within BF16S_MARGIN_BAR of undecided, and
# the per-convolution noise level the gradient bars are made of is BF16S_EPS (not derived from the run's own deviation).
BF16S_FWD_BAR = 1.0e-3
BF16S_MARGIN_BAR = 1.0e-3
BF16S_EPS = 3.0e-5
BF16S_WHOLE_BAR = 1.0e-3
BF16S_FLIP_FRAC = 1.0e-4


def emulated_reference(gc, plan):
    """oracle/bf16_emu.py on gc.problem with the ReLU masks and pooling choices of the HIP run `plan` holds."""
    from oracle import bf16_emu as E
    emu = E.Bf16Emulation(gc.net, dict(gc.problem.cfg), choices=hip_choices(plan))
    return emu.run(gc.problem.x, gc.problem.loss_of)


def check_bf16_storage(gc, net, hip_out, hip_loss=None, hip_heat=None, what='', measure_only=False):
    """Assert a bf16-storage run of `net` (dfl_amd.UNet after backward) against the fp64 emulation of that arithmetic on the run's
    own pattern: forward, loss, every gradient tensor, the whole gradient.  Returns the measurements."""
    plan = train_plan(net)
    assert plan.bf16, 'not a bf16-storage plan'
    ref = emulated_reference(gc, plan)
    info = ref['info']
    d_fwd = rel_l2(hip_out.detach().double().cpu().numpy(), ref['seg'].numpy())
    fl = gc.floor
    bars = {}
    for k, s in fl['s_conv'].items():
        kk = K_WHOLE if k == '*' else K_TENSOR
        bars[k] = kk * ((BF16S_EPS * s) ** 2 + fl['s_bn'][k] ** 2) ** 0.5 + gc.abs_term
    bars['*'] = min(bars['*'], BF16S_WHOLE_BAR)
    flip_frac = (info['relu_flips'] + info['pool_flips']) / max(info['relu_total'], 1)
    res = {'d_fwd': d_fwd, 'info': info, 'bars': bars, 'flip_frac': flip_frac, 'ref': ref}
    if hip_heat is not None and ref['heat'] is not None:
        res['d_heat'] = rel_l2(hip_heat.detach().double().cpu().numpy(), ref['heat'].numpy())
    if hip_loss is not None:
        res['d_loss'] = abs(float(hip_loss) - ref['loss']) / max(abs(ref['loss']), 1e-12)
    worst, worst_k, num_all, den_all = 0.0, None, 0.0, 0.0
    errs = {}
    for k, p in net.named_parameters():
        r = ref['grads'][k]
        if r is None:
            assert p.grad is None, '%s%s: gradient where the reference has none' % (what, k)
            continue
        assert p.grad is not None, '%s%s: no gradient' % (what, k)
        got = p.grad.detach().double().cpu()
        num = float((got - r).pow(2).sum())
        den = max(float(r.pow(2).sum()), 1e-300)
        gn, gn_all = fl['gnorm'][k], fl['gnorm']['*']
        num_all += num
        if den ** 0.5 < 1e-6 * gn_all and gn < 1e-6 * gn_all:        # exact gradient zero (see GradientCheck.check)
            errs[k] = ('abs', num ** 0.5, bars[k] * gn + 1e-12)
            continue
        den_all += den
        e = (num / den) ** 0.5
        errs[k] = ('rel', e, bars[k])
        if e / bars[k] > worst:
            worst, worst_k = e / bars[k], k
    res.update(worst=worst, worst_k=worst_k, whole=(num_all / den_all) ** 0.5, errs=errs)
    if measure_only:
        return res
    assert d_fwd <= BF16S_FWD_BAR, '%sforward output %.3e (relative L2) off the bf16 emulation (bar %.1e)' % (what, d_fwd, BF16S_FWD_BAR)
    if 'd_heat' in res:
        assert res['d_heat'] <= BF16S_FWD_BAR, '%sheat maps %.3e off the bf16 emulation' % (what, res['d_heat'])
    if 'd_loss' in res:
        assert res['d_loss'] <= BF16S_FWD_BAR, '%sloss %.3e off the bf16 emulation' % (what, res['d_loss'])
    assert info['max_margin'] <= BF16S_MARGIN_BAR, \
        '%sa decision differs from the emulation where it is not undecided: margin %.3e of the layer rms (bar %.1e; %d ReLU + %d ' \
        'pooling decisions differ)' % (what, info['max_margin'], BF16S_MARGIN_BAR, info['relu_flips'], info['pool_flips'])
    assert flip_frac <= BF16S_FLIP_FRAC, '%s%.2e of the decisions differ from the emulation (bar %.1e)' % (what, flip_frac, BF16S_FLIP_FRAC)
    for k, (kind, e, bar) in errs.items():
        assert e <= bar, '%s%s: gradient %s error %.3e > bar %.3e against the bf16 emulation (sensitivity %.3g)' % (
            what, k, 'absolute' if kind == 'abs' else 'relative L2', e, bar, fl['s_conv'][k])
    assert res['whole'] <= bars['*'], '%swhole gradient: relative L2 error %.3e > bar %.3e against the bf16 emulation' % (what, res['whole'], bars['*'])
    return res


_EPS_CACHE = {}


def conv_rel_error(mode_name, dev='cuda'):
    """Measured relative error (RMS of the error / RMS of the result) of ONE 3x3 convolution of libdfl_hip.so in the
    current product arithmetic against fp64, on a layer shaped like the network's (64 -> 64 channels, 48x48, batch 2):
    the per-convolution noise level of this arithmetic.  Cached per mode name."""
    if mode_name in _EPS_CACHE:
        return _EPS_CACHE[mode_name]
    import dfl_amd
    g = torch.Generator().manual_seed(7)
    net = dfl_amd.UNet(in_channels=64, n_classes=8, depth=1, wf=6, padding=True, batch_norm=False, do_res=False,
                       block_depth=1, num_lands=0, do_soft_max=False)
    x = torch.randn(2, 64, 48, 48, generator=g)
    w = net.down_path[0].block[0].weight.detach().double()
    b = net.down_path[0].block[0].bias.detach().double()
    ws = net.seg_conv.weight.detach().double()
    ref = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(x.double(), w, b, padding=1)), ws)
    net = net.to(dev).eval()
    with torch.no_grad():
        y = net(x.to(dev)).cpu().double()
    e = _rms(y - ref) / _rms(ref)
    if mode_name == 'fp32':
        # With exact products the error of a result is that of its fp32 accumulation, which grows with the number of terms:
        # this probe sums K = 576 products per output, the deepest layers of the paper network 9216 (sqrt(16) = 4 times the
        # rounding walk).  (The bf16-product modes are the other way round: per-product rounding averages out with K.)
        e *= 4.0
    _EPS_CACHE[mode_name] = e
    return e


_CHECKS = {}


def cached_check(key, make_problem):
    """One GradientCheck per problem and test session (the arithmetic modes a test is parametrised over share the oracle
    and its natural-pattern forward)."""
    if key not in _CHECKS:
        _CHECKS[key] = GradientCheck(make_problem())
    return _CHECKS[key]
