"""The bf16 STORAGE arithmetic (math mode 4, the mode bench.py's headline is quoted in) pinned op by op.

A free-running comparison of two bf16 implementations cannot be tight (rounding amplifies a one-fp32-ulp difference to a full
bf16 rounding within six layers, oracle/bf16_emu.py), so the whole training pass is checked STEP BY STEP instead: the fp64
emulation of the arithmetic recomputes every stored tensor and every fp32 result of the HIP pass from the HIP run's own
inputs of that step ("teacher forcing").  Then
  * every bf16 tensor the HIP pass stored (ReLU outputs, block outputs, down- / up-sampled tensors, every activation gradient)
    must be the correctly rounded value: it may differ on at most FRAC_BAR of the elements (an fp32 sum that lands on the other
    side of a rounding boundary; measured 1e-5 ... 7e-4), by REL_BAR in relative L2 (measured <= 1e-4), and no element may be
    further off than one bf16 ulp + EXCESS_BAR of the tensor's rms (the fp32 accumulation error of sums that cancel);
  * every fp32 result (BatchNorm scale / shift / mean / invstd, the backward coefficients, the network outputs, the loss
    gradient, EVERY parameter gradient) must agree to fp32 accumulation accuracy: FP32_BAR relative L2 (measured <= 1.6e-5).
  Two documented exceptions: tensors written by an ACCUMULATING epilogue (`dxin:` residual 1x1 data gradient on top of the 3x3
  one, `dout:` down-sampling gradient on top of the bridge gradient) were rounded twice, the first value is gone, and a one-ulp
  flip of a large first value that the sum cancels is many ulps of the result: EXCESS_BAR_TWICE; and the matrix-core heads
  multiply fp32 operands as hi + lo bf16 pairs (2^-17 per product, csrc/head_mfma.inc -- the accuracy of the bf16x3 arithmetic),
  which the emulation does not restate: `dfeat` and the three head weight gradients get HEAD_* bars.
Nothing is calibrated on the run under test and no bar depends on the arithmetic's own noise:
a wrong ReLU mask rule, a statistic taken from unrounded values or a 1 % error in one bias gradient fails here.
Reference semantics: train_test_code/unet.py:161-260 (forward), torch autograd at train.py:422 (backward)."""
import os

import numpy as np
import pytest
import torch

import dfl_amd
from dfl_amd import plan as P
from dfl_amd import _native as nat
import noise_floor as NF
import problems as PR
from oracle import bf16_emu as E
from gpu_common import hip_net, math_mode_set, DEV

pytestmark = pytest.mark.gpu

FRAC_BAR, REL_BAR, EXCESS_BAR, EXCESS_BAR_TWICE = 2.0e-3, 2.0e-4, 2.0e-3, 3.0e-2
FP32_BAR = 1.0e-4
HEAD_FRAC_BAR, HEAD_REL_BAR, HEAD_FP32_BAR = 2.0e-2, 5.0e-4, 5.0e-4
HEAD_GRADS = ('grad:seg_conv.weight', 'grad:lands_1x1.0.weight', 'grad:lands_1x1.1.weight')


def teacher_of(plan, outs, douts):
    def nchw(act):
        return plan.act_nchw(act).double().cpu()

    def get(name):
        if name == 'outputs':
            return [o.detach().double().cpu() for o in outs]
        if name == 'doutputs':
            return [d.detach().double().cpu() for d in douts]
        kind, _, rest = name.partition(':')
        if kind == 'r':
            return nchw(plan.relu_out[rest])
        if kind == 'bn':
            return [t.detach().double().cpu() for t in plan.dbg[name]]
        if kind == 'coef':
            if name not in plan.dbg:              # live statistics: the consumers derive the coefficients themselves (nothing is stored)
                return None
            c = plan.dbg[name].detach().double().cpu()
            return list(c.view(3, -1))
        if kind == 'up':
            cat = plan.dbg['cat:' + rest]
            return nchw(cat.chan_slice(0, cat.C // 2))
        act = plan.dbg.get(name)
        return None if act is None else nchw(act)
    return get


def stepwise(pr, what=''):
    """One training pass of the HIP path on problem `pr` in bf16 storage, every step checked against the emulation."""
    gc_net = pr.oracle64()
    with math_mode_set('bf16s'):
        P.KEEP_GRADS = True
        try:
            net = hip_net(pr)
            out = net(pr.x.to(DEV))
        finally:
            P.KEEP_GRADS = False
        outs = list(out) if isinstance(out, tuple) else [out]
        for o in outs:
            o.retain_grad()
        tseg = pr.tseg.to(DEV)
        if pr.theat is not None:
            theat = pr.theat.to(DEV)
            loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)(
                (dfl_amd.center_crop(outs[0], tseg.shape), dfl_amd.center_crop(outs[1], theat.shape)), (tseg, theat))
        else:
            loss = dfl_amd.DiceLoss2D(skip_bg=pr.skip_bg)(dfl_amd.center_crop(outs[0], tseg.shape), tseg)
        loss.backward()
        torch.cuda.synchronize()
        plan = NF.train_plan(net)
        assert plan.bf16 and plan.keep_grads
        emu = E.Bf16Emulation(gc_net, dict(pr.cfg), teacher=teacher_of(plan, outs, [o.grad for o in outs]))
        res = emu.run(pr.x, pr.loss_of)
    rep = res['report']
    # every operand tensor a data-gradient kernel wrote for its layer's weight gradient (dfl_conv_args.x_out) was compared
    nout = sum(1 for st in plan.bwd.structs if isinstance(st, nat.ConvArgs) and st.x_out)
    assert sum(1 for k in rep if k.startswith('dmat:')) == nout, (nout, [k for k in rep if k.startswith('dmat:')])
    rep['loss'] = {'kind': 'fp32', 'rel_l2': abs(loss.item() - res['loss']) / max(abs(res['loss']), 1e-12), 'max_rel': 0.0, 'n': 1}
    for k, p in net.named_parameters():
        e = res['grads'][k]
        if e is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        rep['grad:' + k] = E.compare_fp32(e, p.grad.detach().cpu())
    return rep, res


def summarize(rep):
    b = [(k, v) for k, v in rep.items() if v['kind'] == 'bf16']
    f = [(k, v) for k, v in rep.items() if v['kind'] == 'fp32']
    wb = max(b, key=lambda kv: kv[1]['max_ulps'])
    wf_ = max(b, key=lambda kv: kv[1]['frac'])
    wr = max(b, key=lambda kv: kv[1]['rel_l2'])
    w32 = max(f, key=lambda kv: kv[1]['rel_l2'])
    return ('%d stored tensors: worst %.2f ulps (%s), %.2e differing (%s), rel L2 %.2e (%s); %d fp32 results: worst rel L2 %.2e (%s)' % (
        len(b), wb[1]['max_ulps'], wb[0], wf_[1]['frac'], wf_[0], wr[1]['rel_l2'], wr[0], len(f), w32[1]['rel_l2'], w32[0]))


def assert_report(rep, gnorm_all, what):
    for k, v in rep.items():
        if v['kind'] == 'bf16':
            twice = k.startswith('dxin:') or k.startswith('dout:')          # written by an accumulating epilogue (see the module text)
            head = k == 'dfeat'
            fb, rbar = (HEAD_FRAC_BAR, HEAD_REL_BAR) if head else (FRAC_BAR, REL_BAR)
            xb = EXCESS_BAR_TWICE if twice else EXCESS_BAR
            assert v['frac'] <= fb, '%s%s: %.2e of the elements differ from the correctly rounded value (bar %.0e)' % (what, k, v['frac'], fb)
            assert v['rel_l2'] <= rbar, '%s%s: relative L2 %.2e off the correctly rounded tensor (bar %.0e)' % (what, k, v['rel_l2'], rbar)
            assert v['excess'] <= xb, '%s%s: an element is one bf16 ulp + %.2e of the rms off the correctly rounded value (bar %.0e; %.1f ulps)' % (
                what, k, v['excess'], xb, v['max_ulps'])
        else:
            bar = HEAD_FP32_BAR if k in HEAD_GRADS else FP32_BAR
            # a tensor whose exact value is (numerically) zero has no relative error: its share of the whole gradient counts
            if k.startswith('grad:') and v['norm'] < 1e-6 * gnorm_all:
                assert v['rel_l2'] * v['norm'] <= bar * gnorm_all, '%s%s' % (what, k)
                continue
            assert v['rel_l2'] <= bar, '%s%s: fp32 result %.2e (relative L2) off its definition (bar %.0e)' % (what, k, v['rel_l2'], bar)


KEYS = ['ragged__37x41__mp0', 'ragged__37x41__mp1', 'ragged__50x70__mp0', 'ragged__50x70__mp1', 'ragged__64x96__mp0', 'ragged__64x96__mp1',
        'paper__paper_sc_l14__b2', 'paper__paper_mp_l0__b2', 'paper__paper_sc_l0__b4', 'config3__b1']       # (batch 16 of the paper preset: tests/test_gpu_00_northstar.py)


@pytest.mark.parametrize('key', KEYS)
def test_every_step_of_a_bf16_storage_pass_is_its_definition(key):
    torch.set_num_threads(max(torch.get_num_threads(), min(64, os.cpu_count() or 32)))
    pr = PR.config3_one() if key == 'config3__b1' else PR.REGISTRY[key]()
    if pr is None:
        pytest.skip('rejected architecture')
    rep, res = stepwise(pr, key + ' ')
    print('%s: %s' % (key, summarize(rep)))
    gn = sum(float(g.pow(2).sum()) for g in res['grads'].values() if g is not None) ** 0.5
    assert_report(rep, gn, key + ' ')
