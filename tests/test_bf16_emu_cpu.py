"""oracle/bf16_emu.py (the fp64 restatement of the bf16 STORAGE arithmetic the GPU tests compare the headline mode with) is
itself pinned here: with its two rounding functions replaced by the identity it must BE the fp64 oracle -- same outputs, same
loss, every gradient equal to torch autograd of oracle/ref_cpu.OracleUNet (which tests/test_oracle_golden.py pins to the
reference).  With rounding on, it must sit at bf16 distance from the oracle and its stored tensors must be bf16 values."""
import numpy as np
import pytest
import torch

import noise_floor as NF
import problems as PR
from oracle import bf16_emu as E

KEYS = ['tiny__tiny_sc_l14', 'tiny__tiny_mp_l0', 'tiny__tiny_mp_l14', 'tiny__tiny_nobn_d1', 'tiny__tiny_nobn_nores', 'tiny__tiny_bd3_nosm',
        'ragged__37x41__mp0', 'ragged__50x70__mp1', 'largehead', 'random__3', 'random__7']


def _autograd(pr):
    net = pr.oracle64()
    loss, seg = pr.run(net)
    loss.backward()
    return net, float(loss), seg.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()}


@pytest.mark.parametrize('key', KEYS)
def test_emulation_without_rounding_is_the_oracle(key, monkeypatch):
    pr = PR.REGISTRY[key]()
    if pr is None:
        pytest.skip('rejected architecture')
    if not pr.cfg.get('padding', False) and pr.cfg.get('do_res', True):
        pytest.skip('not a valid architecture')
    net, loss, seg, grads = _autograd(pr)
    monkeypatch.setattr(E, 'rb', lambda t: t.double())
    monkeypatch.setattr(E, 'f32', lambda t: t.double())
    cfg = dict(pr.cfg)
    try:
        emu = E.Bf16Emulation(net, cfg)
    except NotImplementedError:
        pytest.skip('architecture outside the emulation')
    res = emu.run(pr.x, pr.loss_of)
    assert abs(res['loss'] - loss) <= 1e-12 * max(1.0, abs(loss))
    np.testing.assert_allclose(res['seg'].numpy(), seg.numpy(), rtol=1e-10, atol=1e-12)
    for k, g in grads.items():
        if g is None:
            assert res['grads'][k] is None, k
            continue
        assert res['grads'][k] is not None, k
        err = float((res['grads'][k] - g).norm()) / max(float(g.norm()), 1e-30)
        assert err < 1e-9 or float((res['grads'][k] - g).abs().max()) < 1e-14, (k, err)


def test_emulation_with_rounding_is_at_bf16_distance():
    pr = PR.REGISTRY['ragged__64x96__mp0']()
    net, loss, seg, grads = _autograd(pr)
    res = E.Bf16Emulation(net, dict(pr.cfg)).run(pr.x, pr.loss_of)
    d = NF.rel_l2(res['seg'].numpy(), seg.numpy())
    assert 1e-5 < d < 5e-2, d
    num = sum(float((res['grads'][k] - g).pow(2).sum()) for k, g in grads.items() if g is not None)
    den = sum(float(g.pow(2).sum()) for g in grads.values() if g is not None)
    assert 1e-4 < (num / den) ** 0.5 < 0.5


def test_forced_pattern_is_recorded():
    pr = PR.REGISTRY['tiny__tiny_mp_l14']()
    net = pr.oracle64()
    nat = NF.natural_choices(net, pr.run)
    # flip one decision of the first ReLU: the emulation must follow it and report it
    name = sorted(nat['relu'])[0]
    nat['relu'][name] = nat['relu'][name].clone()
    nat['relu'][name].view(-1)[0] ^= True
    res = E.Bf16Emulation(net, dict(pr.cfg), choices=nat).run(pr.x, pr.loss_of)
    assert res['info']['relu_flips'] >= 1 and res['info']['max_margin'] > 0
