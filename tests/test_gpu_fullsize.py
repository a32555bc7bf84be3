"""BASELINE configs[3] and configs[4] at their full image sizes, against the oracle run on the GPU box's host cores
(the oracle is pinned to the reference by tests/test_oracle_golden.py; a 768x768 training step is ~2 TFLOP and a
1440x1440 forward ~1 TFLOP on the CPU -- seconds).  Batch is reduced (2 resp. 1 image, 2 nets): the kernels, tile
configurations, 32-bit offsets and split decisions depend on the image size, not on the batch count.
"""
import os

import numpy as np
import pytest
import torch

import dfl_amd
from conftest import PAPER_CFGS
from oracle import ref_cpu as R
import noise_floor as NF
import problems as PR
from gpu_common import oracle64, label_mask, hip_net, hip_step, math_mode_set

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _pair(cfg, seed, randomize_bn=False):
    torch.manual_seed(seed)
    onet = R.OracleUNet(**cfg)
    if randomize_bn:
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for m in onet.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
                    m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
    net = dfl_amd.UNet(**cfg)
    net.load_state_dict(onet.state_dict())
    return net.to(DEV), onet


@pytest.mark.parametrize('mode', ['fp32', 'bf16x3', 'bf16s'])
def test_config3_736_training_step_matches_oracle(mode):
    """2x-downsampled 736x736 padded to 768 (configs[3]), paper U-Net, dual head: forward, loss and gradients, in the two
    parity modes and in the bf16 storage mode profiles/ quotes this configuration in."""
    torch.set_num_threads(max(torch.get_num_threads(), min(64, os.cpu_count() or 32)))
    gc = NF.cached_check('config3', PR.config3)
    pr = gc.problem
    with math_mode_set(mode):
        net = hip_net(pr)
        out, seg, loss = hip_step(pr, net)
        heat = out[1]
        res = gc.check(net, seg, NF.conv_rel_error(mode), '768x768 %s ' % mode)
    print('768x768 %s: conv noise %.2e, whole-gradient error %.3e, worst per-tensor error / bar %.2f, decisions forced %d ReLU (of %d)' % (
        mode, res['eps_eff'], res['whole'], res['worst'], res['info']['relu_flips'], res['info']['relu_total']))
    oheat, oloss64 = gc.heat, gc.loss
    dev = float((seg.detach().double().cpu() - gc.out).abs().max())
    if mode == 'bf16s':
        assert 1e-5 < dev < 5e-2, 'soft-max deviation %.3e from fp64 in the bf16 storage mode' % dev
        assert abs(loss.item() - oloss64) < 2e-2 * abs(oloss64)
        top2 = gc.out.topk(2, dim=1)[0]
        sure = (top2[:, 0] - top2[:, 1]) > 2.5 * dev
        assert float(sure.float().mean()) > 0.5
        assert bool((seg.detach().argmax(1).cpu() == gc.out.argmax(1))[sure].all())
        return
    np.testing.assert_allclose(seg.detach().cpu().numpy(), gc.out.numpy(), rtol=1e-4, atol=1e-5)
    hs = float(oheat.abs().max())
    np.testing.assert_allclose(heat.detach().cpu().numpy(), oheat.numpy(), rtol=1e-4, atol=1e-4 * hs)
    assert abs(loss.item() - oloss64) < 2e-5
    mask = label_mask(gc.out, seg)
    assert float(mask.float().mean()) < 2e-3
    assert bool((seg.detach().argmax(1).cpu() == gc.out.argmax(1))[~mask].all())


_C4 = {}


@pytest.mark.parametrize('mode', ['fp32', 'bf16x3', 'bf16s'])
def test_config4_1436_ensemble_inference_matches_oracle(mode):
    """Full-resolution 1436x1436 padded to 1440 (configs[4]): eval-mode forward of two nets + the ensemble reduction
    of test_ensemble.py (util.py:318-373): mean softmax -> arg-max labels, per-net min-max normalised heat maps.  fp32 /
    bf16x3: 1e-4 and bit-exact labels outside the margin mask; bf16 storage (the mode bench.py's fwd_ms_per_img quotes): at
    bf16 distance, labels identical wherever the fp64 margin exceeds 2.5 x the deviation of the averaged soft-max."""
    _, cfg = PAPER_CFGS['paper_sc_l14']
    H, P = 1436, 1440
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 1, P, P, generator=g)
    torch.set_num_threads(max(torch.get_num_threads(), min(64, os.cpu_count() or 32)))
    if 'oouts' not in _C4:                                     # the oracle in fp64, once for all modes: also the source of the label mask
        onets = [_pair(cfg, 900 + i, randomize_bn=True)[1].eval() for i in range(2)]
        with torch.no_grad():
            _C4['oouts'] = [o.double()(x.double()) for o in onets]
        _C4['reduced'] = R.ensemble_reduce([o[0] for o in _C4['oouts']], [o[1] for o in _C4['oouts']], (H, H))
    oouts = _C4['oouts']
    olabels, oheats, oavg = _C4['reduced']
    from dfl_amd import util
    with math_mode_set(mode):
        nets = [_pair(cfg, 900 + i, randomize_bn=True)[0].eval() for i in range(2)]
        with torch.no_grad():
            outs = [n(x.to(DEV)) for n in nets]
        labels, heats, avg = util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (H, H), want_avg_seg=True)
    assert labels.shape == (H, H) and heats.shape == (14, H, H)
    if mode == 'bf16s':
        for (s_, h), (os_, oh) in zip(outs, oouts):
            assert float((s_.cpu().double() - os_).abs().max()) < 5e-2
            assert float((h.cpu().double() - oh).abs().max()) < 5e-2 * float(oh.abs().max())
        dev = float((avg.cpu().double() - oavg[0]).abs().max())
        top2 = oavg.topk(2, dim=1)[0]
        sure = ((top2[:, 0] - top2[:, 1]) > 2.5 * dev)[0]
        assert float(sure.float().mean()) > 0.5
        assert bool((labels.cpu() == olabels[0])[sure].all())
        np.testing.assert_allclose(heats.cpu().numpy(), oheats[0].numpy(), rtol=0, atol=5e-2)
        return
    for (s_, h), (os_, oh) in zip(outs, oouts):
        np.testing.assert_allclose(s_.cpu().numpy(), os_.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(h.cpu().numpy(), oh.numpy(), rtol=1e-4, atol=1e-4 * float(oh.abs().max()))
    # labels of the averaged soft-max: bit-exact outside the rounding-margin pixels of the fp64 average
    mask = label_mask(oavg, avg.unsqueeze(0))[0]
    assert float(mask.float().mean()) < 2e-3
    assert bool((labels.cpu() == olabels[0])[~mask].all())
    np.testing.assert_allclose(heats.cpu().numpy(), oheats[0].numpy(), rtol=1e-3, atol=1e-5)
