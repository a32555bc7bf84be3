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
from test_gpu_unet import oracle64, oracle_run, label_mask

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


def test_config3_736_training_step_matches_oracle(math_mode):
    """2x-downsampled 736x736 padded to 768 (configs[3]), paper U-Net, dual head: forward, loss and gradients."""
    _, cfg = PAPER_CFGS['paper_sc_l14']
    net, onet = _pair(cfg, 4242)
    g = torch.Generator().manual_seed(5)
    B, H, P = 2, 736, 768
    x = torch.randn(B, 1, P, P, generator=g)
    lab = torch.randint(0, 7, (B, H, H), generator=g)
    tseg = R.one_hot_masks(lab, 7)
    theat = torch.rand(B, 14, H, H, generator=g) * 0.02
    net.train()
    onet.train()
    seg, heat = net(x.to(DEV))
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg.to(DEV), theat.to(DEV)))
    loss.backward()
    torch.set_num_threads(max(torch.get_num_threads(), min(64, os.cpu_count() or 32)))
    with torch.no_grad():
        oseg, oheat = onet(x)                                 # the oracle in the reference's own fp32
    np.testing.assert_allclose(seg.detach().cpu().numpy(), oseg.numpy(), rtol=1e-4, atol=1e-5)
    hs = float(oheat.abs().max())
    np.testing.assert_allclose(heat.detach().cpu().numpy(), oheat.numpy(), rtol=1e-4, atol=1e-4 * hs)
    # gradients and labels against the fp64 oracle: noise-floor bars (tests/noise_floor.py), rounding-margin label mask
    gf = NF.cached_floor('config3', lambda: NF.GradientFloor(oracle64(cfg, onet.state_dict()), oracle_run(x, tseg, theat), seeds=(1, 2)))
    oloss64 = float(R.dice_and_heatmap_loss_2d((R.center_crop(gf.out, tseg.shape), R.center_crop(oheat.double(), theat.shape)),
                                               (tseg.double(), theat.double()), skip_bg=False, heatmap_wgt=0.5))
    assert abs(loss.item() - oloss64) < 2e-5
    worst, whole, eps_eff = gf.check({k: p.grad for k, p in net.named_parameters()}, seg, NF.conv_rel_error(math_mode), '768x768 ')
    print('768x768 %s: conv noise %.2e, whole-gradient error %.3e, worst per-tensor error / bar %.2f' % (math_mode, eps_eff, whole, worst))
    mask = label_mask(gf.out, seg)
    assert float(mask.float().mean()) < 2e-3
    assert bool((seg.detach().argmax(1).cpu() == gf.out.argmax(1))[~mask].all())


def test_config4_1436_ensemble_inference_matches_oracle(math_mode):
    """Full-resolution 1436x1436 padded to 1440 (configs[4]): eval-mode forward of two nets + the ensemble reduction
    of test_ensemble.py (util.py:318-373): mean softmax -> arg-max labels, per-net min-max normalised heat maps."""
    _, cfg = PAPER_CFGS['paper_sc_l14']
    H, P = 1436, 1440
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 1, P, P, generator=g)
    nets, onets = [], []
    for i in range(2):
        n, o = _pair(cfg, 900 + i, randomize_bn=True)
        n.eval()
        o.eval()
        nets.append(n)
        onets.append(o)
    torch.set_num_threads(max(torch.get_num_threads(), min(64, os.cpu_count() or 32)))
    with torch.no_grad():
        outs = [n(x.to(DEV)) for n in nets]
        oouts = [o.double()(x.double()) for o in onets]       # the oracle in fp64: also the source of the label mask
    for (s, h), (os_, oh) in zip(outs, oouts):
        np.testing.assert_allclose(s.cpu().numpy(), os_.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(h.cpu().numpy(), oh.numpy(), rtol=1e-4, atol=1e-4 * float(oh.abs().max()))
    from dfl_amd import util
    labels, heats, avg = util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (H, H), want_avg_seg=True)
    olabels, oheats, oavg = R.ensemble_reduce([o[0] for o in oouts], [o[1] for o in oouts], (H, H))
    assert labels.shape == (H, H) and heats.shape == (14, H, H)
    # labels of the averaged soft-max: bit-exact outside the rounding-margin pixels of the fp64 average
    mask = label_mask(oavg, avg.unsqueeze(0))[0]
    assert float(mask.float().mean()) < 2e-3
    assert bool((labels.cpu() == olabels[0])[~mask].all())
    np.testing.assert_allclose(heats.cpu().numpy(), oheats[0].numpy(), rtol=1e-3, atol=1e-5)
