"""BASELINE configs[4] at its full image size: the 1436 x 1436 (padded 1440) five-net ensemble forward through test_ensemble.py's own
loop, util.seg_dataset_ensemble (util.py:293-377), against the five-net fp64 oracle run on the GPU box's host cores (the oracle is
pinned to the reference by tests/test_oracle_golden.py).  pytest -m gpu."""
import os

import numpy as np
import pytest
import torch

import dfl_amd
from conftest import PAPER_CFGS
from oracle import ref_cpu as R
from gpu_common import label_mask, math_mode_set

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


_C4 = {}
NNETS = 5


class _FakeH5DS:
    def __init__(self, shape, dtype):
        self.a = np.zeros(shape, dtype=dtype)

    def __setitem__(self, k, v):
        self.a[k] = v


class _FakeH5:
    def __init__(self):
        self.d = {}

    def create_dataset(self, name, shape, dtype='f4', **kw_):
        self.d[name] = _FakeH5DS(shape, dtype)
        return self.d[name]


@pytest.mark.parametrize('mode', ['fp32', 'bf16x3', 'bf16s'])
def test_config4_1436_ensemble_inference_matches_oracle(mode):
    """Full-resolution 1436x1436 padded to 1440 (configs[4]) with the FIVE nets the configuration names, through the loop
    test_ensemble.py runs (util.seg_dataset_ensemble, util.py:293-377: eval-mode forwards -- one hipGraph replay per net --, mean
    softmax -> arg-max labels, per-net min-max normalised heat maps, uint8 / float32 output datasets), against the five-net fp64
    oracle.  fp32 / bf16x3: every net's outputs within 1e-4, labels bit-exact outside the margin mask; bf16 storage (the mode
    bench.py's fwd_ms_per_img quotes): at bf16 distance, labels identical wherever the fp64 margin exceeds 2.5 x the deviation of
    the averaged soft-max."""
    _, cfg = PAPER_CFGS['paper_sc_l14']
    H, P = 1436, 1440
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 1, P, P, generator=g)
    torch.set_num_threads(max(torch.get_num_threads(), min(64, os.cpu_count() or 32)))
    if 'oouts' not in _C4:                                     # the oracle in fp64, once for all modes: also the source of the label mask
        oouts = []
        for i in range(NNETS):
            onet = _pair(cfg, 900 + i, randomize_bn=True)[1].eval().double()
            with torch.no_grad():
                oouts.append(onet(x.double()))
            del onet
        _C4['oouts'] = oouts
        _C4['reduced'] = R.ensemble_reduce([o[0] for o in oouts], [o[1] for o in oouts], (H, H))
    oouts = _C4['oouts']
    olabels, oheats, oavg = _C4['reduced']
    from dfl_amd import util

    class DS(torch.utils.data.Dataset):
        rob_orig_img_shape = (H, H)

        def __len__(self):
            return 1

        def __getitem__(self, i):
            return (x[0], torch.zeros(1), torch.zeros(1), torch.zeros(1))

    with math_mode_set(mode):
        nets = [_pair(cfg, 900 + i, randomize_bn=True)[0].eval() for i in range(NNETS)]
        f = _FakeH5()
        times = []
        util.seg_dataset_ensemble(DS(), nets, f, dev=torch.device(DEV), num_lands=14, times=times)
        with torch.no_grad():
            outs = [n(x.to(DEV)) for n in nets]
        labels2, heats2, avg = util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (H, H), want_avg_seg=True)
    labels, heats = torch.from_numpy(f.d['nn-segs'].a[0]), torch.from_numpy(f.d['nn-heats'].a[0])
    assert labels.dtype == torch.uint8 and tuple(labels.shape) == (H, H) and tuple(heats.shape) == (14, H, H) and len(times) == 1
    assert torch.equal(labels, labels2.cpu()) and torch.equal(heats, heats2.cpu()), 'the loop and a direct reduction of the same forwards differ'
    print('configs[4] %s: %d nets, %.1f ms for the image inside util.seg_dataset_ensemble' % (mode, NNETS, times[0] * 1e3))
    if mode == 'bf16s':
        for (s_, h), (os_, oh) in zip(outs, oouts):
            assert float((s_.cpu().double() - os_).abs().max()) < 5e-2
            assert float((h.cpu().double() - oh).abs().max()) < 5e-2 * float(oh.abs().max())
        dev = float((avg.cpu().double() - oavg[0]).abs().max())
        top2 = oavg.topk(2, dim=1)[0]
        sure = ((top2[:, 0] - top2[:, 1]) > 2.5 * dev)[0]
        assert float(sure.float().mean()) > 0.5
        assert bool((labels == olabels[0])[sure].all())
        np.testing.assert_allclose(heats.numpy(), oheats[0].numpy(), rtol=0, atol=5e-2)
        return
    for (s_, h), (os_, oh) in zip(outs, oouts):
        np.testing.assert_allclose(s_.cpu().numpy(), os_.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(h.cpu().numpy(), oh.numpy(), rtol=1e-4, atol=1e-4 * float(oh.abs().max()))
    # labels of the averaged soft-max: bit-exact outside the rounding-margin pixels of the fp64 average
    mask = label_mask(oavg, avg.unsqueeze(0))[0]
    assert float(mask.float().mean()) < 2e-3
    assert bool((labels == olabels[0])[~mask].all())
    np.testing.assert_allclose(heats.numpy(), oheats[0].numpy(), rtol=1e-3, atol=1e-5)


