"""BASELINE configs[4] at its full image size: the 1436 x 1436 (padded 1440) five-net ensemble forward through test_ensemble.py's own
loop, util.seg_dataset_ensemble (util.py:293-377), against the five-net fp64 oracle -- run in the build container (tools/gen_oracle_fixtures.py
-> tests/golden/config4_oracle.npz; the oracle is pinned to the reference by tests/test_oracle_golden.py).  pytest -m gpu."""
import os

import numpy as np
import pytest
import torch

import dfl_amd
from conftest import PAPER_CFGS, load_golden
from oracle import ref_cpu as R
from gpu_common import math_mode_set

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
    # The five-net fp64 oracle of this image is 80-90 s of host time: it was run in the build container (tools/gen_oracle_fixtures.py,
    # the pinned oracle on the same seeded nets and image) and its outputs are a committed fixture: the full label map,
    # the top-2 margin of the averaged soft-max (exact where the label mask looks, quantised elsewhere), strided samples of the rest.
    G = load_golden('config4_oracle')
    S_AVG, S_NET, S_HEAT = (int(v) for v in G['strides'])
    olabels = torch.from_numpy(G['labels'])
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
    segs_s = [o[0][0][:, ::S_NET, ::S_NET].double().cpu().numpy() for o in outs]
    heats_s = [o[1][0][:, ::S_NET, ::S_NET].double().cpu().numpy() for o in outs]
    avg_s = avg[:, ::S_AVG, ::S_AVG].double().cpu().numpy()
    dev = float(np.abs(avg_s - G['avg_s']).max())              # deviation of the averaged soft-max from fp64 (on the stride-8 grid)
    if mode == 'bf16s':
        for i in range(NNETS):
            assert float(np.abs(segs_s[i] - G['seg_s'][i]).max()) < 5e-2
            assert float(np.abs(heats_s[i] - G['heat_s'][i]).max()) < 5e-2 * float(G['heat_absmax'][i])
        # labels identical wherever the fp64 margin exceeds 2.5 x the deviation (margin >= margin_q / 255: the quantised map errs to "unsure")
        sure = torch.from_numpy(G['margin_q'].astype(np.float64) / 255.0) > 2.5 * dev + 1e-6
        assert float(sure.float().mean()) > 0.4                # (the exact margins gave 0.5-0.6 of the pixels; the quantised map loses those within 1/255 of the threshold)
        assert bool((labels == olabels)[sure].all())
        np.testing.assert_allclose(heats[:, ::S_HEAT, ::S_HEAT].numpy(), G['oheats_s'], rtol=0, atol=5e-2)
        return
    for i in range(NNETS):
        np.testing.assert_allclose(segs_s[i], G['seg_s'][i], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(heats_s[i], G['heat_s'][i], rtol=1e-4, atol=1e-4 * float(G['heat_absmax'][i]))
    # labels of the averaged soft-max: bit-exact outside the rounding-margin pixels of the fp64 average (gpu_common.label_mask's rule:
    # margin below max(1e-5, 2.5 x the deviation); the fixture holds the exact margin wherever it is below 2.5e-4)
    thr = max(1e-5, 2.5 * dev)
    assert thr < 2.5e-4, 'forward deviation %.3e is outside the 1e-4 bar' % (thr / 2.5)
    mask = torch.zeros(H * H, dtype=torch.bool)
    mask[torch.from_numpy(G['low_idx'][G['low_margin'] < thr].astype(np.int64))] = True
    mask = mask.view(H, H)
    assert float(mask.float().mean()) < 2e-3
    assert bool((labels == olabels)[~mask].all())
    np.testing.assert_allclose(heats[:, ::S_HEAT, ::S_HEAT].numpy(), G['oheats_s'], rtol=1e-3, atol=1e-5)
