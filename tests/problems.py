"""The seeded problems whose GRADIENTS the GPU tests check against the fp64 oracle -- defined once, used by the tests and by
tools/calib_floors.py (which computes, in the build container, the sensitivity of every gradient tensor of every problem
to rounding noise and commits it under tests/golden/floors/).  Test infrastructure: imports oracle/.

A problem = (constructor flags, initial state_dict, input, targets).  build(key) is deterministic: seeded generators only.
"""
import os

import numpy as np
import torch

from conftest import TINY_CFGS, PAPER_CFGS, PAPER_BATCH, paper_key, load_golden
from oracle import ref_cpu as R


class Problem:
    def __init__(self, key, cfg, state_dict, x, tseg, theat=None, skip_bg=False):
        self.key, self.cfg, self.sd, self.x, self.tseg, self.theat, self.skip_bg = key, cfg, state_dict, x, tseg, theat, skip_bg

    def oracle64(self):
        """The fp64 oracle carrying this problem's weights, in training mode."""
        o = R.OracleUNet(**self.cfg).double()
        o.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in self.sd.items()})
        return o.train()

    def run(self, net):
        """(loss, seg) of the oracle `net` on this problem in fp64: the loss wiring of train.py:405-421."""
        o = net(self.x.double())
        seg = o[0] if isinstance(o, tuple) else o
        self.last_heat = o[1].detach() if isinstance(o, tuple) else None
        if self.theat is not None:
            loss = R.dice_and_heatmap_loss_2d((R.center_crop(seg, self.tseg.shape), R.center_crop(o[1], self.theat.shape)),
                                              (self.tseg.double(), self.theat.double()), skip_bg=False, heatmap_wgt=0.5)
        else:
            loss = R.dice_loss_2d(R.center_crop(seg, self.tseg.shape), self.tseg.double(), skip_bg=self.skip_bg)
        return loss, seg

    def loss_of(self, seg, heat):
        """The same loss from the two network outputs (fp64; oracle/bf16_emu.py drives its own forward)."""
        if self.theat is not None:
            return R.dice_and_heatmap_loss_2d((R.center_crop(seg, self.tseg.shape), R.center_crop(heat, self.theat.shape)),
                                              (self.tseg.double(), self.theat.double()), skip_bg=False, heatmap_wgt=0.5)
        return R.dice_loss_2d(R.center_crop(seg, self.tseg.shape), self.tseg.double(), skip_bg=self.skip_bg)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _seeded_oracle(seed, cfg, in_channels=None):
    torch.manual_seed(seed)
    return R.OracleUNet(**cfg) if in_channels is None else R.OracleUNet(in_channels, **cfg)


def tiny(name):
    """tests/golden/<name>.npz: weights, input and targets written by the reference run (tools/gen_golden.py)."""
    g = load_golden(name)
    cfg = TINY_CFGS[name]
    sd = {k[4:]: _t(v) for k, v in g.items() if k.startswith('sd0/')}
    return Problem('tiny__' + name, cfg, sd, _t(g['x']), _t(g['tseg']), _t(g['theat']) if cfg['num_lands'] > 0 else None)


def paper(name, batch):
    """Paper presets at 192x192: batch 2 (seed + 1, as the fixtures of tests/golden/paper_*.npz) or BASELINE configs[1]'s
    batch 16 (seed + 16)."""
    seed, cfg = PAPER_CFGS[name]
    onet = _seeded_oracle(seed, cfg)
    gen = torch.Generator().manual_seed(seed + (1 if batch == 2 else batch))
    x = torch.randn(batch, 1, 192, 192, generator=gen)
    lab = torch.randint(0, 7, (batch, 184, 184), generator=gen)
    tseg = R.one_hot_masks(lab, 7)
    theat = torch.rand(batch, 14, 184, 184, generator=gen) * 0.02
    if cfg['num_lands'] == 0:
        theat = None
    return Problem('paper__%s__b%d' % (name, batch), cfg, onet.state_dict(), x, tseg, theat)


RAGGED_CFG = dict(n_classes=5, depth=3, wf=4, batch_norm=True, padding=True, num_lands=6, do_res=True, block_depth=2)


def ragged(H, W, max_pool):
    cfg = dict(RAGGED_CFG, max_pool=bool(max_pool))
    onet = _seeded_oracle(31 + H, cfg, 1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 1, H, W, generator=g)
    try:
        with torch.no_grad():            # (eval mode: a probing forward must not move the BatchNorm running statistics)
            ho, wo = onet.eval()(x[:1])[0].shape[-2:]
    except Exception:
        return None                      # the reference architecture itself rejects this size
    onet.train()
    tseg = torch.softmax(torch.randn(3, 5, ho - 2, wo - 2, generator=g), 1)
    theat = torch.rand(3, 6, ho - 2, wo - 2, generator=g) * 0.02
    return Problem('ragged__%dx%d__mp%d' % (H, W, int(bool(max_pool))), cfg, onet.state_dict(), x, tseg, theat)


def large_head():
    cfg = dict(n_classes=12, depth=3, wf=4, batch_norm=True, padding=True, max_pool=False, num_lands=20, do_res=True, block_depth=2)
    onet = _seeded_oracle(91, cfg, 1)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 1, 40, 48, generator=g)
    tseg = torch.softmax(torch.randn(2, 12, 36, 44, generator=g), 1)
    theat = torch.rand(2, 20, 36, 44, generator=g) * 0.02
    return Problem('largehead', cfg, onet.state_dict(), x, tseg, theat)


def lands_block(lbd, padding=True):
    """(wf = 5: F/2 = 16 channels, so the same problems also run in the bf16 storage arithmetic.)"""
    cfg = dict(n_classes=5, depth=3, wf=5, batch_norm=True, padding=padding, max_pool=False, num_lands=6, do_res=bool(padding),
               block_depth=2, lands_block_depth=lbd)
    onet = _seeded_oracle(123 + lbd, cfg, 1)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 1, 72, 80, generator=g)
    with torch.no_grad():
        so, ho = onet.eval()(x[:1])
    onet.train()
    tseg = torch.softmax(torch.randn(3, 5, ho.shape[-2] - 4, ho.shape[-1] - 4, generator=g), 1)
    theat = torch.rand(3, 6, ho.shape[-2] - 4, ho.shape[-1] - 4, generator=g) * 0.02
    return Problem('landsblock__%d%s' % (lbd, '' if padding else '__valid'), cfg, onet.state_dict(), x, tseg, theat)


def n1x1(n):
    cfg = dict(n_classes=5, depth=3, wf=4, batch_norm=True, padding=True, max_pool=False, num_lands=6, do_res=True,
               block_depth=2, lands_num_1x1=n)
    onet = _seeded_oracle(77 + n, cfg, 1)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 1, 40, 48, generator=g)
    tseg = torch.softmax(torch.randn(2, 5, 36, 44, generator=g), 1)
    theat = torch.rand(2, 6, 36, 44, generator=g) * 0.02
    return Problem('n1x1__%d' % n, cfg, onet.state_dict(), x, tseg, theat)


def random_arch(seed):
    """Seeded sweep over constructor flags and shapes; returns None when the reference architecture rejects the combination."""
    rng = np.random.RandomState(1000 + seed)
    padding = bool(rng.rand() < 0.75)
    cfg = dict(n_classes=int(rng.randint(2, 8)), depth=int(rng.randint(1, 5)), wf=int(rng.randint(2, 5)),
               batch_norm=bool(rng.rand() < 0.7), padding=padding, max_pool=bool(rng.rand() < 0.5),
               num_lands=int(rng.choice([0, 0, 3, 14])), do_res=bool(padding and rng.rand() < 0.7),
               block_depth=int(rng.randint(1, 4)), do_soft_max=bool(rng.rand() < 0.8))
    B = int(rng.randint(1, 4))
    H, W = int(rng.randint(24, 90)), int(rng.randint(24, 90))
    if not padding:                      # valid convolutions shrink every level: keep the deepest level alive
        H, W = H + 60, W + 60
    onet = _seeded_oracle(77 + seed, cfg, 1)
    x = torch.randn(B, 1, H, W, generator=torch.Generator().manual_seed(seed))
    onet.train()
    try:
        with torch.no_grad():
            oout = onet(x)
    except Exception:
        return None
    onet = _seeded_oracle(77 + seed, cfg, 1)             # (the probing forward moved the BatchNorm running statistics)
    oseg = oout[0] if cfg['num_lands'] > 0 else oout
    g = torch.Generator().manual_seed(seed + 1)
    ho, wo = oseg.shape[-2:]
    th, tw = max(ho - 2, 1), max(wo - 2, 1)
    tseg = torch.softmax(torch.randn(B, cfg['n_classes'], th, tw, generator=g), 1)
    theat = torch.rand(B, cfg['num_lands'], th, tw, generator=g) * 0.02 if cfg['num_lands'] > 0 else None
    return Problem('random__%d' % seed, cfg, onet.state_dict(), x, tseg, theat, skip_bg=bool(seed % 2))


def config3():
    """BASELINE configs[3]: 736x736 padded to 768, paper preset, dual head (batch 2 of the 8: kernels, tile configurations,
    32-bit offsets and split decisions depend on the image size, not on the batch count)."""
    _, cfg = PAPER_CFGS['paper_sc_l14']
    onet = _seeded_oracle(4242, cfg)
    g = torch.Generator().manual_seed(5)
    B, H, P = 2, 736, 768
    x = torch.randn(B, 1, P, P, generator=g)
    lab = torch.randint(0, 7, (B, H, H), generator=g)
    tseg = R.one_hot_masks(lab, 7)
    theat = torch.rand(B, 14, H, H, generator=g) * 0.02
    return Problem('config3', cfg, onet.state_dict(), x, tseg, theat)


def config3_one():
    """The upper half (384 rows of the 768 columns) of the first image of config3: the step-by-step check of the bf16 storage
    arithmetic at configs[3]'s row length -- a quarter of the emulation time of the two full images (round 6: the suite's time); the
    full 768 x 768 step is compared with the oracle by tests/test_gpu_fullsize.py."""
    pr = config3()
    return Problem('config3__b1', pr.cfg, pr.sd, pr.x[:1, :, :384].clone(), pr.tseg[:1, :, :352].clone(), pr.theat[:1, :, :352].clone())


def upsample(pad_mode='zeros', wf=4):
    """up_mode='upsample' (unet.py:242-244) and / or pad_mode='circular' (unet.py:211-212): flags no reference CLI selects."""
    cfg = dict(n_classes=5, depth=3, wf=wf, batch_norm=True, padding=True, max_pool=False, num_lands=6, do_res=True,
               block_depth=2, up_mode='upsample', pad_mode=pad_mode)
    onet = _seeded_oracle(55, cfg, 1)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 1, 48, 64, generator=g)
    tseg = torch.softmax(torch.randn(2, 5, 44, 60, generator=g), 1)
    theat = torch.rand(2, 6, 44, 60, generator=g) * 0.02
    return Problem('upsample__%s' % pad_mode + ('' if wf == 4 else '__wf%d' % wf), cfg, onet.state_dict(), x, tseg, theat)


def circular(lbd=0):
    """pad_mode='circular' (unet.py:211-212; with lbd > 0 also the landmark block's convolutions, unet.py:118-120)."""
    cfg = dict(n_classes=5, depth=3, wf=4, batch_norm=True, padding=True, max_pool=(lbd == 0), num_lands=6, do_res=True,
               block_depth=2, pad_mode='circular', lands_block_depth=lbd)
    onet = _seeded_oracle(56 + lbd, cfg, 1)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 1, 40, 56, generator=g)
    tseg = torch.softmax(torch.randn(2, 5, 36, 52, generator=g), 1)
    theat = torch.rand(2, 6, 36, 52, generator=g) * 0.02
    return Problem('circular__lb%d' % lbd, cfg, onet.state_dict(), x, tseg, theat)


# key -> builder.  Every entry gets a floors file (tools/calib_floors.py); a builder may return None (rejected architecture).
REGISTRY = {}
for _n in sorted(TINY_CFGS):
    REGISTRY['tiny__' + _n] = (lambda n=_n: tiny(n))
for _n in sorted(PAPER_CFGS):
    REGISTRY[paper_key(_n)] = (lambda n=_n: paper(n, PAPER_BATCH.get(n, 2)))
REGISTRY['paper__paper_sc_l14__b16'] = lambda: paper('paper_sc_l14', 16)
REGISTRY['paper__paper_sc_l14__b5'] = lambda: paper('paper_sc_l14', 5)     # the paper's own batch (train_test_code/Readme.md:16): 180 pixels at level 5
for _hw in ((50, 70), (37, 41), (64, 96)):
    for _mp in (False, True):
        REGISTRY['ragged__%dx%d__mp%d' % (_hw[0], _hw[1], int(_mp))] = (lambda hw=_hw, mp=_mp: ragged(hw[0], hw[1], mp))
REGISTRY['largehead'] = large_head
for _l in (1, 2):
    REGISTRY['landsblock__%d' % _l] = (lambda l=_l: lands_block(l))
    REGISTRY['landsblock__%d__valid' % _l] = (lambda l=_l: lands_block(l, padding=False))
for _n in (3, 4):
    REGISTRY['n1x1__%d' % _n] = (lambda n=_n: n1x1(n))
for _s in range(12):
    REGISTRY['random__%d' % _s] = (lambda s=_s: random_arch(s))
REGISTRY['config3'] = config3
for _l in (0, 1):
    REGISTRY['circular__lb%d' % _l] = (lambda l=_l: circular(l))
for _m in ('zeros', 'circular'):
    REGISTRY['upsample__%s' % _m] = (lambda m=_m: upsample(m))
REGISTRY['upsample__circular__wf5'] = lambda: upsample('circular', wf=5)     # (wide enough for the bf16 storage arithmetic)

FLOOR_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'floors')
