"""The unrolled 3x3 form of the bf16 convolution (csrc/convq_bf16.hip; tile configurations 40 ... 48 = its nine wave layouts: WM x 8
patch rows, WN x 32 columns, KS k-groups), forced through dfl_conv_force_geometry, under every operand / epilogue form the network
gives a deep-level 3x3 layer (reference: train_test_code/unet.py:211-222 and their autograd): plain + statistics on ragged
images and ragged column counts, BatchNorm affine on load + residual + accumulate + partner statistics through K slices, the
fused BatchNorm + ReLU backward operand with x_out.  Same bars as tests/test_gpu_bf16.py (fp64 PyTorch on the bf16-rounded
operands).  pytest -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dfl_amd import _native as nat
import test_gpu_bf16 as T
from test_gpu_bf16 import rb, nhwc, pack16, conv_bf16, brb_reference, _mode4  # noqa: F401

pytestmark = pytest.mark.gpu
QCFG = {40: (1, 4, 1), 41: (1, 4, 2), 42: (2, 4, 1), 43: (2, 2, 1), 44: (4, 2, 1), 45: (2, 2, 2), 46: (4, 1, 1), 47: (8, 1, 1), 48: (4, 1, 2)}   # csrc/convq_bf16.hip kQ
QCFG.update({t + 9: c for t, c in list(QCFG.items())})          # 49 ... 57: the same layouts, persistent (a workgroup walks several patches)
TILES = tuple(QCFG)


def close_bf16(got, ref, what=''):
    """test_gpu_bf16.close_bf16 (2^-8 of the value + 2e-5 of the largest magnitude) with the allowance long contractions need: the
    fp32 sum of K = 9 * 512 products carries ~1e-5 of sum |x||w|, which at these sizes exceeds the absolute term, so a sum next to a
    bf16 rounding boundary may land on the other side -- at most 2e-4 of the elements, and then by no more than one bf16 step."""
    ref = ref.double()
    err = (got.double() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 2e-5 * float(ref.abs().max())
    bad = err > bound
    assert float(bad.double().mean()) <= 2e-4, '%s: %d of %d elements off; worst |err| %.3e at value %.3e' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(ref.flatten()[err.argmax()]))
    assert not bool((err > 2 * bound).any()), '%s: an element is off by more than one rounding step (|err| %.3e)' % (what, float(err.max()))


class forced:
    def __init__(self, tile, splits):
        self.g = (C.c_int32 * 5)(tile, 1, 8 * QCFG[tile][0], 12, splits)

    def __enter__(self):
        nat.check(nat.lib().dfl_conv_force_geometry(C.addressof(self.g)), 'force')

    def __exit__(self, *exc):
        nat.lib().dfl_conv_force_geometry(None)


_CANDS = {}


def _valid(N, Cin, Cout, H, W, tile, splits):
    """Is (tile, splits) among the layer's candidates?  (The candidate list of a layer is asked for once: a few thousand geometries.)"""
    key = (N, Cin, Cout, H, W)
    if key not in _CANDS:
        _CANDS[key] = set(g for g in T._candidates(N, Cin, Cout, H, W, 3, 1, 1) if g[0] >= 40)
    return (tile, 1, 8 * QCFG[tile][0], 12, splits) in _CANDS[key]


_MEMO = {}


def _memo(key, make):
    """One problem (tensors + its fp64 reference, seconds of host time for the larger ones) per case, shared by the 18 forced forms."""
    if key not in _MEMO:
        if len(_MEMO) >= 3:
            _MEMO.clear()
        _MEMO[key] = make()
    return _MEMO[key]


def _plain_problem(case):
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    b = torch.randn(Cout, generator=g)
    ref = nhwc(F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)))
    return x, w, b, ref, pack16(w, 1)


def _affine_problem(case):
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    other = rb(torch.randn(N, Cout, H, W, generator=g))
    asc, ash = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    y0 = rb(torch.randn(N, Cout, H, W, generator=g))
    partner = rb(torch.randn(N, Cout, H, W, generator=g))
    xa = rb((x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float())     # fmaf: one rounding, then bf16
    ref = F.conv2d(xa.double(), w.double(), b.double(), padding=1)
    ref = ref + other.double() * asc.double().view(1, -1, 1, 1) + ash.double().view(1, -1, 1, 1) + y0.double()
    return x, w, b, sc, sh, other, asc, ash, y0, partner, ref, pack16(w, 1)


def _brb_problem(case, with_bn):
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    dy = rb(torch.randn(N, Cin, H, W, generator=g))
    r = rb(torch.relu(torch.randn(N, Cin, H, W, generator=g)))
    coef = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3, torch.randn(Cin, generator=g) * 0.1]) if with_bn else None
    dpre = brb_reference(dy, r, coef)
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    ref = F.conv2d(dpre.double(), w.double(), padding=1)
    return dy, r, coef, dpre, w, ref, pack16(w, 1)


def test_candidates_list_the_unrolled_form_where_it_applies():
    assert all(_valid(2, 128, 128, 24, 24, t, 1) for t in (40, 41, 42, 43, 44, 45))
    assert not any(_valid(2, 128, 128, 24, 24, t, 1) for t in range(49, 58))         # persistent: only when a workgroup gets more than one patch
    assert _valid(16, 32, 32, 192, 192, 55, 1) and _valid(16, 64, 64, 96, 96, 52, 1) and _valid(8, 128, 128, 192, 192, 49, 1)
    assert _valid(2, 256, 128, 24, 24, 41, 2) and _valid(2, 512, 128, 12, 12, 40, 8)
    assert all(_valid(2, 32, 32, 24, 24, t, 1) for t in (46, 47)) and not any(_valid(2, 32, 32, 24, 24, t, 1) for t in (40, 41, 42, 43, 44, 45))
    assert _valid(2, 64, 64, 24, 24, 43, 1) and _valid(2, 64, 64, 24, 24, 44, 1) and not _valid(2, 64, 64, 24, 24, 46, 1)   # (a 64-column layer takes 64-column tiles)
    assert not any(_valid(2, 16, 32, 24, 24, t, 1) for t in TILES)             # needs 32 resident channels
    assert not any(g[0] >= 39 for g in T._candidates(2, 64, 128, 16, 16, 1, 1, 0))   # 3x3 / stride 1 / pad 1 only


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('case', [(2, 64, 128, 17, 11), (2, 256, 192, 20, 30), (3, 128, 136, 6, 6), (2, 32, 32, 40, 25), (2, 64, 32, 33, 12), (1, 128, 64, 50, 24),
                                  # enough patches for the persistent forms: odd counts per workgroup, one and several channel blocks, K slices
                                  (9, 32, 32, 190, 180), (3, 64, 64, 200, 170), (2, 128, 128, 200, 210), (1, 256, 128, 250, 300)])
def test_convq_plain_and_statistics(case, tile):
    N, Cin, Cout, H, W = case
    if not _valid(N, Cin, Cout, H, W, tile, 1):
        pytest.skip('not a configuration of this layer')
    x, w, b, ref, wp = _memo(('plain',) + case, lambda: _plain_problem(case))
    for splits in (1, 2, 4):
        if not _valid(N, Cin, Cout, H, W, tile, splits):
            continue
        with forced(tile, splits):
            y, st = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, relu=1, stats=True, force_splits=splits)
        close_bf16(y, ref, '%s tile %d splits %d' % (case, tile, splits))
        yd = y.double().reshape(-1, Cout)
        np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
        np.testing.assert_allclose(st[1].numpy(), (yd * yd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * yd).sum(0).max()))


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('case', [(2, 256, 256, 13, 9), (1, 512, 128, 24, 24), (2, 32, 32, 30, 14), (2, 128, 64, 20, 20),
                                  (9, 32, 32, 190, 180), (2, 128, 128, 200, 210)])
def test_convq_affine_residual_epilogue(case, tile):
    """BatchNorm affine on load with zero padding AFTER it, '+ BN(other)', accumulate, statistics against a partner tensor (the
    forward block epilogue, unet.py:229-231, and the fused backward sums), one launch and through K slices."""
    N, Cin, Cout, H, W = case
    if not _valid(N, Cin, Cout, H, W, tile, 1):
        pytest.skip('not a configuration of this layer')
    x, w, b, sc, sh, other, asc, ash, y0, partner, ref, wp = _memo(('aff',) + case, lambda: _affine_problem(case))
    ran = 0
    for splits in (1, 2, 4):
        if not _valid(N, Cin, Cout, H, W, tile, splits):
            continue
        with forced(tile, splits):
            y, st = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, in_aff=(sc, sh), add=other, add_aff=(asc, ash),
                              y_init=y0, accumulate=1, stats=True, stat_other=partner, force_splits=splits)
        close_bf16(y, nhwc(ref), '%s tile %d splits %d' % (case, tile, splits))
        yd = y.double().reshape(-1, Cout)
        pd = nhwc(partner).double().reshape(-1, Cout)
        np.testing.assert_allclose(st[1].numpy(), (yd * pd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * pd).abs().sum(0).max()))
        ran += 1
    assert ran >= 1


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('with_bn', [True])
@pytest.mark.parametrize('case', [(2, 256, 128, 17, 13), (2, 32, 64, 20, 20), (1, 64, 32, 40, 13), (3, 64, 64, 200, 170), (1, 256, 128, 250, 300)])
def test_convq_fused_bn_relu_backward_operand(case, with_bn, tile):
    """dfl_conv_args.x_mode: the data gradient forms [r > 0] * (A dy + B r + C) from (dy, r) while it stages its patches; x_out is
    that operand, every element exactly once, bit for bit (K slices each write their own channels)."""
    N, Cin, Cout, H, W = case
    if not _valid(N, Cin, Cout, H, W, tile, 1):
        pytest.skip('not a configuration of this layer')
    if tile >= 49 and Cin % 128 == 0 and QCFG[tile][0] * QCFG[tile][1] * QCFG[tile][2] == 4:
        pytest.skip('the persistent four-wave forms are not built for 128-channel images of the two-tensor operand (they would spill)')
    dy, r, coef, dpre, w, ref, wp = _memo(('brb', with_bn) + case, lambda: _brb_problem(case, with_bn))
    for splits in (1, 2):
        if not _valid(N, Cin, Cout, H, W, tile, splits):
            continue
        with forced(tile, splits):
            y = conv_bf16(dy, wp, Cout, 3, 3, 1, 1, H, W, brb=(r, coef), force_splits=splits)
            y2, xo = conv_bf16(dy, wp, Cout, 3, 3, 1, 1, H, W, brb=(r, coef), force_splits=splits, x_out=True)
        close_bf16(y, nhwc(ref), 'x_mode %s tile %d splits %d' % (case, tile, splits))
        assert torch.equal(y2, y)
        assert torch.equal(xo[..., :Cin], nhwc(dpre)), 'x_out %s tile %d splits %d: %d elements differ' % (
            case, tile, splits, int((xo[..., :Cin] != nhwc(dpre)).sum()))
        assert bool(torch.isnan(xo[..., Cin:]).all())


def test_convq_is_bit_repeatable_and_agrees_with_the_patch_kernel():
    """Two launches of one geometry give the same bits; the three forms and convp's default agree to fp32 summation order."""
    N, Cin, Cout, H, W = 2, 256, 256, 24, 24
    g = torch.Generator().manual_seed(11)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    wp = pack16(w, 1)
    base = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W)
    for tile in TILES:
        if not _valid(N, Cin, Cout, H, W, tile, 1):
            continue
        with forced(tile, 1):
            y1 = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, force_splits=1)
            y2 = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, force_splits=1)
        assert torch.equal(y1, y2)
        d = (y1.double() - base.double()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(base.abs().max())             # one bf16 rounding step of the largest value at most
        assert float((d > 0).double().mean()) < 2e-2                             # sums on the other side of a rounding boundary
