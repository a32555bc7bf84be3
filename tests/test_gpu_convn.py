"""The narrow 3x3 form of the bf16 convolution (csrc/convn_bf16.hip; tile configurations 58 ... 63 = its six layouts: 32 or 64 pixels
x 12 ... 24 rows per patch, 32 or 64 output columns, the epilogue on the accumulator registers), forced through
dfl_conv_force_geometry, under every operand / epilogue form the network gives a 3x3 layer of its two shallow levels (reference:
train_test_code/unet.py:211-222 and their autograd): plain + bias + ReLU + statistics on ragged images, BatchNorm affine on load
with zero padding after it + partner statistics, the fused BatchNorm + ReLU backward operand with x_out, pixel strides larger than
the channel count on both sides, one to eight channel blocks.  Same bars as tests/test_gpu_bf16.py (fp64 PyTorch on the bf16-rounded
operands).  pytest -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dfl_amd import _native as nat
import test_gpu_bf16 as T
from test_gpu_bf16 import rb, nhwc, pack16, conv_bf16, brb_reference, _mode4  # noqa: F401
from test_gpu_convq import close_bf16

pytestmark = pytest.mark.gpu
NPATCH = {58: (8, 64), 59: (12, 64), 60: (16, 32), 61: (24, 32), 62: (6, 64), 63: (12, 32),    # csrc/convn_bf16.hip kN: rows x pixels
          64: (6, 64), 65: (12, 32)}                                                          # ... and the persistent form of the last two (32 -> 32 layers with more than 512 patches)
TILES = tuple(NPATCH)


class forced:
    def __init__(self, tile):
        self.g = (C.c_int32 * 5)(tile, 1, NPATCH[tile][0], NPATCH[tile][1], 1)

    def __enter__(self):
        nat.check(nat.lib().dfl_conv_force_geometry(C.addressof(self.g)), 'force')

    def __exit__(self, *exc):
        nat.lib().dfl_conv_force_geometry(None)


_CANDS = {}


def _valid(N, Cin, Cout, H, W, tile):
    key = (N, Cin, Cout, H, W)
    if key not in _CANDS:
        _CANDS[key] = set(g for g in T._candidates(N, Cin, Cout, H, W, 3, 1, 1) if g[0] >= 58)
    return (tile, 1, NPATCH[tile][0], NPATCH[tile][1], 1) in _CANDS[key]


_MEMO = {}


def _memo(key, make):
    if key not in _MEMO:
        if len(_MEMO) >= 3:
            _MEMO.clear()
        _MEMO[key] = make()
    return _MEMO[key]


def test_candidates_list_the_narrow_form_where_it_applies():
    assert all(_valid(2, 32, 32, 40, 25, t) for t in TILES[:6]) and not _valid(2, 32, 32, 40, 25, 64)     # (persistent: only when a workgroup gets several patches)
    assert _valid(9, 32, 32, 190, 180, 64) and _valid(9, 32, 32, 190, 180, 65) and not _valid(9, 64, 32, 190, 180, 64) and not _valid(9, 32, 64, 190, 180, 65)
    assert all(_valid(2, 32, 64, 40, 25, t) for t in (58, 60, 62, 63)) and not _valid(2, 32, 64, 40, 25, 59) and not _valid(2, 32, 64, 40, 25, 61)
    assert all(_valid(2, 64, 64, 40, 25, t) for t in (58, 60, 62, 63)) and not _valid(2, 64, 64, 40, 25, 59) and not _valid(2, 64, 64, 40, 25, 61)   # (64 columns: at most four rows per wave)
    assert not any(_valid(2, 64, 128, 24, 24, t) for t in TILES)               # 32 or 64 output columns
    assert not any(_valid(2, 16, 32, 24, 24, t) for t in TILES)                # channel blocks of 32
    assert not any(g[0] >= 58 for g in T._candidates(2, 64, 64, 16, 16, 1, 1, 0))   # 3x3 / stride 1 / pad 1 only


def test_the_two_tensor_operand_of_a_wide_layer_stays_with_the_patch_kernel():
    """convn_shape_ok: 64 columns x several channel blocks x dfl_conv_args.x_mode is not a case of this form -- forcing it fails loudly,
    the unforced call runs (convp)."""
    case = (1, 64, 64, 20, 20)
    dy, r, coef, dpre, w, ref, wp = _brb_problem(case, True)
    with forced(62):
        with pytest.raises(nat.DflError):
            conv_bf16(dy, wp, 64, 3, 3, 1, 1, 20, 20, brb=(r, coef), force_splits=1)
    close_bf16(conv_bf16(dy, wp, 64, 3, 3, 1, 1, 20, 20, brb=(r, coef)), ref, 'x_mode 64 -> 64, library choice')


def _plain_problem(case):
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    b = torch.randn(Cout, generator=g)
    pre = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    return x, w, b, nhwc(F.relu(pre)), nhwc(pre), pack16(w, 1)


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('case', [(2, 32, 32, 40, 25), (2, 64, 32, 33, 12), (1, 128, 64, 50, 24), (3, 32, 64, 70, 100), (2, 64, 64, 96, 96), (1, 256, 64, 20, 70),
                                  (9, 32, 32, 190, 180)])
def test_convn_plain_and_statistics(case, tile):
    N, Cin, Cout, H, W = case
    if not _valid(N, Cin, Cout, H, W, tile):
        pytest.skip('not a configuration of this layer')
    x, w, b, ref, pre, wp = _memo(('plain',) + case, lambda: _plain_problem(case))
    with forced(tile):
        y, st = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, relu=1, stats=True, force_splits=1)
        y0 = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, relu=0, force_splits=1, ldy=Cout + 8, ldx_pad=8)    # no ReLU, no statistics, wider pixel strides
    close_bf16(y, ref, '%s tile %d' % (case, tile))
    close_bf16(y0, pre, '%s tile %d (no ReLU, padded strides)' % (case, tile))
    yd = y.double().reshape(-1, Cout)
    np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
    np.testing.assert_allclose(st[1].numpy(), (yd * yd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * yd).sum(0).max()))


def _affine_problem(case):
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    partner = rb(torch.randn(N, Cout, H, W, generator=g))
    xa = rb(torch.addcmul(sh.view(1, -1, 1, 1).double(), x.double(), sc.view(1, -1, 1, 1).double()).float())    # one rounding to fp32 (fmaf), then bf16
    ref = F.conv2d(xa.double(), w.double(), b.double(), padding=1)
    return x, w, b, sc, sh, partner, nhwc(ref), pack16(w, 1)


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('case', [(2, 32, 32, 30, 14), (2, 128, 64, 20, 20), (3, 64, 32, 65, 130), (9, 32, 32, 190, 180)])
def test_convn_affine_on_load_and_partner_statistics(case, tile):
    """BatchNorm affine on load with zero padding AFTER it (unet.py:211-222 behind a BatchNorm), statistics against a partner tensor
    (the fused backward sums)."""
    N, Cin, Cout, H, W = case
    if not _valid(N, Cin, Cout, H, W, tile):
        pytest.skip('not a configuration of this layer')
    x, w, b, sc, sh, partner, ref, wp = _memo(('aff',) + case, lambda: _affine_problem(case))
    with forced(tile):
        y, st = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, in_aff=(sc, sh), stats=True, stat_other=partner, force_splits=1)
    close_bf16(y, ref, '%s tile %d' % (case, tile))
    yd = y.double().reshape(-1, Cout)
    pd = nhwc(partner).double().reshape(-1, Cout)
    np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
    np.testing.assert_allclose(st[1].numpy(), (yd * pd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * pd).abs().sum(0).max()))


def _brb_problem(case, with_bn):
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    dy = rb(torch.randn(N, Cin, H, W, generator=g))
    r = rb(torch.relu(torch.randn(N, Cin, H, W, generator=g)))
    coef = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3, torch.randn(Cin, generator=g) * 0.1]) if with_bn else None
    dpre = brb_reference(dy, r, coef)
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    ref = F.conv2d(dpre.double(), w.double(), padding=1)
    return dy, r, coef, dpre, w, nhwc(ref), pack16(w, 1)


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('with_bn', [True, False])
@pytest.mark.parametrize('case', [(2, 32, 64, 20, 20), (1, 64, 32, 40, 13), (3, 64, 32, 100, 70), (2, 128, 32, 30, 45), (2, 32, 32, 192, 192), (9, 32, 32, 190, 180)])
def test_convn_fused_bn_relu_backward_operand(case, with_bn, tile):
    """dfl_conv_args.x_mode: the data gradient forms [r > 0] * (A dy + B r + C) from (dy, r) while it stages its patches; x_out is
    that operand, every element exactly once, bit for bit.  (Not taken by this form, csrc/convn_bf16.hip convn_shape_ok: the two-tensor
    operand of a 64-column layer with several channel blocks.)"""
    N, Cin, Cout, H, W = case
    if not _valid(N, Cin, Cout, H, W, tile):
        pytest.skip('not a configuration of this layer')
    if not with_bn and case[3] > 40:
        pytest.skip('the plain ReLU backward operand: the small cases')
    dy, r, coef, dpre, w, ref, wp = _memo(('brb', with_bn) + case, lambda: _brb_problem(case, with_bn))
    with forced(tile):
        y = conv_bf16(dy, wp, Cout, 3, 3, 1, 1, H, W, brb=(r, coef), force_splits=1)
        y2, xo = conv_bf16(dy, wp, Cout, 3, 3, 1, 1, H, W, brb=(r, coef), force_splits=1, x_out=True)
    close_bf16(y, ref, 'x_mode %s tile %d' % (case, tile))
    assert torch.equal(y2, y)
    assert torch.equal(xo[..., :Cin], nhwc(dpre)), 'x_out %s tile %d: %d elements differ' % (case, tile, int((xo[..., :Cin] != nhwc(dpre)).sum()))
    assert bool(torch.isnan(xo[..., Cin:]).all())


def test_convn_is_bit_repeatable_and_agrees_with_the_patch_kernel():
    """Two launches of one geometry give the same bits; the layouts and convp's default agree to fp32 summation order."""
    N, Cin, Cout, H, W = 2, 64, 32, 48, 80
    g = torch.Generator().manual_seed(12)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    wp = pack16(w, 1)
    base = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W)
    ran = 0
    for tile in TILES:
        if not _valid(N, Cin, Cout, H, W, tile):
            continue
        with forced(tile):
            y1, s1 = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, force_splits=1, stats=True)
            y2, s2 = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, force_splits=1, stats=True)
        assert torch.equal(y1, y2) and torch.equal(s1, s2)
        d = (y1.double() - base.double()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(base.abs().max())             # one bf16 rounding step of the largest value at most
        assert float((d > 0).double().mean()) < 2e-2                             # sums on the other side of a rounding boundary
        ran += 1
    assert ran == 6


@pytest.mark.parametrize('tile', TILES)
def test_convn_live_totals(tile):
    """dfl_conv_args.stat_totals -- the form the bf16 storage training step uses: workgroups add their column sums to [8][2][Ntot] doubles
    with fp64 atomics -- against the sums of the stored values, with a partner tensor, on a layer large enough for the persistent tiles."""
    case = (9, 32, 32, 190, 180)
    N, Cin, Cout, H, W = case
    if not _valid(N, Cin, Cout, H, W, tile):
        pytest.skip('not a configuration of this layer')
    x, w, b, sc, sh, partner, ref, wp = _memo(('aff',) + case, lambda: _affine_problem(case))
    with forced(tile):
        y, st = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, in_aff=(sc, sh), stats=True, stat_other=partner, force_splits=1, live_totals=True)
    close_bf16(y, ref, '%s tile %d' % (case, tile))
    yd = y.double().reshape(-1, Cout)
    pd = nhwc(partner).double().reshape(-1, Cout)
    np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
    np.testing.assert_allclose(st[1].numpy(), (yd * pd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * pd).abs().sum(0).max()))


def test_convn_persistent_form_is_bit_repeatable_and_fills_every_statistics_row():
    """The persistent form (tiles 64, 65): two launches give the same bits; its statistics rows (one per patch: the workgroup's first patch
    carries the sum, its other patches zeros) add up to the sums of the stored values even when the buffer held garbage before."""
    N, Cin, Cout, H, W = 9, 32, 32, 190, 180
    g = torch.Generator().manual_seed(13)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    wp = pack16(w, 1)
    for tile in (64, 65):
        assert _valid(N, Cin, Cout, H, W, tile)
        with forced(tile):
            y1, s1 = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, force_splits=1, stats=True)
            y2, s2 = conv_bf16(x, wp, Cout, 3, 3, 1, 1, H, W, force_splits=1, stats=True, stats_fill=float('nan'))
        assert torch.equal(y1, y2) and torch.equal(s1, s2)
        yd = y1.double().reshape(-1, Cout)
        np.testing.assert_allclose(s1[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
