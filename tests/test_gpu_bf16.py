"""Math mode 4, "bf16 storage" (BASELINE configs[1] as named: bf16 activations / weights in HBM, fp32 accumulate, BatchNorm
statistics, losses and master weights): the patch-resident kernels of csrc/convp_bf16.hip / wgradp_bf16.hip and the bf16
variants of the streaming kernels through the C ABI against fp64 PyTorch on the SAME bf16-rounded operands, then the whole
network against the fp64 oracle.  pytest -m gpu.

Bars.  Kernel level: operands are exact bf16 values, products are exact in fp32, so a result differs from the fp64 one
only by fp32 accumulation order (1e-5 relative to sum |x||w|) and by its final rounding to bf16 (2^-9 relative): 2^-8 of
the value + the accumulation term; fp32 outputs (weight gradients, statistics) 2e-5.  Network level: the mode is outside
the 1e-4 forward bar by construction (as mode 3); it is held to the noise-floor gradient bars of tests/noise_floor.py at
its own measured convolution error, to label agreement outside the margin implied by its forward deviation, and -- the
north-star quality bar -- to the reference's hard Dice at a plateau (tests/test_gpu_unet.py::test_plateau_dice...)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dfl_amd
from dfl_amd import _native as nat
from conftest import PAPER_CFGS
from oracle import ref_cpu as R
import noise_floor as NF
import problems as PR
from gpu_common import oracle64, label_mask, hip_net, hip_step

pytestmark = pytest.mark.gpu
DEV = 'cuda'
BF = torch.bfloat16


@pytest.fixture(autouse=True)
def _mode4():
    lib = nat.lib()
    prev = lib.dfl_get_math_mode()
    nat.check(lib.dfl_set_math_mode(4), 'dfl_set_math_mode')
    yield
    nat.check(lib.dfl_set_math_mode(prev), 'dfl_set_math_mode')


def stream():
    return torch.cuda.current_stream().cuda_stream


def rb(t):
    """Round to bf16 and back: the values the kernels see."""
    return t.to(BF).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def pack16(w, kind, flip=0):
    """dfl_pack_weights with split = 2: parameter [A][B][KH][KW] -> bf16 chunk layout [K/16][N][16]."""
    lib = nat.lib()
    src = w.to(DEV).contiguous()
    A, B, KH, KW = w.shape
    Cc = KH * KW
    K = {1: Cc * B, 2: Cc * A, 3: A}[kind]
    N = {1: A, 2: B, 3: Cc * B}[kind]
    dst = torch.full(((K + 15) // 16 * N * 16,), float('nan'), device=DEV, dtype=BF)
    job = nat.PackJob(src=src.data_ptr(), dst=dst.data_ptr(), A=A, B=B, C=Cc, kind=kind, flip=flip, split=2)
    jobs = torch.from_numpy(np.frombuffer(bytes(job), dtype=np.uint8).copy()).to(DEV)
    nat.check(lib.dfl_pack_weights(jobs.data_ptr(), 1, A * B * Cc, stream()))
    torch.cuda.synchronize()
    return dst


def test_pack_bf16_chunk_layout():
    g = torch.Generator().manual_seed(1)
    for (A, B, KK, kind, flip) in ((32, 16, 3, 1, 0), (32, 16, 3, 2, 1), (64, 32, 1, 1, 0), (48, 16, 2, 3, 0), (16, 32, 2, 1, 0)):
        w = torch.randn(A, B, KK, KK, generator=g)
        Cc = KK * KK
        wn = w.reshape(A, B, Cc).numpy()
        if kind == 1:
            W = wn.transpose(2, 1, 0).reshape(Cc * B, A)
        elif kind == 2:
            W = wn[:, :, ::-1].transpose(2, 0, 1).reshape(Cc * A, B) if flip else wn.transpose(2, 0, 1).reshape(Cc * A, B)
        else:
            W = wn.transpose(0, 2, 1).reshape(A, Cc * B)
        K, N = W.shape
        Kc = (K + 15) // 16
        ref = np.zeros((Kc * 16, N), dtype=np.float32)
        ref[:K] = W
        ref = torch.from_numpy(ref.reshape(Kc, 16, N).transpose(0, 2, 1).reshape(-1).copy()).to(BF)
        got = pack16(w, kind, flip).cpu()
        assert torch.equal(got, ref), (A, B, KK, kind, flip)


def conv_bf16(x, wp, Ntot, KH, KW, stride, pad, Hout, Wout, bias=None, in_aff=None, relu=0, add=None, add_aff=None,
              y_init=None, accumulate=0, scatter=0, stats=False, stat_other=None, ldy=None, ldx_pad=0, force_splits=None, brb=None, x_out=False,
              latency=False, stats_fill=0.0, live_totals=False):
    """x: NCHW fp32 cpu tensor (bf16-representable) -> dfl_conv2d with bf16 tensors -> y NHWC fp32 cpu tensor (+ stats).
    x_out (with brb): also returns the operand tensor the kernel wrote (dfl_conv_args.x_out; NHWC with 8 channels of padding)."""
    lib = nat.lib()
    N, Cin, Hin, Win = x.shape
    xh = nhwc(x)
    if ldx_pad:
        xh = F.pad(xh, (0, ldx_pad))
    xd = xh.to(DEV).to(BF).contiguous()
    Cout = Ntot // 4 if scatter else Ntot
    ldy = Cout if ldy is None else ldy
    if y_init is not None:
        yd = nhwc(y_init).to(DEV).to(BF)
        if ldy != Cout:
            yd = F.pad(yd, (0, ldy - Cout))
        yd = yd.contiguous()
    else:
        yd = torch.full((N, Hout, Wout, ldy), float('nan'), device=DEV, dtype=BF)
    a = nat.ConvArgs()
    keep = [xd, yd, wp]
    a.x, a.w, a.y = xd.data_ptr(), wp.data_ptr(), yd.data_ptr()
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    if bias is not None:
        b = bias.to(DEV)
        keep.append(b)
        a.bias = b.data_ptr()
    if in_aff is not None:
        sc, sh = in_aff[0].to(DEV), in_aff[1].to(DEV)
        keep += [sc, sh]
        a.in_scale, a.in_shift = sc.data_ptr(), sh.data_ptr()
    if add is not None:
        ad = nhwc(add).to(DEV).to(BF).contiguous()
        keep.append(ad)
        a.add, a.ldadd = ad.data_ptr(), ad.shape[-1]
        if add_aff is not None:
            asc, ash = add_aff[0].to(DEV), add_aff[1].to(DEV)
            keep += [asc, ash]
            a.add_scale, a.add_shift = asc.data_ptr(), ash.data_ptr()
    if brb is not None:                                 # x is dy; operand = [r > 0] * (A dy + B r + C) formed in the staging (x_mode)
        r_, coef = brb
        rd = F.pad(nhwc(r_), (0, 8)).to(DEV).to(BF).contiguous()       # (its own pixel stride)
        keep.append(rd)
        a.x_mode, a.x2, a.ldx2 = 1, rd.data_ptr(), rd.shape[-1]
        if coef is not None:
            cd = coef.to(DEV).contiguous()
            keep.append(cd)
            a.in_scale = cd.data_ptr()
        if x_out:
            xo = torch.full((N, Hin, Win, Cin + 8), float('nan'), device=DEV, dtype=BF)
            keep.append(xo)
            a.x_out, a.ldxo = xo.data_ptr(), Cin + 8
    a.N, a.Hin, a.Win, a.Cin, a.ldx = N, Hin, Win, Cin, Cin + ldx_pad
    a.KH, a.KW, a.stride, a.pad = KH, KW, stride, pad
    a.Hout, a.Wout, a.Ntot, a.ldy = Hout, Wout, Ntot, ldy
    a.relu, a.accumulate, a.scatter2x2 = relu, accumulate, scatter
    a.latency_form = 1 if latency else 0              # (the latency form of csrc/convs_bf16.hip: tests/test_gpu_latency_form.py)
    sp = force_splits or nat.check(lib.dfl_conv_suggest_splits(C.addressof(a)), 'suggest')
    if sp > 1:
        Mrows = N * (Hin * Win if scatter else Hout * Wout)
        kpart = torch.full((sp * Mrows * Ntot,), float('nan'), device=DEV)
        keep.append(kpart)
        a.splits, a.partial = sp, kpart.data_ptr()
    part = None
    if stats:
        gm = nat.check(lib.dfl_conv_grid_m(C.addressof(a)), 'grid_m')
        part = torch.full((4 * gm, 2, Cout), stats_fill, device=DEV) if scatter else torch.full((gm, 2, Ntot), stats_fill, device=DEV)   # (every row is written by the kernel)
        if live_totals:                                  # dfl_conv_args.stat_totals: [DFL_BN_R = 8][2][Ntot] doubles the workgroups add their sums to (fp64 atomics)
            part = torch.zeros(8, 2, Ntot, device=DEV, dtype=torch.float64)
            a.stat_totals = part.data_ptr()
        else:
            a.stat_partials = part.data_ptr()
        if stat_other is not None:
            so = nhwc(stat_other).to(DEV).to(BF).contiguous()
            keep.append(so)
            a.stat_other, a.ldso = so.data_ptr(), so.shape[-1]
    cfg = nat.check(lib.dfl_conv_config(C.addressof(a)), 'config')
    assert cfg >= 16, 'the patch-resident kernels must take bf16 layers'
    assert not latency or cfg == 16 + LATENCY_CFG, 'the latency form was asked for and is eligible here'
    nat.check(lib.dfl_conv2d(C.addressof(a), stream()), 'dfl_conv2d')
    torch.cuda.synchronize()
    y = yd.float().cpu()[..., :Cout]
    if x_out:
        return y, xo.float().cpu()
    return (y, part.cpu().double().sum(0)) if stats else y


LATENCY_CFG = 39       # csrc/convp.h: CONVS_TILE


def close_bf16(got, ref, what=''):
    """got is a bf16-rounded result of ref (fp64): within 2^-8 of the value + 1e-5 of the largest magnitude."""
    ref = ref.double()
    err = (got.double() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 2e-5 * float(ref.abs().max())
    bad = err > bound
    assert not bool(bad.any()), '%s: %d of %d elements off; worst |err| %.3e at value %.3e' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(ref.flatten()[err.argmax()]))


BCASES = [
    # N, Cin, Cout, H, W, K, stride, pad
    (2, 16, 16, 12, 12, 3, 1, 1),      # the smallest channel count of the mode
    (1, 32, 32, 24, 20, 3, 1, 1),
    (3, 64, 64, 13, 9, 3, 1, 1),       # odd sizes: ragged patches
    (2, 64, 128, 17, 11, 3, 1, 1),
    (2, 32, 16, 10, 10, 3, 1, 0),      # valid conv
    (2, 32, 64, 8, 8, 1, 1, 0),        # 1x1
    (2, 16, 32, 12, 10, 2, 2, 0),      # 2x2 stride 2
    (2, 32, 32, 7, 9, 2, 2, 0),        # 2x2 stride 2, odd input
    (16, 256, 256, 6, 6, 3, 1, 1),     # deepest level shape: several images per patch, K slices
    (4, 512, 256, 12, 12, 3, 1, 1),    # decoder shape, K slices
    (16, 32, 32, 48, 48, 3, 1, 1),     # wide level, 32 columns
    (4, 64, 64, 96, 96, 3, 1, 1),      # 64 columns
    (4, 128, 128, 48, 48, 3, 1, 1),    # 128 columns
    (2, 128, 320, 24, 24, 3, 1, 1),    # columns not a multiple of the tile
]


@pytest.mark.parametrize('case', BCASES)
def test_convp_plain(case):
    N, Cin, Cout, H, W, K, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    y, st = conv_bf16(x, pack16(w, 1), Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, stats=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad))
    close_bf16(y, nhwc(ref), str(case))
    # statistics are those of the STORED values
    yd = y.double().reshape(-1, Cout)
    np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
    np.testing.assert_allclose(st[1].numpy(), (yd * yd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * yd).sum(0).max()))


def _candidates(N, Cin, Cout, H, W, K, stride, pad):
    """Geometry candidates of a layer (dfl_conv_candidates needs pointers only to be non-null)."""
    lib = nat.lib()
    a = nat.ConvArgs()
    a.x = a.w = a.y = 4096
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = N, H, W, Cin, Cin
    a.KH, a.KW, a.stride, a.pad = K, K, stride, pad
    a.Hout, a.Wout = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    a.Ntot = a.ldy = Cout
    out = (C.c_int32 * (5 * 4096))()
    n = nat.check(lib.dfl_conv_candidates(C.addressof(a), C.addressof(out), 4096), 'candidates')
    return [tuple(out[5 * i + j] for j in range(5)) for i in range(min(n, 4096))]


@pytest.mark.parametrize('case', [(2, 64, 128, 17, 11, 3, 1, 1), (4, 128, 256, 24, 24, 3, 1, 1), (3, 64, 64, 13, 9, 3, 1, 1),
                                  (2, 32, 32, 24, 20, 3, 1, 1), (16, 256, 256, 6, 6, 3, 1, 1),
                                  # 1x1 windows: also the configurations that stream A fragments from global memory
                                  (2, 64, 128, 16, 16, 1, 1, 0), (4, 128, 32, 24, 24, 1, 1, 0), (3, 32, 64, 13, 9, 1, 1, 0),
                                  (2, 32, 64, 12, 10, 2, 2, 0), (3, 64, 64, 7, 9, 2, 2, 0)])      # 2x2 / stride 2 (odd input: last row / column unused)
def test_convp_every_tile_configuration(case):
    """The tuning table (dfl_conv_tune_add) may select any candidate of the geometry search: every tile configuration the
    layer admits -- including the two-column-tile ones the cost model never picks -- is forced once per K-slice count
    (dfl_conv_force_geometry) and must give the same result."""
    N, Cin, Cout, H, W, K, stride, pad = case
    lib = nat.lib()
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g)
    wp = pack16(w, 1)
    ref = nhwc(F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)))
    cands = _candidates(*case)
    seen, tiles = set(), set()
    for geom in cands:
        if (geom[0], geom[4]) in seen or geom[4] > 4:
            continue
        seen.add((geom[0], geom[4]))
        tiles.add(geom[0])
        gv = (C.c_int32 * 5)(*geom)
        nat.check(lib.dfl_conv_force_geometry(C.addressof(gv)), 'force')
        try:
            y, st = conv_bf16(x, wp, Cout, K, K, stride, pad, ref.shape[1], ref.shape[2], bias=b, relu=1, stats=True, force_splits=geom[4])
        finally:
            lib.dfl_conv_force_geometry(None)
        close_bf16(y, ref, '%s geometry %s' % (case, geom))
        yd = y.double().reshape(-1, Cout)
        np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
    assert len(tiles) >= 3, tiles


def test_convp_streamed_1x1_forms_with_scatter_and_residual_epilogues():
    """The global-A configurations (tile index >= 22) under the epilogues 1x1 layers use in the network: the transposed
    convolution's 2x2 scatter into a channel half of a wider buffer, and bias + '+ BN(other)' + statistics."""
    lib = nat.lib()
    g = torch.Generator().manual_seed(77)
    N, Ci, Co, H, W = 2, 64, 32, 10, 7
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    wt = rb(torch.randn(Ci, Co, 2, 2, generator=g) / (4 * Ci) ** 0.5)
    bt = torch.randn(Co, generator=g)
    ref_t = nhwc(F.conv_transpose2d(x.double(), wt.double(), bt.double(), stride=2))
    w1 = rb(torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5)
    b1 = torch.randn(Co, generator=g)
    other = rb(torch.randn(N, Co, H, W, generator=g))
    asc, ash = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2
    ref_1 = nhwc(F.conv2d(x.double(), w1.double(), b1.double()) + other.double() * asc.double().view(1, -1, 1, 1) + ash.double().view(1, -1, 1, 1))
    a = nat.ConvArgs()
    a.x = a.w = a.y = 4096
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = N, H, W, Ci, Ci
    a.KH = a.KW = 1
    a.stride, a.pad = 1, 0
    out = (C.c_int32 * (5 * 4096))()
    seen = 0
    for scatter in (1, 0):
        a.scatter2x2 = scatter
        a.Hout, a.Wout = (2 * H, 2 * W) if scatter else (H, W)
        a.Ntot, a.ldy = (4 * Co, 2 * Co) if scatter else (Co, Co)
        n = nat.check(lib.dfl_conv_candidates(C.addressof(a), C.addressof(out), 4096), 'candidates')
        cands = [tuple(out[5 * i + j] for j in range(5)) for i in range(n)]
        tiles = {}
        for c in cands:
            if c[0] >= 22:
                tiles.setdefault(c[0], c)
        assert tiles, 'no streamed configuration among the candidates'
        for geom in tiles.values():
            gv = (C.c_int32 * 5)(*geom)
            nat.check(lib.dfl_conv_force_geometry(C.addressof(gv)), 'force')
            try:
                if scatter:
                    y = conv_bf16(x, pack16(wt, 3), 4 * Co, 1, 1, 1, 0, 2 * H, 2 * W, bias=bt, scatter=1, ldy=2 * Co, force_splits=1)
                    close_bf16(y, ref_t, 'scatter %s' % (geom,))
                else:
                    y, st = conv_bf16(x, pack16(w1, 1), Co, 1, 1, 1, 0, H, W, bias=b1, add=other, add_aff=(asc, ash), stats=True, force_splits=1)
                    close_bf16(y, ref_1, 'residual %s' % (geom,))
                    yd = y.double().reshape(-1, Co)
                    np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
            finally:
                lib.dfl_conv_force_geometry(None)
            seen += 1
    assert seen >= 4


def test_convp_tuning_table_entry_is_used_and_validated():
    lib = nat.lib()
    case = (2, 64, 128, 16, 16, 3, 1, 1)
    a = nat.ConvArgs()
    a.x = a.w = a.y = 4096
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = 2, 16, 16, 64, 64
    a.KH = a.KW = 3
    a.stride, a.pad, a.Hout, a.Wout, a.Ntot, a.ldy = 1, 1, 16, 16, 128, 128
    model = lib.dfl_conv_config(C.addressof(a))
    other = [gm for gm in _candidates(*case) if gm[0] + 16 != model and gm[4] == 1][0]
    key = (C.c_int32 * 10)(2, 16, 16, 64, 128, 3, 3, 1, 1, 0)
    gv = (C.c_int32 * 5)(*other)
    try:
        nat.check(lib.dfl_conv_tune_add(C.addressof(key), C.addressof(gv)), 'tune_add')
        assert lib.dfl_conv_config(C.addressof(a)) == other[0] + 16
        bad = (C.c_int32 * 5)(other[0], 1, 64, 64, 1)            # a patch larger than the tile: ignored, the model decides
        nat.check(lib.dfl_conv_tune_add(C.addressof(key), C.addressof(bad)), 'tune_add')
        assert lib.dfl_conv_config(C.addressof(a)) == model
    finally:
        # leave the table as the library loaded it
        lib.dfl_conv_tune_add(None, None)
        nat.load_tuning(lib, nat.TUNE_PATH)


@pytest.mark.parametrize('case', [(2, 32, 32, 20, 20, 3), (2, 64, 128, 12, 12, 3), (4, 256, 256, 6, 6, 3), (2, 128, 64, 9, 7, 1)])
@pytest.mark.parametrize('splits', [None, 2])
def test_convp_affine_residual_epilogue(case, splits):
    """BatchNorm affine on load with zero padding AFTER it, '+ BN(other)' residual sum, accumulate, statistics against a
    partner tensor -- the forward block epilogue (unet.py:229-231) and the fused backward sums -- also through K slices."""
    N, Cin, Cout, H, W, K = case
    if splits and Cin // 16 < 2 * splits:
        pytest.skip('too few channel blocks to slice')
    pad = K // 2
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    other = rb(torch.randn(N, Cout, H, W, generator=g))
    asc, ash = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    y0 = rb(torch.randn(N, Cout, H, W, generator=g))
    partner = rb(torch.randn(N, Cout, H, W, generator=g))
    xa = rb(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))           # the kernel rounds the affine result to bf16 once
    ref = F.conv2d(xa.double(), w.double(), b.double(), padding=pad)    # padding stays zero: it is applied after the affine
    ref = ref + other.double() * asc.double().view(1, -1, 1, 1) + ash.double().view(1, -1, 1, 1) + y0.double()
    y, st = conv_bf16(x, pack16(w, 1), Cout, K, K, 1, pad, H, W, bias=b, in_aff=(sc, sh), add=other, add_aff=(asc, ash),
                      y_init=y0, accumulate=1, stats=True, stat_other=partner, force_splits=splits)
    close_bf16(y, nhwc(ref), str(case))
    yd = y.double().reshape(-1, Cout)
    pd = nhwc(partner).double().reshape(-1, Cout)
    np.testing.assert_allclose(st[1].numpy(), (yd * pd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * pd).abs().sum(0).max()))


def test_convp_transposed_scatter_and_data_gradients():
    """ConvTranspose2d(k2,s2) as a 1x1 gather with the 2x2-scatter epilogue into the channel half of a wider buffer
    (unet.py:240,255-257), the data gradient of a stride-1 conv (flipped / transposed weights) and of a 2x2/s2 conv
    (scatter form, accumulating)."""
    g = torch.Generator().manual_seed(9)
    N, Ci, Co, H, W = 2, 64, 32, 10, 7
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    w = rb(torch.randn(Ci, Co, 2, 2, generator=g) / (4 * Ci) ** 0.5)
    b = torch.randn(Co, generator=g)
    y = conv_bf16(x, pack16(w, 3), 4 * Co, 1, 1, 1, 0, 2 * H, 2 * W, bias=b, scatter=1, ldy=2 * Co)
    close_bf16(y, nhwc(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)), 'convT')
    # data gradient of a 3x3 / stride 1 / pad 1 conv
    Cin, Cout = 32, 64
    dy = rb(torch.randn(N, Cout, 12, 9, generator=g))
    w3 = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cout) ** 0.5)
    dx = conv_bf16(dy, pack16(w3, 2, flip=1), Cin, 3, 3, 1, 1, 12, 9)
    close_bf16(dx, nhwc(F.conv_transpose2d(dy.double(), w3.double(), padding=1)), 'dgrad 3x3')
    # data gradient of conv 2x2 / stride 2 (scatter form), accumulated onto an existing gradient
    dn = rb(torch.randn(N, 32, 6, 5, generator=g))
    w2 = rb(torch.randn(32, 32, 2, 2, generator=g) / 128 ** 0.5)
    d0 = rb(torch.randn(N, 32, 12, 10, generator=g))
    dd = conv_bf16(dn, pack16(w2, 3), 4 * 32, 1, 1, 1, 0, 12, 10, scatter=1, y_init=d0, accumulate=1)
    close_bf16(dd, nhwc(d0.double() + F.conv_transpose2d(dn.double(), w2.double(), stride=2)), 'dgrad 2x2 s2')
    # ... leaving the BatchNorm-backward sums of the finished gradient against a partner tensor (one pass and through K slices)
    for Cd, sp in ((32, None), (64, None), (64, 2), (128, 2)):
        dn = rb(torch.randn(N, Cd, 6, 5, generator=g))
        w2 = rb(torch.randn(Cd, Cd, 2, 2, generator=g) / (4 * Cd) ** 0.5)
        d0 = rb(torch.randn(N, Cd, 12, 10, generator=g))
        partner = rb(torch.randn(N, Cd, 12, 10, generator=g))
        dd, st = conv_bf16(dn, pack16(w2, 3), 4 * Cd, 1, 1, 1, 0, 12, 10, scatter=1, y_init=d0, accumulate=1, stats=True,
                           stat_other=partner, force_splits=sp)
        close_bf16(dd, nhwc(d0.double() + F.conv_transpose2d(dn.double(), w2.double(), stride=2)), 'dgrad 2x2 s2 + sums %d' % Cd)
        yd, pd = dd.double().reshape(-1, Cd), nhwc(partner).double().reshape(-1, Cd)
        np.testing.assert_allclose(st[0].numpy(), yd.sum(0).numpy(), rtol=2e-5, atol=2e-5 * float(yd.abs().sum(0).max()))
        np.testing.assert_allclose(st[1].numpy(), (yd * pd).sum(0).numpy(), rtol=2e-5, atol=2e-5 * float((yd * pd).abs().sum(0).max()))
    # data gradient of the transposed conv = conv 2x2 / stride 2 over dy
    dyT = rb(torch.randn(N, Co, 2 * H, 2 * W, generator=g))
    du = conv_bf16(dyT, pack16(w, 1), Ci, 2, 2, 2, 0, H, W)
    ref_du = torch.einsum('nchawb,kcab->nkhw', dyT.double().reshape(N, Co, H, 2, W, 2), w.double())
    close_bf16(du, nhwc(ref_du), 'dgrad convT')


def wgrad_bf16(gx, d, KH, KW, stride, pad, Hout, Wout, in_aff=None, force_splits=None, brb=None, bias_plain=False):
    """gx: gathered tensor NCHW, d: dense NCHW (both bf16-representable) -> dw [Cm][Cg][KH][KW] fp32 (cpu)."""
    lib = nat.lib()
    N, Cg, Hin, Win = gx.shape
    Cm = d.shape[1]
    gd = nhwc(gx).to(DEV).to(BF).contiguous()
    dd = nhwc(d).to(DEV).to(BF).contiguous()
    dw = torch.full((Cm, Cg, KH, KW), float('nan'), device=DEV)
    a = nat.WgradArgs()
    keep = [gd, dd, dw]
    a.g, a.d, a.dw = gd.data_ptr(), dd.data_ptr(), dw.data_ptr()
    a.g_bf16, a.d_bf16 = 1, 1
    if in_aff is not None:
        sc, sh = in_aff[0].to(DEV), in_aff[1].to(DEV)
        keep += [sc, sh]
        a.in_scale, a.in_shift = sc.data_ptr(), sh.data_ptr()
    a.N, a.Hin, a.Win, a.Cg, a.ldg = N, Hin, Win, Cg, Cg
    a.KH, a.KW, a.stride, a.pad = KH, KW, stride, pad
    a.Hout, a.Wout, a.Cm, a.ldd = Hout, Wout, Cm, Cm
    if brb is not None:                                 # d is dy; operand = [r > 0] * (A dy + B r + C) formed in the staging (d_mode)
        r_, coef = brb
        rd = F.pad(nhwc(r_), (0, 8)).to(DEV).to(BF).contiguous()
        keep.append(rd)
        a.d_mode, a.d2, a.ldd2 = 1, rd.data_ptr(), rd.shape[-1]
        if coef is not None:
            cd = coef.to(DEV).contiguous()
            keep.append(cd)
            a.coef = cd.data_ptr()
    a.splits = 1
    s = force_splits or nat.check(lib.dfl_wgrad_suggest_splits(C.addressof(a)), 'suggest')
    a.splits = s
    bias_part = None
    if brb is not None or bias_plain:                   # (bias_plain: d is the operand itself, its column sums leave all the same)
        bias_part = torch.full((s, Cm), float('nan'), device=DEV)
        a.bias_partial = bias_part.data_ptr()
    n = Cm * Cg * KH * KW
    if s > 1:
        part = torch.full((s * n,), float('nan'), device=DEV)
        keep.append(part)
        a.partial = part.data_ptr()
    assert nat.check(lib.dfl_wgrad_config(C.addressof(a)), 'config') >= 16
    nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), stream()), 'dfl_conv2d_wgrad')
    if s > 1:
        nat.check(lib.dfl_sum_partials(part.data_ptr(), dw.data_ptr(), n, s, KH * KW, stream()), 'dfl_sum_partials')
    torch.cuda.synchronize()
    if bias_part is not None:
        return dw.cpu(), bias_part.cpu().double().sum(0)
    return dw.cpu()


def brb_reference(dy, r, coef):
    """d(pre-activation) of BatchNorm + ReLU backward as the kernels form it: fp32 arithmetic on bf16 values, rounded to bf16."""
    if coef is None:
        return rb(torch.where(r > 0, dy, torch.zeros(())))
    A, B, Cc = (coef[i].view(1, -1, 1, 1).double() for i in range(3))
    t = (B * r.double() + Cc).float()                       # fmaf(B, r, C): one rounding (the fp64 sum of an fp32 product is exact here)
    v = (A * dy.double() + t.double()).float()              # fmaf(A, dy, t)
    return rb(torch.where(r > 0, v, torch.zeros(())))


@pytest.mark.parametrize('case', [(2, 32, 32, 20, 20), (2, 64, 128, 12, 12), (4, 256, 256, 6, 6), (3, 128, 64, 9, 7), (2, 32, 32, 96, 96),
                                  (2, 32, 32, 192, 192), (2, 64, 64, 96, 96)])
@pytest.mark.parametrize('with_bn', [True, False])
def test_fused_bn_relu_backward_operand(case, with_bn):
    """dfl_conv_args.x_mode / dfl_wgrad_args.d_mode: the data-gradient convolution and the weight gradient form
    [r > 0] * (A dy + B r + C) from (dy, r) while they stage their patches -- what dfl_bn_relu_bwd_apply would have written to
    HBM -- and the weight gradient leaves the column sums of it (the bias gradient).  Against fp64 on the bf16-rounded operand."""
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    dy = rb(torch.randn(N, Cin, H, W, generator=g))
    r = rb(torch.relu(torch.randn(N, Cin, H, W, generator=g)))
    coef = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3, torch.randn(Cin, generator=g) * 0.1]) if with_bn else None
    dpre = brb_reference(dy, r, coef)
    # data gradient: a 3x3 convolution over dpre (zero padding applies to dpre: outside pixels contribute nothing)
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    ref = F.conv2d(dpre.double(), w.double(), padding=1)
    for splits in (None, 2 if Cin >= 128 else None):
        y = conv_bf16(dy, pack16(w, 1), Cout, 3, 3, 1, 1, H, W, brb=(r, coef), force_splits=splits)
        close_bf16(y, nhwc(ref), 'x_mode %s splits %s' % (str(case), splits))
        # x_out: the operand itself, written once by the kernel that forms it -- every element, BIT-EXACT, nothing beyond Cin
        y2, xo = conv_bf16(dy, pack16(w, 1), Cout, 3, 3, 1, 1, H, W, brb=(r, coef), force_splits=splits, x_out=True)
        assert torch.equal(y2, y)
        assert torch.equal(xo[..., :Cin], nhwc(dpre)), 'x_out %s splits %s: %d elements differ' % (
            str(case), splits, int((xo[..., :Cin] != nhwc(dpre)).sum()))
        assert bool(torch.isnan(xo[..., Cin:]).all())
    # weight gradient with dpre as the dense operand + the bias gradient
    x = rb(torch.randn(N, Cout, H, W, generator=g))                  # the layer input (gathered operand), Cg = Cout here
    refb = dpre.double().sum(dim=(0, 2, 3))
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    for in_aff in (None, (sc, sh)):                                   # ... without / with the BatchNorm affine on the gathered operand
        xa = x if in_aff is None else rb((x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float())   # fmaf: one rounding
        dw, bias = wgrad_bf16(x, dy, 3, 3, 1, 1, H, W, brb=(r, coef), in_aff=in_aff)
        refw = torch.nn.grad.conv2d_weight(xa.double(), (Cin, Cout, 3, 3), dpre.double(), padding=1)
        np.testing.assert_allclose(dw.numpy(), refw.numpy(), rtol=2e-5, atol=3e-5 * float(refw.abs().max()))
        np.testing.assert_allclose(bias.numpy(), refb.numpy(), rtol=2e-5, atol=3e-5 * float(dpre.double().abs().sum(dim=(0, 2, 3)).max()))
        # ... and from the materialised operand (d_mode 0 + bias_partial: what follows a data gradient with x_out)
        dw2, bias2 = wgrad_bf16(x, dpre, 3, 3, 1, 1, H, W, in_aff=in_aff, bias_plain=True)
        np.testing.assert_allclose(dw2.numpy(), refw.numpy(), rtol=2e-5, atol=3e-5 * float(refw.abs().max()))
        np.testing.assert_allclose(bias2.numpy(), refb.numpy(), rtol=2e-5, atol=3e-5 * float(dpre.double().abs().sum(dim=(0, 2, 3)).max()))


@pytest.mark.parametrize('case', [(2, 32, 20, 20, 1), (3, 16, 13, 9, 1), (1, 64, 4, 7, 1), (2, 32, 192, 192, 1), (2, 32, 12, 14, 0)])
@pytest.mark.parametrize('with_bn', [True, False])
def test_fused_bn_relu_backward_first_layer(case, with_bn):
    """The same operand in the network's first layer (1-channel fp32 input, direct 3x3 weight-gradient kernel in its row form):
    d_mode with a bf16 (dy, r) pair, weight gradient and bias-gradient sums against fp64 on the bf16-rounded operand."""
    lib = nat.lib()
    N, Cm, H, W, pad = case
    Hin, Win = H + 2 - 2 * pad, W + 2 - 2 * pad
    g = torch.Generator().manual_seed(sum(case) + 3)
    dy = rb(torch.randn(N, Cm, H, W, generator=g))
    r = rb(torch.relu(torch.randn(N, Cm, H, W, generator=g)))
    coef = torch.stack([torch.rand(Cm, generator=g) + 0.5, torch.randn(Cm, generator=g) * 0.3, torch.randn(Cm, generator=g) * 0.1]) if with_bn else None
    dpre = brb_reference(dy, r, coef)
    x = torch.randn(N, 1, Hin, Win, generator=g)
    xd = nhwc(x).to(DEV).contiguous()
    dyd = nhwc(dy).to(DEV).to(BF).contiguous()
    rd = F.pad(nhwc(r), (0, 8)).to(DEV).to(BF).contiguous()
    dw = torch.full((Cm, 1, 3, 3), float('nan'), device=DEV)
    a = nat.WgradArgs()
    a.g, a.d, a.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
    a.g_bf16, a.d_bf16 = 0, 1
    a.N, a.Hin, a.Win, a.Cg, a.ldg = N, Hin, Win, 1, 1
    a.KH, a.KW, a.stride, a.pad = 3, 3, 1, pad
    a.Hout, a.Wout, a.Cm, a.ldd = H, W, Cm, Cm
    a.d_mode, a.d2, a.ldd2 = 1, rd.data_ptr(), rd.shape[-1]
    cd = coef.to(DEV).contiguous() if coef is not None else None
    a.coef = nat.ptr(cd)
    a.splits = 1
    s = nat.check(lib.dfl_wgrad_suggest_splits(C.addressof(a)), 'suggest')
    a.splits = s
    bias_part = torch.full((s, Cm), float('nan'), device=DEV)
    a.bias_partial = bias_part.data_ptr()
    part = torch.full((max(s, 2) * Cm * 9,), float('nan'), device=DEV)
    a.partial = part.data_ptr()
    nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), stream()), 'dfl_conv2d_wgrad')
    if s > 1:
        nat.check(lib.dfl_sum_partials(part.data_ptr(), dw.data_ptr(), Cm * 9, s, 9, stream()), 'dfl_sum_partials')
    torch.cuda.synchronize()
    refw = torch.nn.grad.conv2d_weight(x.double(), (Cm, 1, 3, 3), dpre.double(), padding=pad)
    np.testing.assert_allclose(dw.cpu().numpy(), refw.numpy(), rtol=2e-5, atol=3e-5 * float(refw.abs().max()))
    refb = dpre.double().sum(dim=(0, 2, 3))
    np.testing.assert_allclose(bias_part.cpu().double().sum(0).numpy(), refb.numpy(), rtol=2e-5,
                               atol=3e-5 * float(dpre.double().abs().sum(dim=(0, 2, 3)).max()))


WCASES = [
    # N, Cg, Cm, H, W, K, stride, pad
    (2, 32, 32, 24, 20, 3, 1, 1),
    (3, 16, 16, 12, 12, 3, 1, 1),
    (2, 64, 32, 13, 9, 3, 1, 1),       # ragged patches, two tile pairs
    (2, 32, 64, 17, 11, 3, 1, 1),
    (4, 128, 128, 12, 12, 3, 1, 1),    # several workgroup tiles
    (16, 256, 128, 6, 6, 3, 1, 1),     # several images per patch
    (2, 64, 64, 8, 8, 1, 1, 0),        # 1x1
    (2, 32, 64, 12, 10, 2, 2, 0),      # 2x2 stride 2
    (2, 32, 32, 96, 96, 3, 1, 1),      # wide image: row patches, many pixel slices
    (2, 64, 64, 40, 300, 3, 1, 1),     # rows longer than a patch
    (2, 96, 96, 12, 12, 3, 1, 1),      # 3 x 3 tiles: with 2 slices the XCD-aware order has a tail (18 workgroups)
]


@pytest.mark.parametrize('case', WCASES)
def test_wgradp(case):
    N, Cg, Cm, H, W, K, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(N, Cg, H, W, generator=g))
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    d = rb(torch.randn(N, Cm, Ho, Wo, generator=g))
    dw = wgrad_bf16(x, d, K, K, stride, pad, Ho, Wo)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cm, Cg, K, K), d.double(), stride=stride, padding=pad)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(dw.numpy(), ref.numpy(), rtol=2e-5, atol=3e-5 * scale)
    if K == 3:      # one slot (direct torch-order store) and an explicit slice count
        for s in (1, 2, 3):
            if Cm > 32 and Cg > 32 or s > 1:
                dws = wgrad_bf16(x, d, K, K, stride, pad, Ho, Wo, force_splits=s)
                np.testing.assert_allclose(dws.numpy(), ref.numpy(), rtol=2e-5, atol=3e-5 * scale)


@pytest.mark.parametrize('case', [
    # N, Cg, Cm, H, W
    (2, 64, 32, 14, 10),
    (3, 32, 64, 9, 21),        # ragged: width not a multiple of 8, odd height
    (2, 128, 64, 24, 24),      # several cg tiles, whole-image patches
    (16, 64, 64, 6, 6),        # several images per patch, every pixel on a border
    (1, 32, 32, 70, 40),       # interior patches, border patches of all nine kinds
    (2, 96, 96, 12, 12),
])
def test_wgradp_affine_on_load(case):
    """BatchNorm affine of the gathered operand, applied while the patch is staged: the operand is bf16(fma(x, scale, shift)) inside
    the image and zero outside (the padding is applied to the BatchNorm output, unet.py:214-222) -- ragged widths, whole-image
    patches, images that are all border."""
    N, Cg, Cm, H, W = case
    g = torch.Generator().manual_seed(4 + sum(case))
    x = rb(torch.randn(N, Cg, H, W, generator=g))
    d = rb(torch.randn(N, Cm, H, W, generator=g) + 0.25)
    sc, sh = torch.rand(Cg, generator=g) + 0.5, torch.randn(Cg, generator=g) * 0.3
    xa = rb((x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float())     # fmaf: one rounding to fp32, then bf16
    ref = torch.nn.grad.conv2d_weight(xa.double(), (Cm, Cg, 3, 3), d.double(), padding=1)
    for s in (None, 1, 3):
        dw = wgrad_bf16(x, d, 3, 3, 1, 1, H, W, in_aff=(sc, sh), force_splits=s)
        np.testing.assert_allclose(dw.numpy(), ref.numpy(), rtol=2e-5, atol=3e-5 * float(ref.abs().max()))
    dw, bias = wgrad_bf16(x, d, 3, 3, 1, 1, H, W, in_aff=(sc, sh), bias_plain=True)
    np.testing.assert_allclose(dw.numpy(), ref.numpy(), rtol=2e-5, atol=3e-5 * float(ref.abs().max()))
    refb = d.double().sum(dim=(0, 2, 3))
    np.testing.assert_allclose(bias.double().numpy(), refb.numpy(), rtol=2e-5, atol=3e-5 * float(d.double().abs().sum(dim=(0, 2, 3)).max()))


def test_streaming_kernels_bf16():
    """dfl_bn_relu_bwd_apply, dfl_colstats, dfl_affine_copy, max-pool forward / backward with bf16 tensors."""
    lib = nat.lib()
    g = torch.Generator().manual_seed(11)
    M, Cc = 1000, 48
    dy = rb(torch.randn(M, Cc, generator=g))
    r = rb(torch.relu(torch.randn(M, Cc, generator=g)))
    coef = torch.randn(3, Cc, generator=g)
    dyd, rd = dy.to(DEV).to(BF), r.to(DEV).to(BF)
    dpre = torch.empty(M, Cc, device=DEV, dtype=BF)
    nb = lib.dfl_rowblock_count(M, Cc)
    part = torch.zeros(nb, Cc, device=DEV)
    cd = coef.to(DEV)
    nat.call('dfl_bn_relu_bwd_apply', nat.BnReluBwdArgs(dy=dyd.data_ptr(), r=rd.data_ptr(), coef=cd.data_ptr(), dpre=dpre.data_ptr(),
                                                        partials=part.data_ptr(), M=M, C=Cc, lddy=Cc, ldr=Cc, ldo=Cc, nblocks=nb,
                                                        bf16=1), stream())
    ref = torch.where(r > 0, coef[0] * dy + coef[1] * r + coef[2], torch.zeros(()))
    close_bf16(dpre.float().cpu(), ref, 'bn_relu_bwd')
    np.testing.assert_allclose(part.sum(0).cpu().numpy(), dpre.float().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)
    parts = torch.zeros(nb, 2, Cc, device=DEV)
    nat.call('dfl_colstats', nat.ColstatsArgs(a=dyd.data_ptr(), b=rd.data_ptr(), partials=parts.data_ptr(), M=M, C=Cc, lda=Cc, ldb=Cc,
                                             nblocks=nb, bf16=1), stream())
    s = parts.sum(0).cpu().double()
    np.testing.assert_allclose(s[0].numpy(), dy.double().sum(0).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(s[1].numpy(), (dy.double() * r.double()).sum(0).numpy(), rtol=1e-5, atol=1e-4)
    # affine copy with crop window, then max-pool forward / backward
    N, H, W = 2, 10, 8
    x = rb(torch.randn(N, H, W, Cc, generator=g))
    xd = x.to(DEV).to(BF)
    sc, sh = (torch.rand(Cc, generator=g) + 0.5).to(DEV), torch.randn(Cc, generator=g).to(DEV)
    y = torch.zeros(N, 6, 4, Cc, device=DEV, dtype=BF)
    nat.call('dfl_affine_copy', nat.AffineCopyArgs(x=xd.data_ptr(), y=y.data_ptr(), scale=sc.data_ptr(), shift=sh.data_ptr(), N=N, H=6,
                                                   W=4, C=Cc, ldx=Cc, xH=H, xW=W, xoy=2, xox=3, ldy=Cc, yH=6, yW=4, bf16=1), stream())
    close_bf16(y.float().cpu(), x[:, 2:8, 3:7] * sc.cpu() + sh.cpu(), 'affine_copy')
    p = torch.empty(N, H // 2, W // 2, Cc, device=DEV, dtype=BF)
    nat.call('dfl_maxpool2x2_fwd', nat.PoolArgs(x=xd.data_ptr(), y=p.data_ptr(), N=N, H=H, W=W, C=Cc, ldx=Cc, ldy=Cc, bf16=1), stream())
    pref = F.max_pool2d(x.permute(0, 3, 1, 2), 2)
    assert torch.equal(p.float().cpu(), pref.permute(0, 2, 3, 1))
    gp = rb(torch.randn(N, H // 2, W // 2, Cc, generator=g))
    gpd = gp.to(DEV).to(BF)
    dx = torch.zeros(N, H, W, Cc, device=DEV, dtype=BF)
    nat.call('dfl_maxpool2x2_bwd', nat.PoolArgs(x=xd.data_ptr(), y=gpd.data_ptr(), dx=dx.data_ptr(), N=N, H=H, W=W, C=Cc, ldx=Cc, ldy=Cc,
                                                lddx=Cc, bf16=1), stream())
    xt = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.max_pool2d(xt, 2).backward(gp.permute(0, 3, 1, 2))
    assert torch.equal(dx.float().cpu(), xt.grad.permute(0, 2, 3, 1))


# ---------------------------------------------------------------------------------------------------- whole network
@pytest.mark.parametrize('NC,L,two,softmax', [(7, 14, True, 1), (7, 0, True, 1), (5, 9, False, 1), (5, 9, True, 1), (8, 16, True, 0),
                                              (3, 0, True, 0), (2, 3, True, 1)])
def test_head_backward_fused_weight_gradients(NC, L, two, softmax):
    """dfl_head_fwd / dfl_head_bwd with bf16 features (F = 32; unet.py:176-191 under autograd): outputs, dx and the three head
    weight gradients without the per-pixel scratch -- on the matrix cores when the head has two landmark layers or none (fp32
    weights and computed operands as hi + lo bf16 pairs: the outputs keep the fp32 kernels' bar), thread-per-pixel otherwise.
    The weight gradients round [dlogits | dmid | dheat] and [logits | mid] to bf16 (x is bf16 already): 2^-9 relative per
    product, averaged over the pixels -- the bar is 2^-8 of the largest gradient magnitude of each tensor, dx keeps the bar
    of a bf16 result."""
    lib = nat.lib()
    g = torch.Generator().manual_seed(NC + L + 1)
    N, H, W, F_ = 3, 37, 29, 32                                     # 3219 pixels: ragged last tile, images that straddle tiles
    x = rb(torch.randn(N, F_, H, W, generator=g)).double().requires_grad_(True)
    wseg = (torch.randn(NC, F_, 1, 1, generator=g, dtype=torch.float64) / 3).float().double().requires_grad_(True)
    NM = (NC + L if two else L) if L > 0 else 0
    logits = F.conv2d(x, wseg)
    seg = torch.softmax(logits, 1) if softmax else logits
    outs = [seg]
    w1 = w2 = None
    if L > 0:
        w1 = (torch.randn(NM, F_ + NC, 1, 1, generator=g, dtype=torch.float64) / 3).float().double().requires_grad_(True)
        mid = F.conv2d(torch.cat((x, logits), 1), w1)
        heat = mid
        if two:
            w2 = (torch.randn(L, NM, 1, 1, generator=g, dtype=torch.float64) / 3).float().double().requires_grad_(True)
            heat = F.conv2d(mid, w2)
        outs.append(heat)
    gouts = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in outs]
    torch.autograd.backward(outs, gouts)
    dv = lambda t: t.detach().float().contiguous().to(DEV)
    xd = F.pad(nhwc(x.detach().float()), (0, 8)).to(DEV).to(BF).contiguous()      # pixel stride 40
    wsd, w1d, w2d = dv(wseg), (dv(w1) if w1 is not None else None), (dv(w2) if w2 is not None else None)
    seg_out = torch.full((N, NC, H, W), float('nan'), device=DEV)
    heat_out = torch.full((N, L, H, W), float('nan'), device=DEV) if L > 0 else None
    nat.call('dfl_head_fwd', nat.HeadFwdArgs(
        x=xd.data_ptr(), w_seg=wsd.data_ptr(), w_l1=nat.ptr(w1d), w_l2=nat.ptr(w2d), seg=seg_out.data_ptr(), heat=nat.ptr(heat_out),
        N=N, H=H, W=W, F=F_, ldx=F_ + 8, NC=NC, NM=NM, L=L, softmax=softmax, x_bf16=1), stream())
    torch.cuda.synchronize()
    # (three bf16 products per product: 2^-17 of the terms' magnitude -- the bar of the bf16x3 mode)
    np.testing.assert_allclose(seg_out.cpu().numpy(), seg.detach().numpy(), rtol=1e-4, atol=1e-5 * float(seg.detach().abs().max()))
    if L > 0:
        np.testing.assert_allclose(heat_out.cpu().numpy(), outs[1].detach().numpy(), rtol=1e-4, atol=1e-5 * float(outs[1].detach().abs().max()))
    segd = dv(seg)
    dsegd = dv(gouts[0])
    dheatd = dv(gouts[1]) if L > 0 else None
    M = N * H * W
    nb = nat.check(lib.dfl_head_wgrad_blocks(M), 'blocks')
    part = torch.full((nb * 4096,), float('nan'), device=DEV)
    dxd = torch.full((N, H, W, F_), float('nan'), device=DEV, dtype=BF)
    dws = torch.full((NC, F_), float('nan'), device=DEV)
    dw1 = torch.full((NM, F_ + NC), float('nan'), device=DEV) if L > 0 else None
    dw2 = torch.full((L, NM), float('nan'), device=DEV) if (L > 0 and two) else None
    nat.call('dfl_head_bwd', nat.HeadBwdArgs(
        x=xd.data_ptr(), seg=segd.data_ptr(), dseg=dsegd.data_ptr(), dheat=nat.ptr(dheatd), w_seg=wsd.data_ptr(),
        w_l1=nat.ptr(w1d), w_l2=nat.ptr(w2d), dx=dxd.data_ptr(), N=N, H=H, W=W, F=F_, ldx=F_ + 8, lddx=F_, NC=NC, NM=NM, L=L,
        softmax=softmax, x_bf16=1, dw_seg=dws.data_ptr(), dw_l1=nat.ptr(dw1), dw_l2=nat.ptr(dw2), wg_partial=part.data_ptr()), stream())
    torch.cuda.synchronize()
    close_bf16(dxd.float().cpu(), nhwc(x.grad), 'dx')
    for got, ref, name in ((dws, wseg.grad, 'seg_conv'), (dw1, None if w1 is None else w1.grad, 'lands_1x1.0'),
                           (dw2, None if w2 is None else w2.grad, 'lands_1x1.1')):
        if got is None:
            continue
        r = ref[:, :, 0, 0]
        err = float((got.cpu().double() - r).abs().max())
        assert err <= 2.0 ** -8 * float(r.abs().max()), (name, err, float(r.abs().max()))


def _eps4():
    return NF.conv_rel_error('bf16s')


@pytest.mark.parametrize('cfgname', ['paper_sc_l14', 'paper_mp_l0'])
def test_network_bf16_storage_against_fp64_oracle(cfgname):
    """Paper presets, batch 2, free running against the CLEAN fp64 oracle: forward deviation at the bf16 level, labels identical
    outside the margin that deviation implies, whole gradient within what bf16 rounding of 25 layers amounts to (a sanity bar:
    the parity gate of this arithmetic is the step-by-step check of the same problems, tests/test_gpu_bf16_stepwise.py)."""
    torch.set_num_threads(max(torch.get_num_threads(), 32))
    gf = NF.cached_check('paper__%s__b2' % cfgname, lambda: PR.paper(cfgname, 2))
    pr = gf.problem
    net = hip_net(pr)
    out, seg, loss = hip_step(pr, net)
    plan = NF.train_plan(net)
    assert plan.bf16 and plan.feat.t.dtype == torch.bfloat16, 'the recorded program must hold bf16 activations'
    dev = float((seg.detach().double().cpu() - gf.out).abs().max())
    assert dev < 5e-2, 'soft-max deviation %.3e from fp64' % dev
    assert dev > 1e-5, 'the bf16 storage mode does not seem to be in effect'
    top2 = gf.out.topk(2, dim=1)[0]
    sure = (top2[:, 0] - top2[:, 1]) > 2.5 * dev
    assert float(sure.float().mean()) > 0.5
    assert bool((seg.detach().argmax(1).cpu() == gf.out.argmax(1))[sure].all())
    res = gf.whole_error(net, seg)
    print('%s bf16 storage against clean fp64: forward %.3e, whole-gradient error %.3e, decisions forced %d ReLU / %d pool (of %d), largest '
          'margin %.2e' % (cfgname, res['d_fwd'], res['whole'], res['info']['relu_flips'], res['info']['pool_flips'], res['info']['relu_total'],
                           res['info']['max_margin']))
    assert res['whole'] <= 5e-2 and res['d_fwd'] <= 2.4e-2


@pytest.mark.parametrize('key', ['upsample__circular__wf5', 'landsblock__1__valid', 'landsblock__2'])
def test_constructor_flags_in_bf16_storage(key):
    """The constructor values no reference CLI selects -- up_mode='upsample' + pad_mode='circular' (bf16 forms of dfl_upsample2x_*
    and of the frame copies), lands_block_depth with and without padding (two head calls) -- in the
    bf16 storage arithmetic: forward at bf16 distance from the fp64 oracle, every gradient inside its bar on the run's pattern."""
    pr = PR.REGISTRY[key]()
    if any(2 ** (pr.cfg['wf'] + i) % 16 for i in range(pr.cfg['depth'])):
        pytest.skip('channel counts below 16')
    gc = NF.cached_check(key, lambda: pr)
    net = hip_net(gc.problem)
    out, seg, loss = hip_step(gc.problem, net)
    assert NF.train_plan(net).bf16
    dev = float((seg.detach().double().cpu() - gc.out).abs().max())
    assert 1e-6 < dev < 5e-2, 'soft-max deviation %.3e from fp64' % dev
    assert float((out[1].detach().double().cpu() - gc.heat).abs().max()) < 5e-2 * float(gc.heat.abs().max())
    assert abs(loss.item() - gc.loss) < 2e-2 * abs(gc.loss)
    res = gc.check(net, seg, _eps4(), key + ' bf16s ', margin_cap=1.0)     # (free running at bf16 distance: these flags are not restated by the emulation)
    print('%s bf16 storage: conv noise %.2e, whole-gradient error %.3e, worst error / bar %.2f' % (key, res['eps_eff'], res['whole'], res['worst']))


def test_eval_and_inference_graph_bf16_storage():
    """Eval-mode forward (running statistics) and its hipGraph replay in the bf16 storage mode."""
    seed, cfg = PAPER_CFGS['paper_sc_l14']
    torch.manual_seed(seed)
    net = dfl_amd.UNet(**cfg).to(DEV)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    net.eval()
    x = torch.randn(1, 1, 192, 192, device=DEV)
    lib = nat.lib()
    with torch.no_grad():
        s4, h4 = net(x)
        s4b, h4b = net(x)
        assert torch.equal(s4, s4b) and torch.equal(h4, h4b)
        net.use_graphs = False
        s4p, h4p = net(x)
        net.use_graphs = True
        assert torch.equal(s4, s4p) and torch.equal(h4, h4p)
        nat.check(lib.dfl_set_math_mode(0), 'mode')
        s0, h0 = net(x)
        nat.check(lib.dfl_set_math_mode(4), 'mode')
    assert float((s0 - s4).abs().max()) < 5e-2 and float((h0 - h4).abs().max()) < 5e-2 * float(h0.abs().max())
    assert float((s4.sum(1) - 1).abs().max()) < 1e-5


def test_small_channel_counts_are_refused():
    net = dfl_amd.UNet(n_classes=3, depth=2, wf=3, padding=True, batch_norm=True).to(DEV)
    with pytest.raises(RuntimeError, match='multiples of 16'):
        net(torch.zeros(1, 1, 16, 16, device=DEV))


def _jobs_dev(arr):
    return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(DEV)


@pytest.mark.parametrize('C_,rows', [(32, 1536), (256, 36), (1024, 8)])
def test_live_batchnorm_batched_finalize_equals_the_per_layer_kernels(C_, rows):
    """dfl_bn_finalize_live / dfl_bn_bwd_finalize_live (one launch for a batch of layers, statistics as [DFL_BN_R][2][C] fp64 totals)
    against dfl_bn_finalize / dfl_bn_bwd_finalize on the partial rows the totals were added up from: bit for bit (the fp64 sum of
    fp32 addends of similar size is exact, so its order does not matter)."""
    lib = nat.lib()
    g = torch.Generator().manual_seed(C_ + rows)
    R_ = 8
    count = rows * 384
    part = torch.rand(rows, 2, C_, generator=g) * 300.0 + 50.0                 # sums of 384 non-negative values, and of their squares
    part[:, 1] *= 2.0
    gamma, beta = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g)
    rm, rv = torch.randn(C_, generator=g), torch.rand(C_, generator=g) + 0.5
    tot = torch.zeros(R_, 2, C_, dtype=torch.float64)
    for r_ in range(rows):
        tot[r_ % R_] += part[r_].double()
    dev = lambda t: t.clone().to(DEV)
    outs = {}
    for which in ('old', 'live'):
        d = dict(gamma=dev(gamma), beta=dev(beta), rm=dev(rm), rv=dev(rv), nbt=torch.tensor(3, dtype=torch.int64, device=DEV),
                 scale=torch.empty(C_, device=DEV), shift=torch.empty(C_, device=DEV), mean=torch.empty(C_, device=DEV),
                 invstd=torch.empty(C_, device=DEV))
        if which == 'old':
            pd = dev(part)
            nat.call('dfl_bn_finalize', nat.BnFinalizeArgs(
                partials=pd.data_ptr(), gamma=d['gamma'].data_ptr(), beta=d['beta'].data_ptr(), running_mean=d['rm'].data_ptr(),
                running_var=d['rv'].data_ptr(), num_batches_tracked=d['nbt'].data_ptr(), scale=d['scale'].data_ptr(), shift=d['shift'].data_ptr(),
                save_mean=d['mean'].data_ptr(), save_invstd=d['invstd'].data_ptr(), count=count, nblocks=rows, C=C_, eps=1e-5, momentum=0.1), stream())
        else:
            td = dev(tot)
            job = (nat.BnLiveJob * 1)()
            job[0].totals, job[0].gamma, job[0].beta = td.data_ptr(), d['gamma'].data_ptr(), d['beta'].data_ptr()
            job[0].running_mean, job[0].running_var, job[0].num_batches_tracked = d['rm'].data_ptr(), d['rv'].data_ptr(), d['nbt'].data_ptr()
            job[0].scale, job[0].shift, job[0].save_mean, job[0].save_invstd = (d[k].data_ptr() for k in ('scale', 'shift', 'mean', 'invstd'))
            job[0].count, job[0].C, job[0].eps, job[0].momentum = count, C_, 1e-5, 0.1
            jd = _jobs_dev(job)
            nat.check(lib.dfl_bn_finalize_live(jd.data_ptr(), 1, C_, stream()), 'dfl_bn_finalize_live')
        torch.cuda.synchronize()
        outs[which] = {k: d[k].cpu() for k in ('scale', 'shift', 'mean', 'invstd', 'rm', 'rv', 'nbt')}
    for k in outs['old']:
        assert torch.equal(outs['old'][k], outs['live'][k]), k
    # backward: (sum dy, sum dy*r) partials of mixed sign
    part2 = torch.randn(rows, 2, C_, generator=g) * 3.0
    tot2 = torch.zeros(R_, 2, C_, dtype=torch.float64)
    for r_ in range(rows):
        tot2[r_ % R_] += part2[r_].double()
    mean, invstd = dev(outs['old']['mean']), dev(outs['old']['invstd'])
    gd = dev(gamma)
    dgo, dbo, coef = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV), torch.empty(3, C_, device=DEV)
    p2 = dev(part2)
    nat.call('dfl_bn_bwd_finalize', nat.BnBwdFinalizeArgs(partials=p2.data_ptr(), gamma=gd.data_ptr(), save_mean=mean.data_ptr(),
                                                          save_invstd=invstd.data_ptr(), dgamma=dgo.data_ptr(), dbeta=dbo.data_ptr(),
                                                          coef=coef.data_ptr(), count=count, nblocks=rows, C=C_), stream())
    dgl, dbl, sm = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV), torch.empty(C_, device=DEV)
    t2 = dev(tot2)
    job = (nat.BnBwdLiveJob * 1)()
    job[0].totals, job[0].save_mean, job[0].save_invstd = t2.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    job[0].dgamma, job[0].dbeta, job[0].sum_out, job[0].C = dgl.data_ptr(), dbl.data_ptr(), sm.data_ptr(), C_
    jd = _jobs_dev(job)
    nat.check(lib.dfl_bn_bwd_finalize_live(jd.data_ptr(), 1, C_, stream()), 'dfl_bn_bwd_finalize_live')
    torch.cuda.synchronize()
    # (signed partials: the fp64 sums agree to the last bit unless a partial is tiny against the total -- allow one fp32 ulp)
    np.testing.assert_allclose(dgl.cpu().numpy(), dgo.cpu().numpy(), rtol=2e-7, atol=1e-6)
    np.testing.assert_allclose(dbl.cpu().numpy(), dbo.cpu().numpy(), rtol=2e-7, atol=1e-6)
    assert torch.equal(dbl, sm)


@pytest.mark.parametrize('key', ['paper__paper_sc_l14__b2', 'ragged__37x41__mp1'])
def test_live_batchnorm_statistics_change_nothing(key):
    """A training step with the BatchNorm statistics completed by their consumers (dfl_conv_args.stat_totals / in_tot / add_tot,
    dfl_wgrad_args.coef_tot: the default) against the same step with a dfl_bn_finalize / dfl_bn_bwd_finalize launch per layer
    (UNetPlan.LIVE_BN = False): outputs, loss, running statistics and EVERY gradient are BIT-IDENTICAL -- the producers add the
    same fp32 workgroup sums, and their fp64 total is exact whatever the order.  (The head's backward kernel is the one producer
    that groups its fp32 sums differently from the pass it replaces, dfl_colstats: with it on -- the default -- the forward pass
    is still bit-identical and the gradients are a second valid bf16 rounding of the same step, compared loosely here and step
    by step in tests/test_gpu_bf16_stepwise.py.)"""
    from dfl_amd import plan as P_
    pr = PR.REGISTRY[key]()
    res = {}
    for mode in ('live', 'live_nohead', 'off'):
        prev = (P_.UNetPlan.LIVE_BN, P_.UNetPlan.LIVE_HEAD)
        P_.UNetPlan.LIVE_BN, P_.UNetPlan.LIVE_HEAD = mode != 'off', mode == 'live'
        try:
            net = hip_net(pr)
            out, seg, loss = hip_step(pr, net)
            plan = NF.train_plan(net)
            nfin = sum(1 for st in plan.fwd.structs + plan.bwd.structs if isinstance(st, (nat.BnFinalizeArgs, nat.BnBwdFinalizeArgs)))
        finally:
            P_.UNetPlan.LIVE_BN, P_.UNetPlan.LIVE_HEAD = prev
        res[mode] = dict(seg=seg.detach().clone(), loss=loss.item(), nfin=nfin,
                         grads={k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                         bufs={k: v.clone() for k, v in net.named_buffers()})
    a, a2, b = res['live'], res['live_nohead'], res['off']
    print('%s: %d -> %d finalize launches per step (%d without the head kernel\'s sums)' % (key, b['nfin'], a['nfin'], a2['nfin']))
    assert a['nfin'] <= 2 or a['nfin'] * 3 <= b['nfin'], (a['nfin'], b['nfin'])     # (max-pool boundaries keep a statistics pass)
    # same workgroup sums, exact totals: bit-identical
    assert torch.equal(a2['seg'], b['seg']) and a2['loss'] == b['loss']
    for k, v in b['bufs'].items():
        assert torch.equal(a2['bufs'][k], v), k
    for k, v in b['grads'].items():
        assert torch.equal(a2['grads'][k], v), k
    # with the head kernel's own sums: identical forward pass, gradients at bf16 distance (whole gradient, relative L2)
    assert torch.equal(a['seg'], b['seg']) and a['loss'] == b['loss']
    num = sum(float((a['grads'][k] - v).double().pow(2).sum()) for k, v in b['grads'].items())
    den = sum(float(v.double().pow(2).sum()) for v in b['grads'].values())
    assert (num / den) ** 0.5 <= 3e-2, (num / den) ** 0.5


@pytest.mark.parametrize('key', ['paper__paper_sc_l14__b2', 'ragged__37x41__mp1'])
def test_update_inside_the_weight_relayout_changes_nothing(key):
    """optimizer.step() of dfl_amd.SGD in the bf16 storage mode (train.py:333-334,424): the update runs INSIDE the tiled weight
    re-layout (dfl_sgd_pack_tiled: one pass over the weights, no dfl_sgd_step launch) -- parameters, momentum buffers and the
    next forward passes (which read the bf16 layouts written there) are BIT-IDENTICAL to update launches followed by the
    re-layout (SGD.FUSE_PACK = False), and within fp32 rounding of torch.optim.SGD."""
    pr = PR.REGISTRY[key]()
    nets, opts, outs = [], [], []
    for fuse in (True, False, None):
        net = hip_net(pr)
        if fuse is None:
            opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
        else:
            opt = dfl_amd.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
            opt.FUSE_PACK = fuse
        calls = {'dfl_sgd_step': 0, 'dfl_sgd_pack_tiled': 0}
        if fuse is not None:
            real = opt._lib

            class Spy:
                def __getattr__(self, k, real=real, calls=calls):
                    f = getattr(real, k)
                    if k in calls:
                        def counted(*a):
                            calls[k] += 1
                            return f(*a)
                        return counted
                    return f
            opt._lib = Spy()
        seq = []
        for step in range(3):
            opt.zero_grad()
            out, seg, loss = hip_step(pr, net)
            seq.append((seg.detach().clone(), loss.item()))
            opt.step()
            if step == 0:
                first = [p.detach().clone() for p in net.parameters()]
                opt.param_groups[0]['lr'] = 0.02          # schedulers write the group between steps
        net.eval()
        with torch.no_grad():
            o = net(pr.x.to(DEV))
        seq.append(((o[0] if isinstance(o, tuple) else o).clone(), 0.0))
        torch.cuda.synchronize()
        nets.append(net); opts.append(opt); outs.append((seq, calls, first))
    (sa, ca, fa), (sb, cb, _), (sc, _, fc) = outs
    plan = NF.train_plan(nets[0])
    assert plan._tiled_host is not None, 'no tiled layouts in this problem: nothing was tested'
    assert ca == {'dfl_sgd_step': 0, 'dfl_sgd_pack_tiled': 3}, ca
    assert cb['dfl_sgd_pack_tiled'] == 0 and cb['dfl_sgd_step'] >= 3, cb
    for (s1, l1), (s2, l2) in zip(sa, sb):
        assert torch.equal(s1, s2) and l1 == l2
    for (k, pa), pb in zip(nets[0].named_parameters(), nets[1].parameters()):
        assert torch.equal(pa, pb), k
        if pa.grad is not None:
            assert torch.equal(opts[0].state[pa]['momentum_buffer'], opts[1].state[pb]['momentum_buffer']), k
    # torch's multi-tensor kernels round a*b+c twice where the HIP update uses fma: the first update (same gradients) agrees
    # within fp32 rounding; later steps read bf16 layouts of slightly different masters and are not comparable bit by bit
    for (k, _), pa, pc in zip(nets[0].named_parameters(), fa, fc):
        d = (pa - pc).abs().max().item()
        assert d <= 2e-6 * max(1.0, pc.abs().max().item()), (k, d)


@pytest.mark.parametrize('key', ['paper__paper_sc_l14__b2', 'paper__paper_sc_l0__b4', 'ragged__37x41__mp1', 'ragged__50x70__mp0'])
def test_operand_written_by_the_data_gradient_changes_nothing(key):
    """dfl_conv_args.x_out: the data-gradient kernel of a 3x3 layer writes the BatchNorm + ReLU backward operand it forms in its
    staging path (every element once: patch interiors, K slices, column tile 0; bit-exact, test_fused_bn_relu_backward_operand)
    and the layer's weight gradient reads that tensor (d_mode 0 + bias_partial) instead of forming the same bf16 values again
    from (dy, r).  Same operands, same products; what differs is the ORDER of the fp32 sums over pixels (a plain operand
    allows larger patches): the loss and every data gradient are bit-identical, the parameter gradients agree within fp32
    summation noise -- with the switch at 'every layer', at its default (small tensors only) and off (SURVEY Appendix F:
    unet.py:211-222 backward)."""
    from dfl_amd import plan as P_
    pr = PR.REGISTRY[key]()
    res = {}
    for name, mb in (('all', 1 << 40), ('default', None), ('off', 0)):
        prev = P_.UNetPlan.DPRE_OUT_BYTES
        if mb is not None:
            P_.UNetPlan.DPRE_OUT_BYTES = mb
        try:
            net = hip_net(pr)
            out, seg, loss = hip_step(pr, net)
            plan = NF.train_plan(net)
            nout = sum(1 for st in plan.bwd.structs if isinstance(st, nat.ConvArgs) and st.x_out)
        finally:
            P_.UNetPlan.DPRE_OUT_BYTES = prev
        res[name] = dict(loss=loss.item(), nout=nout, grads={k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    print('%s: layers whose data gradient writes the operand: all %d, default %d, off %d' % (key, res['all']['nout'], res['default']['nout'], res['off']['nout']))
    assert res['off']['nout'] == 0 and res['all']['nout'] >= 2 and res['all']['nout'] >= res['default']['nout']
    for name in ('all', 'default'):
        assert res[name]['loss'] == res['off']['loss']
        for k, v in res['off']['grads'].items():
            d = float((res[name]['grads'][k] - v).double().norm()) / max(float(v.double().norm()), 1e-30)
            assert d <= 2e-6, (name, k, d)
