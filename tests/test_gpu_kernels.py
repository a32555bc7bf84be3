"""Per-kernel parity: every C-ABI entry point of libdfl_hip.so against a plain PyTorch fp32/fp64 CPU computation of
the same op (and, where one exists, against the oracle).  Runs on the GPU box: pytest -m gpu."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dfl_amd
from dfl_amd import _native as nat
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

TOLK = [1.0]


@pytest.fixture(autouse=True)
def _both_math_modes(math_mode):
    """Every kernel test runs with fp32 products (tolerances as written) and with split-bf16 products (x8: 2^-16 per
    product instead of 2^-24; the non-GEMM kernels are unaffected by the mode)."""
    TOLK[0] = 1.0 if math_mode == 'fp32' else 8.0
    yield
    TOLK[0] = 1.0


def aclose(actual, desired, rtol=1e-7, atol=0.0, err_msg=''):
    np.testing.assert_allclose(actual, desired, rtol=rtol * TOLK[0], atol=atol * TOLK[0], err_msg=err_msg)

DEV = 'cuda'


def stream():
    return torch.cuda.current_stream().cuda_stream


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def pack(w, kind, flip=0, split=0):
    """Run one dfl_pack_weights job on the device: parameter [A][B][KH][KW] -> quad-packed GEMM operand."""
    lib = nat.lib()
    src = w.to(DEV).contiguous()
    A, B, KH, KW = w.shape
    Cc = KH * KW
    K = {1: Cc * B, 2: Cc * A, 3: A}[kind]
    N = {1: A, 2: B, 3: Cc * B}[kind]
    dst = torch.full(((K + 3) // 4 * N * 4,), float('nan'), device=DEV)
    job = nat.PackJob(src=src.data_ptr(), dst=dst.data_ptr(), A=A, B=B, C=Cc, kind=kind, flip=flip, split=split)
    jobs = torch.from_numpy(np.frombuffer(bytes(job), dtype=np.uint8).copy()).to(DEV)
    nat.check(lib.dfl_pack_weights(jobs.data_ptr(), 1, A * B * Cc, stream()))
    torch.cuda.synchronize()
    return dst


def conv_call(x, wp, Ntot, KH, KW, stride, pad, Hout, Wout, bias=None, in_aff=None, relu=0, add=None, add_aff=None,
              y_init=None, accumulate=0, scatter=0, stats=False, stat_other=None, ldy=None, ldx_pad=0,
              force_splits=None, w_split=0, x_split=0, latency=False, out_aff=None):
    """x: NCHW cpu tensor -> runs dfl_conv2d -> returns y as NHWC cpu tensor [N,Hout,Wout,Cout] (+ stats).
    latency: the latency form (dfl_conv_args.latency_form, csrc/convs_f32.hip; tests/test_gpu_latency_form_f32.py)."""
    lib = nat.lib()
    N, Cin, Hin, Win = x.shape
    xh = nhwc(x)
    if ldx_pad:
        xh = F.pad(xh, (0, ldx_pad))
    xd = xh.to(DEV)
    if x_split:                 # the 16-byte slot of 4 values holds their 4 hi bf16, then their 4 lo bf16
        hi = xd.bfloat16()
        lo = (xd - hi.float()).bfloat16()
        q = torch.cat([hi.reshape(-1, 4), lo.reshape(-1, 4)], dim=1).contiguous()
        xd = q.view(torch.float32).reshape(xd.shape).contiguous()
    Cout = Ntot // 4 if scatter else Ntot
    ldy = Cout if ldy is None else ldy
    if y_init is not None:
        yd = nhwc(y_init).to(DEV)
        if ldy != Cout:
            yd = F.pad(yd, (0, ldy - Cout)).contiguous()
    else:
        yd = torch.full((N, Hout, Wout, ldy), float('nan'), device=DEV)
    a = nat.ConvArgs()
    keep = [xd, yd, wp]
    a.x, a.w, a.y = xd.data_ptr(), wp.data_ptr(), yd.data_ptr()
    if bias is not None:
        b = bias.to(DEV)
        keep.append(b)
        a.bias = b.data_ptr()
    if in_aff is not None:
        sc, sh = in_aff[0].to(DEV), in_aff[1].to(DEV)
        keep += [sc, sh]
        a.in_scale, a.in_shift = sc.data_ptr(), sh.data_ptr()
    if add is not None:
        ad = nhwc(add).to(DEV)
        keep.append(ad)
        a.add, a.ldadd = ad.data_ptr(), ad.shape[-1]
        if add_aff is not None:
            asc, ash = add_aff[0].to(DEV), add_aff[1].to(DEV)
            keep += [asc, ash]
            a.add_scale, a.add_shift = asc.data_ptr(), ash.data_ptr()
    a.N, a.Hin, a.Win, a.Cin, a.ldx = N, Hin, Win, Cin, Cin + ldx_pad
    a.KH, a.KW, a.stride, a.pad = KH, KW, stride, pad
    a.Hout, a.Wout, a.Ntot, a.ldy = Hout, Wout, Ntot, ldy
    a.relu, a.accumulate, a.scatter2x2 = relu, accumulate, scatter
    a.w_split, a.x_split = w_split, x_split
    if out_aff is not None:
        osc, osh = out_aff[0].to(DEV), out_aff[1].to(DEV)
        keep += [osc, osh]
        a.out_scale, a.out_shift = osc.data_ptr(), osh.data_ptr()
    if latency:
        a.latency_form = 1
        assert lib.dfl_conv_config(C.addressof(a)) == 16 + 39, 'the latency form was asked for and is eligible here'
    sp = force_splits or nat.check(lib.dfl_conv_suggest_splits(C.addressof(a)))
    if sp > 1:
        Mrows = N * (Hin * Win if scatter else Hout * Wout)
        kpart = torch.full((sp * Mrows * Ntot,), float('nan'), device=DEV)
        keep.append(kpart)
        a.splits, a.partial = sp, kpart.data_ptr()
    part = None
    if stats:
        gm = nat.check(lib.dfl_conv_grid_m(C.addressof(a)))
        part = torch.zeros(gm, 2, Ntot, device=DEV)
        a.stat_partials = part.data_ptr()
        if stat_other is not None:
            so = nhwc(stat_other).to(DEV)
            keep.append(so)
            a.stat_other, a.ldso = so.data_ptr(), so.shape[-1]
    nat.check(lib.dfl_conv2d(C.addressof(a), stream()), 'dfl_conv2d')
    torch.cuda.synchronize()
    y = yd.cpu()[..., :Cout]
    return (y, part.cpu().double().sum(0)) if stats else y


@pytest.mark.parametrize('shape', [(64, 32, 3), (48, 20, 3), (7, 5, 3), (32, 1, 3), (128, 64, 1), (16, 16, 2), (6, 10, 2)])
def test_pack_weights_layout(shape):
    """All three index maps of dfl_pack_weights (tiled and element-wise paths) against a numpy construction."""
    A, B, KK = shape
    g = torch.Generator().manual_seed(A + B)
    w = torch.randn(A, B, KK, KK, generator=g)
    Cc = KK * KK
    wn = w.reshape(A, B, Cc).numpy()
    for kind, flip in ((1, 0), (2, 1), (2, 0), (3, 0)):
        if kind == 1:
            W = wn.transpose(2, 1, 0).reshape(Cc * B, A)                       # k = c*B + b, n = a
        elif kind == 2:
            src = wn[:, :, ::-1] if flip else wn
            W = src.transpose(2, 0, 1).reshape(Cc * A, B)                      # k = c'*A + a, n = b
        else:
            W = wn.transpose(0, 2, 1).reshape(A, Cc * B)                       # k = a, n = c*B + b
        K, N = W.shape
        Kq = (K + 3) // 4
        ref = np.zeros((Kq * 4, N), dtype=np.float32)
        ref[:K] = W
        ref = ref.reshape(Kq, 4, N).transpose(0, 2, 1).reshape(-1)
        got = pack(w, kind, flip).cpu().numpy()
        assert np.array_equal(got, ref), (shape, kind, flip)


@pytest.mark.parametrize('shape', [(64, 32, 3), (32, 64, 1), (128, 96, 2), (64, 64, 3)])
@pytest.mark.parametrize('split', [0, 1])
def test_pack_weights_tiled_quad_layouts(shape, split):
    """dfl_pack_weights_tiled writes the fp32 quad layouts (and the split hi | lo bf16 quads) too (round 5: the update inside the
    weight re-layout for the parity arithmetics): both layouts of a job, bit for bit what dfl_pack_weights writes."""
    lib = nat.lib()
    A, B, KK = shape
    w = torch.randn(A, B, KK, KK, generator=torch.Generator().manual_seed(A * B + KK))
    src = w.to(DEV).contiguous()
    Cc = KK * KK
    for (k1, f1), (k2, f2) in (((1, 0), (2, 1)), ((3, 0), (1, 0)), ((2, 0), (3, 0))):
        def size(kind):
            K = {1: Cc * B, 2: Cc * A, 3: A}[kind]
            N = {1: A, 2: B, 3: Cc * B}[kind]
            return (K + 3) // 4 * N * 4
        d1 = torch.full((size(k1),), float('nan'), device=DEV)
        d2 = torch.full((size(k2),), float('nan'), device=DEV)
        job = nat.PackJob(src=src.data_ptr(), dst=d1.data_ptr(), A=A, B=B, C=Cc, kind=k1, flip=f1, split=split, dst2=d2.data_ptr(),
                          kind2=k2, flip2=f2, first_tile=0, split2=split)
        jobs = torch.from_numpy(np.frombuffer(bytes(job), dtype=np.uint8).copy()).to(DEV)
        nat.check(lib.dfl_pack_weights_tiled(jobs.data_ptr(), 1, (A // 32) * (B // 32), stream()), 'dfl_pack_weights_tiled')
        torch.cuda.synchronize()
        r1, r2 = pack(w, k1, f1, split), pack(w, k2, f2, split)
        assert torch.equal(d1.view(torch.int32), r1.view(torch.int32)), (shape, split, k1, f1)
        assert torch.equal(d2.view(torch.int32), r2.view(torch.int32)), (shape, split, k2, f2)


CONV_CASES = [
    # N, Cin, Cout, H, W, K, stride, pad
    (2, 8, 16, 12, 12, 3, 1, 1),
    (1, 32, 32, 24, 20, 3, 1, 1),
    (2, 64, 128, 13, 9, 3, 1, 1),      # odd sizes, several tile configs
    (2, 16, 8, 10, 10, 3, 1, 0),       # valid conv
    (3, 1, 32, 16, 16, 3, 1, 1),       # first layer: Cin = 1 (scalar gather)
    (2, 3, 4, 9, 11, 3, 1, 1),         # odd Cin/Cout
    (2, 32, 64, 8, 8, 1, 1, 0),        # 1x1
    (2, 16, 16, 12, 10, 2, 2, 0),      # 2x2 stride 2
    (2, 8, 8, 7, 9, 2, 2, 0),          # 2x2 stride 2, odd input
    (1, 256, 256, 6, 6, 3, 1, 1),      # deep level
    (16, 32, 32, 48, 48, 3, 1, 1),     # enough rows for the 256x32 tile
    (4, 64, 64, 96, 96, 3, 1, 1),      # 128x64 tile
    (4, 128, 128, 64, 64, 1, 1, 0),    # 128x128 tile
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_plain(case):
    N, Cin, Cout, H, W, K, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / math.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g)
    T = K * K
    wp = pack(w, 1)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    y = conv_call(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b)
    aclose(nchw(y).numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)


def test_conv_fwd_fused_epilogue_and_prologue():
    """BN-on-load with zero padding after the affine, bias, ReLU, statistics; then the residual form."""
    g = torch.Generator().manual_seed(5)
    N, Cin, Cout, H, W = 2, 16, 32, 14, 10
    x = torch.randn(N, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 12
    b = torch.randn(Cout, generator=g)
    wp = pack(w, 1)
    xa = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref = F.relu(F.conv2d(xa, w.double(), b.double(), padding=1))
    y, st = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, in_aff=(sc, sh), relu=1, stats=True)
    aclose(nchw(y).numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    aclose(st[0].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    aclose(st[1].numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    # residual: y = conv1x1(x) + bias + (r*s2 + t2), into a wider buffer (ldy > Cout), input with ldx > Cin
    w1 = torch.randn(Cout, Cin, 1, 1, generator=g) / 4
    w1p = pack(w1, 1)
    r = torch.randn(N, Cout, H, W, generator=g)
    s2, t2 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref2 = F.conv2d(x.double(), w1.double(), b.double()) + r.double() * s2.double().view(1, -1, 1, 1) + t2.double().view(1, -1, 1, 1)
    y2 = conv_call(x, w1p, Cout, 1, 1, 1, 0, H, W, bias=b, add=r, add_aff=(s2, t2), ldy=2 * Cout, ldx_pad=8)
    aclose(nchw(y2).numpy(), ref2.numpy(), rtol=2e-5, atol=2e-5)
    # accumulate + statistics against another tensor (backward use)
    y0 = torch.randn(N, Cout, H, W, generator=g)
    other = torch.randn(N, Cout, H, W, generator=g)
    ref3 = F.conv2d(x.double(), w.double(), None, padding=1) + y0.double()
    y3, st3 = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, y_init=y0, accumulate=1, stats=True, stat_other=other)
    aclose(nchw(y3).numpy(), ref3.numpy(), rtol=2e-5, atol=2e-5)
    aclose(st3[1].numpy(), (ref3 * other.double()).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)


def test_conv_split_k():
    """Deep-level shape (few pixels, long K): split-K partial sums + finish kernel with the full epilogue."""
    g = torch.Generator().manual_seed(8)
    N, Cin, Cout, H, W = 2, 256, 128, 6, 6
    x = torch.randn(N, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 48
    b = torch.randn(Cout, generator=g)
    wp = pack(w, 1)
    xa = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref = F.relu(F.conv2d(xa, w.double(), b.double(), padding=1))
    for fs in (None, 2, 7):
        y, st = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, in_aff=(sc, sh), relu=1, stats=True, force_splits=fs)
        aclose(nchw(y).numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
        aclose(st[0].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
        aclose(st[1].numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    # residual-add + accumulate + statistics against another tensor through the finish kernel
    r = torch.randn(N, Cout, H, W, generator=g)
    s2, t2 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    y0 = torch.randn(N, Cout, H, W, generator=g)
    other = torch.randn(N, Cout, H, W, generator=g)
    ref2 = F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double() * s2.double().view(1, -1, 1, 1) \
        + t2.double().view(1, -1, 1, 1) + y0.double()
    y2, st2 = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, add=r, add_aff=(s2, t2), y_init=y0, accumulate=1, stats=True,
                        stat_other=other, force_splits=4)
    aclose(nchw(y2).numpy(), ref2.numpy(), rtol=2e-5, atol=2e-5)
    aclose(st2[1].numpy(), (ref2 * other.double()).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    # transposed conv through split-K
    wt = torch.randn(Cin, 64, 2, 2, generator=g) / 16
    bt = torch.randn(64, generator=g)
    wtp = pack(wt, 3)
    reft = F.conv_transpose2d(x.double(), wt.double(), bt.double(), stride=2)
    yt = conv_call(x, wtp, 4 * 64, 1, 1, 1, 0, 2 * H, 2 * W, bias=bt, scatter=1, force_splits=2)
    aclose(nchw(yt).numpy(), reft.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('shape', [(2, 16, 8, 5, 7), (1, 64, 32, 6, 6), (2, 8, 4, 3, 3)])
def test_conv_transpose_scatter(shape):
    N, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, generator=g) / 4
    b = torch.randn(Cout, generator=g)
    wp = pack(w, 3)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    y = conv_call(x, wp, 4 * Cout, 1, 1, 1, 0, 2 * H, 2 * W, bias=b, scatter=1, ldy=2 * Cout)
    aclose(nchw(y).numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)


def test_dgrad_forms():
    """Data gradients of conv3x3 (pad 1 / pad 0), conv2x2s2 (scatter) and convT2x2s2 vs torch autograd."""
    g = torch.Generator().manual_seed(11)
    N, Ci, Co, H, W = 2, 8, 16, 9, 7
    for pad in (1, 0):
        x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        w = torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64) / 6
        y = F.conv2d(x, w, padding=pad)
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(dy)
        wd = pack(w.float(), 2, flip=1)
        dx = conv_call(dy.float(), wd, Ci, 3, 3, 1, 2 - pad, H, W)
        aclose(nchw(dx).numpy(), x.grad.numpy(), rtol=2e-5, atol=2e-5)
    # conv2x2 stride 2 (odd input: last row/col get no gradient; accumulate keeps what was there)
    x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Ci, Ci, 2, 2, generator=g, dtype=torch.float64) / 3
    y = F.conv2d(x, w, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    wd = pack(w.float(), 3)
    base = torch.randn(N, Ci, H, W, generator=g)
    dx = conv_call(dy.float(), wd, 4 * Ci, 1, 1, 1, 0, H, W, scatter=1, y_init=base, accumulate=1)
    aclose(nchw(dx).numpy(), (x.grad + base.double()).numpy(), rtol=2e-5, atol=2e-5)
    # convT
    x = torch.randn(N, Ci, 4, 5, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Ci, Co, 2, 2, generator=g, dtype=torch.float64) / 3
    y = F.conv_transpose2d(x, w, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    wd = pack(w.float(), 1)
    dx = conv_call(dy.float(), wd, Ci, 2, 2, 2, 0, 4, 5)
    aclose(nchw(dx).numpy(), x.grad.numpy(), rtol=2e-5, atol=2e-5)


def wgrad_call(gx, d, KH, KW, stride, pad, Hout, Wout, in_aff=None, force_splits=None):
    lib = nat.lib()
    N, Cg, Hin, Win = gx.shape
    Cm = d.shape[1]
    gd, dd = nhwc(gx).to(DEV), nhwc(d).to(DEV)
    T = KH * KW
    dw = torch.full((Cm, Cg, KH, KW), float('nan'), device=DEV)
    a = nat.WgradArgs(g=gd.data_ptr(), d=dd.data_ptr(), dw=dw.data_ptr(), N=N, Hin=Hin, Win=Win, Cg=Cg, ldg=Cg, KH=KH, KW=KW,
                      stride=stride, pad=pad, Hout=Hout, Wout=Wout, Cm=Cm, ldd=Cm, splits=1)
    keep = []
    if in_aff is not None:
        sc, sh = in_aff[0].to(DEV), in_aff[1].to(DEV)
        keep += [sc, sh]
        a.in_scale, a.in_shift = sc.data_ptr(), sh.data_ptr()
    s = force_splits or nat.check(lib.dfl_wgrad_suggest_splits(C.addressof(a)))
    a.splits = s
    if s > 1:
        part = torch.empty(s * Cm * Cg * T, device=DEV)
        a.partial = part.data_ptr()
    nat.check(lib.dfl_conv2d_wgrad(C.addressof(a), stream()), 'dfl_conv2d_wgrad')
    if s > 1:
        nat.check(lib.dfl_sum_partials(part.data_ptr(), dw.data_ptr(), Cm * Cg * T, s, T, stream()))
    torch.cuda.synchronize()
    return dw.cpu(), s


WG_CASES = [
    # N, Cin, Cout, H, W, K, stride, pad
    (2, 8, 16, 12, 12, 3, 1, 1),
    (2, 32, 32, 20, 24, 3, 1, 1),       # 9-tap kernel
    (1, 128, 256, 6, 6, 3, 1, 1),       # tap-in-grid 64x64
    (1, 256, 512, 6, 6, 3, 1, 1),       # 128x128
    (3, 1, 32, 16, 16, 3, 1, 1),        # first layer
    (2, 16, 8, 10, 10, 3, 1, 0),
    (2, 32, 64, 8, 8, 1, 1, 0),
    (2, 128, 128, 9, 9, 1, 1, 0),
    (2, 16, 16, 12, 10, 2, 2, 0),
    (2, 128, 128, 7, 9, 2, 2, 0),
    (2, 7, 39, 9, 9, 1, 1, 0),          # head-like odd channel counts (scalar paths)
    (2, 32, 32, 5, 48, 3, 1, 1),        # kernel-row variant with the shared 18-pixel halo (W % 16 == 0), image borders
    (1, 64, 32, 3, 32, 3, 1, 1),
    (3, 32, 64, 2, 16, 3, 1, 1),        # every chunk is a whole image row
    (16, 64, 32, 6, 6, 3, 1, 1),        # chunks that span 3-4 image rows and cross image boundaries (shared halo per segment)
    (4, 128, 128, 12, 12, 3, 1, 1),     # wide layers, narrow images: one tap per workgroup
    (2, 128, 256, 24, 24, 3, 1, 1),
    (1, 128, 128, 3, 48, 3, 1, 1),      # wide layers, W % 16 == 0: 64 x 64 tile with three taps per workgroup
    (3, 32, 32, 8, 10, 3, 1, 1),
    (1, 32, 32, 4, 36, 3, 1, 1),
]


@pytest.mark.parametrize('case', WG_CASES)
def test_wgrad(case):
    N, Cin, Cout, H, W, K, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.zeros(Cout, Cin, K, K, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    for fs in (None, 1, 3):
        dw, s = wgrad_call(x.float(), dy.float(), K, K, stride, pad, y.shape[2], y.shape[3], force_splits=fs)
        scale = w.grad.abs().max().item()
        aclose(dw.numpy(), w.grad.numpy(), rtol=1e-4, atol=2e-5 * max(scale, 1.0))


def test_wgrad_affine_and_convT():
    g = torch.Generator().manual_seed(2)
    N, Ci, Co, H, W = 2, 16, 32, 10, 8
    r = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64)
    sc, sh = torch.rand(Ci, generator=g, dtype=torch.float64) + 0.5, torch.randn(Ci, generator=g, dtype=torch.float64)
    z = r * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(z, w, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dw, _ = wgrad_call(r.float(), dy.float(), 3, 3, 1, 1, H, W, in_aff=(sc.float(), sh.float()))
    aclose(dw.numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-4)
    # the same through the shared-halo kernel-row variant (W % 16 == 0): zero padding must stay zero behind the affine
    N2, H2, W2 = 2, 6, 32
    r2 = torch.randn(N2, 32, H2, W2, generator=g, dtype=torch.float64)
    sc2, sh2 = torch.rand(32, generator=g, dtype=torch.float64) + 0.5, torch.randn(32, generator=g, dtype=torch.float64)
    w2 = torch.zeros(32, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    y2 = F.conv2d(r2 * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1), w2, padding=1)
    dy2 = torch.randn(y2.shape, generator=g, dtype=torch.float64)
    y2.backward(dy2)
    for fs in (None, 3):
        dw2, _ = wgrad_call(r2.float(), dy2.float(), 3, 3, 1, 1, H2, W2, in_aff=(sc2.float(), sh2.float()), force_splits=fs)
        aclose(dw2.numpy(), w2.grad.numpy(), rtol=1e-4, atol=2e-4)
    # ... and with chunks that span several image rows and two images (6-pixel rows)
    N3, H3, W3 = 8, 6, 6
    r3 = torch.randn(N3, 64, H3, W3, generator=g, dtype=torch.float64)
    sc3, sh3 = torch.rand(64, generator=g, dtype=torch.float64) + 0.5, torch.randn(64, generator=g, dtype=torch.float64)
    w3 = torch.zeros(32, 64, 3, 3, dtype=torch.float64, requires_grad=True)
    y3 = F.conv2d(r3 * sc3.view(1, -1, 1, 1) + sh3.view(1, -1, 1, 1), w3, padding=1)
    dy3 = torch.randn(y3.shape, generator=g, dtype=torch.float64)
    y3.backward(dy3)
    for fs in (None, 2):
        dw3, _ = wgrad_call(r3.float(), dy3.float(), 3, 3, 1, 1, H3, W3, in_aff=(sc3.float(), sh3.float()), force_splits=fs)
        aclose(dw3.numpy(), w3.grad.numpy(), rtol=1e-4, atol=2e-4)
    # ConvTranspose2d weight gradient: gathered = dy (stride 2), dense = x  -> [Cin][Cout][2][2]
    x = torch.randn(N, Ci, 5, 6, generator=g, dtype=torch.float64)
    wt = torch.zeros(Ci, Co, 2, 2, dtype=torch.float64, requires_grad=True)
    yt = F.conv_transpose2d(x, wt, stride=2)
    dyt = torch.randn(yt.shape, generator=g, dtype=torch.float64)
    yt.backward(dyt)
    dwt, _ = wgrad_call(dyt.float(), x.float(), 2, 2, 2, 0, 5, 6)
    aclose(dwt.numpy(), wt.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('C_,M', [(32, 5000), (8, 777), (1024, 576), (4, 33)])
def test_batchnorm_forward_backward(C_, M):
    """colstats + bn_finalize + bn_bwd_finalize + bn_relu_bwd_apply against torch's batch_norm autograd (fp64)."""
    lib = nat.lib()
    g = torch.Generator().manual_seed(C_ + M)
    r = F.relu(torch.randn(M, C_, generator=g)) * 2.0
    gamma, beta = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g)
    rm, rv = torch.randn(C_, generator=g), torch.rand(C_, generator=g) + 0.5
    nbt = torch.tensor(7, dtype=torch.int64)
    rd = r.to(DEV)
    nb = lib.dfl_rowblock_count(M, C_)
    part = torch.empty(nb, 2, C_, device=DEV)
    nat.call('dfl_colstats', nat.ColstatsArgs(a=rd.data_ptr(), b=None, partials=part.data_ptr(), M=M, C=C_,
                                                            lda=C_, ldb=0, nblocks=nb), stream())
    dv = {k: v.to(DEV) for k, v in dict(gamma=gamma, beta=beta, rm=rm.clone(), rv=rv.clone(), nbt=nbt.clone()).items()}
    scale, shift, mean, invstd = (torch.empty(C_, device=DEV) for _ in range(4))
    nat.call('dfl_bn_finalize', nat.BnFinalizeArgs(
        partials=part.data_ptr(), gamma=dv['gamma'].data_ptr(), beta=dv['beta'].data_ptr(), running_mean=dv['rm'].data_ptr(),
        running_var=dv['rv'].data_ptr(), num_batches_tracked=dv['nbt'].data_ptr(), scale=scale.data_ptr(),
        shift=shift.data_ptr(), save_mean=mean.data_ptr(), save_invstd=invstd.data_ptr(), count=M, nblocks=nb, C=C_,
        eps=1e-5, momentum=0.1), stream())
    # torch reference (double)
    r64 = r.double().t().reshape(1, C_, M, 1).requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    z = F.batch_norm(r64, rm64, rv64, gamma.double(), beta.double(), training=True, momentum=0.1, eps=1e-5)
    z_gpu = (rd * scale + shift).cpu()
    aclose(z_gpu.numpy(), z[0, :, :, 0].t().detach().numpy(), rtol=1e-4, atol=1e-4)
    aclose(dv['rm'].cpu().numpy(), rm64.numpy(), rtol=1e-5, atol=1e-6)
    aclose(dv['rv'].cpu().numpy(), rv64.numpy(), rtol=1e-5, atol=1e-6)
    assert int(dv['nbt'].cpu()) == 8
    # eval prepare
    es, et = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV)
    nat.check(lib.dfl_bn_eval_prepare(dv['gamma'].data_ptr(), dv['beta'].data_ptr(), dv['rm'].data_ptr(), dv['rv'].data_ptr(),
                                      es.data_ptr(), et.data_ptr(), C_, 1e-5, stream()))
    ze = F.batch_norm(r.double().t().reshape(1, C_, M, 1), rm64, rv64, gamma.double(), beta.double(), training=False, eps=1e-5)
    aclose((rd * es + et).cpu().numpy(), ze[0, :, :, 0].t().numpy(), rtol=1e-4, atol=1e-4)
    # backward through BN then ReLU (the ReLU that produced r): d pre-activation
    dz = torch.randn(M, C_, generator=g)
    pre = r64  # treat r as relu(pre) with pre = r where r > 0; mask = r > 0
    z.backward(dz.double().t().reshape(1, C_, M, 1))
    dr_ref = r64.grad[0, :, :, 0].t() * (r.double() > 0)
    dzd = dz.to(DEV)
    part2 = torch.empty(nb, 2, C_, device=DEV)
    nat.call('dfl_colstats', nat.ColstatsArgs(a=dzd.data_ptr(), b=rd.data_ptr(), partials=part2.data_ptr(), M=M,
                                                            C=C_, lda=C_, ldb=C_, nblocks=nb), stream())
    dgamma, dbeta, coef = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV), torch.empty(3, C_, device=DEV)
    nat.call('dfl_bn_bwd_finalize', nat.BnBwdFinalizeArgs(
        partials=part2.data_ptr(), gamma=dv['gamma'].data_ptr(), save_mean=mean.data_ptr(), save_invstd=invstd.data_ptr(),
        dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), coef=coef.data_ptr(), count=M, nblocks=nb, C=C_), stream())
    dpre = torch.empty(M, C_, device=DEV)
    bpart = torch.empty(nb, C_, device=DEV)
    nat.call('dfl_bn_relu_bwd_apply', nat.BnReluBwdArgs(
        dy=dzd.data_ptr(), r=rd.data_ptr(), coef=coef.data_ptr(), dpre=dpre.data_ptr(), partials=bpart.data_ptr(), M=M, C=C_,
        lddy=C_, ldr=C_, ldo=C_, nblocks=nb), stream())
    bsum = torch.empty(C_, device=DEV)
    nat.check(lib.dfl_reduce_partials(bpart.data_ptr(), bsum.data_ptr(), nb, C_, C_, stream()))
    torch.cuda.synchronize()
    # gamma/beta grads from torch need parameters with grad: recompute analytically in fp64
    xhat = (r.double() - r.double().mean(0)) / torch.sqrt(r.double().var(0, unbiased=False) + 1e-5)
    aclose(dgamma.cpu().numpy(), (dz.double() * xhat).sum(0).numpy(), rtol=1e-3, atol=1e-3)
    aclose(dbeta.cpu().numpy(), dz.double().sum(0).numpy(), rtol=1e-3, atol=1e-3)
    aclose(dpre.cpu().numpy(), dr_ref.numpy(), rtol=1e-3, atol=2e-5)
    aclose(bsum.cpu().numpy(), dr_ref.sum(0).numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('shape', [(2, 8, 10, 12), (1, 32, 7, 9), (2, 3, 6, 6)])
def test_maxpool(shape):
    lib = nat.lib()
    N, C_, H, W = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 4, shape, generator=g).float()        # many ties -> exercises the first-max rule
    x.requires_grad_(True)
    y = F.max_pool2d(x, 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = nhwc(x.detach()).to(DEV)
    yd = torch.empty(N, H // 2, W // 2, C_, device=DEV)
    nat.call('dfl_maxpool2x2_fwd', nat.PoolArgs(x=xd.data_ptr(), y=yd.data_ptr(), N=N, H=H, W=W, C=C_, ldx=C_,
                                                              ldy=C_), stream())
    base = torch.randn(N, H, W, C_, generator=g)
    dxd = base.to(DEV)
    dyd = nhwc(dy).to(DEV)
    nat.call('dfl_maxpool2x2_bwd', nat.PoolArgs(x=xd.data_ptr(), y=dyd.data_ptr(), dx=dxd.data_ptr(), N=N, H=H,
                                                              W=W, C=C_, ldx=C_, ldy=C_, lddx=C_), stream())
    torch.cuda.synchronize()
    assert torch.equal(nchw(yd.cpu()), y.detach())
    aclose(nchw(dxd.cpu() - base).numpy(), x.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_affine_copy_window():
    lib = nat.lib()
    g = torch.Generator().manual_seed(4)
    N, C_, H, W = 2, 8, 9, 11
    x = torch.randn(N, H, W, C_, generator=g)
    sc, sh = torch.rand(C_, generator=g), torch.randn(C_, generator=g)
    xd, scd, shd = x.to(DEV), sc.to(DEV), sh.to(DEV)
    yd = torch.zeros(N, 5, 6, 2 * C_, device=DEV)
    a = nat.AffineCopyArgs(x=xd.data_ptr(), y=yd.data_ptr() + 4 * C_, scale=scd.data_ptr(), shift=shd.data_ptr(), N=N, H=5, W=6,
                           C=C_, ldx=C_, xH=H, xW=W, xoy=2, xox=3, ldy=2 * C_, yH=5, yW=6)
    nat.check(lib.dfl_affine_copy(C.addressof(a), stream()))
    torch.cuda.synchronize()
    ref = x[:, 2:7, 3:9, :] * sc + sh
    aclose(yd.cpu()[..., C_:].numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)
    assert float(yd.cpu()[..., :C_].abs().max()) == 0.0


@pytest.mark.parametrize('NC,L,two,softmax,F_', [(7, 14, True, True, 32), (3, 14, False, True, 8), (5, 0, True, False, 8),
                                                 (7, 14, True, True, 4),
                                                 # beyond 8 classes / 16 landmarks / 24 mid channels: the large-capacity build of the kernels
                                                 (12, 20, True, True, 32), (16, 32, True, True, 16), (10, 0, True, True, 8), (3, 30, False, False, 8)])
def test_heads_forward_backward(NC, L, two, softmax, F_):
    lib = nat.lib()
    g = torch.Generator().manual_seed(NC + L)
    N, H, W = 2, 11, 13
    x = torch.randn(N, F_, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    wseg = (torch.randn(NC, F_, 1, 1, generator=g, dtype=torch.float64) / 3).requires_grad_(True)
    NM = (NC + L if two else L) if L > 0 else 0
    params = [wseg]
    logits = F.conv2d(x, wseg)
    seg = torch.softmax(logits, 1) if softmax else logits
    outs = [seg]
    if L > 0:
        w1 = (torch.randn(NM, F_ + NC, 1, 1, generator=g, dtype=torch.float64) / 3).requires_grad_(True)
        params.append(w1)
        mid = F.conv2d(torch.cat((x, logits), 1), w1)
        heat = mid
        if two:
            w2 = (torch.randn(L, NM, 1, 1, generator=g, dtype=torch.float64) / 3).requires_grad_(True)
            params.append(w2)
            heat = F.conv2d(mid, w2)
        outs.append(heat)
    gouts = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in outs]
    torch.autograd.backward(outs, gouts)
    dv = lambda t: t.detach().float().contiguous().to(DEV)
    xd = nhwc(x.detach().float()).to(DEV)
    wsd = dv(wseg)
    w1d = dv(w1) if L > 0 else None
    w2d = dv(w2) if (L > 0 and two) else None
    segd = torch.empty(N, NC, H, W, device=DEV)
    heatd = torch.empty(N, L, H, W, device=DEV) if L > 0 else None
    nat.call('dfl_head_fwd', nat.HeadFwdArgs(
        x=xd.data_ptr(), w_seg=wsd.data_ptr(), w_l1=nat.ptr(w1d), w_l2=nat.ptr(w2d), seg=segd.data_ptr(), heat=nat.ptr(heatd),
        N=N, H=H, W=W, F=F_, ldx=F_, NC=NC, NM=NM, L=L, softmax=int(softmax)), stream())
    torch.cuda.synchronize()
    aclose(segd.cpu().numpy(), seg.detach().numpy(), rtol=1e-5, atol=1e-6)
    if L > 0:
        aclose(heatd.cpu().numpy(), heat.detach().numpy(), rtol=1e-5, atol=1e-5)
    # backward
    sld = lib.dfl_head_scratch_ld_for(F_, NC, NM, L)
    scratch = torch.full((N * H * W, sld), float('nan'), device=DEV)
    dxd = torch.empty(N, H, W, F_, device=DEV)
    dsegd = dv(gouts[0])
    dheatd = dv(gouts[1]) if L > 0 else None
    nat.call('dfl_head_bwd', nat.HeadBwdArgs(
        x=xd.data_ptr(), seg=segd.data_ptr(), dseg=dsegd.data_ptr(), dheat=nat.ptr(dheatd), w_seg=wsd.data_ptr(),
        w_l1=nat.ptr(w1d), w_l2=nat.ptr(w2d), dx=dxd.data_ptr(), scratch=scratch.data_ptr(), N=N, H=H, W=W, F=F_, ldx=F_,
        lddx=F_, NC=NC, NM=NM, L=L, softmax=int(softmax), scratch_ld=sld), stream())
    torch.cuda.synchronize()
    aclose(nchw(dxd.cpu()).numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-5)
    sc = scratch.cpu().double()
    assert torch.isfinite(sc).all()
    off = [lib.dfl_head_scratch_off_for(F_, NC, NM, L, k) for k in range(5)]
    dwseg = sc[:, off[1]:off[1] + NC].t() @ sc[:, off[0]:off[0] + F_]
    aclose(dwseg.numpy(), wseg.grad[:, :, 0, 0].numpy(), rtol=1e-4, atol=1e-4)
    if L > 0:
        dw1 = sc[:, off[2]:off[2] + NM].t() @ sc[:, off[0]:off[0] + F_ + NC]
        aclose(dw1.numpy(), w1.grad[:, :, 0, 0].numpy(), rtol=1e-4, atol=1e-4)
        if two:
            dw2 = sc[:, off[4]:off[4] + L].t() @ sc[:, off[3]:off[3] + NM]
            aclose(dw2.numpy(), w2.grad[:, :, 0, 0].numpy(), rtol=1e-4, atol=1e-4)


def test_losses_against_golden_and_oracle(golden):
    g = golden('losses')
    s = torch.from_numpy(g['dice_in']).float().to(DEV).requires_grad_(True)
    t = torch.from_numpy(g['dice_tgt']).float().to(DEV)
    for sb in (True, False):
        s.grad = None
        l = dfl_amd.DiceLoss2D(skip_bg=sb)(s, t)
        l.backward()
        assert abs(l.item() - float(g['dice_sb%d' % int(sb)])) < 2e-6
        aclose(s.grad.cpu().numpy(), g['dice_sb%d_grad' % int(sb)], rtol=1e-4, atol=1e-8)
    assert abs(dfl_amd.DiceLoss2D(skip_bg=False)(t, t).item() - float(g['dice_perfect'])) < 2e-6
    X = torch.from_numpy(g['ncc_x']).float().to(DEV)
    Y = torch.from_numpy(g['ncc_y']).float().to(DEV)
    aclose(dfl_amd.ncc_2d(X, Y).cpu().numpy(), g['ncc'], rtol=1e-4, atol=1e-6)
    aclose(dfl_amd.ncc_2d(Y, Y).cpu().numpy(), g['ncc_self'], rtol=1e-5)
    X.requires_grad_(True)
    s.grad = None
    l = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.3)((s, X), (t, Y))
    l.backward()
    assert abs(l.item() - float(g['dh_loss'])) < 2e-6
    aclose(s.grad.cpu().numpy(), g['dh_gseg'], rtol=1e-4, atol=1e-8)
    aclose(X.grad.cpu().numpy(), g['dh_gheat'], rtol=2e-4, atol=1e-8)


def test_ncc_2d_is_differentiable_in_both_arguments():
    """ncc.py:12-38 under autograd: value and gradients with respect to X and Y against the oracle (fp64), for an
    arbitrary incoming gradient per image."""
    g = torch.Generator().manual_seed(21)
    X = torch.randn(3, 5, 17, 23, generator=g, dtype=torch.float64)
    Y = (0.6 * X + 0.4 * torch.randn(3, 5, 17, 23, generator=g, dtype=torch.float64))
    w = torch.randn(3, 5, generator=g, dtype=torch.float64)
    Xr, Yr = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
    (R.ncc_2d(Xr, Yr) * w).sum().backward()
    Xd, Yd = X.float().to(DEV).requires_grad_(True), Y.float().to(DEV).requires_grad_(True)
    out = dfl_amd.ncc_2d(Xd, Yd)
    (out * w.float().to(DEV)).sum().backward()
    aclose(out.detach().cpu().numpy(), R.ncc_2d(X, Y).numpy(), rtol=1e-5, atol=1e-6)
    aclose(Xd.grad.cpu().numpy(), Xr.grad.numpy(), rtol=1e-4, atol=1e-8)
    aclose(Yd.grad.cpu().numpy(), Yr.grad.numpy(), rtol=1e-4, atol=1e-8)
    # only one argument needs a gradient
    X2 = X.float().to(DEV).requires_grad_(True)
    dfl_amd.ncc_2d(X2, Y.float().to(DEV)).sum().backward()
    Xr.grad = None
    R.ncc_2d(Xr, Y).sum().backward()
    aclose(X2.grad.cpu().numpy(), Xr.grad.numpy(), rtol=1e-4, atol=1e-8)


def test_loss_on_cropped_views_full_size():
    """BASELINE config-2 sized loss on center-cropped (strided) views vs the oracle."""
    g = torch.Generator().manual_seed(9)
    B = 4
    seg = torch.softmax(torch.randn(B, 7, 192, 192, generator=g), 1)
    heat = torch.randn(B, 14, 192, 192, generator=g) * 0.01
    lab = torch.randint(0, 7, (B, 184, 184), generator=g)
    tseg = R.one_hot_masks(lab, 7)
    theat = torch.rand(B, 14, 184, 184, generator=g) * 0.02
    sr, hr = seg.clone().requires_grad_(True), heat.clone().requires_grad_(True)
    lr = R.dice_and_heatmap_loss_2d((R.center_crop(sr, tseg.shape), R.center_crop(hr, theat.shape)), (tseg, theat), False, 0.5)
    lr.backward()
    sd, hd = seg.to(DEV).requires_grad_(True), heat.to(DEV).requires_grad_(True)
    ld = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)(
        (dfl_amd.center_crop(sd, tseg.shape), dfl_amd.center_crop(hd, theat.shape)), (tseg.to(DEV), theat.to(DEV)))
    ld.backward()
    assert abs(ld.item() - lr.item()) < 2e-6
    aclose(sd.grad.cpu().numpy(), sr.grad.numpy(), rtol=1e-3, atol=1e-10)
    aclose(hd.grad.cpu().numpy(), hr.grad.numpy(), rtol=1e-3, atol=1e-9)


def test_two_losses_on_one_cropped_output_accumulate():
    """The loss gradient is written into a pooled zero-bordered full-size tensor (dice._zero_border_buffer) and handed to
    util._Crop's backward as is.  Two losses on the same network output in one backward pass must get two different
    buffers (a buffer still referenced by autograd is not handed out again) and their gradients must add up."""
    g = torch.Generator().manual_seed(31)
    B = 2
    seg = torch.softmax(torch.randn(B, 7, 48, 48, generator=g), 1)
    heat = torch.randn(B, 14, 48, 48, generator=g) * 0.01
    lab1, lab2 = torch.randint(0, 7, (B, 44, 44), generator=g), torch.randint(0, 7, (B, 44, 44), generator=g)
    t1, t2 = R.one_hot_masks(lab1, 7), R.one_hot_masks(lab2, 7)
    th = torch.rand(B, 14, 44, 44, generator=g) * 0.02
    sr, hr = seg.clone().requires_grad_(True), heat.clone().requires_grad_(True)
    lr = (R.dice_and_heatmap_loss_2d((R.center_crop(sr, t1.shape), R.center_crop(hr, th.shape)), (t1, th), False, 0.5)
          + 3.0 * R.dice_and_heatmap_loss_2d((R.center_crop(sr, t2.shape), R.center_crop(hr, th.shape)), (t2, th), True, 0.25))
    lr.backward()
    sd, hd = seg.to(DEV).requires_grad_(True), heat.to(DEV).requires_grad_(True)
    s_out, h_out = sd * 1.0, hd * 1.0                                  # non-leaf outputs, as a network's are
    l1 = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)(
        (dfl_amd.center_crop(s_out, t1.shape), dfl_amd.center_crop(h_out, th.shape)), (t1.to(DEV), th.to(DEV)))
    l2 = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=True, heatmap_wgt=0.25)(
        (dfl_amd.center_crop(s_out, t2.shape), dfl_amd.center_crop(h_out, th.shape)), (t2.to(DEV), th.to(DEV)))
    (l1 + 3.0 * l2).backward()
    assert abs((l1 + 3.0 * l2).item() - lr.item()) < 5e-6
    aclose(sd.grad.cpu().numpy(), sr.grad.numpy(), rtol=1e-3, atol=1e-9)
    aclose(hd.grad.cpu().numpy(), hr.grad.numpy(), rtol=1e-3, atol=1e-9)
    # a second pass re-uses the pooled buffers: same result
    sd.grad = hd.grad = None
    s_out, h_out = sd * 1.0, hd * 1.0
    l1 = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)(
        (dfl_amd.center_crop(s_out, t1.shape), dfl_amd.center_crop(h_out, th.shape)), (t1.to(DEV), th.to(DEV)))
    l1.backward()
    sr.grad = hr.grad = None
    R.dice_and_heatmap_loss_2d((R.center_crop(sr, t1.shape), R.center_crop(hr, th.shape)), (t1, th), False, 0.5).backward()
    aclose(sd.grad.cpu().numpy(), sr.grad.numpy(), rtol=1e-3, atol=1e-9)
    aclose(hd.grad.cpu().numpy(), hr.grad.numpy(), rtol=1e-3, atol=1e-9)


def test_sgd_step():
    lib = nat.lib()
    g = torch.Generator().manual_seed(6)
    n = 10007
    p = torch.randn(n, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    pd, buf = p.to(DEV), torch.zeros(n, device=DEV)
    for step in range(3):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone()
        opt.step()
        grd = gr.to(DEV)
        nat.check(lib.dfl_sgd_step(pd.data_ptr(), grd.data_ptr(), buf.data_ptr(), n, 0.1, 0.9, 1e-4, 1.0, 1, int(step == 0), stream()))
    torch.cuda.synchronize()
    aclose(pd.cpu().numpy(), pr.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_reduce_batch():
    """dfl_reduce_batch: many sums of different shapes in one launch (bias gradients, weight-gradient slices)."""
    import ctypes as C
    lib = nat.lib()
    g = torch.Generator().manual_seed(17)
    # (n, stride, count, T): T > 1 = tap-major slices of dfl_conv2d_wgrad, transposed to [..][T] while summing
    shapes = [(32, 64, 9216, 1), (9216, 9216, 1366, 9), (5, 7, 3, 1), (1, 1, 200, 1), (300000, 300000, 2, 4),
              (819, 819, 70, 1), (64, 128, 64, 1), (36, 36, 5, 9), (1024, 1028, 17, 4), (4096, 4096, 15, 1), (72, 72, 33, 9)]
    srcs, dsts, arr, blocks = [], [], (nat.ReduceJob * len(shapes))(), 0
    for i, (n, stride, count, T) in enumerate(shapes):
        src = torch.randn(count * stride, generator=g).to(DEV)
        dst = torch.full((n,), float('nan'), device=DEV)
        srcs.append(src)
        dsts.append(dst)
        arr[i].src, arr[i].dst, arr[i].n, arr[i].stride, arr[i].count, arr[i].first_block = (
            src.data_ptr(), dst.data_ptr(), n, stride, count, blocks)
        arr[i].T = T
        blocks += nat.check(lib.dfl_reduce_job_blocks(n, count))
    jobs = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(DEV)
    nat.check(lib.dfl_reduce_batch(jobs.data_ptr(), len(shapes), blocks, stream()), 'dfl_reduce_batch')
    torch.cuda.synchronize()
    for (n, stride, count, T), src, dst in zip(shapes, srcs, dsts):
        want = src.cpu().double().view(count, stride)[:, :n].sum(0).view(T, n // T).t().reshape(-1)
        aclose(dst.cpu().numpy(), want.numpy(), rtol=2e-6, atol=1e-5)


@pytest.mark.parametrize('shape', [(2, 128, 64, 20, 28, 3), (1, 64, 96, 13, 9, 3), (2, 32, 32, 33, 17, 3), (3, 64, 160, 6, 6, 3),
                                   (2, 64, 32, 12, 10, 1), (2, 32, 64, 16, 12, 2)])
def test_conv_presplit_operands(shape, math_mode):
    """bf16x3 only: weights packed as split quads (dfl_pack_job.split), alone and together with a pre-split input tensor
    (x_split), through the simple and the general epilogue, one pass and split-K -- against the fp32 convolution at the
    bf16x3 bar."""
    if math_mode != 'bf16x3':
        pytest.skip('split operands exist for bf16x3 products')
    N, Cin, Cout, H, W, K = shape
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    stride, pad = (2, 0) if K == 2 else (1, K // 2)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    other = torch.randn(N, Cout, Ho, Wo, generator=g)

    def ref(xin):
        return nhwc(F.relu(F.conv2d(xin, w, b, stride=stride, padding=pad)))

    # the affine happens before zero padding in the reference order BN -> conv: emulate by transforming x, padding after
    x_aff = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    for wsp in (1,):
        wp = pack(w, 1, split=wsp)
        y = conv_call(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, w_split=wsp)
        aclose(y.numpy(), ref(x).numpy(), rtol=2e-5, atol=2e-5, err_msg='w_split %d' % wsp)
        y, st = conv_call(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, in_aff=(sc, sh), stats=True, w_split=wsp)
        r = ref(x_aff)
        aclose(y.numpy(), r.numpy(), rtol=2e-5, atol=3e-5, err_msg='w_split %d + affine' % wsp)
        aclose(st[0].numpy(), r.double().sum((0, 1, 2)).numpy(), rtol=1e-5, atol=1e-3)
        # both operands pre-split, general epilogue (residual add), statistics against a partner tensor
        y, st = conv_call(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, add=other, stats=True, stat_other=other,
                          w_split=wsp, x_split=1)
        r = ref(x) + nhwc(other)
        aclose(y.numpy(), r.numpy(), rtol=2e-5, atol=3e-5, err_msg='w_split %d + x_split' % wsp)
        aclose(st[1].numpy(), (r.double() * nhwc(other).double()).sum((0, 1, 2)).numpy(), rtol=1e-5, atol=2e-3)
        if Cin * K * K >= 512:
            y = conv_call(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, w_split=wsp, x_split=1, force_splits=3)
            aclose(y.numpy(), ref(x).numpy(), rtol=2e-5, atol=2e-5, err_msg='w_split %d + x_split, split-K' % wsp)


@pytest.fixture
def rows_everywhere():
    """Take the row-tiled kernels also for problems far below the size they are meant for."""
    lib = nat.lib()
    old = lib.dfl_set_conv_rows_min_tiles(1)
    yield
    lib.dfl_set_conv_rows_min_tiles(old)


@pytest.mark.parametrize('shape', [(2, 32, 32, 5, 192), (1, 64, 32, 3, 192), (2, 32, 64, 4, 96), (1, 64, 64, 3, 96),
                                   (1, 128, 64, 2, 192), (1, 32, 20, 3, 384), (2, 64, 48, 1, 96),
                                   # several complete image rows per tile, tiles that straddle two images
                                   (2, 32, 64, 48, 48), (4, 64, 128, 24, 24), (8, 32, 32, 12, 12), (16, 64, 96, 6, 6),
                                   (2, 32, 32, 24, 48), (3, 32, 64, 8, 32),
                                   # 96 x 32 tiles: narrow layers whose rows do not divide into 192 pixels
                                   (1, 32, 32, 3, 96), (1, 32, 16, 2, 1440)])
def test_conv_row_tiles(shape, math_mode, rows_everywhere):
    """The row-tiled 3x3 kernels (conv_rows.hip: wide images, <= 64 output channels; tiles of 192 / 96 pixels of one
    image row, each kernel row's pixels staged once for its three taps): plain, with the BatchNorm affine on load (zero
    padding after the affine), with statistics against a partner tensor, and -- bf16x3 -- with pre-split operands;
    image borders, several images and ragged channel counts included."""
    lib = nat.lib()
    N, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(Cin * 7 + Cout + W)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    other = torch.randn(N, Cout, H, W, generator=g)
    bf = math_mode == 'bf16x3'
    wp = pack(w, 1, split=1 if bf else 0)
    # the path under test is the one taken
    a = nat.ConvArgs(x=wp.data_ptr(), w=wp.data_ptr(), y=wp.data_ptr(), N=N, Hin=H, Win=W, Cin=Cin, ldx=Cin, KH=3, KW=3,
                     stride=1, pad=1, Hout=H, Wout=W, Ntot=Cout, ldy=Cout, w_split=int(bf))
    def fits(bm):
        return (W % bm == 0 or (bm % W == 0 and bm // W <= 16)) and (N * H * W) % bm == 0
    want = (6 if fits(192) else 8) if (Cout <= 32 and (fits(192) or fits(96))) else 7      # 192x32, 96x32, 96x64 tiles
    if not bf:
        pytest.skip('the row-tiled kernels exist for bf16x3 products (fp32 products are matrix-pipe bound and keep the generic tiles)')
    assert nat.check(lib.dfl_conv_config(C.addressof(a))) == want
    ref = nhwc(F.relu(F.conv2d(x, w, b, padding=1)))
    y, st = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, relu=1, stats=True, w_split=int(bf))
    aclose(y.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    aclose(st[0].numpy(), ref.double().sum((0, 1, 2)).numpy(), rtol=1e-5, atol=2e-3)
    aclose(st[1].numpy(), (ref.double() ** 2).sum((0, 1, 2)).numpy(), rtol=1e-5, atol=2e-3)
    x_aff = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    refa = nhwc(F.conv2d(x_aff, w, b, padding=1))
    y, st = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, in_aff=(sc, sh), stats=True, stat_other=other, w_split=int(bf))
    aclose(y.numpy(), refa.numpy(), rtol=2e-5, atol=4e-5)
    aclose(st[1].numpy(), (refa.double() * nhwc(other).double()).sum((0, 1, 2)).numpy(), rtol=1e-5, atol=4e-3)
    if bf:
        y = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, relu=1, w_split=1, x_split=1)
        aclose(y.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    # K slices ((dy, channel chunk) pairs dealt out over blockIdx.z) + finish kernel, which then also carries the
    # epilogues the one-pass form leaves to the generic kernel (residual add, accumulate)
    for sp in (2, 3):
        if (3 * Cin // 16) % (2 * sp) != 0:
            continue
        a.splits = sp
        assert nat.check(lib.dfl_conv_config(C.addressof(a))) == want
        y, st = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, in_aff=(sc, sh), stats=True, stat_other=other,
                          w_split=int(bf), force_splits=sp)
        aclose(y.numpy(), refa.numpy(), rtol=2e-5, atol=4e-5)
        aclose(st[1].numpy(), (refa.double() * nhwc(other).double()).sum((0, 1, 2)).numpy(), rtol=1e-5, atol=4e-3)
        y = conv_call(x, wp, Cout, 3, 3, 1, 1, H, W, bias=b, relu=1, add=other, y_init=other * 0.5, accumulate=1,
                      w_split=int(bf), force_splits=sp)
        aclose(y.numpy(), (ref + 1.5 * nhwc(other)).numpy(), rtol=2e-5, atol=4e-5)
