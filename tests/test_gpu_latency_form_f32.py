"""The latency form of the convolution for fp32 tensors (csrc/convs_f32.hip: math modes fp32 and bf16x3 -- the arithmetics that hold
the 1e-4 forward bar; reference loops: train_test_code/util.py:116-165, :318-356) through the C ABI against fp64 PyTorch at the
bars of tests/test_gpu_kernels.py and against the GEMM kernels on the same argument block; pairs, the output affine, the first
layer; then a whole batch-1 inference forward with and without it.  pytest -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dfl_amd
from dfl_amd import _native as nat
from conftest import by_mode
from test_gpu_kernels import nhwc, pack, conv_call, aclose, stream, _both_math_modes  # noqa: F401  (fixture: both arithmetics)

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# N, Cin, Cout, H, W, K, stride, pad
FCASES = [
    (1, 32, 32, 96, 96, 3, 1, 1),       # one wave per tile
    (1, 64, 64, 48, 48, 3, 1, 1),
    (1, 128, 128, 24, 24, 3, 1, 1),
    (1, 256, 256, 12, 12, 3, 1, 1),     # eight waves per tile
    (1, 512, 512, 12, 12, 3, 1, 1),     # + K slices (finish kernel)
    (1, 1024, 1024, 6, 6, 3, 1, 1),
    (2, 64, 40, 13, 9, 3, 1, 1),        # ragged pixels, 40 columns
    (1, 32, 16, 10, 10, 3, 1, 0),       # valid convolution
    (1, 64, 32, 96, 96, 1, 1, 0),       # 1x1
    (1, 64, 64, 48, 48, 2, 2, 0),       # 2x2 stride 2
    (3, 32, 32, 7, 9, 2, 2, 0),
    (1, 16, 16, 12, 12, 3, 1, 1),
]


@pytest.mark.parametrize('case', FCASES)
def test_latency_form_f32_plain(case, math_mode):
    N, Cin, Cout, H, W, K, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    split = 1 if (math_mode == 'bf16x3' and Cin % 16 == 0) else 0
    wp = pack(w, 1, split=split)
    y = conv_call(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, w_split=split, latency=True)
    ref = nhwc(F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)))
    aclose(y.double().numpy(), ref.numpy(), rtol=2e-5, atol=2e-6 * float(ref.abs().max()) * (Cin * K * K) ** 0.5)
    y0 = conv_call(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, w_split=split)
    aclose(y.numpy(), y0.numpy(), rtol=2e-5, atol=2e-6 * float(ref.abs().max()) * (Cin * K * K) ** 0.5)


@pytest.mark.parametrize('case', [(1, 32, 32, 20, 20, 3), (1, 64, 128, 12, 12, 3), (1, 256, 256, 6, 6, 3), (2, 128, 64, 9, 7, 1), (1, 512, 512, 12, 12, 3)])
def test_latency_form_f32_affine_residual_epilogue(case, math_mode):
    N, Cin, Cout, H, W, K = case
    pad = K // 2
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    other = torch.randn(N, Cout, H, W, generator=g)
    asc, ash = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    y0 = torch.randn(N, Cout, H, W, generator=g)
    split = 1 if math_mode == 'bf16x3' else 0
    xa = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref = F.conv2d(xa, w.double(), b.double(), padding=pad)       # padding stays zero: it is applied after the affine
    ref = ref + other.double() * asc.double().view(1, -1, 1, 1) + ash.double().view(1, -1, 1, 1)
    tol = 2e-6 * float(ref.abs().max()) * (Cin * K * K) ** 0.5
    y = conv_call(x, pack(w, 1, split=split), Cout, K, K, 1, pad, H, W, bias=b, in_aff=(sc, sh), add=other, add_aff=(asc, ash), w_split=split, latency=True)
    aclose(y.double().numpy(), nhwc(ref).numpy(), rtol=2e-5, atol=tol)
    ya = conv_call(x, pack(w, 1, split=split), Cout, K, K, 1, pad, H, W, bias=b, in_aff=(sc, sh), add=other, add_aff=(asc, ash), y_init=y0, accumulate=1,
                   w_split=split, latency=True)
    aclose(ya.double().numpy(), nhwc(ref + y0.double()).numpy(), rtol=2e-5, atol=tol)


def test_latency_form_f32_transposed_scatter(math_mode):
    g = torch.Generator().manual_seed(9)
    split = 1 if math_mode == 'bf16x3' else 0
    for (N, Ci, Co, H, W) in ((1, 64, 32, 10, 7), (1, 1024, 512, 6, 6), (1, 64, 32, 48, 48)):
        x = torch.randn(N, Ci, H, W, generator=g)
        w = torch.randn(Ci, Co, 2, 2, generator=g) / (4 * Ci) ** 0.5
        b = torch.randn(Co, generator=g)
        y = conv_call(x, pack(w, 3, split=split), 4 * Co, 1, 1, 1, 0, 2 * H, 2 * W, bias=b, scatter=1, ldy=2 * Co, w_split=split, latency=True)
        ref = nhwc(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2))
        aclose(y.double().numpy(), ref.numpy(), rtol=2e-5, atol=2e-6 * float(ref.abs().max()) * Ci ** 0.5)


def _args(x_dev, w_dev, y_dev, N, H, W, Cin, Cout, K, pad, split, ldy=None):
    a = nat.ConvArgs()
    a.x, a.w, a.y = x_dev.data_ptr(), w_dev.data_ptr(), y_dev.data_ptr()
    a.N, a.Hin, a.Win, a.Cin, a.ldx = N, H, W, Cin, Cin
    a.KH, a.KW, a.stride, a.pad = K, K, 1, pad
    a.Hout, a.Wout, a.Ntot, a.ldy = H, W, Cout, ldy or Cout
    a.w_split = split
    a.latency_form = 1
    return a


def _run(lib, q, N, H, W, C_, mult=1):
    sp = nat.check(lib.dfl_conv_suggest_splits(C.addressof(q)), 'suggest')
    if sp > 1:
        q.splits = sp
        q._part = torch.full((mult * sp * N * H * W * C_,), float('nan'), device=DEV)
        q.partial = q._part.data_ptr()
    return sp


def test_output_affine_f32_is_the_consumers_affine_on_load(math_mode):
    """dfl_conv_args.out_scale with fp32 tensors: nothing is rounded, the producer's fma IS the consumer's affine on load: the same bits."""
    lib = nat.lib()
    split = 1 if math_mode == 'bf16x3' else 0
    for (Cin, C_, H, W) in ((32, 64, 40, 40), (128, 128, 24, 20), (256, 512, 12, 12)):
        g = torch.Generator().manual_seed(Cin + H)
        x = nhwc(torch.randn(1, Cin, H, W, generator=g)).to(DEV).contiguous()
        wa = pack(torch.randn(C_, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, 1, split=split)
        wb = pack(torch.randn(C_, C_, 3, 3, generator=g) / (9 * C_) ** 0.5, 1, split=split)
        ba, bb = torch.randn(C_, generator=g).to(DEV), torch.randn(C_, generator=g).to(DEV)
        sc, sh = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.3).to(DEV)

        def run(producer_side):
            r = torch.full((1, H, W, C_), float('nan'), device=DEV)
            y = torch.full((1, H, W, C_), float('nan'), device=DEV)
            a = _args(x, wa, r, 1, H, W, Cin, C_, 3, 1, split)
            a.bias, a.relu = ba.data_ptr(), 1
            b = _args(r, wb, y, 1, H, W, C_, C_, 3, 1, split)
            b.bias, b.relu = bb.data_ptr(), 1
            if producer_side:
                a.out_scale, a.out_shift = sc.data_ptr(), sh.data_ptr()
                assert lib.dfl_conv_config(C.addressof(a)) == 16 + 39
            else:
                b.in_scale, b.in_shift = sc.data_ptr(), sh.data_ptr()
            for q in (a, b):
                _run(lib, q, 1, H, W, C_)
                nat.check(lib.dfl_conv2d(C.addressof(q), stream()), 'conv')
            torch.cuda.synchronize()
            return y.cpu()
        assert torch.equal(run(True), run(False)), (Cin, C_, H, W)


@pytest.mark.parametrize('case', [(1, 32, 32, 40, 40), (1, 64, 128, 48, 48), (1, 256, 512, 12, 12), (2, 96, 64, 13, 9), (1, 0, 32, 48, 40), (1, 512, 1024, 6, 6),
                                  (1, 1024, 512, 12, 12)])
def test_pair_f32_is_the_two_launches(case, math_mode):
    """dfl_conv2d_pair with fp32 tensors against dfl_conv2d(a); dfl_conv2d(b): y1 bit for bit, y2 up to the 1x1 product's summation order
    (Cres = 0: the first block's 1-channel image, bit for bit)."""
    N, Cres, C_, H, W = case
    lib = nat.lib()
    split = 1 if math_mode == 'bf16x3' else 0
    g = torch.Generator().manual_seed(sum(case))
    r1 = nhwc(torch.randn(N, C_, H, W, generator=g)).to(DEV).contiguous()
    w = torch.randn(C_, C_, 3, 3, generator=g) / (9 * C_) ** 0.5
    wp = pack(w, 1, split=split)
    b1, b3 = torch.randn(C_, generator=g).to(DEV), torch.randn(C_, generator=g).to(DEV)
    asc, ash = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.2).to(DEV)
    if Cres:
        xin = torch.randn(N, Cres, H, W, generator=g)
        w3 = torch.randn(C_, Cres, 1, 1, generator=g) / Cres ** 0.5
        xind = nhwc(xin).to(DEV).contiguous()
        w3p = pack(w3, 1, split=split)
    else:
        xin = torch.randn(N, 1, H, W, generator=g)
        w3 = torch.randn(C_, 1, 1, 1, generator=g)
        xind = nhwc(xin).to(DEV).contiguous()
        w3p = pack(w3, 1)

    def run(paired):
        y1 = torch.full((N, H, W, C_), float('nan'), device=DEV)
        y2 = torch.full((N, H, W, 2 * C_), float('nan'), device=DEV)
        a = _args(r1, wp, y1, N, H, W, C_, C_, 3, 1, split)
        a.bias, a.relu = b1.data_ptr(), 1
        b = _args(xind, w3p, y2, N, H, W, Cres or 1, C_, 1, 0, split if Cres else 0, ldy=2 * C_)
        b.bias, b.add, b.ldadd, b.add_scale, b.add_shift = b3.data_ptr(), y1.data_ptr(), C_, asc.data_ptr(), ash.data_ptr()
        sp = _run(lib, a, N, H, W, C_, mult=2)
        ok = lib.dfl_conv_pair_ok(C.addressof(a), C.addressof(b))
        if paired:
            assert ok == (2 if sp > 1 else 1), 'these two convolutions form a pair'
            nat.check(lib.dfl_conv2d_pair(C.addressof(a), C.addressof(b), stream()), 'pair')
        else:
            nat.check(lib.dfl_conv2d(C.addressof(a), stream()), 'a')
            nat.check(lib.dfl_conv2d(C.addressof(b), stream()), 'b')
        torch.cuda.synchronize()
        return y1.cpu(), y2.cpu()[..., :C_]
    y1p, y2p = run(True)
    y1s, y2s = run(False)
    assert torch.equal(y1p, y1s)
    if not Cres:
        assert torch.equal(y2p, y2s)
    y1ref = nhwc(F.relu(F.conv2d(r1.cpu().permute(0, 3, 1, 2).double(), w.double(), b1.cpu().double(), padding=1)))
    aclose(y1p.double().numpy(), y1ref.numpy(), rtol=2e-5, atol=2e-6 * float(y1ref.abs().max()) * (9 * C_) ** 0.5)
    y2ref = nhwc(F.conv2d(xin.double(), w3.double(), b3.cpu().double())) + y1p.double() * asc.cpu().double() + ash.cpu().double()
    aclose(y2p.double().numpy(), y2ref.numpy(), rtol=2e-5, atol=2e-6 * float(y2ref.abs().max()) * max(Cres, 1) ** 0.5)


def test_first_layer_f32_latency_form_is_the_direct_kernel(math_mode):
    lib = nat.lib()
    for (N, H, W, C_) in ((1, 192, 192, 32), (2, 37, 41, 16)):
        g = torch.Generator().manual_seed(H + C_)
        x = torch.randn(N, H, W, 1, generator=g).to(DEV).contiguous()
        w = torch.randn(C_, 1, 3, 3, generator=g)
        wq = pack(w, 1)
        b = torch.randn(C_, generator=g).to(DEV)
        outs = []
        for hint in (1, 0):
            y = torch.full((N, H, W, C_), float('nan'), device=DEV)
            a = nat.ConvArgs()
            a.x, a.w, a.y, a.bias = x.data_ptr(), wq.data_ptr(), y.data_ptr(), b.data_ptr()
            a.N, a.Hin, a.Win, a.Cin, a.ldx = N, H, W, 1, 1
            a.KH, a.KW, a.stride, a.pad = 3, 3, 1, 1
            a.Hout, a.Wout, a.Ntot, a.ldy = H, W, C_, C_
            a.relu, a.latency_form = 1, hint
            assert (lib.dfl_conv_config(C.addressof(a)) == 16 + 39) == bool(hint)
            nat.check(lib.dfl_conv2d(C.addressof(a), stream()), 'conv')
            torch.cuda.synchronize()
            outs.append(y.cpu())
        assert torch.equal(outs[0], outs[1]), (N, H, W, C_)


def test_inference_forward_f32_with_and_without_the_latency_form(math_mode):
    """Eval-mode forward of the paper network at 192x192, batch 1, in the parity arithmetics: the plan's hint against plans without it."""
    import bench
    torch.manual_seed(5)
    net = dfl_amd.UNet(**bench.PAPER).to(DEV).eval()
    x = torch.randn(1, 1, 192, 192, device=DEV)
    with torch.no_grad():
        seg1, heat1 = net(x)
        plan = [p for ps in net._plans.values() for p in ps if not p.need_grad and p.math == nat.lib().dfl_get_math_mode()][0]
        assert sum(1 for st in plan.fwd.structs if isinstance(st, nat.ConvPairArgs)) >= 11
        seg1, heat1 = seg1.clone(), heat1.clone()
        net2 = dfl_amd.UNet(**bench.PAPER).to(DEV).eval()
        net2.load_state_dict(net.state_dict())
        old = os.environ.get('DFL_PLAN_LATENCY_FORM')
        os.environ['DFL_PLAN_LATENCY_FORM'] = '0'
        try:
            seg0, heat0 = net2(x)
        finally:
            if old is None:
                del os.environ['DFL_PLAN_LATENCY_FORM']
            else:
                os.environ['DFL_PLAN_LATENCY_FORM'] = old
    tol = by_mode(math_mode, 2e-5, 1e-4)
    assert float((seg1 - seg0).abs().max()) < tol
    assert float((heat1 - heat0).abs().max()) <= tol * max(1.0, float(heat0.abs().max()))
