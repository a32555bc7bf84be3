"""The latency form of the bf16 convolution (csrc/convs_bf16.hip, dfl_conv_args.latency_form: the kernel inference plans ask
for on the small problems of a batch-1 forward; reference loops: train_test_code/util.py:116-165, :318-356) through the C ABI
against fp64 PyTorch on the same bf16 operands -- the bars of tests/test_gpu_bf16.py -- and against the patch-resident kernel
on the same argument block; then a whole inference forward with and without it.  pytest -m gpu."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

import dfl_amd
from dfl_amd import _native as nat
from test_gpu_bf16 import rb, nhwc, pack16, conv_bf16, close_bf16, _mode4  # noqa: F401  (the fixture switches to bf16 storage)

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# N, Cin, Cout, H, W, K, stride, pad: every window of the network, ragged pixel counts, columns that are no multiple of 32, the
# wave splits 1 / 2 / 4 / 8 of a tile's k-steps and K slices over workgroups (the deep levels of a 192x192 image)
LCASES = [
    (1, 32, 32, 192, 192, 3, 1, 1),     # level 0: one wave per tile
    (1, 32, 64, 96, 96, 3, 1, 1),       # level 1
    (1, 64, 64, 80, 80, 3, 1, 1),       # two waves per tile
    (1, 128, 128, 48, 48, 3, 1, 1),     # four
    (1, 256, 256, 24, 24, 3, 1, 1),     # eight
    (1, 512, 512, 12, 12, 3, 1, 1),     # eight + K slices (finish kernel)
    (1, 1024, 1024, 6, 6, 3, 1, 1),     # 36 pixels: the second pixel tile is nearly empty
    (1, 1024, 512, 12, 12, 3, 1, 1),    # decoder, 576 k-steps
    (2, 64, 40, 13, 9, 3, 1, 1),        # ragged pixels, 40 columns
    (1, 32, 16, 10, 10, 3, 1, 0),       # valid convolution
    (1, 64, 32, 192, 192, 1, 1, 0),     # 1x1
    (1, 32, 64, 17, 11, 1, 1, 0),
    (1, 64, 64, 96, 96, 2, 2, 0),       # 2x2 stride 2
    (3, 32, 32, 7, 9, 2, 2, 0),         # ... odd input
    (1, 16, 16, 12, 12, 3, 1, 1),       # one chunk per tap
]


@pytest.mark.parametrize('case', LCASES)
def test_latency_form_plain(case):
    N, Cin, Cout, H, W, K, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    wp = pack16(w, 1)
    y = conv_bf16(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1, latency=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad))
    close_bf16(y, nhwc(ref), str(case))
    # ... and against the patch-resident kernel: the same values up to the fp32 summation order in front of the one rounding
    y0 = conv_bf16(x, wp, Cout, K, K, stride, pad, Ho, Wo, bias=b, relu=1)
    differing = float((y != y0).double().mean())
    assert differing < 0.02, 'fraction of elements that round differently: %.4f' % differing
    assert float((y - y0).abs().max()) <= 2.0 ** -7 * float(y0.abs().max())


@pytest.mark.parametrize('case', [(1, 32, 32, 20, 20, 3), (1, 64, 128, 12, 12, 3), (1, 256, 256, 6, 6, 3), (2, 128, 64, 9, 7, 1), (1, 512, 512, 12, 12, 3)])
def test_latency_form_affine_residual_epilogue(case):
    """BatchNorm affine on load with zero padding AFTER it and the '+ BN(other)' residual sum (unet.py:211-231 in eval mode)."""
    N, Cin, Cout, H, W, K = case
    pad = K // 2
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    other = rb(torch.randn(N, Cout, H, W, generator=g))
    asc, ash = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    y0 = rb(torch.randn(N, Cout, H, W, generator=g))
    xa = rb(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = F.conv2d(xa.double(), w.double(), b.double(), padding=pad)
    ref = ref + other.double() * asc.double().view(1, -1, 1, 1) + ash.double().view(1, -1, 1, 1)
    y = conv_bf16(x, pack16(w, 1), Cout, K, K, 1, pad, H, W, bias=b, in_aff=(sc, sh), add=other, add_aff=(asc, ash), latency=True)
    close_bf16(y, nhwc(ref), str(case))
    ya = conv_bf16(x, pack16(w, 1), Cout, K, K, 1, pad, H, W, bias=b, in_aff=(sc, sh), add=other, add_aff=(asc, ash), y_init=y0, accumulate=1,
                   latency=True)
    close_bf16(ya, nhwc(ref + y0.double()), str(case) + ' accumulate')


def test_latency_form_transposed_scatter():
    """ConvTranspose2d(k2,s2) as a 1x1 gather with the 2x2-scatter epilogue into the channel half of a wider buffer (unet.py:240,255-257)."""
    g = torch.Generator().manual_seed(9)
    for (N, Ci, Co, H, W) in ((1, 64, 32, 10, 7), (1, 1024, 512, 6, 6), (1, 64, 32, 96, 96)):
        x = rb(torch.randn(N, Ci, H, W, generator=g))
        w = rb(torch.randn(Ci, Co, 2, 2, generator=g) / (4 * Ci) ** 0.5)
        b = torch.randn(Co, generator=g)
        y = conv_bf16(x, pack16(w, 3), 4 * Co, 1, 1, 1, 0, 2 * H, 2 * W, bias=b, scatter=1, ldy=2 * Co, latency=True)
        close_bf16(y, nhwc(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)), 'convT %d' % Ci)


def test_output_affine_is_the_consumers_affine_on_load():
    """dfl_conv_args.out_scale: producer applies the next layer's eval-mode BatchNorm -- conv(a) with out_scale, then conv(b) plain,
    against conv(a), then conv(b) with the affine on load: the same bits (the same k-step split in b)."""
    lib = nat.lib()
    BF = torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    for (Cin, C_, H, W) in ((32, 64, 40, 40), (128, 128, 24, 20), (256, 512, 12, 12)):
        g = torch.Generator().manual_seed(Cin + H)
        x = nhwc(rb(torch.randn(1, Cin, H, W, generator=g))).to(DEV).to(BF).contiguous()
        wa = pack16(rb(torch.randn(C_, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5), 1)
        wb = pack16(rb(torch.randn(C_, C_, 3, 3, generator=g) / (9 * C_) ** 0.5), 1)
        ba, bb = torch.randn(C_, generator=g).to(DEV), torch.randn(C_, generator=g).to(DEV)
        sc, sh = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.3).to(DEV)

        def run(producer_side):
            r = torch.full((1, H, W, C_), float('nan'), device=DEV, dtype=BF)
            y = torch.full((1, H, W, C_), float('nan'), device=DEV, dtype=BF)
            a = _conv_args(x, wa, r, 1, H, W, Cin, C_, 3, 1)
            a.bias, a.relu = ba.data_ptr(), 1
            b = _conv_args(r, wb, y, 1, H, W, C_, C_, 3, 1)
            b.bias, b.relu = bb.data_ptr(), 1
            if producer_side:
                a.out_scale, a.out_shift = sc.data_ptr(), sh.data_ptr()
                assert lib.dfl_conv_config(C.addressof(a)) == 16 + 39
            else:
                b.in_scale, b.in_shift = sc.data_ptr(), sh.data_ptr()
            for q in (a, b):
                sp = nat.check(lib.dfl_conv_suggest_splits(C.addressof(q)), 'suggest')
                if sp > 1:
                    q.splits = sp
                    q._part = torch.empty(sp * H * W * C_, device=DEV)
                    q.partial = q._part.data_ptr()
                nat.check(lib.dfl_conv2d(C.addressof(q), st), 'conv')
            torch.cuda.synchronize()
            return y.float().cpu()
        assert torch.equal(run(True), run(False)), (Cin, C_, H, W)


def test_first_layer_latency_form_is_the_row_kernel():
    """1-channel 3x3 first layer (unet.py:211, in_channels = 1): the latency form against direct_conv3_rows_kernel, bit for bit."""
    lib = nat.lib()
    BF = torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    for (N, H, W, C_) in ((1, 192, 192, 32), (2, 37, 41, 16), (1, 64, 48, 64)):
        g = torch.Generator().manual_seed(H + C_)
        x = torch.randn(N, H, W, 1, generator=g).to(DEV).contiguous()
        w = torch.randn(C_, 1, 3, 3, generator=g)
        wq = torch.zeros(3, C_, 4)                            # quad-packed fp32 operand [ceil(9/4)][N][4]
        for k in range(9):
            wq[k >> 2, :, k & 3] = w.reshape(C_, 9)[:, k]
        wq = wq.to(DEV).contiguous()
        b = torch.randn(C_, generator=g).to(DEV)
        outs = []
        for hint in (1, 0):
            y = torch.full((N, H, W, C_), float('nan'), device=DEV, dtype=BF)
            a = nat.ConvArgs()
            a.x, a.w, a.y, a.bias = x.data_ptr(), wq.data_ptr(), y.data_ptr(), b.data_ptr()
            a.x_bf16, a.y_bf16 = 0, 1
            a.N, a.Hin, a.Win, a.Cin, a.ldx = N, H, W, 1, 1
            a.KH, a.KW, a.stride, a.pad = 3, 3, 1, 1
            a.Hout, a.Wout, a.Ntot, a.ldy = H, W, C_, C_
            a.relu, a.latency_form = 1, hint
            assert (lib.dfl_conv_config(C.addressof(a)) == 16 + 39) == bool(hint)
            nat.check(lib.dfl_conv2d(C.addressof(a), st), 'conv')
            torch.cuda.synchronize()
            outs.append(y.float().cpu())
        assert torch.equal(outs[0], outs[1]), (N, H, W, C_)
        ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double(), b.double().cpu(), padding=1))
        close_bf16(outs[0], nhwc(ref), 'first layer')


def _conv_args(x_dev, w_dev, y_dev, N, H, W, Cin, Cout, K, pad, ldx=None, ldy=None):
    a = nat.ConvArgs()
    a.x, a.w, a.y = x_dev.data_ptr(), w_dev.data_ptr(), y_dev.data_ptr()
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = N, H, W, Cin, ldx or Cin
    a.KH, a.KW, a.stride, a.pad = K, K, 1, pad
    a.Hout, a.Wout, a.Ntot, a.ldy = H, W, Cout, ldy or Cout
    a.latency_form = 1
    return a


@pytest.mark.parametrize('case', [(1, 32, 32, 40, 40), (1, 64, 128, 48, 48), (1, 128, 256, 24, 24), (1, 256, 512, 12, 12), (2, 96, 64, 13, 9), (1, 0, 32, 48, 40),
                                  (1, 512, 1024, 6, 6), (1, 1024, 512, 12, 12)])
def test_pair_is_the_two_launches(case):
    """dfl_conv2d_pair(a, b) -- the last 3x3 convolution of a residual block (ReLU) and the block's 1x1
    convolution with '+ BN(y1)' (unet.py:218-231) -- against dfl_conv2d(a); dfl_conv2d(b): y1 bit for bit (same k-step split), y2 up
    to the summation order of the 1x1 product; Cres = 0: the first block's 1-channel fp32 image (direct 1x1 kernel)."""
    N, Cres, C_, H, W = case
    lib = nat.lib()
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(sum(case))
    r1 = rb(torch.randn(N, C_, H, W, generator=g))
    w = rb(torch.randn(C_, C_, 3, 3, generator=g) / (9 * C_) ** 0.5)
    b1, b3 = torch.randn(C_, generator=g).to(DEV), torch.randn(C_, generator=g).to(DEV)
    sc, sh = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.3).to(DEV)
    asc, ash = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.2).to(DEV)
    r1d = nhwc(r1).to(DEV).to(BF).contiguous()
    wp = pack16(w, 1)
    if Cres:
        xin = rb(torch.randn(N, Cres, H, W, generator=g))
        w3 = rb(torch.randn(C_, Cres, 1, 1, generator=g) / Cres ** 0.5)
        xind = nhwc(xin).to(DEV).to(BF).contiguous()
        w3p = pack16(w3, 1)
    else:
        xin = torch.randn(N, 1, H, W, generator=g)
        w3 = torch.randn(C_, 1, 1, 1, generator=g)
        xind = nhwc(xin).to(DEV).contiguous()
        w3p = torch.zeros(C_, 4, device=DEV)                 # quad-packed fp32 operand [ceil(K/4)][N][4], K = 1
        w3p[:, 0] = w3.reshape(-1).to(DEV)

    def run(paired):
        y1 = torch.full((N, H, W, C_), float('nan'), device=DEV, dtype=BF)
        y2 = torch.full((N, H, W, 2 * C_), float('nan'), device=DEV, dtype=BF)        # (the concat buffer's half: ldy = 2 C)
        a = _conv_args(r1d, wp, y1, N, H, W, C_, C_, 3, 1)
        a.bias, a.relu = b1.data_ptr(), 1                    # (its operand is plain: the producer applied the BatchNorm, out_scale)
        if Cres:
            b = _conv_args(xind, w3p, y2, N, H, W, Cres, C_, 1, 0, ldy=2 * C_)
        else:
            b = _conv_args(xind, w3p, y2, N, H, W, 1, C_, 1, 0, ldy=2 * C_)
            b.x_bf16, b.w_split = 0, 0
        b.bias, b.add, b.ldadd, b.add_scale, b.add_shift = b3.data_ptr(), y1.data_ptr(), C_, asc.data_ptr(), ash.data_ptr()
        sp = nat.check(lib.dfl_conv_suggest_splits(C.addressof(a)), 'suggest')
        part = None
        if sp > 1:                                           # K slices: a finish launch (for the pair: one for both outputs)
            part = torch.full((2 * sp * N * H * W * C_,), float('nan'), device=DEV)
            a.splits, a.partial = sp, part.data_ptr()
        ok = lib.dfl_conv_pair_ok(C.addressof(a), C.addressof(b))
        st = torch.cuda.current_stream().cuda_stream
        if paired:
            assert ok == (2 if sp > 1 else 1), 'these two convolutions form a pair'
            nat.check(lib.dfl_conv2d_pair(C.addressof(a), C.addressof(b), st), 'pair')
        else:
            nat.check(lib.dfl_conv2d(C.addressof(a), st), 'a')
            nat.check(lib.dfl_conv2d(C.addressof(b), st), 'b')
        torch.cuda.synchronize()
        return y1.float().cpu(), y2.float().cpu()[..., :C_]
    y1p, y2p = run(True)
    y1s, y2s = run(False)
    assert torch.equal(y1p, y1s)
    if not Cres:
        assert torch.equal(y2p, y2s)                         # fp32 multiply-adds in the same order
    else:
        assert float((y2p != y2s).double().mean()) < 0.02
    # and against fp64 on the same operands
    y1ref = F.relu(F.conv2d(r1.double(), w.double(), b1.cpu().double(), padding=1))
    close_bf16(y1p, nhwc(y1ref), 'y1 %s' % (case,))
    y1r = nhwc(y1p).permute(0, 1, 2, 3)                      # NHWC values as stored
    y2ref = F.conv2d(xin.double(), w3.double(), b3.cpu().double()) + y1p.permute(0, 3, 1, 2).double() * asc.cpu().double().view(1, -1, 1, 1) + \
        ash.cpu().double().view(1, -1, 1, 1)
    close_bf16(y2p, nhwc(y2ref), 'y2 %s' % (case,))


def test_latency_form_is_a_hint():
    """Statistics, large problems and the fused backward operand keep the patch-resident kernels whatever the hint says."""
    lib = nat.lib()
    g = torch.Generator().manual_seed(3)
    x = rb(torch.randn(16, 64, 96, 96, generator=g))
    w = rb(torch.randn(64, 64, 3, 3, generator=g) / 24.0)
    a = nat.ConvArgs()
    xd = nhwc(x).to(DEV).to(torch.bfloat16).contiguous()
    yd = torch.empty(16, 96, 96, 64, device=DEV, dtype=torch.bfloat16)
    wp = pack16(w, 1)
    a.x, a.w, a.y = xd.data_ptr(), wp.data_ptr(), yd.data_ptr()
    a.x_bf16, a.y_bf16, a.w_split = 1, 1, 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = 1, 48, 48, 64, 64
    a.KH, a.KW, a.stride, a.pad = 3, 3, 1, 1
    a.Hout, a.Wout, a.Ntot, a.ldy = 48, 48, 64, 64
    a.latency_form = 1
    assert lib.dfl_conv_config(C.addressof(a)) == 16 + 39
    part = torch.zeros(4096, 2, 64, device=DEV)
    a.stat_partials = part.data_ptr()
    assert 16 <= lib.dfl_conv_config(C.addressof(a)) != 16 + 39   # statistics: the patch kernels (any of their tile configurations)
    a.stat_partials = None
    a.Hin = a.Win = a.Hout = a.Wout = 96
    a.N = 16
    assert 16 <= lib.dfl_conv_config(C.addressof(a)) != 16 + 39   # 10.9 GFLOP: a throughput problem


def test_inference_forward_with_and_without_the_latency_form():
    """The eval-mode forward of the paper network at 192x192, batch 1 (the shape of the per-image loops): the plan's hint against
    DFL_CONVS=0-style plans (hint cleared), output for output."""
    import bench
    torch.manual_seed(5)
    net = dfl_amd.UNet(**bench.PAPER).to(DEV).eval()
    x = torch.randn(1, 1, 192, 192, device=DEV)
    with torch.no_grad():
        seg1, heat1 = net(x)
        plan = [p for ps in net._plans.values() for p in ps if not p.need_grad][0]
        convs = [st for st in plan.fwd.structs if isinstance(st, nat.ConvArgs)] + [st for st in plan.fwd.keep if isinstance(st, nat.ConvArgs)]
        lean = [st for st in convs if st.latency_form]
        assert len(lean) >= 40
        lib = nat.lib()
        taken = sum(1 for st in lean if lib.dfl_conv_config(C.addressof(st)) == 16 + 39)
        assert taken >= 42, 'latency form taken by %d of %d convolutions' % (taken, len(lean))
        assert sum(1 for st in plan.fwd.structs if isinstance(st, nat.ConvPairArgs)) >= 11
        seg1, heat1 = seg1.clone(), heat1.clone()
        # the same plan with the hint cleared (K slices re-planned by the library for the patch kernels)
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
        plan2 = [p for ps in net2._plans.values() for p in ps if not p.need_grad][0]
        assert not any(st.latency_form for st in plan2.fwd.structs if isinstance(st, nat.ConvArgs)) and not any(isinstance(st, nat.ConvPairArgs) for st in plan2.fwd.structs)
    # two bf16-storage forwards that differ in fp32 summation order: a few last-bit roundings per tensor, amplified by 44 layers
    assert float((seg1 - seg0).abs().max()) < 3e-2
    assert float((heat1 - heat0).abs().max()) <= 3e-2 * max(1.0, float(heat0.abs().max()))
    agree = float((seg1.argmax(1) == seg0.argmax(1)).float().mean())
    assert agree > 0.995, 'label agreement %.4f' % agree


def test_ensemble_forwards_on_a_stream_per_net_are_the_serial_ones():
    """util.forward_nets: the nets of an ensemble on one small image run on a stream each (util.py:326-330 calls them one after the
    other): the same bits as the serial calls, replay after replay; large inputs stay on the caller's stream."""
    import bench
    from dfl_amd import util
    nets = []
    for i in range(3):
        torch.manual_seed(20 + i)
        nets.append(dfl_amd.UNet(**bench.PAPER).to(DEV).eval())
    with torch.no_grad():
        for size in (192, 96):
            x = torch.randn(1, 1, size, size, device=DEV)
            serial = [n(x) for n in nets]
            for rep in range(3):
                outs = util.forward_nets(nets, x, 14)
                labels, heats, _ = util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (size - 8, size - 8))
                torch.cuda.synchronize()
                for (s0, h0), (s1, h1) in zip(serial, outs):
                    assert torch.equal(s0, s1) and torch.equal(h0, h1)
            ref = util.ensemble_reduce([o[0] for o in serial], [o[1] for o in serial], (size - 8, size - 8))
            assert torch.equal(ref[0], labels) and torch.equal(ref[1], heats)
        assert util.ENSEMBLE_STREAMS_MAX_PIXELS < 1440 * 1440
