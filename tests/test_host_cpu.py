"""CPU-side checks of the product's host layer (no kernel is launched): the C ABI loads and exports what the header
declares, the struct mirrors match, the drop-in module reproduces the reference's state_dict / seeded init / schedule,
plans build for every preset, and the product refuses to run without a GPU instead of falling back."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest
import torch

import dfl_amd
from conftest import TINY_CFGS, PAPER_CFGS, load_golden, ROOT
from dfl_amd import _native as nat
from dfl_amd.plan import UNetPlan


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'dfl_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(dfl_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 30
    lib = C.CDLL(nat.LIB_PATH)
    missing = [f for f in sorted(declared) if not hasattr(lib, f)]
    assert not missing, missing
    assert set(nat.EXPORTS) <= declared
    L = nat.lib()                              # also verifies every ctypes struct mirror against dfl_sizeof()
    assert L.dfl_version() >= 100
    assert L.dfl_sizeof(999) == -1


def test_argument_validation_reports_errors_without_a_gpu():
    L = nat.lib()
    a = nat.ConvArgs()                         # all NULL
    assert L.dfl_conv2d(C.addressof(a), None) == -1
    assert b'required' in L.dfl_last_error()
    a = nat.ConvArgs(x=16, w=16, y=16, N=1, Hin=8, Win=8, Cin=4, ldx=4, KH=3, KW=3, stride=1, pad=1, Hout=9, Wout=8,
                     Ntot=8, ldy=8)
    assert L.dfl_conv_grid_m(C.addressof(a)) == -1 and b'do not match' in L.dfl_last_error()
    assert L.dfl_rowblock_count(5000, 32) == 20
    assert L.dfl_head_scratch_ld(32) % 4 == 0


def test_cpu_tensors_are_refused():
    net = dfl_amd.UNet(n_classes=7, depth=2, wf=2, padding=True, batch_norm=True)
    with pytest.raises(RuntimeError, match='GPU only'):
        net(torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError):
        dfl_amd.DiceLoss2D()(torch.rand(1, 2, 4, 4), torch.rand(1, 2, 4, 4))
    with pytest.raises(RuntimeError):
        dfl_amd.get_device(no_gpu=True)
    with pytest.raises(NotImplementedError):
        dfl_amd.UNet(pad_mode='reflect')              # torch accepts it; the reference's flag values are 'zeros' and 'circular'
    with pytest.raises(AssertionError):
        dfl_amd.UNet(up_mode='nearest')               # unet.py:72: assert up_mode in ('upconv', 'upsample')


@pytest.mark.parametrize('flags', [dict(up_mode='upsample', padding=True), dict(pad_mode='circular', padding=True),
                                   dict(padding=False, do_res=False, num_lands=6, lands_block_depth=2),
                                   dict(padding=True, num_lands=6, lands_block_depth=1, lands_num_1x1=3, up_mode='upsample', pad_mode='circular')])
def test_every_constructor_flag_builds_the_reference_module_tree(flags):
    """The constructor values no reference CLI selects (unet.py:41-45: up_mode='upsample', pad_mode='circular', lands_block_depth
    with and without padding): same state_dict keys, shapes and seeded initialisation as the oracle's module tree, which
    tests/test_oracle_golden.py pins to the reference."""
    from oracle import ref_cpu as R
    cfg = dict(n_classes=5, depth=3, wf=3, batch_norm=True, max_pool=False)
    cfg.update(flags)
    torch.manual_seed(11)
    a = dfl_amd.UNet(1, **cfg).state_dict()
    torch.manual_seed(11)
    b = R.OracleUNet(1, **cfg).state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    # ... and the programs record for them (addresses only, nothing is launched), in the fp32 and the bf16 storage arithmetic
    L = nat.lib()
    prev = L.dfl_get_math_mode()
    try:
        for mode, wf in ((0, 3), (4, 5)):      # (bf16 storage: channel counts, F/2 of the landmark block included, are multiples of 16)
            nat.check(L.dfl_set_math_mode(mode), 'dfl_set_math_mode')
            net = dfl_amd.UNet(1, **dict(cfg, wf=wf))
            P, B = net._state()
            plan = UNetPlan(net._cfg, P, B, 2, 64, 64, True, True, torch.device('cpu'))
            seg, heat = plan.new_outputs()
            with torch.no_grad():
                o = R.OracleUNet(1, **dict(cfg, wf=wf))(torch.zeros(2, 1, 64, 64))
            oseg, oheat = (o if isinstance(o, tuple) else (o, None))
            assert tuple(seg.shape) == tuple(oseg.shape) and (heat is None) == (oheat is None)
            assert heat is None or tuple(heat.shape) == tuple(oheat.shape)
            live = [k for k in plan.grad_names if k not in plan.dead_params]
            assert set(plan.grad_ready_op) == set(live)
    finally:
        nat.check(L.dfl_set_math_mode(prev), 'dfl_set_math_mode')


@pytest.mark.parametrize('name', sorted(PAPER_CFGS))
def test_seeded_init_and_state_dict_match_reference(name):
    seed, cfg = PAPER_CFGS[name]
    g = load_golden(name)
    torch.manual_seed(seed)
    net = dfl_amd.UNet(**cfg)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g['sd_names'])
    sha = lambda t: hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()
    assert [sha(v) for v in sd.values()] == list(g['sd_sha'])
    assert [k for k, _ in net.named_parameters()] == list(g['param_names'])


@pytest.mark.parametrize('name', sorted(TINY_CFGS))
def test_plans_build_for_every_preset(name):
    cfg = TINY_CFGS[name]
    g = load_golden(name)
    net = dfl_amd.UNet(**cfg)
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd0/')})
    P, B = net._state()
    N, _, H, W = g['x'].shape
    plan = UNetPlan(net._cfg, P, B, N, H, W, True, True, torch.device('cpu'))   # addresses recorded, nothing launched
    assert plan.out_hw == tuple(g['seg'].shape[-2:])
    assert len(plan.fwd) > 0 and len(plan.bwd) > len(plan.fwd)
    live = [k for k in plan.grad_names if k not in plan.dead_params]
    assert set(plan.grad_ready_op) == set(live)
    ev = UNetPlan(net._cfg, P, B, 1, H, W, False, False, torch.device('cpu'))
    assert ev.bwd is None and len(ev.fwd) > 0


def test_center_crop_and_schedule_against_reference_goldens():
    x = torch.arange(2 * 3 * 9 * 8).view(2, 3, 9, 8)
    c = dfl_amd.center_crop(x, (5, 4))
    assert c.shape == (2, 3, 5, 4) and c.data_ptr() == x[:, :, 2:, 2:].data_ptr()   # a view, start = int(diff/2)
    assert dfl_amd.center_crop(x, x.shape) is x
    g = load_golden('sched')
    import contextlib
    import io
    for tag, (period, growth) in {'p2g2': (2, 2), 'p3g1': (3, 1)}.items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.1)
        with contextlib.redirect_stdout(io.StringIO()):
            s = dfl_amd.WarmRestartLR(opt, init_run_period_epochs=period, growth_factor=growth)
            trace, restarts = [], []
            for ep in range(9):
                for k in range(4):
                    s.intra_epoch_step((k + 1) / 4)
                    trace.append(opt.param_groups[0]['lr'])
                s.step()
                trace.append(opt.param_groups[0]['lr'])
                restarts.append(int(s.just_restarted))
        np.testing.assert_allclose(trace, g[tag], rtol=1e-12, atol=1e-15)
        assert restarts == list(g[tag + '_restarts'])


def test_dataset_refuses_cpu_and_pad_arithmetic():
    from dfl_amd import dataset as D
    g = load_golden('dataset')
    assert D.calc_pad_amount(48, 46) == int(g['pad_48_46'])
    assert D.calc_pad_amount(192, 184) == int(g['pad_192_184'])
    assert D.calc_pad_amount(193, 180) == int(g['pad_193_180'])
    with pytest.raises(nat.DflError):
        D.DeviceDataSet(torch.zeros(1, 1, 8, 8), torch.zeros(1, 8, 8, dtype=torch.uint8), num_classes=2, device='cpu')


def test_module_copies_and_pickles_without_its_recorded_programs():
    """deepcopy / torch.save(net) work (plans hold ctypes structures and raw addresses: they are dropped and rebuilt) and
    every copy keeps its parameters in one arena that knows its owner (sgd.SGD's pre-pack hook)."""
    import copy
    import io
    from dfl_amd.unet import owner_of
    net = dfl_amd.UNet(1, n_classes=3, depth=2, wf=2, padding=True, batch_norm=True, max_pool=False)
    P, B = net._state()
    from dfl_amd.plan import UNetPlan
    net._plans[('dummy',)] = [UNetPlan(net._cfg, P, B, 1, 16, 16, True, True, torch.device('cpu'))]
    n2 = copy.deepcopy(net)
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    n3 = torch.load(buf, weights_only=False)
    for m in (n2, n3):
        assert m._plans == {} and m._param_flat is not None
        assert owner_of(next(m.parameters())) is m
        for a, b in zip(net.state_dict().values(), m.state_dict().values()):
            assert torch.equal(a, b)
        base = m._param_flat.data_ptr()
        assert all(base <= p.data_ptr() < base + 4 * m._param_flat.numel() for p in m.parameters())
    assert owner_of(next(net.parameters())) is net


def test_late_scalars_deliver_every_value_in_order():
    """util.LateScalars on CPU tensors degenerates to .item(); the queue logic (depth, flush) is device independent."""
    from dfl_amd.util import LateScalars
    late = LateScalars(depth=1)
    got = [late.push(torch.tensor(float(i))) for i in range(4)]
    assert got == [0.0, 1.0, 2.0, 3.0] and late.flush() == []          # CPU tensors are read at once
    sync = LateScalars(depth=0)
    assert sync.push(torch.tensor(2.5)) == 2.5 and sync.flush() == []


def test_center_crop_keeps_the_plain_view_off_the_training_path():
    """Tensors that need no gradient (targets, inference outputs) and CPU tensors get the reference's plain slice; the
    autograd-aware window (util._Crop) is for GPU tensors that require a gradient."""
    t = torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).view(2, 3, 8, 8)
    c = dfl_amd.center_crop(t, (2, 3, 4, 6))
    assert c.shape == (2, 3, 4, 6) and c._base is not None and c.data_ptr() == t[..., 2:, 1:].data_ptr()      # a view, no copy
    t.requires_grad_(True)
    c = dfl_amd.center_crop(t * 1.0, (4, 4))                              # CPU: generic autograd slice
    c.sum().backward()
    assert float(t.grad.sum()) == 2 * 3 * 16
    # the generic branch of the custom backward (a gradient that is not the loss kernels' zero-bordered window)
    from dfl_amd.util import _Crop
    x = torch.randn(1, 2, 6, 6, requires_grad=True)
    y = _Crop.apply(x * 1.0, 1, 2, 3, 3)
    (y * torch.arange(9.0).view(3, 3)).sum().backward()
    want = torch.zeros(1, 2, 6, 6)
    want[..., 1:4, 2:5] = torch.arange(9.0).view(3, 3)
    assert torch.equal(x.grad, want)


def test_loss_stage_and_stride_arguments_are_validated():
    L = nat.lib()
    a = nat.LossArgs()
    a.loss = a.sums = a.seg = a.tseg = 4096
    a.B, a.C, a.h, a.w = 1, 2, 4, 4
    a.stage = 3
    assert L.dfl_dice_ncc_loss(C.addressof(a), None) < 0 and b'stage' in L.dfl_last_error()
    a.stage = 2
    a.dseg_sN = 64                                                        # a stride without its partners
    assert L.dfl_dice_ncc_loss(C.addressof(a), None) < 0 and b'strides' in L.dfl_last_error()


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: only tests/, tools/ (fixture generation), __graft_entry__.smoke() and bench.py's cpu_baseline
    leg may import it -- never the package or the host scripts (a product path that routes through the CPU restatement would void
    every parity claim)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r'^\s*(from\s+oracle\b|import\s+oracle\b)', re.M)
    product = [os.path.join(root, f) for f in ('train.py', 'test_ensemble.py', 'est_lands_csv.py', 'compute_actual_dice_on_test.py', 'dfl_amd.py')]
    pkg = os.path.join(root, 'deepfluorolabeling-ipcai2020_amd')
    for d, _, files in os.walk(pkg):
        product += [os.path.join(d, f) for f in files if f.endswith('.py')]
    for f in product:
        if os.path.exists(f):
            assert not pat.search(open(f).read()), '%s imports the oracle' % f
    # the two allowed root files use it in exactly one function each
    for f, fn in (('bench.py', 'cpu_baseline'), ('__graft_entry__.py', 'smoke')):
        src = open(os.path.join(root, f)).read()
        hits = [m.start() for m in pat.finditer(src)]
        assert len(hits) == 1, (f, len(hits))
        head = src[:hits[0]]
        assert head.rfind('def ') == head.rfind('def ' + fn), '%s: the oracle import is not inside %s()' % (f, fn)


def test_missing_library_fails_loudly():
    """No fallback: without libdfl_hip.so the first call into the library raises (a subprocess, so that this process keeps its
    loaded library)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import dfl_amd\n"
            "from dfl_amd import _native as nat\n"
            "try:\n"
            "    nat.lib()\n"
            "except nat.DflError as e:\n"
            "    print('RAISED', e)\n" % root)
    env = dict(os.environ, DFL_LIB_OVERRIDE='/nonexistent/libdfl_hip.so')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert 'RAISED' in out.stdout and 'no fallback' in out.stdout, (out.stdout[-400:], out.stderr[-400:])


def test_integration_doc_binding_matches_the_header_mirror():
    """INTEGRATION.md shows the ctypes mirror of dfl_conv_args a maintainer of the reference would write: it has to be the
    struct include/dfl_hip.h declares (same fields, same order, same size as dfl_amd._native.ConvArgs, whose size the library's
    dfl_sizeof pins on the GPU box)."""
    import ctypes as C
    from dfl_amd import _native as nat
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    i = text.index('class ConvArgs(C.Structure)')
    j = text.index('lib = C.CDLL', i)
    ns = {'C': C}
    exec(text[i:j], ns)
    doc = ns['ConvArgs']
    assert [f[0] for f in doc._fields_] == [f[0] for f in nat.ConvArgs._fields_]
    assert C.sizeof(doc) == C.sizeof(nat.ConvArgs)
