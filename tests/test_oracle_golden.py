"""Pins oracle/ref_cpu.py against fixtures produced by the reference itself (tools/gen_golden.py)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import PAPER_BATCH, TINY_CFGS, PAPER_CFGS, load_golden
from oracle import ref_cpu as R


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _load_net(g, cfg, prefix='sd0/'):
    net = R.OracleUNet(**cfg)
    sd = {k[len(prefix):]: _t(v) for k, v in g.items() if k.startswith(prefix)}
    assert list(sd.keys()) == list(net.state_dict().keys())       # names AND order (checkpoint compat)
    net.load_state_dict(sd)
    return net


@pytest.mark.parametrize('name', sorted(TINY_CFGS))
def test_tiny_forward_backward(name):
    cfg = TINY_CFGS[name]
    g = load_golden(name)
    net = _load_net(g, cfg)
    net.train()
    x = _t(g['x'])
    taps = {}
    out = net(x, taps)
    nl = cfg['num_lands']
    seg = out[0] if nl > 0 else out
    assert seg.shape == g['seg'].shape
    np.testing.assert_allclose(seg.detach().numpy(), g['seg'], rtol=1e-5, atol=1e-6)
    for k in [k for k in g if k.startswith('act/')]:
        np.testing.assert_allclose(taps[k[4:]].detach().numpy(), g[k], rtol=1e-5, atol=1e-5)
    tseg = _t(g['tseg'])
    if nl > 0:
        np.testing.assert_allclose(out[1].detach().numpy(), g['heat'], rtol=1e-5, atol=1e-6)
        theat = _t(g['theat'])
        loss = R.dice_and_heatmap_loss_2d((R.center_crop(seg, tseg.shape), R.center_crop(out[1], theat.shape)),
                                          (tseg, theat), skip_bg=False, heatmap_wgt=0.5)
    else:
        loss = R.dice_loss_2d(R.center_crop(seg, tseg.shape), tseg, skip_bg=False)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    has_grads = any(k.startswith('grad/') for k in g)
    if has_grads:
        loss.backward()
        for k, p in net.named_parameters():
            ref = g['grad/' + k]
            if ref.size == 0:
                assert p.grad is None          # dead downsample conv
                continue
            np.testing.assert_allclose(p.grad.numpy(), ref, rtol=2e-4, atol=2e-6)
    for k in [k for k in g if k.startswith('sd1/')]:
        np.testing.assert_allclose(net.state_dict()[k[4:]].numpy(), g[k], rtol=1e-5, atol=1e-6)
    net.eval()
    with torch.no_grad():
        oe = net(x)
    np.testing.assert_allclose((oe[0] if nl > 0 else oe).numpy(), g['seg_eval'], rtol=1e-5, atol=1e-6)
    if nl > 0:
        np.testing.assert_allclose(oe[1].numpy(), g['heat_eval'], rtol=1e-5, atol=1e-6)


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize('name', sorted(PAPER_CFGS))
def test_paper_init_and_forward(name):
    seed, cfg = PAPER_CFGS[name]
    g = load_golden(name)
    torch.manual_seed(seed)
    net = R.OracleUNet(**cfg)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g['sd_names'])
    assert [_sha(v) for v in sd.values()] == list(g['sd_sha'])       # identical seeded init
    B = PAPER_BATCH.get(name, 2)
    gen = torch.Generator().manual_seed(seed + (1 if B == 2 else B))
    x = torch.randn(B, 1, 192, 192, generator=gen)
    assert _sha(x) == str(g['x_sha'])
    lab = torch.randint(0, 7, (B, 184, 184), generator=gen)
    assert np.array_equal(lab.numpy().astype(np.uint8), g['lab'])
    tseg = R.one_hot_masks(lab, 7)
    theat = torch.rand(B, 14, 184, 184, generator=gen) * 0.02
    net.train()
    out = net(x)
    nl = cfg['num_lands']
    seg = out[0] if nl > 0 else out
    np.testing.assert_allclose(seg[:, :, ::16, ::16].detach().numpy(), g['seg_s16'], rtol=1e-4, atol=1e-6)
    if nl > 0:
        np.testing.assert_allclose(out[1][:, :, ::16, ::16].detach().numpy(), g['heat_s16'], rtol=1e-4, atol=1e-5)
        loss = R.dice_and_heatmap_loss_2d((R.center_crop(seg, tseg.shape), R.center_crop(out[1], theat.shape)),
                                          (tseg, theat), skip_bg=False)
    else:
        loss = R.dice_loss_2d(R.center_crop(seg, tseg.shape), tseg, skip_bg=False)
    assert abs(loss.item() - float(g['loss32'])) < 2e-6
    # bit-exact labels wherever the fp64 reference's top-2 margin is not tiny (hard parts)
    am = torch.max(seg, dim=1)[1].numpy().astype(np.uint8)
    close = np.unpackbits(g['margin_lt_1e5'])[:am.size].reshape(am.shape).astype(bool)
    assert np.array_equal(am[~close], g['argmax64'][~close])


def test_losses_known_answers():
    g = load_golden('losses')
    s, t = _t(g['dice_in']).requires_grad_(True), _t(g['dice_tgt'])
    for sb in (True, False):
        s.grad = None
        l = R.dice_loss_2d(s, t, skip_bg=sb)
        l.backward()
        assert abs(l.item() - float(g['dice_sb%d' % int(sb)])) < 1e-12
        np.testing.assert_allclose(s.grad.numpy(), g['dice_sb%d_grad' % int(sb)], rtol=1e-9, atol=1e-15)
    assert abs(R.dice_loss_2d(t, t, skip_bg=False).item() - float(g['dice_perfect'])) < 1e-12
    X, Y = _t(g['ncc_x']).requires_grad_(True), _t(g['ncc_y'])
    n = R.ncc_2d(X, Y)
    n.sum().backward()
    np.testing.assert_allclose(n.detach().numpy(), g['ncc'], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(X.grad.numpy(), g['ncc_grad_of_sum'], rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(R.ncc_2d(Y, Y).numpy(), g['ncc_self'], rtol=1e-12)
    N = Y.shape[-1] * Y.shape[-2]
    np.testing.assert_allclose(g['ncc_self'], (N - 1) / N, rtol=1e-5)      # SURVEY section 4 known answer (the 1e-8 in the denominator shows at this scale)
    X.grad = None
    s.grad = None
    l = R.dice_and_heatmap_loss_2d((s, X), (t, Y), skip_bg=False, heatmap_wgt=0.3)
    l.backward()
    assert abs(l.item() - float(g['dh_loss'])) < 1e-12
    np.testing.assert_allclose(s.grad.numpy(), g['dh_gseg'], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(X.grad.numpy(), g['dh_gheat'], rtol=1e-8, atol=1e-15)


def test_sched_trace():
    g = load_golden('sched')
    np.testing.assert_allclose(R.warm_restart_lr_trace(0.1, 2, 2, 0.0, 9, 4), g['p2g2'], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(R.warm_restart_lr_trace(0.1, 3, 1, 0.0, 9, 4), g['p3g1'], rtol=1e-12, atol=1e-15)


def test_dataset_items():
    g = load_golden('dataset')
    assert R.calc_pad_amount(48, 46) == int(g['pad_48_46'])
    assert R.calc_pad_amount(192, 184) == int(g['pad_192_184']) == 4
    assert R.calc_pad_amount(193, 180) == int(g['pad_193_180'])
    projs, segs, lands = _t(g['projs']), _t(g['segs']), _t(g['lands'])
    H, W = projs.shape[-2:]
    lands_m = R.mark_oob_landmarks(lands, H, W)
    masks = R.one_hot_masks(segs, 7)
    pad = R.calc_pad_amount(48, W)
    for i in range(3):
        p = R.preprocess_proj(projs[i:i + 1], pad)
        np.testing.assert_allclose(p.numpy(), g['item%d_p' % i], rtol=1e-6, atol=1e-6)
        assert np.array_equal(masks[i].numpy(), g['item%d_s' % i])
        assert np.array_equal(lands_m[i].numpy(), g['item%d_l' % i])
        h = R.gaussian_heatmaps(lands_m[i], H, W)
        np.testing.assert_allclose(h.numpy(), g['item%d_h' % i], rtol=1e-6, atol=1e-9)
        assert g['item%d_h' % i][13].max() == 0.0                  # out-of-bounds landmark -> zero map


def test_ensemble():
    g = load_golden('ensemble')
    cfg = dict(n_classes=7, depth=3, wf=2, batch_norm=True, padding=True, max_pool=False, num_lands=14,
               do_res=True, block_depth=2)
    nets = []
    for i in range(3):
        n = _load_net(g, cfg, prefix='net%d/' % i)
        n.eval()
        nets.append(n)
    imgs = _t(g['imgs'])
    with torch.no_grad():
        for j in range(imgs.shape[0]):
            outs = [n(imgs[j:j + 1]) for n in nets]
            labels, heats, _ = R.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (28, 28))
            assert np.array_equal(labels[0].numpy(), g['nn_segs'][j])
            np.testing.assert_allclose(heats[0].numpy(), g['nn_heats'][j], rtol=1e-5, atol=1e-6)


def test_trajectory():
    g = load_golden('trajectory')
    cfg = dict(n_classes=7, depth=3, wf=3, batch_norm=True, padding=True, max_pool=False, num_lands=14,
               do_res=True, block_depth=2)
    net = _load_net(g, cfg)
    projs, segs, lands = _t(g['projs']), _t(g['segs']), _t(g['lands'])
    H, W = projs.shape[-2:]
    lm = R.mark_oob_landmarks(lands, H, W)
    pad = R.calc_pad_amount(48, W)
    P = torch.stack([R.preprocess_proj(projs[i:i + 1], pad) for i in range(8)])
    S = R.one_hot_masks(segs, 7)
    Hm = torch.stack([R.gaussian_heatmaps(lm[i], H, W) for i in range(8)]).view(8, 14, H, W)
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    net.train()
    losses = []
    for step in range(30):
        idx = [(step * 4 + j) % 8 for j in range(4)]
        losses.append(R.train_step(net, opt, P[idx], S[idx], Hm[idx], 0.5))
    np.testing.assert_allclose(losses[:10], g['losses'][:10], rtol=0, atol=2e-5)
    np.testing.assert_allclose(losses, g['losses'], rtol=0, atol=5e-3)
    net.eval()
    with torch.no_grad():
        out = net(P)
    labels = torch.max(R.center_crop(out[0], S.shape), dim=1)[1]
    d = R.hard_dice(labels, segs.long(), 7)
    np.testing.assert_allclose(d, g['hard_dice'], atol=0.02)


def test_landmark_extraction():
    """oracle.est_landmarks == the reference script est_lands_csv.py (run by tools/gen_golden.py), with and without mask."""
    g = load_golden('est_lands')
    heats, segs = _t(g['heats']), _t(g['segs'])
    labels = [int(v) for v in g['label_for_land']]
    rc, ncc = R.est_landmarks(heats, segs, labels, return_ncc=True)
    assert np.array_equal(rc.numpy(), g['rc_masked'])
    rc2, ncc2 = R.est_landmarks(heats, None, None, return_ncc=True)
    assert np.array_equal(rc2.numpy(), g['rc_plain'])
    found = g['rc_plain'][..., 0] >= 0
    assert found.any() and (~found).any()                         # both outcomes are covered ...
    assert float((ncc2 - 0.9).abs().min()) > 2e-4                 # ... and none sits on the 0.9 threshold (fp32 noise ~1e-6)
    assert (g['rc_masked'] != g['rc_plain']).any()                # the mask changes some answers


VAL_BASE = dict(n_classes=7, depth=3, wf=2, batch_norm=True, padding=True, max_pool=False, do_res=True, block_depth=2)


def validation_items(g):
    t = lambda a: torch.from_numpy(np.asarray(a))
    return [(t(g['x'][i]), t(g['masks'][i]), t(g['lands'][i]), t(g['heats'][i])) for i in range(g['x'].shape[0])]


def validation_nets(g, tag, num_lands, n, make):
    nets = []
    for i in range(n):
        net = make(**dict(VAL_BASE, num_lands=num_lands))
        pre = '%s_net%d/' % (tag, i)
        net.load_state_dict({k[len(pre):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(pre)})
        nets.append(net)
    return nets


def test_validation_loops():
    """util.test_dataset / util.test_dataset_ensemble (util.py:116-241) of the reference, run by tools/gen_golden.py: the
    oracle's restatement returns the same (mean, std) -- with and without landmarks, both dice_only values."""
    g = load_golden('validation')
    items = validation_items(g)
    n14 = validation_nets(g, 'l14', 14, 3, R.OracleUNet)
    n0 = validation_nets(g, 'l0', 0, 2, R.OracleUNet)
    got = {'single_l14': R.validation_loss(n14[0], items, 14), 'single_l0': R.validation_loss(n0[0], items, 0),
           'ens_l14': R.validation_loss_ensemble(n14, items, 14), 'ens_l14_dice_only': R.validation_loss_ensemble(n14, items, 14, dice_only=True),
           'ens_l0': R.validation_loss_ensemble(n0, items, 0)}
    for k, (m, s) in got.items():
        np.testing.assert_allclose([float(m), float(s)], g['result/' + k], rtol=0, atol=2e-6, err_msg=k)
