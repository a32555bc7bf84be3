"""North-star parity tests -- collected FIRST (file name) so that a time limit or an unrelated failure further down cannot
hide them: the paper presets against the reference's own fixtures and the fp64 oracle, BASELINE configs[1]'s
batch-16 step in every arithmetic mode bench.py quotes (fp32, bf16x3 and the bf16 STORAGE mode it times; files
test_gpu_00a_paper_batch16.py / test_gpu_00b_paper_batch5.py), the 30-step
trajectory against the reference's run, the ensemble loop and the validation loops against the reference's outputs, and
hard Dice at a training plateau within +-0.005 of the reference (north_star's bar).  pytest -m gpu.

Tolerances: forward 1e-4 relative in the fp32-tensor modes; labels bit-exact outside the pixels whose fp64 top-2 margin is
at rounding level; gradients against the fp64 oracle ON THE HIP RUN'S ACTIVATION PATTERN inside bars made of sensitivities
measured offline and committed (tests/noise_floor.py, tests/golden/floors/)."""
import hashlib
import os

import numpy as np
import pytest
import torch

import dfl_amd
from dfl_amd import _native as nat
from conftest import PAPER_CFGS, PAPER_BATCH, paper_key, load_golden, by_mode
from oracle import ref_cpu as R
import noise_floor as NF
import problems as PR
from gpu_common import DEV, _t, oracle64, load_net, hip_net, hip_step, label_mask, rel_close, math_mode_set

pytestmark = pytest.mark.gpu


def _oracle_threads():
    torch.set_num_threads(max(torch.get_num_threads(), min(64, os.cpu_count() or 32)))


@pytest.mark.parametrize('name', sorted(PAPER_CFGS))
def test_paper_golden(name, math_mode):
    """Paper preset (depth 6, wf 5): seeded init reproduces the reference bit for bit, forward within 1e-4 of the
    reference's fp32 run and of its fp64 run, labels identical outside the tiny-margin pixels, gradients against the fp64
    oracle on the run's pattern and against the numbers of the reference's own fp64 run."""
    seed, cfg = PAPER_CFGS[name]
    g = load_golden(name)
    torch.manual_seed(seed)
    net = dfl_amd.UNet(**cfg)
    sha = lambda t: hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()
    assert list(net.state_dict().keys()) == list(g['sd_names'])
    assert [sha(v) for v in net.state_dict().values()] == list(g['sd_sha'])
    _oracle_threads()
    gc = NF.cached_check(paper_key(name), lambda: PR.paper(name, PAPER_BATCH.get(name, 2)))
    pr = gc.problem
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), pr.sd.values()))   # the problem IS this seeded net
    net = net.to(DEV).train()
    out, seg, loss = hip_step(pr, net)
    nl = cfg['num_lands']
    s16 = seg[:, :, ::16, ::16].detach().cpu().numpy()
    np.testing.assert_allclose(s16, g['seg_s16'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(s16, g['seg64_s16'], rtol=1e-4, atol=1e-6)
    if nl > 0:
        rel_close(out[1][:, :, ::16, ::16].detach().cpu().numpy(), g['heat64_s16'], 1e-4, 'heat maps vs fp64 reference')
    assert abs(loss.item() - float(g['loss64'])) < 5e-6
    am = torch.max(seg, dim=1)[1].cpu().numpy().astype(np.uint8)
    close = np.unpackbits(g['margin_lt_1e5'])[:am.size].reshape(am.shape).astype(bool)
    assert np.array_equal(am[~close], g['argmax64'][~close])
    res = gc.check(net, seg, NF.conv_rel_error(math_mode), what=name + ' ')
    print('%s %s: conv noise %.2e, whole-gradient error %.3e, worst per-tensor error / bar %.2f, decisions forced %d ReLU %d pool' % (
        name, math_mode, res['eps_eff'], res['whole'], res['worst'], res['info']['relu_flips'], res['info']['pool_flips']))
    # ... and against the numbers of the REFERENCE's own fp64 run (tests/golden): per-tensor norms, small tensors in full.  The
    # reference ran on its own pattern; the oracle's distance between the two patterns is added to the bar of each tensor.
    names = list(g['param_names'])
    bars, ref = res['bars'], res['ref']
    for k, p in net.named_parameters():
        gn = float(g['gradnorm64'][names.index(k)])
        if gn < 0:
            assert p.grad is None, k
            continue
        n_ref = float(ref[k].norm())
        n_got = p.grad.double().norm().item()
        shift = abs(n_ref - gn)                       # pattern difference, measured on the oracle
        assert abs(n_got - gn) <= bars[k] * max(gn, 1e-12) + shift, '%s: grad norm %.6e vs fp64 reference %.6e (bar %.2e, pattern shift %.2e)' % (k, n_got, gn, bars[k], shift)
        gk = 'g64/' + k
        if gk in g:
            l2 = NF.rel_l2(p.grad.cpu().numpy(), g[gk])
            pshift = NF.rel_l2(ref[k].numpy(), g[gk])
            assert l2 <= bars[k] + pshift, '%s: relative L2 error %.3e vs the reference fp64 gradient (bar %.2e + pattern shift %.2e)' % (k, l2, bars[k], pshift)


@pytest.mark.parametrize('mode', ['fp32', 'bf16x3', 'bf16', 'bf16s'])
def test_plateau_dice_matches_reference(mode):
    """North-star quality bar: hard Dice within +-0.005 of the REFERENCE.  tests/golden/plateau.npz holds a run of the
    reference itself (tools/gen_golden.py: 400 SGD steps on 16 toy-ellipses images, learning rate cut 10x for the last
    100, train.py:405-430 wiring) -- twice, with 8 and 1 CPU threads, which shows the reference's own run-to-run spread
    at the plateau (mean Dice 0.9972 / 0.9955, single classes up to 0.007 apart).  The HIP path, same data, same steps:
    mean Dice of the training images within 0.005 of the reference's runs, every class within 0.005 + the reference's own
    spread on that class, and the plateau loss not more than 5e-3 above the reference's."""
    # (bf16s = math mode 4, bf16 STORAGE, needs >= 16 channels: its reference run is the 16..64-channel network of
    # tests/golden/plateau_wf4.npz, same data and schedule)
    g = load_golden('plateau_wf4' if mode == 'bf16s' else 'plateau')
    cfg = dict(n_classes=7, depth=3, wf=int(g['wf']) if 'wf' in g else 3, batch_norm=True, padding=True, max_pool=False,
               num_lands=14, do_res=True, block_depth=2)
    lib = nat.lib()
    prev = lib.dfl_get_math_mode()
    nat.check(lib.dfl_set_math_mode({'fp32': 0, 'bf16x3': 1, 'bf16': 3, 'bf16s': 4}[mode]), 'dfl_set_math_mode')
    try:
        net = load_net(g, cfg)
        projs, segs, lands = _t(g['projs']), _t(g['segs']), _t(g['lands'])
        n_train, steps = int(g['n_train']), int(g['steps'])
        H, W = projs.shape[-2:]
        lm = R.mark_oob_landmarks(lands, H, W)
        pad = R.calc_pad_amount(48, W)
        n = projs.shape[0]
        P = torch.stack([R.preprocess_proj(projs[i:i + 1], pad) for i in range(n)]).to(DEV)
        S = R.one_hot_masks(segs, 7).to(DEV)
        Hm = torch.stack([R.gaussian_heatmaps(lm[i], H, W) for i in range(n)]).view(n, 14, H, W).to(DEV)
        opt = dfl_amd.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
        crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
        net.train()
        losses = []
        for step in range(steps):
            if step == 300:
                for gr in opt.param_groups:
                    gr['lr'] = 0.005
            idx = [(step * 4 + j) % n_train for j in range(4)]
            opt.zero_grad()
            out = net(P[idx])
            loss = crit((dfl_amd.center_crop(out[0], S[idx].shape), dfl_amd.center_crop(out[1], Hm[idx].shape)), (S[idx], Hm[idx]))
            loss.backward()
            opt.step()
            losses.append(loss.item())
        net.eval()
        with torch.no_grad():
            out = net(P[:n_train])
        labels = torch.max(dfl_amd.center_crop(out[0], S[:n_train].shape), dim=1)[1].cpu()
        d = R.hard_dice(labels, segs[:n_train].long(), 7)
    finally:
        nat.check(lib.dfl_set_math_mode(prev), 'dfl_set_math_mode')
    ref8, ref1 = g['dice_train'], g['dice_train_1thread']
    runs = g['dice_train_runs']                        # the reference's own runs (8 / 1 / 4 / 2 CPU threads: four summation orders)
    assert runs.shape[0] >= 4
    lo, hi = float(runs.mean(1).min()), float(runs.mean(1).max())
    print('plateau %s: mean Dice %.4f (reference runs %s), per class %s' % (mode, float(np.mean(d)), np.round(runs.mean(1), 4), np.round(d, 4)))
    assert lo - 0.005 <= float(np.mean(d)) <= hi + 0.005, 'mean hard Dice %.4f vs the reference\'s runs %s' % (float(np.mean(d)), np.round(runs.mean(1), 4))
    # per class: 0.005 around the band the REFERENCE's own runs span on that class -- the same bar in every arithmetic.
    # (Round 5 gave the bf16 modes a flat 0.01 after one class of one trajectory had landed 0.0003 outside 0.005 around TWO reference
    # runs; four runs show what the width of that band is: class 1 of this fixture moves by 0.014 between the reference's own thread
    # counts, the other classes by 0.001-0.004.)
    for c in range(6):
        a, b = float(runs[:, c].min()), float(runs[:, c].max())
        assert a - 0.005 <= d[c] <= b + 0.005, 'class %d: hard Dice %.4f vs the reference\'s runs %s' % (c + 1, d[c], np.round(runs[:, c], 4))
    # plateau loss (mean of the last 20 steps; the trajectories are chaotic at this level: the reference's own two runs -- 8 / 1
    # CPU threads -- end 0.0016 apart for the wf = 3 fixture and 0.0042 for wf = 4, two builds of this library 0.003): not
    # more than 5e-3 above the worse of the reference's runs, and not implausibly far below the better one
    l_hip, l8, l1 = float(np.mean(losses[-20:])), float(g['losses'][-20:].mean()), float(g['losses_1thread'][-20:].mean())
    print('plateau %s: loss %.4f (reference %.4f / %.4f)' % (mode, l_hip, l8, l1))
    assert min(l8, l1) - 2e-2 <= l_hip <= max(l8, l1) + 5e-3, 'plateau loss %.4f vs reference %.4f / %.4f' % (l_hip, l8, l1)



def _paper_plateau_run(g, mode):
    """400 steps of the paper preset from the fixture's seeded weights in arithmetic `mode`; (training Dice per class, held-out
    Dice per class, mean loss of the last 20 steps)."""
    _, cfg = PAPER_CFGS['paper_sc_l14']
    with math_mode_set(mode):
        torch.manual_seed(int(g['seed']))
        net = dfl_amd.UNet(**cfg).to(DEV)
        projs, segs, lands = _t(g['projs']), _t(g['segs']), _t(g['lands'])
        n_train, steps, cut = int(g['n_train']), int(g['steps']), int(g['cut'])
        H, W = projs.shape[-2:]
        lm = R.mark_oob_landmarks(lands, H, W)
        pad = R.calc_pad_amount(192, W)
        n = projs.shape[0]
        P = torch.stack([R.preprocess_proj(projs[i:i + 1], pad) for i in range(n)]).to(DEV)
        S = R.one_hot_masks(segs, 7).to(DEV)
        Hm = torch.stack([R.gaussian_heatmaps(lm[i], H, W) for i in range(n)]).view(n, 14, H, W).to(DEV)
        opt = dfl_amd.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
        crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
        net.train()
        losses = []
        for step in range(steps):
            if step == cut:
                for gr in opt.param_groups:
                    gr['lr'] = 0.01
            idx = [(step * 4 + j) % n_train for j in range(4)]
            opt.zero_grad()
            out = net(P[idx])
            loss = crit((dfl_amd.center_crop(out[0], S[idx].shape), dfl_amd.center_crop(out[1], Hm[idx].shape)), (S[idx], Hm[idx]))
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        losses = [float(l) for l in losses]
        net.eval()
        with torch.no_grad():
            seg = torch.cat([net(P[i:i + 4])[0] for i in range(0, n, 4)])
        labels = torch.max(dfl_amd.center_crop(seg, S.shape), dim=1)[1].cpu()
    d = R.hard_dice(labels[:n_train], segs[:n_train].long(), 7)
    dv = R.hard_dice(labels[n_train:], segs[n_train:].long(), 7)
    return np.asarray(d, dtype=np.float64), np.asarray(dv, dtype=np.float64), float(np.mean(losses[-20:]))


@pytest.mark.parametrize('mode', ['fp32', 'bf16x3', 'bf16s'])
def test_paper_preset_plateau_dice_matches_reference(mode):
    """north_star's quality bar ON THE CONFIGURATION IT NAMES (missing #1): the paper preset -- depth 6, 32 ... 1024
    channels, BatchNorm, zero padding, strided convolutions, 14 landmarks (train_test_code/Readme.md:16) -- at the 8x-downsampled
    size (184 x 184 padded to 192), batch 4, SGD 0.1 / 0.9 / nesterov / 1e-4 with the learning rate cut 10x for the last quarter,
    the step body of train.py:405-430, scored by hard Dice per class (compute_actual_dice_on_test.py:63-93).
    tests/golden/plateau_paper.npz holds the REFERENCE's own four runs (8 / 1 / 4 / 2 CPU threads = four summation orders inside its
    convolutions; tools/gen_golden.py fixture_plateau_paper, --extend-plateau-runs): mean training Dice 0.9937 ... 0.9956, single
    classes up to 0.0065 apart.  The HIP path from the same seeded initial weights (their SHA-256 is pinned by test_paper_golden),
    same data, same 400 steps: mean Dice within +-0.005 of the reference's runs, every class within 0.005 of the band the reference's
    runs span on it -- in the two parity arithmetics and in the bf16 STORAGE arithmetic the headline is quoted in (1024-channel /
    6 x 6-pixel levels included), the same bar for all three.

    Training is chaotic, and a change in the ORDER of an fp32 sum is enough to pick another trajectory -- in the reference (the four
    runs above) as in this library (fourteen builds / switch settings that differ in nothing else ended between 0.9934 and 0.9954
    mean Dice in bf16 storage).  The bf16 storage mode runs two trajectories (the default build; the operand
    written by every data gradient + the head's statistics from colstats) and EVERY one is held to the bars above."""
    from dfl_amd import plan as P_
    g = load_golden('plateau_paper')
    ref8, ref1 = g['dice_train'], g['dice_train_1thread']
    refs = g['dice_train_runs']                        # the reference's own runs (8 / 1 / 4 / 2 CPU threads), tools/gen_golden.py --extend-plateau-runs
    assert refs.shape[0] >= 4
    lo, hi = float(refs.mean(1).min()), float(refs.mean(1).max())
    l8, l1 = float(g['losses'][-20:].mean()), float(g['losses_1thread'][-20:].mean())
    runs = []
    # bf16 storage: two trajectories -- the default build and the one with the operand written by the data gradient at every layer
    # and the head's statistics from colstats (round 5 ran four; the bar below no longer needs an average over them)
    settings = [(None, None)] if mode != 'bf16s' else [(None, None), (1 << 40, False)]
    for dpre, live_head in settings:
        prev = (P_.UNetPlan.DPRE_OUT_BYTES, P_.UNetPlan.LIVE_HEAD)
        if dpre is not None:
            P_.UNetPlan.DPRE_OUT_BYTES, P_.UNetPlan.LIVE_HEAD = dpre, live_head
        try:
            d, dv, l_hip = _paper_plateau_run(g, mode)
        finally:
            P_.UNetPlan.DPRE_OUT_BYTES, P_.UNetPlan.LIVE_HEAD = prev
        print('paper-preset plateau %s%s: mean training Dice %.4f (reference runs %s), per class %s; held-out %.4f (reference %.4f / %.4f); '
              'loss %.4f (reference %.4f / %.4f)' % (mode, '' if dpre is None else ' [operand written: %s, head sums: %s]' % (bool(dpre), live_head),
                                                     float(d.mean()), np.round(refs.mean(1), 4), np.round(d, 4), float(dv.mean()),
                                                     g['dice_valid'].mean(), g['dice_valid_1thread'].mean(), l_hip, l8, l1))
        assert lo - 0.005 <= float(d.mean()) <= hi + 0.005, 'mean hard Dice %.4f vs the reference\'s runs %s' % (float(d.mean()), np.round(refs.mean(1), 4))
        runs.append((d, l_hip))
    # per class: EVERY trajectory of EVERY arithmetic within 0.005 of the band the reference's own four runs span on
    # that class (its spread is 0.001-0.0065 here) -- no wider bar for the bf16 modes, no averaging over trajectories
    for c in range(6):
        a, b = float(refs[:, c].min()), float(refs[:, c].max())
        for d, _ in runs:
            assert a - 0.005 <= d[c] <= b + 0.005, 'class %d: hard Dice %.4f vs the reference\'s runs %s' % (c + 1, d[c], np.round(refs[:, c], 4))
    # plateau loss (mean of the last 20 steps): not more than 5e-3 (bf16 storage: 1e-2) above the worse of the reference's runs and not
    # implausibly far below the better one
    for _, l in runs:
        assert min(l8, l1) - 2e-2 <= l <= max(l8, l1) + (1e-2 if mode == 'bf16s' else 5e-3), 'plateau loss %.4f vs reference %.4f / %.4f' % (l, l8, l1)


@pytest.mark.parametrize('optimizer', ['torch', 'dfl'])
def test_training_trajectory_matches_reference(optimizer, math_mode):
    """30 SGD steps wired as train.py:405-430 on the toy-ellipses set: per-step loss vs the reference's run, with
    torch.optim.SGD and with the one-launch dfl_amd.SGD."""
    g = load_golden('trajectory')
    cfg = dict(n_classes=7, depth=3, wf=3, batch_norm=True, padding=True, max_pool=False, num_lands=14, do_res=True,
               block_depth=2)
    net = load_net(g, cfg)
    projs, segs, lands = _t(g['projs']), _t(g['segs']), _t(g['lands'])
    H, W = projs.shape[-2:]
    lm = R.mark_oob_landmarks(lands, H, W)
    pad = R.calc_pad_amount(48, W)
    P = torch.stack([R.preprocess_proj(projs[i:i + 1], pad) for i in range(8)]).to(DEV)
    S = R.one_hot_masks(segs, 7).to(DEV)
    Hm = torch.stack([R.gaussian_heatmaps(lm[i], H, W) for i in range(8)]).view(8, 14, H, W).to(DEV)
    SGD = torch.optim.SGD if optimizer == 'torch' else dfl_amd.SGD
    opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
    net.train()
    losses = []
    for step in range(30):
        idx = [(step * 4 + j) % 8 for j in range(4)]
        opt.zero_grad()
        out = net(P[idx])
        loss = crit((dfl_amd.center_crop(out[0], S[idx].shape), dfl_amd.center_crop(out[1], Hm[idx].shape)), (S[idx], Hm[idx]))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    # a 30-step run amplifies rounding differences step by step: the bars are per mode
    np.testing.assert_allclose(losses[:10], g['losses'][:10], rtol=0, atol=by_mode(math_mode, 5e-5, 3e-4))
    np.testing.assert_allclose(losses, g['losses'], rtol=0, atol=by_mode(math_mode, 5e-3, 2e-2))
    net.eval()
    with torch.no_grad():
        out = net(P)
    labels = torch.max(dfl_amd.center_crop(out[0], S.shape), dim=1)[1].cpu()
    d = R.hard_dice(labels, segs.long(), 7)
    # 30 steps in, the network is still moving fast (Dice 0.6-0.7): a sanity band only; the +-0.005 bar of north_star is
    # checked where it is defined, at a plateau (test_plateau_dice_matches_reference)
    assert abs(float(np.mean(d)) - float(np.mean(g['hard_dice']))) < 0.03



class _FakeH5DS:
    def __init__(self, shape, dtype):
        self.a = np.zeros(shape, dtype=dtype)

    def __setitem__(self, k, v):
        self.a[k] = v


class _FakeH5:
    def __init__(self):
        self.d = {}

    def create_dataset(self, name, shape, dtype='f4', **kw):
        self.d[name] = _FakeH5DS(shape, dtype)
        return self.d[name]


def test_ensemble_golden():
    """seg_dataset_ensemble with three nets on two images vs the reference's outputs (util.py:293-377)."""
    g = load_golden('ensemble')
    cfg = dict(n_classes=7, depth=3, wf=2, batch_norm=True, padding=True, max_pool=False, num_lands=14, do_res=True,
               block_depth=2)
    nets = [load_net(g, cfg, prefix='net%d/' % i) for i in range(3)]
    imgs = _t(g['imgs'])

    class DS(torch.utils.data.Dataset):
        rob_orig_img_shape = (28, 28)

        def __len__(self):
            return 2

        def __getitem__(self, i):
            return (imgs[i], torch.zeros(1), torch.zeros(1), torch.zeros(1))

    from dfl_amd import util
    f = _FakeH5()
    times = []
    util.seg_dataset_ensemble(DS(), nets, f, dev=torch.device(DEV), num_lands=14, times=times)
    assert len(times) == 2
    segs = f.d['nn-segs'].a
    assert segs.dtype == np.uint8
    # labels: bit-exact against the reference's file outside the pixels whose averaged soft-max has a rounding-level
    # top-2 margin in fp64 (the oracle, pinned to the reference, recomputes that margin here)
    o64 = []
    for i in range(3):
        o = oracle64(cfg, {k[5:]: _t(v) for k, v in g.items() if k.startswith('net%d/' % i)}).eval()
        with torch.no_grad():
            o64.append(o(imgs.double()))
    avg64 = R.center_crop(sum(o[0] for o in o64) / 3.0, (28, 28))
    mask = label_mask(avg64).numpy()
    assert mask.mean() < 5e-3
    assert np.array_equal(segs[~mask], g['nn_segs'][~mask]), 'labels differ from the reference outside the rounding-margin mask'
    assert np.array_equal(segs[~mask], avg64.argmax(1).numpy().astype(np.uint8)[~mask])
    np.testing.assert_allclose(f.d['nn-heats'].a, g['nn_heats'], rtol=1e-3, atol=1e-5)
    # single-net path and validation loops run and agree with the oracle
    f2 = _FakeH5()
    util.seg_dataset(DS(), nets[0], f2, dev=torch.device(DEV), num_lands=14)
    onet = R.OracleUNet(**cfg)
    onet.load_state_dict({k[5:]: _t(v) for k, v in g.items() if k.startswith('net0/')})
    onet.eval()
    with torch.no_grad():
        o = onet(imgs)
    lab = torch.max(R.center_crop(o[0], (28, 28)), dim=1)[1].numpy()
    m1 = label_mask(R.center_crop(o64[0][0], (28, 28))).numpy()
    assert m1.mean() < 5e-3 and np.array_equal(f2.d['nn-segs'].a[~m1], lab.astype(np.uint8)[~m1])
    np.testing.assert_allclose(f2.d['nn-heats'].a, R.center_crop(o[1], (28, 28)).numpy(), rtol=1e-4, atol=1e-5)




def test_validation_loops_match_reference():
    """dfl_amd.util.test_dataset / test_dataset_ensemble against the (mean, std) the REFERENCE's util.py:116-241 returned
    for the same nets and items (tests/golden/validation.npz, tools/gen_golden.py): single net and ensembles, with and
    without landmarks (the fixed 0.5 heat-map weight of the validation loss), both dice_only values."""
    from dfl_amd import util
    from test_oracle_golden import validation_items, validation_nets
    g = load_golden('validation')
    items = validation_items(g)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            return items[i]
    ds = DS()
    dev = torch.device(DEV)
    n14 = [n.to(dev) for n in validation_nets(g, 'l14', 14, 3, dfl_amd.UNet)]
    n0 = [n.to(dev) for n in validation_nets(g, 'l0', 0, 2, dfl_amd.UNet)]
    got = {'single_l14': util.test_dataset(ds, n14[0], dev, 14), 'single_l0': util.test_dataset(ds, n0[0], dev, 0),
           'ens_l14': util.test_dataset_ensemble(ds, n14, dev, 14), 'ens_l14_dice_only': util.test_dataset_ensemble(ds, n14, dev, 14, dice_only=True),
           'ens_l0': util.test_dataset_ensemble(ds, n0, dev, 0)}
    for k, (m, s_) in got.items():
        np.testing.assert_allclose([float(m), float(s_)], g['result/' + k], rtol=0, atol=1e-5, err_msg=k)
    assert not n14[0].training                      # the loop leaves the net in eval mode, like the reference
