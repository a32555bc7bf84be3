#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

Dev-only: needs /root/reference (absent on the GPU box).  It imports the reference's own modules
(unet, dice, ncc, util, warm_restarts_lr, dataset) from /root/reference/train_test_code with stub
h5py/torchvision modules (neither is installed; recipe: SURVEY.md Appendix G) and stores inputs and
expected outputs only -- no reference source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import hashlib
import math
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/train_test_code'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
sys.dont_write_bytecode = True


# ----------------------------------------------------------------------------- stubs
class _FakeDS:
    def __init__(self, arr):
        self._a = arr

    def __getitem__(self, k):
        return self._a[k] if not (isinstance(k, tuple) and len(k) == 0) else self._a

    @property
    def shape(self):
        return self._a.shape


class _FakeGroup(dict):
    def __getitem__(self, k):
        if '/' in k:
            head, rest = k.split('/', 1)
            return dict.__getitem__(self, head)[rest]
        return dict.__getitem__(self, k)


FAKE_FILES = {}


class _FakeFile(_FakeGroup):
    def __init__(self, path, mode='r'):
        super().__init__(FAKE_FILES[path])

    def close(self):
        pass


def install_stubs():
    h5 = types.ModuleType('h5py')
    h5.File = _FakeFile
    sys.modules['h5py'] = h5
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvf = types.ModuleType('torchvision.transforms.functional')
    tvt.InterpolationMode = object
    tvt.functional = tvf
    tv.transforms = tvt
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt
    sys.modules['torchvision.transforms.functional'] = tvf


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def np32(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- synthetic data
def toy_ellipses(n, H, W, seed, num_lands=14):
    """Seeded synthetic 'fluoro' images: background noise + 6 axis-aligned ellipses (labels 1..6).
    Returns projs [n,H,W] f32, segs [n,H,W] u8, lands [n,2,L] f32 (row 0 = col, row 1 = row)."""
    g = torch.Generator().manual_seed(seed)
    projs = 0.1 * torch.randn(n, H, W, generator=g)
    segs = torch.zeros(n, H, W, dtype=torch.uint8)
    lands = torch.zeros(n, 2, num_lands)
    Y, X = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    for i in range(n):
        for c in range(1, 7):
            cx = float(torch.rand(1, generator=g)) * (W * 0.6) + W * 0.2
            cy = float(torch.rand(1, generator=g)) * (H * 0.6) + H * 0.2
            rx = float(torch.rand(1, generator=g)) * (W * 0.10) + W * 0.05
            ry = float(torch.rand(1, generator=g)) * (H * 0.10) + H * 0.05
            m = ((X - cx) / rx) ** 2 + ((Y - cy) / ry) ** 2 <= 1.0
            segs[i][m] = c
            projs[i][m] += 0.3 * c
            lands[i, 0, c - 1] = cx
            lands[i, 1, c - 1] = cy
            lands[i, 0, 6 + c - 1] = cx
            lands[i, 1, 6 + c - 1] = cy - ry
        lands[i, 0, 12], lands[i, 1, 12] = W * 0.25, H * 0.25
        lands[i, 0, 13], lands[i, 1, 13] = -5.0, H * 0.5        # out of bounds on purpose
    return projs, segs, lands


# ----------------------------------------------------------------------------- fixtures
def fixture_tiny(unet, dice, util, name, seed, hp=24, bwd=True, **kw):
    """Tiny preset: full state_dict, input, every block output, outputs, losses, all grads."""
    torch.manual_seed(seed)
    net = unet.UNet(**kw)
    num_lands = kw.get('num_lands', 0)
    ncls = kw['n_classes']
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(2, 1, hp, hp, generator=g)
    net.train()
    # perturb BN affine + running stats so they are not the trivial 1/0
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if p.dim() == 1 and ('block.2' in n_ or 'block.5' in n_):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    acts = {}
    hooks = []
    for i, m in enumerate(net.down_path):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, k='down%d' % i: acts.__setitem__(k, out.detach().clone())))
    for i, m in enumerate(net.up_path):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, k='up%d' % i: acts.__setitem__(k, out.detach().clone())))
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    out = net(x)
    for h in hooks:
        h.remove()
    seg = out[0] if num_lands > 0 else out
    Ho, Wo = seg.shape[-2], seg.shape[-1]
    Ht, Wt = Ho - 4, Wo - 4           # targets smaller than the output => exercises center_crop
    lab = torch.randint(0, ncls, (2, Ht, Wt), generator=g)
    tseg = torch.stack([(lab == c) for c in range(ncls)], 1).float()
    res = {'x': np32(x), 'tseg': np32(tseg), 'seg': np32(seg)}
    if num_lands > 0:
        theat = torch.rand(2, num_lands, Ht, Wt, generator=g) * 0.02
        crit = dice.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
        loss = crit((util.center_crop(seg, tseg.shape), util.center_crop(out[1], theat.shape)), (tseg, theat))
        res['theat'] = np32(theat)
        res['heat'] = np32(out[1])
    else:
        crit = dice.DiceLoss2D(skip_bg=False)
        loss = crit(util.center_crop(seg, tseg.shape), tseg)
    if bwd:
        loss.backward()   # (the reference cannot back-propagate without BN when do_res=True: in-place add on a ReLU output)
    res['loss'] = np.array(loss.item(), dtype=np.float64)
    for k, v in sd0.items():
        res['sd0/' + k] = np32(v)
    for k, v in net.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            res['sd1/' + k] = np32(v)       # BN buffers after one training forward
    keep = ('down0', 'down%d' % (kw['depth'] - 1), 'up%d' % (kw['depth'] - 2))
    for k, v in acts.items():
        if k in keep:
            res['act/' + k] = np32(v)
    if bwd:
        for k, p in net.named_parameters():
            res['grad/' + k] = np32(p.grad) if p.grad is not None else np.zeros(0, dtype=np.float32)
    # eval-mode forward with the updated running stats
    net.eval()
    with torch.no_grad():
        oe = net(x)
    res['seg_eval'] = np32(oe[0] if num_lands > 0 else oe)
    if num_lands > 0:
        res['heat_eval'] = np32(oe[1])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **res)
    print(name, 'loss', loss.item(), 'keys', len(res))


def fixture_paper(unet, dice, util, name, seed, max_pool, num_lands, batch=2):
    """Paper preset: init hashes, strided output samples, argmax map, loss, grad norms (fp32 + fp64)."""
    kw = dict(n_classes=7, depth=6, wf=5, batch_norm=True, padding=True, max_pool=max_pool,
              num_lands=num_lands, do_res=True, block_depth=2)
    torch.manual_seed(seed)
    net = unet.UNet(**kw)
    res = {}
    names = list(net.state_dict().keys())
    res['sd_names'] = np.array(names)
    res['sd_sha'] = np.array([sha(v) for v in net.state_dict().values()])
    g = torch.Generator().manual_seed(seed + (1 if batch == 2 else batch))       # (as tests/problems.py: paper())
    x = torch.randn(batch, 1, 192, 192, generator=g)
    lab = torch.randint(0, 7, (batch, 184, 184), generator=g)
    tseg = torch.stack([(lab == c) for c in range(7)], 1).float()
    theat = torch.rand(batch, 14, 184, 184, generator=g) * 0.02
    res['x_sha'] = np.array(sha(x))
    res['lab'] = lab.to(torch.uint8).numpy()
    res['theat_sha'] = np.array(sha(theat))

    def run(net_, dt):
        net_.train()
        for p in net_.parameters():
            p.grad = None
        out = net_(x.to(dt))
        seg = out[0] if num_lands > 0 else out
        if num_lands > 0:
            crit = dice.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
            loss = crit((util.center_crop(seg, tseg.shape), util.center_crop(out[1], theat.shape)),
                        (tseg.to(dt), theat.to(dt)))
        else:
            loss = dice.DiceLoss2D(skip_bg=False)(util.center_crop(seg, tseg.shape), tseg.to(dt))
        loss.backward()
        return out, loss

    out32, loss32 = run(net, torch.float32)
    seg32 = out32[0] if num_lands > 0 else out32
    gn32 = {k: p.grad.double().norm().item() if p.grad is not None else -1.0 for k, p in net.named_parameters()}
    import copy
    net64 = copy.deepcopy(net).double()
    # deepcopy after a training forward has updated running stats; harmless for a train-mode forward
    out64, loss64 = run(net64, torch.float64)
    seg64 = out64[0] if num_lands > 0 else out64
    gn64 = {k: p.grad.norm().item() if p.grad is not None else -1.0 for k, p in net64.named_parameters()}
    res['seg_s16'] = np32(seg32[:, :, ::16, ::16])
    res['seg64_s16'] = seg64[:, :, ::16, ::16].detach().numpy()
    top2 = torch.topk(seg64, 2, dim=1)[0]
    res['argmax64'] = torch.max(seg64, dim=1)[1].to(torch.uint8).numpy()
    res['argmax32'] = torch.max(seg32, dim=1)[1].to(torch.uint8).numpy()
    res['margin_lt_1e5'] = np.packbits(((top2[:, 0] - top2[:, 1]) < 1e-5).numpy())
    if num_lands > 0:
        res['heat_s16'] = np32(out32[1][:, :, ::16, ::16])
        res['heat64_s16'] = out64[1][:, :, ::16, ::16].detach().numpy()
    res['loss32'] = np.array(loss32.item())
    res['loss64'] = np.array(loss64.item())
    pn = [k for k, _ in net.named_parameters()]
    res['param_names'] = np.array(pn)
    res['gradnorm32'] = np.array([gn32[k] for k in pn])
    res['gradnorm64'] = np.array([gn64[k] for k in pn])
    # a few full small gradients from the fp64 run (heads + deepest BN) for direct comparison
    for k, p in net64.named_parameters():
        if p.grad is not None and p.numel() <= 4096:
            res['g64/' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **res)
    print(name, 'loss32', loss32.item(), 'loss64', loss64.item())


def fixture_losses(dice, ncc):
    g = torch.Generator().manual_seed(7)
    res = {}
    s = torch.softmax(torch.randn(2, 7, 10, 12, generator=g), 1).double().requires_grad_(True)
    lab = torch.randint(0, 7, (2, 10, 12), generator=g)
    lab[0][lab[0] == 6] = 5           # class 6 empty in the target of image 0
    t = torch.stack([(lab == c) for c in range(7)], 1).double()
    for sb in (True, False):
        s.grad = None
        l = dice.DiceLoss2D(skip_bg=sb)(s, t)
        l.backward()
        res['dice_sb%d' % int(sb)] = np.array(l.item())
        res['dice_sb%d_grad' % int(sb)] = s.grad.numpy().copy()
    res['dice_in'] = s.detach().numpy()
    res['dice_tgt'] = t.numpy()
    perfect = dice.DiceLoss2D(skip_bg=False)(t, t)
    res['dice_perfect'] = np.array(perfect.item())
    X = torch.randn(2, 14, 10, 12, generator=g).double().requires_grad_(True)
    Y = (torch.rand(2, 14, 10, 12, generator=g) * 0.02).double()
    n = ncc.ncc_2d(X, Y)
    n.sum().backward()
    res['ncc_x'] = X.detach().numpy()
    res['ncc_y'] = Y.numpy()
    res['ncc'] = n.detach().numpy()
    res['ncc_grad_of_sum'] = X.grad.numpy().copy()
    res['ncc_self'] = ncc.ncc_2d(Y, Y).numpy()
    X.grad = None
    s.grad = None
    l = dice.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.3)((s, X), (t, Y))
    l.backward()
    res['dh_loss'] = np.array(l.item())
    res['dh_gseg'] = s.grad.numpy().copy()
    res['dh_gheat'] = X.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'losses.npz'), **res)
    print('losses', {k: float(v) for k, v in res.items() if v.ndim == 0})


def fixture_sched(wr):
    import torch.optim as optim
    res = {}
    for tag, (period, growth) in {'p2g2': (2, 2), 'p3g1': (3, 1)}.items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = optim.SGD([p], lr=0.1)
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            s = wr.WarmRestartLR(opt, init_run_period_epochs=period, growth_factor=growth)
            trace = []
            restarts = []
            for ep in range(9):
                for k in range(4):
                    s.intra_epoch_step((k + 1) / 4)
                    trace.append(opt.param_groups[0]['lr'])
                s.step()
                trace.append(opt.param_groups[0]['lr'])
                restarts.append(int(s.just_restarted))
        res[tag] = np.array(trace)
        res[tag + '_restarts'] = np.array(restarts)
    np.savez_compressed(os.path.join(OUT, 'sched.npz'), **res)
    print('sched', res['p2g2'][:6])


def fixture_dataset(dataset):
    projs, segs, lands = toy_ellipses(3, 46, 46, seed=11)
    FAKE_FILES['fake.h5'] = {
        '01': {'projs': _FakeDS(projs[:2].numpy()), 'segs': _FakeDS(segs[:2].numpy()), 'lands': _FakeDS(lands[:2].numpy())},
        '02': {'projs': _FakeDS(projs[2:].numpy()), 'segs': _FakeDS(segs[2:].numpy()), 'lands': _FakeDS(lands[2:].numpy())},
        'land-names': {'num-lands': _FakeDS(np.array(14)),
                       **{'land-%02d' % l: _FakeDS(np.bytes_(('L%02d' % l).encode())) for l in range(14)}},
    }
    ds = dataset.get_dataset('fake.h5', [1, 2], num_classes=7, pad_img_dim=48)
    res = {'projs': projs.numpy(), 'segs': segs.numpy(), 'lands': lands.numpy(),
           'num_lands': np.array(dataset.get_num_lands_from_dataset('fake.h5')),
           'pad_48_46': np.array(dataset.calc_pad_amount(48, 46)),
           'pad_192_184': np.array(dataset.calc_pad_amount(192, 184)),
           'pad_193_180': np.array(dataset.calc_pad_amount(193, 180)),
           'orig_shape': np.array(ds.rob_orig_img_shape), 'len': np.array(len(ds))}
    for i in range(3):
        p, s, l, h = ds[i]
        res['item%d_p' % i] = p.numpy()
        res['item%d_s' % i] = s.numpy()
        res['item%d_l' % i] = l.numpy()
        res['item%d_h' % i] = h.numpy()
    np.savez_compressed(os.path.join(OUT, 'dataset.npz'), **res)
    print('dataset item shapes', [tuple(t.shape) for t in ds[0]])


def fixture_ensemble(unet, util):
    """seg_dataset_ensemble (util.py:293-377) with 3 tiny reference nets on 2 images."""
    kw = dict(n_classes=7, depth=3, wf=2, batch_norm=True, padding=True, max_pool=False,
              num_lands=14, do_res=True, block_depth=2)
    nets = []
    sds = []
    for s in (21, 22, 23):
        torch.manual_seed(s)
        n = unet.UNet(**kw)
        g = torch.Generator().manual_seed(s)
        with torch.no_grad():
            for b in n.buffers():
                if b.dtype.is_floating_point:
                    b.add_(0.2 * torch.rand(b.shape, generator=g))
        nets.append(n)
        sds.append({k: v.clone() for k, v in n.state_dict().items()})
    g = torch.Generator().manual_seed(5)
    imgs = torch.randn(2, 1, 32, 32, generator=g)

    class DS(torch.utils.data.Dataset):
        rob_orig_img_shape = (28, 28)

        def __len__(self):
            return 2

        def __getitem__(self, i):
            return (imgs[i], torch.zeros(1), torch.zeros(1), torch.zeros(1))

    class FakeH5DS:
        def __init__(self, shape, dtype):
            self.a = np.zeros(shape, dtype=dtype)

        def __setitem__(self, k, v):
            self.a[k] = v

    class FakeH5:
        def __init__(self):
            self.d = {}

        def create_dataset(self, name, shape, dtype='f4', **kw_):
            self.d[name] = FakeH5DS(shape, dtype)
            return self.d[name]

    f = FakeH5()
    util.seg_dataset_ensemble(DS(), nets, f, dev=None, num_lands=14, times=[])
    res = {'imgs': imgs.numpy(), 'nn_segs': f.d['nn-segs'].a, 'nn_heats': f.d['nn-heats'].a}
    for i, sd in enumerate(sds):
        for k, v in sd.items():
            res['net%d/%s' % (i, k)] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'ensemble.npz'), **res)
    print('ensemble labels hist', np.bincount(res['nn_segs'].ravel(), minlength=7))


def fixture_validation(unet, util):
    """util.test_dataset (util.py:116-165) and util.test_dataset_ensemble (util.py:167-241) of the REFERENCE on seeded tiny
    nets and four toy images: (mean, std) of the per-image losses with and without landmarks (D11: the fixed 0.5 heat-map
    weight of the validation loss) and both dice_only values."""
    base = dict(n_classes=7, depth=3, wf=2, batch_norm=True, padding=True, max_pool=False, do_res=True, block_depth=2)

    def make(seed, num_lands):
        torch.manual_seed(seed)
        n = unet.UNet(num_lands=num_lands, **base)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for b in n.buffers():
                if b.dtype.is_floating_point:
                    b.add_(0.2 * torch.rand(b.shape, generator=g))
        return n
    nets14 = [make(s, 14) for s in (31, 32, 33)]
    nets0 = [make(s, 0) for s in (41, 42)]
    H = W = 28
    projs, segs, lands = toy_ellipses(4, H, W, seed=77)
    g = torch.Generator().manual_seed(78)
    x = torch.nn.functional.pad(projs[:, None], (2, 2, 2, 2), mode='reflect')
    x = (x - x.mean(dim=(1, 2, 3), keepdim=True)) / x.std(dim=(1, 2, 3), keepdim=True)
    masks = torch.stack([(segs == c) for c in range(7)], 1).float()
    Y, X = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    heats = torch.zeros(4, 14, 1, H, W)
    for i in range(4):
        for l in range(14):
            cx, cy = float(lands[i, 0, l]), float(lands[i, 1, l])
            if 0 <= cx <= W - 1 and 0 <= cy <= H - 1:
                heats[i, l, 0] = torch.exp(((X - cx) ** 2 + (Y - cy) ** 2) / (-2 * 2.5 * 2.5)) / (2 * math.pi * 2.5 * 2.5)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return (x[i], masks[i], lands[i], heats[i])

    res = {'x': x.numpy(), 'masks': masks.numpy(), 'lands': lands.numpy(), 'heats': heats.numpy()}
    for tag, nets in (('l14', nets14), ('l0', nets0)):
        for i, n in enumerate(nets):
            for k, v in n.state_dict().items():
                res['%s_net%d/%s' % (tag, i, k)] = v.numpy()
    ds = DS()
    cpu = torch.device('cpu')       # (with dev=None the reference skips the [B,L,1,H,W] -> [B,L,H,W] view of the heat targets)
    out = {
        'single_l14': util.test_dataset(ds, nets14[0], cpu, 14),
        'single_l0': util.test_dataset(ds, nets0[0], cpu, 0),
        'ens_l14': util.test_dataset_ensemble(ds, nets14, cpu, 14, dice_only=False),
        'ens_l14_dice_only': util.test_dataset_ensemble(ds, nets14, cpu, 14, dice_only=True),
        'ens_l0': util.test_dataset_ensemble(ds, nets0, cpu, 0),
    }
    for k, (m, sd) in out.items():
        res['result/' + k] = np.array([float(m), float(sd)], dtype=np.float64)
        print('validation %-20s mean %.6f std %.6f' % (k, float(m), float(sd)))
    np.savez_compressed(os.path.join(OUT, 'validation.npz'), **res)


def fixture_trajectory(unet, dice, util):
    """Short SGD trajectory (train.py:405-430 wiring) on the toy-ellipses set, tiny-ish net."""
    import torch.optim as optim
    projs, segs, lands = toy_ellipses(8, 40, 40, seed=3)
    kw = dict(n_classes=7, depth=3, wf=3, batch_norm=True, padding=True, max_pool=False,
              num_lands=14, do_res=True, block_depth=2)
    torch.manual_seed(99)
    net = unet.UNet(**kw)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    opt = optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    crit = dice.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
    # loader math restated inline from the reference's own functions is avoided: use its dataset class
    import dataset
    FAKE_FILES['traj.h5'] = {'01': {'projs': _FakeDS(projs.numpy()), 'segs': _FakeDS(segs.numpy()),
                                    'lands': _FakeDS(lands.numpy())},
                             'land-names': {'num-lands': _FakeDS(np.array(14))}}
    ds = dataset.get_dataset('traj.h5', [1], num_classes=7, pad_img_dim=48)
    items = [ds[i] for i in range(8)]
    P = torch.stack([it[0] for it in items])
    S = torch.stack([it[1] for it in items])
    Hm = torch.stack([it[3] for it in items]).view(8, 14, 40, 40)
    losses = []
    net.train()
    for step in range(30):
        idx = [(step * 4 + j) % 8 for j in range(4)]
        opt.zero_grad()
        out = net(P[idx])
        loss = crit((util.center_crop(out[0], S[idx].shape), util.center_crop(out[1], Hm[idx].shape)),
                    (S[idx], Hm[idx]))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    net.eval()
    with torch.no_grad():
        out = net(P)
    labels = torch.max(util.center_crop(out[0], S.shape), dim=1)[1]
    d = []
    for c in range(1, 7):
        a, b = labels == c, segs.long() == c
        den = int(a.sum()) + int(b.sum())
        d.append(2.0 * int((a & b).sum()) / den if den > 0 else 1.0)
    res = {'projs': projs.numpy(), 'segs': segs.numpy(), 'lands': lands.numpy(),
           'losses': np.array(losses), 'hard_dice': np.array(d)}
    for k, v in sd0.items():
        res['sd0/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'trajectory.npz'), **res)
    print('trajectory losses', losses[0], losses[-1], 'dice', np.mean(d))


def _add_reference_runs(path, run_dice, threads):
    """the per-class bar of the plateau tests is 0.005 + the REFERENCE's own spread, so the fixture has to hold more than
    two reference runs.  Adds runs with the given thread counts (other summation orders inside the reference's convolutions) to an
    existing fixture -- `dice_train_runs` / `dice_valid_runs` [run][class] and `run_threads` -- keeping every array already there (the
    8- and 1-thread runs are rows 0 and 1).  run_dice(threads) -> (train dice per class, held-out dice per class)."""
    old = dict(np.load(path, allow_pickle=False))
    tr = [old['dice_train'], old['dice_train_1thread']]
    va = [old['dice_valid'], old['dice_valid_1thread']]
    th = [8, 1]
    if 'run_threads' in old:
        tr, va, th = list(old['dice_train_runs']), list(old['dice_valid_runs']), [int(t) for t in old['run_threads']]
    for t in threads:
        if t in th:
            continue
        dtr, dva = run_dice(t)
        tr.append(dtr)
        va.append(dva)
        th.append(t)
        print('  %s: run with %d threads: train dice %s (mean %.4f), held-out mean %.4f' % (os.path.basename(path), t, np.round(dtr, 4), dtr.mean(), dva.mean()), flush=True)
    old['dice_train_runs'], old['dice_valid_runs'], old['run_threads'] = np.stack(tr), np.stack(va), np.array(th)
    np.savez_compressed(path, **old)
    sp = np.stack(tr).max(0) - np.stack(tr).min(0)
    print('  %s: %d reference runs (threads %s); per-class spread of the training dice %s' % (os.path.basename(path), len(th), th, np.round(sp, 4)))
    torch.set_num_threads(8)


def fixture_plateau(unet, dice, util, wf=3, out_name='plateau.npz', extend=None):
    """The reference trained to a plateau on the toy-ellipses set (16 training + 8 held-out images, 400 SGD steps wired as
    train.py:405-430, learning rate cut 10x for the last 100): hard Dice per class on both sets (formula of
    compute_actual_dice_on_test.py:63-93) -- the quantity north_star's "+-0.005 of reference" is about.  The reference is
    run TWICE (8 threads / 1 thread: different summation orders in its convolutions) so that the fixture also records the
    reference's own run-to-run spread at the plateau."""
    import torch.optim as optim
    import dataset
    projs, segs, lands = toy_ellipses(24, 40, 40, seed=11)
    kw = dict(n_classes=7, depth=3, wf=wf, batch_norm=True, padding=True, max_pool=False,
              num_lands=14, do_res=True, block_depth=2)
    FAKE_FILES['plateau.h5'] = {'01': {'projs': _FakeDS(projs.numpy()), 'segs': _FakeDS(segs.numpy()),
                                       'lands': _FakeDS(lands.numpy())},
                                'land-names': {'num-lands': _FakeDS(np.array(14))}}
    ds = dataset.get_dataset('plateau.h5', [1], num_classes=7, pad_img_dim=48)
    items = [ds[i] for i in range(24)]
    P = torch.stack([it[0] for it in items])
    S = torch.stack([it[1] for it in items])
    Hm = torch.stack([it[3] for it in items]).view(24, 14, 40, 40)
    NTR, STEPS = 16, 400

    def hard_dice(labels, gt):
        d = []
        for c in range(1, 7):
            a, b = labels == c, gt == c
            den = int(a.sum()) + int(b.sum())
            d.append(2.0 * int((a & b).sum()) / den if den > 0 else 1.0)
        return np.array(d)

    def run(threads):
        torch.set_num_threads(threads)
        torch.manual_seed(4242)
        net = unet.UNet(**kw)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        opt = optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
        crit = dice.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
        net.train()
        losses = []
        for step in range(STEPS):
            if step == 300:
                for gr in opt.param_groups:
                    gr['lr'] = 0.005
            idx = [(step * 4 + j) % NTR for j in range(4)]
            opt.zero_grad()
            out = net(P[idx])
            loss = crit((util.center_crop(out[0], S[idx].shape), util.center_crop(out[1], Hm[idx].shape)),
                        (S[idx], Hm[idx]))
            loss.backward()
            opt.step()
            losses.append(loss.item())
        net.eval()
        with torch.no_grad():
            out = net(P)
        labels = torch.max(util.center_crop(out[0], S.shape), dim=1)[1]
        gt = segs.long()
        return sd0, np.array(losses), hard_dice(labels[:NTR], gt[:NTR]), hard_dice(labels[NTR:], gt[NTR:])

    if extend is not None:
        _add_reference_runs(os.path.join(OUT, out_name), lambda t: run(t)[2:], extend)
        return
    sd0, l8, dtr8, dva8 = run(8)
    _, l1, dtr1, dva1 = run(1)
    torch.set_num_threads(8)
    res = {'projs': projs.numpy(), 'segs': segs.numpy(), 'lands': lands.numpy(), 'n_train': np.array(NTR),
           'steps': np.array(STEPS), 'losses': l8, 'losses_1thread': l1,
           'dice_train': dtr8, 'dice_valid': dva8, 'dice_train_1thread': dtr1, 'dice_valid_1thread': dva1}
    for k, v in sd0.items():
        res['sd0/' + k] = v.numpy()
    res['wf'] = np.array(wf)
    np.savez_compressed(os.path.join(OUT, out_name), **res)
    print('plateau: loss %.4f -> %.4f; train dice %s (mean %.4f / 1 thread %.4f); valid dice %s (mean %.4f / %.4f)' % (
        l8[0], l8[-20:].mean(), np.round(dtr8, 4), dtr8.mean(), dtr1.mean(), np.round(dva8, 4), dva8.mean(), dva1.mean()))
    print('  per-class |8 threads - 1 thread|: train', np.round(np.abs(dtr8 - dtr1), 4), 'valid', np.round(np.abs(dva8 - dva1), 4))

def fixture_plateau_paper(unet, dice, util, steps=400, cut=300, threads=(8, 1), out_name='plateau_paper.npz', extend=None):
    """The PAPER preset (train_test_code/Readme.md:16: depth 6, 32 initial features, BatchNorm, padding, strided convolutions,
    14 landmarks, SGD 0.1 / 0.9 / nesterov / 1e-4) trained by the reference to a plateau on toy-ellipses images of the paper's
    8x-downsampled size (184 x 184 padded to 192; 16 training + 4 held-out images, batch 4, the step body of train.py:405-430,
    learning rate cut 10x for the last quarter), scored by hard Dice per class (compute_actual_dice_on_test.py:63-93).  The
    initial weights are NOT stored (152 MB): torch.manual_seed(1234) + the constructor, whose per-tensor SHA-256 is in
    paper_sc_l14.npz and is checked here.  Two runs (8 threads / 1 thread) record the reference's own spread."""
    import torch.optim as optim
    import dataset
    NTR, NALL, H = 16, 20, 184
    projs, segs, lands = toy_ellipses(NALL, H, H, seed=21)
    kw = dict(n_classes=7, depth=6, wf=5, batch_norm=True, padding=True, max_pool=False, num_lands=14, do_res=True, block_depth=2)
    FAKE_FILES['plateau_paper.h5'] = {'01': {'projs': _FakeDS(projs.numpy()), 'segs': _FakeDS(segs.numpy()),
                                             'lands': _FakeDS(lands.numpy())},
                                      'land-names': {'num-lands': _FakeDS(np.array(14))}}
    ds = dataset.get_dataset('plateau_paper.h5', [1], num_classes=7, pad_img_dim=192)
    items = [ds[i] for i in range(NALL)]
    P = torch.stack([it[0] for it in items])
    S = torch.stack([it[1] for it in items])
    Hm = torch.stack([it[3] for it in items]).view(NALL, 14, H, H)
    shas = dict(np.load(os.path.join(OUT, 'paper_sc_l14.npz'), allow_pickle=False))

    def hard_dice(labels, gt):
        d = []
        for c in range(1, 7):
            a, b = labels == c, gt == c
            den = int(a.sum()) + int(b.sum())
            d.append(2.0 * int((a & b).sum()) / den if den > 0 else 1.0)
        return np.array(d)

    def run(nthreads):
        import time
        torch.set_num_threads(nthreads)
        torch.manual_seed(1234)
        net = unet.UNet(**kw)
        names = [str(n) for n in shas['sd_names']] if 'sd_names' in shas else None
        if names is not None:
            for n_, h_ in zip(names, shas['sd_sha']):
                assert sha(net.state_dict()[n_]) == str(h_), n_
        opt = optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
        crit = dice.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
        net.train()
        losses = []
        t0 = time.time()
        for step in range(steps):
            if step == cut:
                for gr in opt.param_groups:
                    gr['lr'] = 0.01
            idx = [(step * 4 + j) % NTR for j in range(4)]
            opt.zero_grad()
            out = net(P[idx])
            loss = crit((util.center_crop(out[0], S[idx].shape), util.center_crop(out[1], Hm[idx].shape)), (S[idx], Hm[idx]))
            loss.backward()
            opt.step()
            losses.append(loss.item())
            if step % 20 == 0:
                print('  [%d threads] step %d loss %.4f (%.0f s)' % (nthreads, step, losses[-1], time.time() - t0), flush=True)
        net.eval()
        with torch.no_grad():
            out = torch.cat([net(P[i:i + 4])[0] for i in range(0, NALL, 4)])
        labels = torch.max(util.center_crop(out, S.shape), dim=1)[1]
        gt = segs.long()
        return np.array(losses), hard_dice(labels[:NTR], gt[:NTR]), hard_dice(labels[NTR:], gt[NTR:])

    if extend is not None:
        _add_reference_runs(os.path.join(OUT, out_name), lambda t: run(t)[1:], extend)
        return
    runs = [run(t) for t in threads]
    torch.set_num_threads(8)
    (l8, dtr8, dva8), (l1, dtr1, dva1) = runs[0], runs[-1]
    res = {'projs': projs.numpy(), 'segs': segs.numpy(), 'lands': lands.numpy(), 'n_train': np.array(NTR), 'steps': np.array(steps),
           'cut': np.array(cut), 'seed': np.array(1234), 'losses': l8, 'losses_1thread': l1,
           'dice_train': dtr8, 'dice_valid': dva8, 'dice_train_1thread': dtr1, 'dice_valid_1thread': dva1}
    np.savez_compressed(os.path.join(OUT, out_name), **res)
    print('plateau (paper preset): loss %.4f -> %.4f; train dice %s (mean %.4f / %.4f); valid dice %s (mean %.4f / %.4f)' % (
        l8[0], l8[-20:].mean(), np.round(dtr8, 4), dtr8.mean(), dtr1.mean(), np.round(dva8, 4), dva8.mean(), dva1.mean()))
    print('  per-class |run A - run B|: train', np.round(np.abs(dtr8 - dtr1), 4), 'valid', np.round(np.abs(dva8 - dva1), 4))


EST_LAND_NAMES = ['FH-l', 'FH-r', 'GSN-l', 'GSN-r', 'IOF-l', 'IOF-r', 'MOF-l', 'MOF-r', 'SPS-l', 'SPS-r', 'IPS-l', 'IPS-r',
                  'ASIS-l', 'ASIS-r']
EST_LABEL_FOR = [5, 6, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2]     # est_lands_csv.py:56-73


def fixture_est_lands(util):
    """Landmark extraction (est_lands_csv.py:96-124) run as the reference runs it: the script itself, executed with
    runpy on a fake HDF5 file, with and without the segmentation mask; the CSV rows are the expected outputs."""
    import runpy
    import tempfile
    g = torch.Generator().manual_seed(77)
    B, H, W, L = 3, 44, 52, 14
    heats = 2e-4 * torch.rand(B, L, H, W, generator=g)
    segs = torch.zeros(B, H, W, dtype=torch.uint8)
    segs[:, :, :W // 2] = 1
    segs[:, :, W // 2:] = 2
    segs[:, H // 2:, : W // 4] = 5
    segs[:, H // 2:, 3 * W // 4:] = 6
    for i in range(B):
        for l in range(L):
            kind = (i * L + l) % 7
            want = EST_LABEL_FOR[l]
            while True:                                   # most blobs sit inside the label the masked rule looks at
                r = int(torch.randint(0, H, (1,), generator=g))
                c = int(torch.randint(0, W, (1,), generator=g))
                if kind == 2 or int(segs[i, r, c]) == want:
                    break
            if kind == 5:
                continue                                  # noise only: NCC far below 0.9
            if kind == 6:
                r, c = (0, W - 1) if l % 2 else (H - 1, 0)   # blob on the border: the reflect padding matters
            blob = util.get_gaussian_2d_heatmap(H, W, 2.5, peak_row=r, peak_col=c)
            if kind == 4:
                blob = util.get_gaussian_2d_heatmap(H, W, 6.0, peak_row=r, peak_col=c) * 3   # wrong width: NCC < 0.9?
            heats[i, l] += blob
            if kind == 3:                                 # a second, higher peak outside the label
                while True:
                    r2 = int(torch.randint(0, H, (1,), generator=g))
                    c2 = int(torch.randint(0, W, (1,), generator=g))
                    if int(segs[i, r2, c2]) != want:
                        break
                heats[i, l] += 1.5 * util.get_gaussian_2d_heatmap(H, W, 2.5, peak_row=r2, peak_col=c2)
    FAKE_FILES['heat.h5'] = {
        'nn-heats': _FakeDS(heats.numpy()), 'nn-segs': _FakeDS(segs.numpy()),
        'land-names': {'num-lands': _FakeDS(np.array(L)),
                       **{'land-%02d' % l: _FakeDS(np.bytes_(EST_LAND_NAMES[l].encode())) for l in range(L)}},
    }
    res = {'heats': heats.numpy(), 'segs': segs.numpy(), 'label_for_land': np.array(EST_LABEL_FOR, dtype=np.int32)}
    script = os.path.join(REF, 'est_lands_csv.py')
    for tag, extra in (('masked', ['--use-seg', 'nn-segs']), ('plain', [])):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, 'l.csv')
            argv = sys.argv
            sys.argv = [script, 'heat.h5', 'nn-heats', '--out', out, '--pat', '1'] + extra
            try:
                ns = runpy.run_path(script, run_name='__main__')
                ns['csv_out'].close()       # the script leaves the file to interpreter exit
            finally:
                sys.argv = argv
            rows = [ln.strip().split(',') for ln in open(out)][1:]
        rc = np.full((B, L, 2), -7, dtype=np.int32)
        for pat, proj, land, row, col, _t in rows:
            rc[int(proj), int(land)] = (int(row), int(col))
        assert (rc != -7).all(), (len(rows), rows[:3], rows[-1])
        res['rc_' + tag] = rc
        print('est_lands', tag, 'found', int((rc[..., 0] >= 0).sum()), 'of', B * L)
    np.savez_compressed(os.path.join(OUT, 'est_lands.npz'), **res)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    sys.path.insert(0, REF)
    import ncc
    import dice
    import util
    import unet
    import warm_restarts_lr as wr
    import dataset
    torch.set_num_threads(8)
    if '--extend-plateau-runs' in sys.argv:          # two more reference runs (4 and 2 threads) per plateau fixture
        fixture_plateau(unet, dice, util, extend=(4, 2))
        fixture_plateau(unet, dice, util, wf=4, out_name='plateau_wf4.npz', extend=(4, 2))
        fixture_plateau_paper(unet, dice, util, extend=(4, 2))
        return
    if '--only-plateau' in sys.argv:
        fixture_plateau(unet, dice, util)
        fixture_plateau(unet, dice, util, wf=4, out_name='plateau_wf4.npz')     # 16..64 channels: the bf16 storage mode's minimum
        return
    if '--only-plateau-paper' in sys.argv:
        kw = {}
        if '--quick' in sys.argv:          # (look at the trajectory first: one 8-thread run, nothing kept)
            kw = dict(threads=(8,), out_name='plateau_paper_quick.npz')
        fixture_plateau_paper(unet, dice, util, **kw)
        return
    if '--only-paper-sc-l0' in sys.argv:
        fixture_paper(unet, dice, util, 'paper_sc_l0', 1236, max_pool=False, num_lands=0, batch=4)
        return
    if '--only-validation' in sys.argv:
        fixture_validation(unet, util)
        return
    if '--only-est-lands' in sys.argv:
        sys.argv.remove('--only-est-lands')
        fixture_est_lands(util)
        return
    base = dict(n_classes=7, depth=3, wf=2, batch_norm=True, padding=True, do_res=True, block_depth=2)
    fixture_tiny(unet, dice, util, 'tiny_sc_l14', 101, max_pool=False, num_lands=14, **base)
    fixture_tiny(unet, dice, util, 'tiny_mp_l0', 102, max_pool=True, num_lands=0, **base)
    fixture_tiny(unet, dice, util, 'tiny_mp_l14', 103, max_pool=True, num_lands=14, **base)
    fixture_tiny(unet, dice, util, 'tiny_valid_nores', 104, hp=32, max_pool=True, num_lands=0,
                 n_classes=7, depth=2, wf=2, batch_norm=True, padding=False, do_res=False, block_depth=2)
    fixture_tiny(unet, dice, util, 'tiny_nobn_d1', 105, bwd=False, max_pool=False, num_lands=14,
                 n_classes=3, depth=2, wf=3, batch_norm=False, padding=True, do_res=True, block_depth=1)
    fixture_tiny(unet, dice, util, 'tiny_nobn_nores', 107, max_pool=False, num_lands=0,
                 n_classes=3, depth=2, wf=3, batch_norm=False, padding=True, do_res=False, block_depth=2)
    fixture_tiny(unet, dice, util, 'tiny_bd3_nosm', 106, max_pool=True, num_lands=0,
                 n_classes=5, depth=2, wf=3, batch_norm=True, padding=True, do_res=True, block_depth=3,
                 do_soft_max=False)
    fixture_losses(dice, ncc)
    fixture_sched(wr)
    fixture_dataset(dataset)
    fixture_ensemble(unet, util)
    fixture_validation(unet, util)
    fixture_trajectory(unet, dice, util)
    fixture_plateau(unet, dice, util)
    fixture_plateau(unet, dice, util, wf=4, out_name='plateau_wf4.npz')
    fixture_est_lands(util)
    fixture_paper(unet, dice, util, 'paper_sc_l14', 1234, max_pool=False, num_lands=14)
    fixture_paper(unet, dice, util, 'paper_mp_l0', 1235, max_pool=True, num_lands=0)
    fixture_paper(unet, dice, util, 'paper_sc_l0', 1236, max_pool=False, num_lands=0, batch=4)     # BASELINE configs[0]'s program
    fixture_plateau_paper(unet, dice, util)         # (needs paper_sc_l14.npz: the SHA-256 of the seeded initial weights)


if __name__ == '__main__':
    main()
