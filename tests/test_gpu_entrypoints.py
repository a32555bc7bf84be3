"""train.py and test_ensemble.py end to end on the GPU, as a user of the reference would call them: a small synthetic data
file in the reference's layout (hdf5_layouts/Readme.md:105-117, here as .npz: h5py is absent), two epochs of training,
a resumed third, then a two-net ensemble.  Checks the files they leave against the reference's formats
(train.py:463-513 checkpoint dictionary, util.py:72-74 loss logs, util.py:300-310 + test_ensemble.py:126-132 outputs)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from test_entrypoints_cpu import CHECKPOINT_KEYS

pytestmark = pytest.mark.gpu

H = W = 44
L, NC = 3, 4
LAND_NAMES = ['GSN-l', 'GSN-r', 'IOF-l']        # landmarks tied to labels 1, 2, 1 (est_lands_csv.py:57-74)


def make_file(path, n_per_pat=(8, 4)):
    g = torch.Generator().manual_seed(3)
    d = {'land-names/num-lands': np.int64(L)}
    for l, name in enumerate(LAND_NAMES):
        d['land-names/land-%02d' % l] = np.array(name)
    Y, X = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    for pat, n in enumerate(n_per_pat, 1):
        projs = 0.1 * torch.randn(n, H, W, generator=g)
        segs = torch.zeros(n, H, W, dtype=torch.uint8)
        lands = torch.zeros(n, 2, L)
        for i in range(n):
            for c in range(1, NC):
                cx, cy = float(torch.rand(1, generator=g)) * 24 + 10, float(torch.rand(1, generator=g)) * 24 + 10
                m = ((X - cx) / 6) ** 2 + ((Y - cy) / 5) ** 2 <= 1
                segs[i][m] = c
                projs[i][m] += 0.4 * c
                lands[i, :, c - 1] = torch.tensor([cx, cy])
        d['%02d/projs' % pat], d['%02d/segs' % pat], d['%02d/lands' % pat] = projs.numpy(), segs.numpy(), lands.numpy()
    np.savez(path, **d)


def run(script, args, cwd):
    p = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    return p.stdout


def test_train_resume_and_ensemble(tmp_path):
    cwd = str(tmp_path)
    make_file(os.path.join(cwd, 'data.npz'))
    common = ['data.npz', '--train-pats', '1', '--valid-pats', '2', '--num-classes', str(NC), '--unet-img-dim', '48',
              '--batch-size', '4', '--unet-num-lvls', '3', '--unet-init-feats-exp', '3', '--unet-batch-norm', '--unet-padding',
              '--unet-no-max-pool', '--use-lands', '--nesterov', '--wgt-decay', '1e-4', '--init-lr', '0.05',
              '--cos-anneal-epochs', '1', '--cos-growth', '1', '--save-restart-net', 'restart', '--checkpoint-net', 'ck.pt',
              '--best-net', 'best.pt', '--train-loss-txt', 'tl.txt', '--valid-loss-txt', 'vl.txt']
    out = run('train.py', common + ['--max-num-epochs', '2'], cwd)
    assert 'num. lands read from file: 3' in out and 'Epoch: 001' in out and 'Exiting - maximum number of epochs performed!' in out
    ck = torch.load(os.path.join(cwd, 'ck.pt'), map_location='cpu', weights_only=False)
    assert list(ck.keys()) == CHECKPOINT_KEYS
    assert ck['epoch'] == 2 and ck['num-lands'] == L and ck['depth'] == 3 and ck['no-max-pool'] is True and ck['lrs-meth'] == 'cos'
    assert ck['lrs-num-restarts'] == 2 and ck['pad-img-size'] == 48 and ck['batch-size'] == 4
    assert isinstance(ck['loss'], torch.Tensor) and ck['loss'].dim() == 0
    import dfl_amd
    ref_keys = list(dfl_amd.UNet(n_classes=NC, depth=3, wf=3, batch_norm=True, padding=True, max_pool=False, num_lands=L).state_dict().keys())
    assert list(ck['model-state-dict'].keys()) == ref_keys
    assert 'momentum_buffer' in next(iter(ck['optimizer-state-dict']['state'].values()))
    fl = re.compile(r'^-?\d+\.\d{6}$')
    tl = open(os.path.join(cwd, 'tl.txt')).read().split('\n')
    assert tl[-1] == '' and len(tl) - 1 == 2 * 2 and all(fl.match(x) for x in tl[:-1])      # 8 images / batch 4, 2 epochs
    vl = open(os.path.join(cwd, 'vl.txt')).read().split('\n')
    assert len(vl) - 1 == 2 and all(fl.match(x) for x in vl[:-1])
    assert os.path.exists(os.path.join(cwd, 'best.pt')) and os.path.exists(os.path.join(cwd, 'restart_00.pt'))
    assert float(vl[1]) < float(vl[0]) + 0.05                                                 # it does learn something
    # resume: the checkpoint overrides the command line, logs are appended, the epoch counter goes on
    out = run('train.py', common + ['--max-num-epochs', '3'], cwd)
    assert 'loading state from checkpoint...' in out and 'Epoch: 002' in out and 'Epoch: 001' not in out
    ck3 = torch.load(os.path.join(cwd, 'ck.pt'), map_location='cpu', weights_only=False)
    assert ck3['epoch'] == 3 and ck3['lrs-num-restarts'] == 3
    assert len(open(os.path.join(cwd, 'tl.txt')).read().split('\n')) - 1 == 3 * 2
    # two-net ensemble on patient 2
    out = run('test_ensemble.py', ['data.npz', 'out.npz', '--pats', '2', '--nets', 'ck.pt', 'best.pt', '--times', 't.txt'], cwd)
    assert 'Length of testing dataset: 4' in out
    z = np.load(os.path.join(cwd, 'out.npz'))
    assert z['nn-segs'].dtype == np.uint8 and z['nn-segs'].shape == (4, H, W) and int(z['nn-segs'].max()) < NC
    assert z['nn-heats'].dtype == np.float32 and z['nn-heats'].shape == (4, L, H, W)
    assert float(z['nn-heats'].min()) >= 0.0 and float(z['nn-heats'].max()) <= 1.0 + 1e-6
    assert int(z['land-names/num-lands']) == L and str(z['land-names/land-01']) == 'GSN-r'
    times = open(os.path.join(cwd, 't.txt')).read().split('\n')
    assert len(times) - 1 == 4 and all(fl.match(x) for x in times[:-1])
    # the file holds what the library computes for the same nets and images
    from dfl_amd import dataset, util
    nets = []
    for f in ('ck.pt', 'best.pt'):
        st = torch.load(os.path.join(cwd, f), map_location='cpu', weights_only=False)
        n = dfl_amd.UNet(n_classes=NC, depth=3, wf=3, batch_norm=True, padding=True, max_pool=False, num_lands=L)
        n.load_state_dict(st['model-state-dict'])
        nets.append(n.to('cuda').eval())
    ds = dataset.get_dataset(os.path.join(cwd, 'data.npz'), [2], num_classes=NC, pad_img_dim=48, no_seg=True)
    with torch.no_grad():
        x = ds[1][0][None]
        outs = [n(x) for n in nets]
        labels, heats, _ = util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (H, W))
    assert np.array_equal(labels.cpu().numpy().reshape(H, W), z['nn-segs'][1])
    np.testing.assert_allclose(heats.cpu().numpy().reshape(L, H, W), z['nn-heats'][1], rtol=0, atol=1e-6)

    # hard Dice CSV (compute_actual_dice_on_test.py:61,93) and landmark CSV (est_lands_csv.py:77,127) from those outputs
    run('compute_actual_dice_on_test.py', ['data.npz', 'out.npz', 'nn-segs', 'dice.csv', '2', '--num-classes', str(NC)], cwd)
    rows = open(os.path.join(cwd, 'dice.csv')).read().split('\n')
    assert rows[0] == 'pat,proj,label,dice' and rows[-1] == '' and len(rows) - 2 == 4 * (NC - 1)
    gt = np.load(os.path.join(cwd, 'data.npz'))['02/segs']
    k = 1
    for proj in range(4):
        for l in range(1, NC):
            e, g_ = z['nn-segs'][proj] == l, gt[proj] == l
            tot = int(e.sum()) + int(g_.sum())
            dsc = 1.0 if tot == 0 else 2.0 * int((e & g_).sum()) / tot
            assert rows[k] == '{},{},{},{:.2f}'.format(2, proj, l, dsc), (rows[k], dsc)
            k += 1
    run('est_lands_csv.py', ['out.npz', 'nn-heats', '--use-seg', 'nn-segs', '--pat', '2', '--out', 'lands.csv'], cwd)
    rows = open(os.path.join(cwd, 'lands.csv')).read().split('\n')
    assert rows[0] == 'pat,proj,land,row,col,time' and len(rows) - 2 == 4 * L
    want = util.est_lands(torch.from_numpy(z['nn-heats']).cuda(), torch.from_numpy(z['nn-segs']).cuda(), [1, 2, 1]).cpu()
    k = 1
    for proj in range(4):
        for l in range(L):
            f_ = rows[k].split(',')
            assert [int(v) for v in f_[:5]] == [2, proj, l, int(want[proj, l, 0]), int(want[proj, l, 1])] and float(f_[5]) >= 0
            k += 1
