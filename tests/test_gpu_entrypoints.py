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


def read_container(path):
    """name -> array of an .npz or (through the dependency-free reader) an .h5 file."""
    if path.endswith('.npz'):
        return np.load(path)
    from dfl_amd import h5lite

    class _R:
        def __init__(self):
            self.f = h5lite.File(path, 'r')

        def __getitem__(self, k):
            v = self.f[k][()]
            return np.array(v.decode()) if isinstance(v, bytes) else np.asarray(v)
    return _R()


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
    if path.endswith('.npz'):
        np.savez(path, **d)
        return
    from dfl_amd import h5lite                  # the reference's layout as real HDF5 (hdf5_layouts/Readme.md:105-117)
    with h5lite.File(path, 'w') as f:
        for k, v in d.items():
            if k.endswith('/segs'):
                f.create_dataset(k, data=v, chunks=(1, H, W), compression='gzip', compression_opts=9)
            elif v.dtype.kind == 'U':
                f[k] = str(v)
            else:
                f[k] = v


def run(script, args, cwd):
    p = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    return p.stdout


@pytest.mark.parametrize('ext', ['npz', 'h5'])
def test_train_resume_and_ensemble(tmp_path, ext):
    cwd = str(tmp_path)
    data, outf = 'data.' + ext, 'out.' + ext
    make_file(os.path.join(cwd, data))
    common = [data, '--train-pats', '1', '--valid-pats', '2', '--num-classes', str(NC), '--unet-img-dim', '48',
              '--batch-size', '4', '--unet-num-lvls', '3', '--unet-init-feats-exp', '3', '--unet-batch-norm', '--unet-padding',
              '--unet-no-max-pool', '--use-lands', '--nesterov', '--wgt-decay', '1e-4', '--init-lr', '0.05',
              '--cos-anneal-epochs', '1', '--cos-growth', '1', '--save-restart-net', 'restart', '--checkpoint-net', 'ck.pt',
              '--best-net', 'best.pt', '--train-loss-txt', 'tl.txt', '--valid-loss-txt', 'vl.txt']
    out = run('train.py', common + ['--max-num-epochs', '2'], cwd)
    assert 'num. lands read from file: 3' in out and 'Epoch: 001' in out and 'Exiting - maximum number of epochs performed!' in out
    assert 'Saving network before restart 1 to restart_00.pt' in out
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
    out = run('test_ensemble.py', [data, outf, '--pats', '2', '--nets', 'ck.pt', 'best.pt', '--times', 't.txt'], cwd)
    assert 'Length of testing dataset: 4' in out
    z = read_container(os.path.join(cwd, outf))
    if ext == 'h5':
        from dfl_amd import h5lite
        assert h5lite.is_hdf5(os.path.join(cwd, outf))
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
    ds = dataset.get_dataset(os.path.join(cwd, data), [2], num_classes=NC, pad_img_dim=48, no_seg=True)
    with torch.no_grad():
        x = ds[1][0][None]
        outs = [n(x) for n in nets]
        labels, heats, _ = util.ensemble_reduce([o[0] for o in outs], [o[1] for o in outs], (H, W))
    assert np.array_equal(labels.cpu().numpy().reshape(H, W), z['nn-segs'][1])
    np.testing.assert_allclose(heats.cpu().numpy().reshape(L, H, W), z['nn-heats'][1], rtol=0, atol=1e-6)

    # hard Dice CSV (compute_actual_dice_on_test.py:61,93) and landmark CSV (est_lands_csv.py:77,127) from those outputs
    run('compute_actual_dice_on_test.py', [data, outf, 'nn-segs', 'dice.csv', '2', '--num-classes', str(NC)], cwd)
    rows = open(os.path.join(cwd, 'dice.csv')).read().split('\n')
    assert rows[0] == 'pat,proj,label,dice' and rows[-1] == '' and len(rows) - 2 == 4 * (NC - 1)
    gt = read_container(os.path.join(cwd, data))['02/segs']
    k = 1
    for proj in range(4):
        for l in range(1, NC):
            e, g_ = z['nn-segs'][proj] == l, gt[proj] == l
            tot = int(e.sum()) + int(g_.sum())
            dsc = 1.0 if tot == 0 else 2.0 * int((e & g_).sum()) / tot
            assert rows[k] == '{},{},{},{:.2f}'.format(2, proj, l, dsc), (rows[k], dsc)
            k += 1
    run('est_lands_csv.py', [outf, 'nn-heats', '--use-seg', 'nn-segs', '--pat', '2', '--out', 'lands.csv'], cwd)
    rows = open(os.path.join(cwd, 'lands.csv')).read().split('\n')
    assert rows[0] == 'pat,proj,land,row,col,time' and len(rows) - 2 == 4 * L
    want = util.est_lands(torch.from_numpy(np.asarray(z['nn-heats'])).cuda(), torch.from_numpy(np.asarray(z['nn-segs'])).cuda(), [1, 2, 1]).cpu()
    k = 1
    for proj in range(4):
        for l in range(L):
            f_ = rows[k].split(',')
            assert [int(v) for v in f_[:5]] == [2, proj, l, int(want[proj, l, 0]), int(want[proj, l, 1])] and float(f_[5]) >= 0
            k += 1


def test_two_rank_train_matches_the_sequential_emulation(tmp_path):
    """train.py under torchrun with two ranks (gloo transport, both on GPU 0; RCCL refuses two ranks on one device): its
    training-loss and validation-loss logs must equal a single-process emulation of the same data-parallel run -- the
    same seed, every global minibatch cut into the two contiguous shards, gradients of the shards averaged, BatchNorm
    statistics per shard with rank 0's running statistics kept -- and rank 0 alone writes the files."""
    import random
    import dfl_amd
    from dfl_amd import dataset, util
    cwd = str(tmp_path)
    make_file(os.path.join(cwd, 'data.npz'), n_per_pat=(12, 5))
    B, SEED, EPOCHS = 3, 77, 2
    args = ['data.npz', '--train-pats', '1', '--valid-pats', '2', '--num-classes', str(NC), '--unet-img-dim', '48',
            '--batch-size', str(B), '--unet-num-lvls', '3', '--unet-init-feats-exp', '3', '--unet-batch-norm', '--unet-padding',
            '--unet-no-max-pool', '--use-lands', '--nesterov', '--wgt-decay', '1e-4', '--init-lr', '0.05',
            '--cos-anneal-epochs', '2', '--checkpoint-net', 'ck.pt', '--best-net', 'best.pt', '--train-loss-txt', 'tl.txt',
            '--valid-loss-txt', 'vl.txt', '--max-num-epochs', str(EPOCHS), '--seed', str(SEED), '--dist-backend', 'gloo']
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(ROOT, 'train.py')] + args
    p = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    assert p.stdout.count('Epoch: 000') == 1 and '(2 ranks, global batch 6)' in p.stdout      # rank 0 alone reports
    tl = [float(v) for v in open(os.path.join(cwd, 'tl.txt')).read().split()]
    vl = [float(v) for v in open(os.path.join(cwd, 'vl.txt')).read().split()]
    assert len(tl) == EPOCHS * 2 and len(vl) == EPOCHS                                       # 12 images / (3 x 2 ranks)
    ck = torch.load(os.path.join(cwd, 'ck.pt'), map_location='cpu', weights_only=False)
    assert list(ck.keys()) == CHECKPOINT_KEYS and ck['batch-size'] == B and ck['epoch'] == EPOCHS

    # ---- the same run, one process, shards in sequence
    dev = torch.device('cuda')
    random.seed(SEED)
    torch.manual_seed(SEED)
    tr = dataset.get_dataset(os.path.join(cwd, 'data.npz'), [1], num_classes=NC, pad_img_dim=48, device=dev)
    va = dataset.get_dataset(os.path.join(cwd, 'data.npz'), [2], num_classes=NC, pad_img_dim=48, device=dev)
    net = dfl_amd.UNet(n_classes=NC, depth=3, wf=3, batch_norm=True, padding=True, max_pool=False, num_lands=L).to(dev)
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
    opt = dfl_amd.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=True)
    sched = dfl_amd.WarmRestartLR(opt, init_run_period_epochs=2, growth_factor=2)
    etl, evl = [], []
    for epoch in range(EPOCHS):
        st = random.getstate()
        shards = []
        for r in range(2):
            random.setstate(st)
            shards.append(list(tr.batches(B, shuffle=True, shard=(r, 2))))
        net.train()
        seen = 0
        for step in range(len(shards[0])):
            opt.zero_grad()
            vals = []
            keep = None
            for r in range(2):
                projs, masks, _, heats = shards[r][step]
                heats = util._squeeze_heats(heats)
                out = net(projs)
                loss = crit((dfl_amd.center_crop(out[0], masks.shape), dfl_amd.center_crop(out[1], heats.shape)), (masks, heats))
                (loss * 0.5).backward()
                vals.append(loss.item())
                if r == 0:
                    keep = [b.clone() for b in net.buffers()]
            with torch.no_grad():
                for b, k in zip(net.buffers(), keep):          # rank 0's running statistics
                    b.copy_(k)
            opt.step()
            seen += 2 * B
            sched.intra_epoch_step(seen / len(tr))
            etl.append(sum(vals) / 2)
        m, _ = util.test_dataset(va, net, dev=dev, num_lands=L)
        evl.append(float(m))
        sched.step()
    np.testing.assert_allclose(tl, etl, rtol=0, atol=2e-5)
    np.testing.assert_allclose(vl, evl, rtol=0, atol=5e-4)      # eval-mode losses also see the running statistics' rounding
    for k, v in net.state_dict().items():
        if v.dtype.is_floating_point:
            np.testing.assert_allclose(ck['model-state-dict'][k].numpy(), v.cpu().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)
