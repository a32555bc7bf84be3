"""Host entry points (train.py, test_ensemble.py): command-line surface and file formats against the reference's
(train_test_code/train.py:24-100, :463-513; test_ensemble.py:20-37), without a GPU."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)

# option -> default, as declared by the reference's parser (train.py:29-100); store_true flags default to False
TRAIN_FLAGS = {
    'train_pats': None, 'valid_pats': None, 'num_classes': None, 'batch_size': 1, 'unet_img_dim': 364,
    'checkpoint_net': 'zz_checkpoint.pt', 'best_net': 'zz_best_valid.pt', 'checkpoint_freq': 1,
    'no_save_best_valid': False, 'optim': 'sgd', 'lr_sched': 'cos', 'init_lr': 1.0e-2, 'lr_patience': 20,
    'lr_cooldown': 20, 'nesterov': False, 'momentum': 0.9, 'wgt_decay': 0, 'cos_anneal_epochs': 10, 'cos_growth': 2,
    'save_restart_net': None, 'save_after_n_restarts': 0, 'max_num_restarts': -1, 'max_num_epochs': 200,
    'train_loss_txt': 'train_iter_loss.txt', 'valid_loss_txt': 'valid_loss.txt', 'no_gpu': False, 'max_hours': -1.0,
    'unet_num_lvls': 5, 'unet_init_feats_exp': 4, 'unet_batch_norm': False, 'unet_padding': False,
    'unet_no_max_pool': False, 'unet_block_depth': 2, 'data_aug': False, 'use_lands': False, 'heat_coeff': 0.5,
    'dice_valid': False, 'unet_no_res': False, 'train_valid_split': -1.0,
}
ENSEMBLE_FLAGS = {'nets': None, 'pats': None, 'no_gpu': False, 'times': ''}
# train.py:465-510, in order
CHECKPOINT_KEYS = ['epoch', 'model-state-dict', 'optim-type', 'optimizer-state-dict', 'scheduler-state-dict', 'loss',
                   'best-valid-loss', 'save-best-valid', 'num-classes', 'depth', 'init-feats-exp', 'batch-norm', 'padding',
                   'no-max-pool', 'pad-img-size', 'batch-size', 'data-aug', 'opt-nesterov', 'opt-momentum',
                   'opt-wgt-decay', 'num-lands', 'heat-coeff', 'use-dice-valid', 'unet-use-res', 'unet-block-depth',
                   'lrs-meth', 'lrs-num-epochs', 'lrs-growth-factor', 'lrs-max-num-restarts',
                   'lrs-save-restart-net-prefix', 'lrs-save-after-n-restarts', 'lrs-num-restarts', 'lrs-patience',
                   'lrs-cooldown', 'checkpoint-freq', 'train-idx', 'valid-idx']


def test_train_parser_matches_the_reference_flags():
    import train
    ns = vars(train.build_parser().parse_args(['data.h5']))
    assert ns.pop('input_data_file_path') == 'data.h5'
    assert ns == TRAIN_FLAGS
    assert train.CHECKPOINT_KEYS == CHECKPOINT_KEYS
    # the parser main() uses = the reference's flags + this build's extensions, all optional and inert by default
    full = vars(train.build_full_parser().parse_args(['data.h5']))
    assert {k: full[k] for k in TRAIN_FLAGS} == TRAIN_FLAGS
    assert {k: v for k, v in full.items() if k not in TRAIN_FLAGS and k != 'input_data_file_path'} == train.EXTENSION_FLAGS
    # every settings key of a checkpoint is either run state or produced from the command line
    cfg = train.Settings.from_args(train.build_full_parser().parse_args(['data.h5', '--use-lands']))
    assert set(cfg) | set(train.RUN_STATE_KEYS) == set(CHECKPOINT_KEYS)
    a = train.build_parser().parse_args(['d.h5', '--train-pats', '1,2', '--valid-pats', '3', '--num-classes', '7', '--use-lands',
                                         '--unet-no-max-pool', '--unet-batch-norm', '--unet-padding', '--nesterov',
                                         '--wgt-decay', '1e-4', '--max-num-restarts', '4'])
    assert (a.train_pats, a.num_classes, a.use_lands, a.unet_no_max_pool, a.wgt_decay, a.max_num_restarts) == \
        ('1,2', 7, True, True, 1e-4, 4)


def test_ensemble_parser_matches_the_reference_flags():
    import test_ensemble
    ns = vars(test_ensemble.build_parser().parse_args(['in.h5', 'out.h5']))
    assert (ns.pop('input_data_file_path'), ns.pop('output_data_file_path')) == ('in.h5', 'out.h5')
    assert ns == ENSEMBLE_FLAGS
    a = test_ensemble.build_parser().parse_args(['in.h5', 'out.h5', '--nets', 'a.pt', 'b.pt', '--pats', '1,4'])
    assert a.nets == ['a.pt', 'b.pt'] and a.pats == '1,4'


def test_entry_points_refuse_to_run_without_a_gpu(tmp_path):
    import train
    import test_ensemble
    from dfl_amd._native import DflError
    with pytest.raises(DflError):
        train.main(['d.npz', '--train-pats', '1', '--valid-pats', '2', '--no-gpu'])
    with pytest.raises(DflError):
        test_ensemble.main(['d.npz', 'o.npz', '--pats', '1', '--nets', 'a.pt', '--no-gpu'])


def test_snapshot_policy_serialises_once_per_epoch(tmp_path):
    """One serialisation per epoch; the other destinations of that epoch are copies (the reference's rule, train.py:
    515-577); nothing is written on ranks other than 0."""
    import train
    writes = []

    def writer(path):
        writes.append(path)
        open(path, 'w').write('net %d' % len(writes))
    snap = train.Snapshots(writer)
    ck, best, rst = (str(tmp_path / n) for n in ('ck.pt', 'best.pt', 'r_00.pt'))
    snap.new_epoch()
    snap.put(ck)
    snap.put(best)
    snap.put(rst)
    snap.put(ck)
    assert len(writes) == 1 and open(best).read() == open(ck).read() == open(rst).read() == 'net 1'
    snap.new_epoch()
    snap.put(best)                    # an epoch without a regular checkpoint: the best copy is written directly ...
    snap.put(ck)                      # ... and the final checkpoint is a copy of it
    assert len(writes) == 2 and open(ck).read() == 'net 2' and not os.path.exists(best + '.tmp')
    off = train.Snapshots(writer, enabled=False)
    off.new_epoch()
    off.put(str(tmp_path / 'never.pt'))
    assert len(writes) == 2 and not os.path.exists(str(tmp_path / 'never.pt'))


def test_sharded_batches_partition_the_global_minibatch():
    """Data-parallel batching: every rank draws the same permutation and takes the r-th contiguous slice of
    each global minibatch; a ragged tail is cut into equal non-empty slices."""
    import random
    from dfl_amd.dataset import DeviceDataSet

    class Probe(DeviceDataSet):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def _prepare(self, idx):
            return list(idx)
    for n, b, world in ((32, 4, 2), (37, 4, 4), (10, 4, 2), (7, 4, 2), (5, 1, 4)):
        per_rank = []
        for r in range(world):
            random.seed(11)
            per_rank.append(list(Probe(n).batches(b, shuffle=True, shard=(r, world))))
        random.seed(11)
        order = list(range(n))
        random.shuffle(order)
        steps = len(per_rank[0])
        assert all(len(p) == steps for p in per_rank)
        seen = []
        for s in range(steps):
            sizes = {len(p[s]) for p in per_rank}
            assert len(sizes) == 1 and sizes.pop() >= 1            # equal, non-empty shards
            glob = [i for p in per_rank for i in p[s]]
            assert glob == order[len(seen):len(seen) + len(glob)]   # contiguous slices of the same permutation, in rank order
            seen += glob
        assert len(seen) == len(set(seen)) and n - len(seen) < world * 1 + (n % (b * world)) % world + 1
    random.seed(3)
    single = list(Probe(10).batches(4, shuffle=True))
    assert [len(x) for x in single] == [4, 4, 2]                   # one rank: exactly the DataLoader's batches


def test_output_container_and_land_names_round_trip(tmp_path):
    """The .npz stand-in for the HDF5 output file keeps the reference's dataset names, dtypes and group paths
    (test_ensemble.py:126-132, util.py:300-310) and is readable by the loader's container reader."""
    from dfl_amd import dataset
    p = str(tmp_path / 'out.npz')
    f = dataset.open_output_container(p)
    g = f.create_group('land-names')
    g['num-lands'] = 2
    g['land-00'], g['land-01'] = 'FH-l', 'GSN-r'
    segs = f.create_dataset('nn-segs', (3, 5, 6), dtype='u1', chunks=(1, 5, 6), compression='gzip', compression_opts=9)
    heats = f.create_dataset('nn-heats', (3, 2, 5, 6), chunks=(1, 1, 5, 6), compression='gzip', compression_opts=9)
    segs[1, :, :] = 4
    heats[2, 1, :, :] = 0.25
    f.flush()
    f.close()
    z = np.load(p)
    assert z['nn-segs'].dtype == np.uint8 and z['nn-segs'].shape == (3, 5, 6) and z['nn-segs'][1].min() == 4
    assert z['nn-heats'].dtype == np.float32 and float(z['nn-heats'][2, 1, 0, 0]) == 0.25
    assert dataset.get_num_lands_from_dataset(p) == 2
    assert dataset.get_land_names_from_dataset(p) == ['FH-l', 'GSN-r']
    # any other path is written as real HDF5 by the dependency-free writer (tests/test_h5lite_cpu.py)
    from dfl_amd import h5lite
    f5 = dataset.open_output_container(str(tmp_path / 'out.h5'))
    f5.create_dataset('nn-segs', (1, 2, 2), dtype='u1', chunks=(1, 2, 2), compression='gzip', compression_opts=9)[0] = 3
    f5.close()
    assert h5lite.is_hdf5(str(tmp_path / 'out.h5')) and int(h5lite.File(str(tmp_path / 'out.h5'))['nn-segs'][0, 1, 1]) == 3
