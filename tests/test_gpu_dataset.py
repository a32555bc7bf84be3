"""GPU-side input pipeline (dfl_prep_batch behind dfl_amd.dataset) against the reference loader's items (golden
'dataset', written by tools/gen_golden.py from train_test_code/dataset.py) and against the oracle on random data."""
import math
import os

import numpy as np
import pytest
import torch

import dfl_amd
from dfl_amd import dataset as D
from conftest import load_golden
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_items_match_reference_loader():
    g = load_golden('dataset')
    projs, segs, lands = _t(g['projs']), _t(g['segs']), _t(g['lands'])
    H, W = projs.shape[-2:]
    lands_m = R.mark_oob_landmarks(lands, H, W)             # what get_dataset hands to the dataset class
    ds = D.DeviceDataSet(projs.unsqueeze(1), segs, lands_m, proj_pad_dim=48, num_classes=7, device=DEV)
    assert len(ds) == int(g['len']) and ds.extra_pad == int(g['pad_48_46'])
    for i in range(3):
        p, s, l, h = ds[i]
        np.testing.assert_allclose(p.cpu().numpy(), g['item%d_p' % i], rtol=1e-5, atol=2e-6)
        assert np.array_equal(s.cpu().numpy(), g['item%d_s' % i])
        assert np.array_equal(l.cpu().numpy(), g['item%d_l' % i])
        np.testing.assert_allclose(h.cpu().numpy(), g['item%d_h' % i], rtol=2e-6, atol=1e-9)
    assert float(ds[0][3][13].abs().max()) == 0.0            # out-of-view landmark -> zero map
    # a whole batch in one call == the items
    x, m, l, h = next(ds.batches(3))
    for i in range(3):
        np.testing.assert_allclose(x[i].cpu().numpy(), g['item%d_p' % i], rtol=1e-5, atol=2e-6)
        assert np.array_equal(m[i].cpu().numpy(), g['item%d_s' % i])
        np.testing.assert_allclose(h[i].cpu().numpy(), g['item%d_h' % i], rtol=2e-6, atol=1e-9)
    assert h.shape == (3, 14, 1, H, W)                       # train.py:399-402 squeezes this form


@pytest.mark.parametrize('H,W,pad_dim,C,L', [(184, 184, 192, 7, 14), (37, 53, 60, 4, 3), (20, 20, 0, 2, 1)])
def test_batch_matches_oracle(H, W, pad_dim, C, L):
    g = torch.Generator().manual_seed(H * 1000 + W)
    B = 5
    projs = torch.rand(B, 1, H, W, generator=g) * 3000 + 100
    segs = torch.randint(0, C, (B, H, W), generator=g)
    lands = torch.stack([torch.rand(B, L, generator=g) * (W + 20) - 10, torch.rand(B, L, generator=g) * (H + 20) - 10], 1)
    lands[0, :, 0] = torch.tensor([0.0, float(H - 1)])[:2] if L >= 1 else lands[0, :, 0]   # border values stay in view
    ds = D.DeviceDataSet(projs, segs, lands, proj_pad_dim=pad_dim, num_classes=C, device=DEV)
    x, m, l, h = ds._prepare(list(range(B)))
    pad = R.calc_pad_amount(pad_dim, W) if pad_dim else 0
    lm = R.mark_oob_landmarks(lands, H, W)
    for i in range(B):
        want = R.preprocess_proj(projs[i], pad)
        np.testing.assert_allclose(x[i].cpu().numpy(), want.numpy(), rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(h[i].cpu().numpy(), R.gaussian_heatmaps(lm[i], H, W).numpy(), rtol=3e-6, atol=1e-9)
    assert np.array_equal(m.cpu().numpy(), R.one_hot_masks(segs, C).numpy())
    assert abs(float(x.mean())) < 1e-5 and abs(float(x[0].std()) - 1.0) < 1e-4


def test_get_dataset_from_npz(tmp_path):
    g = load_golden('dataset')
    path = os.path.join(tmp_path, 'toy.npz')
    raw = np.where(np.isinf(g['lands']), -5.0, g['lands']).astype(np.float32)   # the file holds finite coordinates; out of view here
    np.savez(path, **{'01/projs': g['projs'][:2], '01/segs': g['segs'][:2], '01/lands': raw[:2],
                      '03/projs': g['projs'][2:], '03/segs': g['segs'][2:], '03/lands': raw[2:],
                      'land-names/num-lands': np.int64(14)})
    assert D.get_num_lands_from_dataset(path) == 14
    ds = D.get_dataset(path, [1, 3], num_classes=7, pad_img_dim=48, device=DEV)
    assert len(ds) == 3 and ds.rob_orig_img_shape == (46, 46) and not ds.rob_data_is_scaled
    for i in range(3):
        p, s, l, h = ds[i]
        np.testing.assert_allclose(p.cpu().numpy(), g['item%d_p' % i], rtol=1e-5, atol=2e-6)
        assert np.array_equal(l.cpu().numpy(), g['item%d_l' % i])          # out-of-view landmarks marked inf by get_dataset
    tr, va, ti, vi = D.get_dataset(path, [1, 3], num_classes=7, pad_img_dim=48, train_valid_split=0.6, device=DEV)
    assert len(tr) == 2 and len(va) == 1 and sorted(ti + vi) == [0, 1, 2]
    with pytest.raises(NotImplementedError):
        D.get_dataset(path, [1], num_classes=7, data_aug=True, device=DEV)
