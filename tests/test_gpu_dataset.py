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


def test_landmark_extraction_matches_reference_script():
    """dfl_est_lands against the rows est_lands_csv.py wrote for the same heat maps (golden 'est_lands')."""
    from dfl_amd import util as U
    g = load_golden('est_lands')
    heats, segs = _t(g['heats']).to(DEV), _t(g['segs']).to(DEV)
    labels = [int(v) for v in g['label_for_land']]
    rc, ncc = U.est_lands(heats, segs, labels, return_ncc=True)
    assert np.array_equal(rc.cpu().numpy(), g['rc_masked'])
    assert np.array_equal(U.est_lands(heats).cpu().numpy(), g['rc_plain'])
    want, wncc = R.est_landmarks(_t(g['heats']), _t(g['segs']), labels, return_ncc=True)
    np.testing.assert_allclose(ncc.cpu().numpy(), wncc.numpy(), atol=2e-5)
    # 5-D heat maps as the network's loader hands them, and a label no pixel carries
    assert np.array_equal(U.est_lands(heats.unsqueeze(2)).cpu().numpy(), g['rc_plain'])
    none = U.est_lands(heats, segs, [9] * 14)
    assert int(none.max()) == -1


def test_landmark_extraction_random_vs_oracle():
    from dfl_amd import util as U
    g = torch.Generator().manual_seed(3)
    B, L, H, W = 2, 5, 64, 80
    heats = 1e-4 * torch.rand(B, L, H, W, generator=g)
    segs = torch.randint(0, 3, (B, H, W), generator=g)
    for i in range(B):
        for l in range(L):
            r, c = int(torch.randint(14, H - 14, (1,), generator=g)), int(torch.randint(14, W - 14, (1,), generator=g))
            heats[i, l] += R.gaussian_heatmaps(torch.tensor([[float(c)], [float(r)]]), H, W)[0, 0]
    heats[0, 0] = heats[0, 0].max()          # constant map: ties -> index 0, zero variance -> ncc 0 -> rejected
    labels = [1, None, 2, -1, 0]
    got = U.est_lands(heats.to(DEV), segs.to(DEV), labels).cpu()
    want = R.est_landmarks(heats, segs, [-1 if v is None else v for v in labels])
    assert np.array_equal(got.numpy(), want.numpy())


def test_hard_dice_exact():
    from dfl_amd import util as U
    g = torch.Generator().manual_seed(9)
    B, H, W, C = 3, 70, 90, 7
    est = torch.randint(0, C, (B, H, W), generator=g)
    gt = torch.randint(0, C, (B, H, W), generator=g)
    est[1][est[1] == 4] = 0
    gt[1][gt[1] == 4] = 0                      # label 4 absent from both in image 1 -> 1.0
    gt[2] = est[2]                             # identical -> 1.0 everywhere
    d, counts = U.hard_dice(est.to(DEV), gt.to(DEV), C, return_counts=True)
    for i in range(B):
        want = R.hard_dice(est[i], gt[i], C)
        assert d[i].cpu().tolist() == want     # integer counts, one division: exact
    assert d[1, 3].item() == 1.0 and float(d[2].min()) == 1.0
    assert int(counts[0, :, 0].sum()) == H * W and int(counts[0, :, 1].sum()) == H * W
    t = load_golden('trajectory')              # the value the reference pipeline produced for the toy run is a mean of these
    assert 'hard_dice' in t


def test_end_to_end_example():
    """examples/train_toy.py: loader -> two short trainings with warm restarts -> validation -> ensemble -> hard Dice ->
    landmark extraction, all through the library; it must learn something in a few epochs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('train_toy', os.path.join(os.path.dirname(__file__), '..', 'examples', 'train_toy.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main(epochs=30, images=24, math='bf16x3', quiet=True)
    first, last = res['train_loss_first_last']
    assert last < first - 0.15, res
    assert res['ensemble_mean_hard_dice'] > 0.2, res        # 24 random toy images do not generalise far; chance is ~0.1
    assert res['landmarks_total'] == 8 * 14
