"""GPU-resident counterpart of the reference loader (train_test_code/dataset.py), deterministic part only.

The reference's ``RandomDataAugDataSet.__getitem__`` (dataset.py:98-328) reflect-pads and standardises the projection,
synthesises L Gaussian heat maps and returns float one-hot masks -- on the host, per item, every step; train.py then
copies the float tensors to the GPU (train.py:395-403: 45 MB per batch-16 step).  Here the RAW arrays (fp32
projections, uint8 labels, 2xL landmark coordinates) live in HBM and ``dfl_prep_batch`` builds a whole batch of
network inputs and targets in two launches.  Same item protocol -- ``ds[i]`` and the batches of
``ds.batches(...)`` are the tuples ``(proj, mask, lands, heat)`` train.py:393 unpacks -- so the training loop is
unchanged.  Random augmentation (dataset.py:107-283: RNG-order and PIL dependent) is out of scope: asking for it raises.
"""
import math
import random

import numpy as np
import torch

from . import _native as nat


def calc_pad_amount(padded_dim, cur_dim):
    """dataset.py:26-40 -- odd differences round up."""
    assert padded_dim > cur_dim
    pad = (padded_dim - cur_dim) / 2
    return int(pad) + 1 if pad != int(pad) else int(pad)


class DeviceDataSet(torch.utils.data.Dataset):
    """projs [N,1,H,W] fp32, segs [N,H,W] integer labels (or [N,C,H,W] one-hot, converted), lands [N,2,L] (row 0 = x)."""

    def __init__(self, projs, segs, lands=None, proj_pad_dim=0, num_classes=None, device=None, sigma=2.5):
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if dev.type != 'cuda':
            raise nat.DflError('dataset.DeviceDataSet builds its batches with dfl_prep_batch on the GPU (no CPU path)')
        projs = torch.as_tensor(projs)
        assert projs.dim() == 4 and projs.shape[1] == 1
        self.projs = projs.to(dev, torch.float32).contiguous()
        N, _, H, W = projs.shape
        self.segs = None
        self.num_classes = num_classes
        if segs is not None:
            segs = torch.as_tensor(segs)
            if segs.dim() == 4:                      # the reference keeps float one-hot masks; labels are 1 byte per pixel
                self.num_classes = segs.shape[1]
                segs = segs.argmax(dim=1)
            assert segs.shape == (N, H, W)
            assert self.num_classes is not None and 0 < self.num_classes <= 255
            self.segs = segs.to(dev, torch.uint8).contiguous()
        self.lands = None
        if lands is not None:
            lands = torch.as_tensor(lands)
            assert lands.shape[0] == N and lands.shape[1] == 2
            self.lands = lands.to(dev, torch.float32).contiguous()
        self.extra_pad = calc_pad_amount(proj_pad_dim, W) if proj_pad_dim > 0 else 0
        self.do_norm_01_scale = True
        self.include_heat_map = self.lands is not None
        self.heat_sigma = float(sigma)
        self.prob_of_aug = 0.0
        self.dev = dev
        self._lib = nat.lib()
        self._scratch = {}

    def __len__(self):
        return self.projs.shape[0]

    def _prepare(self, idx):
        if self.prob_of_aug > 0:
            raise NotImplementedError('random data augmentation is outside the HIP path (DESIGN.md section 7)')
        idx = torch.as_tensor(idx, dtype=torch.long, device=self.dev)
        B = int(idx.numel())
        _, _, H, W = self.projs.shape
        p = self.extra_pad
        raw = self.projs.index_select(0, idx)
        x = torch.empty((B, 1, H + 2 * p, W + 2 * p), dtype=torch.float32, device=self.dev)
        a = nat.PrepArgs(proj=raw.data_ptr(), x=x.data_ptr(), B=B, H=H, W=W, pad=p, standardize=int(self.do_norm_01_scale),
                         sigma=self.heat_sigma)
        keep = [raw]
        masks = lands = heats = None
        if self.segs is not None:
            lab = self.segs.index_select(0, idx)
            masks = torch.empty((B, self.num_classes, H, W), dtype=torch.float32, device=self.dev)
            a.labels, a.masks, a.C = lab.data_ptr(), masks.data_ptr(), self.num_classes
            keep.append(lab)
        if self.lands is not None:
            lands = self.lands.index_select(0, idx)
            if self.include_heat_map:
                L = lands.shape[-1]
                heats = torch.empty((B, L, 1, H, W), dtype=torch.float32, device=self.dev)
                a.lands, a.heats, a.L = lands.data_ptr(), heats.data_ptr(), L
        sc = self._scratch.get(B)
        if sc is None:
            sc = self._scratch[B] = torch.empty(self._lib.dfl_prep_scratch_doubles(B), dtype=torch.float64, device=self.dev)
        a.scratch = sc.data_ptr()
        nat.call('dfl_prep_batch', a, torch.cuda.current_stream(self.dev).cuda_stream)
        return x, masks, lands, heats

    def __getitem__(self, i):
        x, masks, lands, heats = self._prepare([int(i)])
        return (x[0], masks[0] if masks is not None else None, lands[0] if lands is not None else None,
                heats[0] if heats is not None else None)

    def batches(self, batch_size, shuffle=False, drop_last=False, shard=None):
        """What ``DataLoader(ds, batch_size, shuffle)`` yields for the reference's dataset (train.py:365-372), already on
        the GPU.  Data parallel: ``shard=(rank, world)`` -- every rank must draw the SAME permutation (same ``random``
        seed); a step's global minibatch is ``batch_size * world`` consecutive entries of it and rank r takes the r-th
        contiguous slice of ``batch_size``.  A ragged tail is cut into equal non-empty slices (both losses
        are means over images, so equal shards keep mean-of-means exact); what does not divide is left out this epoch."""
        order = list(range(len(self)))
        if shuffle:
            random.shuffle(order)
        rank, world = shard if shard is not None else (0, 1)
        step = batch_size * world
        for s in range(0, len(order), step):
            glob = order[s:s + step]
            if len(glob) < step and drop_last:
                break
            per = batch_size if len(glob) == step else len(glob) // world
            if per == 0:
                break
            yield self._prepare(glob[rank * per:(rank + 1) * per])


RandomDataAugDataSet = DeviceDataSet      # the reference's class name (dataset.py:42)


def _open_container(path):
    """name -> array view of the preprocessed container: the reference's HDF5 layout (hdf5_layouts/Readme.md:105-117:
    '<pat>/projs', '<pat>/segs', '<pat>/lands', 'land-names/num-lands'), read with the dependency-free reader of
    h5lite.py (h5py is absent from the build and GPU images; when it IS installed it takes the files whose HDF5 features
    h5lite refuses), or an .npz with the same names (slashes kept)."""
    if str(path).endswith('.npz'):
        z = np.load(path)
        return (lambda k: z[k]), (lambda: None)
    from . import h5lite
    try:
        f = h5lite.File(path, 'r')
        f.keys()
    except h5lite.H5Error as e:
        try:
            import h5py
        except ImportError:
            raise e
        f = h5py.File(path, 'r')
    return (lambda k: f[k][()]), f.close


def get_num_lands_from_dataset(h5_file_path):
    """dataset.py:339-346."""
    get, close = _open_container(h5_file_path)
    n = int(get('land-names/num-lands'))
    close()
    return n


def get_land_names_from_dataset(h5_file_path):
    """dataset.py:348-365: the strings 'land-names/land-XX' (bytes or str in the file)."""
    get, close = _open_container(h5_file_path)
    names = []
    for l in range(int(get('land-names/num-lands'))):
        s = get('land-names/land-{:02d}'.format(l))
        if isinstance(s, np.ndarray):
            s = s.item() if s.ndim == 0 else s.tobytes()
        if isinstance(s, (bytes, np.bytes_)):
            s = s.decode()
        assert isinstance(s, str)
        names.append(s)
    close()
    return names


class NpzFile:
    """Write side of the container for hosts without h5py: the few methods of an ``h5py.File`` opened for writing that
    test_ensemble.py / util.seg_dataset* use (create_group, item assignment, create_dataset with the chunking /
    compression keywords accepted and ignored, flush, close), collected in memory and written as ONE compressed .npz with
    the same dataset names (slashes kept) on close()."""

    class _Group:
        def __init__(self, owner, prefix):
            self._o, self._p = owner, prefix

        def __setitem__(self, k, v):
            self._o._d[self._p + '/' + k] = np.asarray(v)

        def create_dataset(self, name, shape, dtype='f4', **kw):
            return self._o.create_dataset(self._p + '/' + name, shape, dtype=dtype, **kw)

    def __init__(self, path, mode='w'):
        assert mode == 'w', 'NpzFile is the write side; dataset._open_container reads'
        self._path, self._d, self._closed = path, {}, False

    def create_group(self, name):
        return NpzFile._Group(self, name)

    def __setitem__(self, k, v):
        self._d[k] = np.asarray(v)

    def create_dataset(self, name, shape, dtype='f4', **kw):
        self._d[name] = np.zeros(shape, dtype=np.dtype(dtype))
        return self._d[name]

    def flush(self):
        pass

    def close(self):
        if not self._closed:
            with open(self._path, 'wb') as f:          # (np.savez would append '.npz' to a bare path)
                np.savez_compressed(f, **self._d)
            self._closed = True


def open_output_container(path):
    """The reference writes its results with ``h5.File(path, 'w')`` (test_ensemble.py:121): here the same calls go to
    h5lite.File (HDF5 written without h5py: chunked + gzip datasets streamed chunk by chunk, readable by h5py / h5dump);
    a path ending in .npz gets an NpzFile with the same dataset names."""
    if str(path).endswith('.npz'):
        return NpzFile(path)
    from . import h5lite
    return h5lite.File(path, 'w')


def get_dataset(h5_file_path, pat_inds, num_classes, pad_img_dim=0, no_seg=False, minmax=None, data_aug=False,
                train_valid_split=None, train_valid_idx=None, dup_data_w_left_right_flip=False, device=None):
    """dataset.py:367-555 without augmentation: concatenates the patients' arrays, marks out-of-view landmarks with inf
    (:421-429), optional min/max scaling (:384-395, :513-516), optional train/validation split (:524-551)."""
    if data_aug:
        raise NotImplementedError('random data augmentation is outside the HIP path (DESIGN.md section 7)')
    if dup_data_w_left_right_flip:
        raise NotImplementedError('dup_data_w_left_right_flip is not implemented (no reference CLI default selects it)')
    get, close = _open_container(h5_file_path)
    projs, segs, lands = [], [], []
    for pat_idx in pat_inds:
        g = '{:02d}'.format(pat_idx)
        p = np.asarray(get(g + '/projs'), dtype=np.float32)
        assert p.ndim == 3
        projs.append(p)
        if not no_seg:
            segs.append(np.asarray(get(g + '/segs')))
        lands.append(np.asarray(get(g + '/lands'), dtype=np.float32))
    close()
    projs = torch.from_numpy(np.concatenate(projs))
    H, W = projs.shape[-2:]
    lands = torch.from_numpy(np.concatenate(lands))
    assert torch.all(torch.isfinite(lands))
    x, y = lands[:, 0], lands[:, 1]
    oob = (x < 0) | (x > W - 1) | (y < 0) | (y > H - 1)
    x[oob] = math.inf
    y[oob] = math.inf
    segs = torch.from_numpy(np.concatenate(segs)) if segs else None
    scaled = None
    if minmax is not None and minmax is not False:
        lo, hi = (float(projs.min()), float(projs.max())) if minmax is True else (float(minmax[0]), float(minmax[1]))
        assert hi - lo > 1.0e-6
        projs = (projs - lo) / (hi - lo)
        scaled = (lo, hi)
    projs = projs.unsqueeze(1)

    def make(sel):
        ds = DeviceDataSet(projs[sel] if sel is not None else projs,
                           None if segs is None else (segs[sel] if sel is not None else segs),
                           lands[sel] if sel is not None else lands, proj_pad_dim=pad_img_dim, num_classes=num_classes,
                           device=device)
        ds.rob_orig_img_shape = (H, W)
        ds.rob_data_is_scaled = scaled is not None
        if scaled is not None:
            ds.rob_minmax = scaled
        return ds

    if train_valid_split is not None and train_valid_split > 0:
        assert 0.0 < train_valid_split < 1.0
        n = projs.shape[0]
        num_train = int(math.ceil(train_valid_split * n))
        inds = list(range(n))
        if train_valid_idx is None or train_valid_idx[0] is None or train_valid_idx[1] is None:
            random.shuffle(inds)
            train_inds, valid_inds = inds[:num_train], inds[num_train:]
        else:
            train_inds, valid_inds = train_valid_idx
            assert len(train_inds) == num_train and len(valid_inds) == n - num_train
        return make(train_inds), make(valid_inds), train_inds, valid_inds
    return make(None)
