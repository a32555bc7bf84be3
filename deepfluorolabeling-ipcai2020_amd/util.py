"""Utility layer: centre crop, device pick, loss logs, validation and (ensemble) inference loops.

Counterpart of the reference's train_test_code/util.py.  ``center_crop`` (util.py:92-114) is index math and stays a view;
the loops (``test_dataset`` :116-165, ``test_dataset_ensemble`` :167-241, ``seg_dataset`` :243-290,
``seg_dataset_ensemble`` :293-377) keep their signatures and file outputs but run the network through the HIP
programs and do the per-image ensemble arithmetic (mean over nets, first-max argmax, per-net min-max normalised heat
maps) in one ``dfl_ensemble_reduce`` call instead of a chain of torch ops.
"""
import ctypes as C
import time

import torch
from torch.utils.data import DataLoader

from . import _native as nat
from .dice import DiceLoss2D, DiceAndHeatMapLoss2D

__all__ = ['get_device', 'center_crop', 'RunningFloatWriter', 'write_floats_to_txt', 'read_floats_from_txt',
           'test_dataset', 'test_dataset_ensemble', 'seg_dataset', 'seg_dataset_ensemble', 'ensemble_reduce']


def get_device(no_gpu=False):
    """Reference: util.py:17-36.  The HIP path needs the GPU; asking for the CPU is an error here, not a fallback."""
    if no_gpu:
        raise RuntimeError('--no-gpu: this build has no CPU path (the hot path is HIP kernels for MI355X)')
    if not torch.cuda.is_available():
        raise RuntimeError('no GPU visible: this build has no CPU path (the hot path is HIP kernels for MI355X)')
    return torch.device('cuda', torch.cuda.current_device())


class _Crop(torch.autograd.Function):
    """The centre window as a view, with a backward that knows the loss kernels: `dice._LossFn` writes its gradient
    straight into the interior of a full-size tensor whose border is (and stays) zero and returns that interior as a
    view -- which IS the gradient of the uncropped tensor, so no zero fill and no copy are left to do here (autograd's own
    slice backward spends two passes over the full-size tensor per output and step).  Any other incoming gradient takes
    the generic path."""

    @staticmethod
    def forward(ctx, img, r0, c0, h, w):
        ctx.geom = (tuple(img.shape), r0, c0, h, w)
        return img[..., r0:r0 + h, c0:c0 + w]

    @staticmethod
    def backward(ctx, g):
        shape, r0, c0, h, w = ctx.geom
        base = g._base
        if base is not None and tuple(base.shape) == shape and getattr(base, '_dfl_zero_border', None) == (r0, c0, h, w) \
                and g.stride() == base.stride() and g.storage_offset() == base.storage_offset() + r0 * shape[-1] + c0:
            return base, None, None, None, None
        out = g.new_zeros(shape)
        out[..., r0:r0 + h, c0:c0 + w] = g
        return out, None, None, None, None


def center_crop(img, dst_shape):
    """Centre window of the last two dims as a view; ``img`` itself when the sizes already match (util.py:92-114)."""
    rows, cols = img.shape[-2], img.shape[-1]
    want_r, want_c = dst_shape[-2], dst_shape[-1]
    if rows == want_r and cols == want_c:
        return img
    assert img.dim() in (2, 3, 4)
    r0 = int((rows - want_r) / 2)
    c0 = int((cols - want_c) / 2)
    if img.dim() == 4 and img.is_cuda and img.requires_grad and torch.is_grad_enabled():
        return _Crop.apply(img, r0, c0, want_r, want_c)
    return img[..., r0:r0 + want_r, c0:c0 + want_c]


def write_floats_to_txt(file_path, floats):
    with open(file_path, 'w') as out:
        out.writelines('{:.6f}\n'.format(f) for f in floats)


def read_floats_from_txt(file_path):
    with open(file_path) as f:
        return torch.Tensor([float(line.strip()) for line in f])


class RunningFloatWriter:
    """One '%.6f' line per value, flushed per write; append mode when resuming (util.py:62-89)."""

    def __init__(self, file_path, new_file=True):
        self.out = open(file_path, 'w' if new_file else 'a')

    def write(self, x):
        self.out.write('{:.6f}\n'.format(x))
        self.out.flush()

    def close(self):
        if self.out:
            self.out.flush()
            self.out.close()
            self.out = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


# ------------------------------------------------------------------------------------------------ ensemble kernel
class LateScalars:
    """Device scalars read on the host one step late.

    The reference's loop reads `loss.item()` right after `optimizer.step()` (train.py:430): the host then waits for the
    whole step and only afterwards starts queueing the next one, so the GPU idles for the host's launch latency at every
    step boundary (measured 0.15 ms of a 5.6 ms step on MI355X).  push() instead queues a copy of the scalar into pinned
    host memory behind the step and returns the value pushed `depth` calls earlier (None until then) -- by the time it is
    read the GPU is busy with the following step.  flush() waits for what is left.  Every value arrives, in order; only
    the moment the host looks at it moves.  depth = 0 is the reference's behaviour (synchronise at once)."""

    def __init__(self, depth=1):
        self.depth = int(depth)
        self._slots = []
        self._free = []

    def push(self, t):
        t = t.detach()
        if self.depth <= 0 or not t.is_cuda:
            return float(t.item())
        hit = next((k for k, (b, _) in enumerate(self._free) if b.dtype == t.dtype), None)
        if hit is not None:
            buf, ev = self._free.pop(hit)
        else:
            buf, ev = torch.empty(1, dtype=t.dtype).pin_memory(), torch.cuda.Event()
        buf.copy_(t.reshape(1), non_blocking=True)      # same dtype on both sides: one small device-to-host copy, no conversion kernel
        ev.record()
        self._slots.append((buf, ev))
        if len(self._slots) > self.depth:
            return self._pop()
        return None

    def _pop(self):
        buf, ev = self._slots.pop(0)
        ev.synchronize()
        v = float(buf[0])
        self._free.append((buf, ev))
        return v

    def flush(self):
        out = []
        while self._slots:
            out.append(self._pop())
        return out


_PTR_TABLES = {}


def _ptr_table(ptrs, dev):
    """Device table of tensor addresses.  A pageable host -> device copy blocks the host until everything queued before it is done --
    one host / GPU round trip per image of the ensemble loop (round 5: 1.78 -> 1.65 ms per 192x192 five-net image); the tables go up
    through pinned memory without blocking, and since the allocator hands the same few blocks round and round the tables are kept."""
    # (keyed by the stream too -- ADVICE r05: the copy is ordered on the stream that was current when the table was made, a consumer
    # on another stream could read it before it has landed; evicted tables are kept alive until the device is idle at the next wrap)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, tuple(ptrs))
    t = _PTR_TABLES.get(key)
    if t is None:
        if len(_PTR_TABLES) >= 256:
            torch.cuda.synchronize(dev)             # kernels queued on other streams may still read the tables about to go
            _PTR_TABLES.clear()
        t = torch.tensor(list(ptrs), dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        _PTR_TABLES[key] = t
    return t


def ensemble_reduce(seg_list, heat_list, orig_shape, raw_heat=False, want_avg_seg=False):
    """One image: list of per-net outputs [1,C,Hp,Wp] / [1,L,Hp,Wp] -> (labels uint8 [h,w], heats [L,h,w] or None,
    avg_seg [C,h,w] or None).  Arithmetic of util.py:326-373 (or :204-229 with raw_heat) in one library call."""
    lib = nat.lib()
    n = len(seg_list)
    s0 = seg_list[0]
    dev = s0.device
    Cc, Hp, Wp = s0.shape[1], s0.shape[2], s0.shape[3]
    h, w = orig_shape[-2], orig_shape[-1]
    segs = [s.detach().float().contiguous() for s in seg_list]
    a = nat.EnsembleArgs()
    seg_ptrs = _ptr_table([s.data_ptr() for s in segs], dev)
    a.seg_ptrs = seg_ptrs.data_ptr()
    labels = torch.empty((h, w), dtype=torch.uint8, device=dev)
    a.labels = labels.data_ptr()
    avg = torch.empty((Cc, h, w), dtype=torch.float32, device=dev) if want_avg_seg else None
    a.avg_seg = nat.ptr(avg)
    heats_out = None
    L = 0
    keep = [segs, seg_ptrs]
    if heat_list:
        heats = [t.detach().float().contiguous() for t in heat_list]
        L = heats[0].shape[1]
        heat_ptrs = _ptr_table([t.data_ptr() for t in heats], dev)
        heats_out = torch.empty((L, h, w), dtype=torch.float32, device=dev)
        minmax = torch.empty(2 * n * 65, dtype=torch.float32, device=dev)
        a.heat_ptrs, a.heat_out, a.minmax = heat_ptrs.data_ptr(), heats_out.data_ptr(), minmax.data_ptr()
        keep += [heats, heat_ptrs, minmax]
    a.nnets, a.C, a.L, a.Hp, a.Wp, a.h, a.w = n, Cc, L, Hp, Wp, h, w
    a.oy, a.ox = int((Hp - h) / 2), int((Wp - w) / 2)
    a.raw_heat = 1 if raw_heat else 0
    nat.check(lib.dfl_ensemble_reduce(C.addressof(a), torch.cuda.current_stream().cuda_stream), 'dfl_ensemble_reduce')
    return labels, heats_out, avg


_ENS_STREAMS = {}
ENSEMBLE_STREAMS_MAX_PIXELS = 1 << 16      # (one 192x192 image: 36864; a 1440x1440 net fills the GPU on its own)


def forward_nets(nets, x, num_lands=0):
    """[(seg, heat)] of every net of an ensemble for one batch x (util.py:326-330 calls them one after the other).  The forwards are
    independent and, for small images, latency-bound chains of ~40 short launches each: up to ENSEMBLE_STREAMS_MAX_PIXELS pixels
    every net runs on a stream of its own (round 5: five nets on one 192x192 image 1.65 -> 1.08 ms; same bits -- each net replays its
    own recorded program on its own buffers); larger inputs keep the caller's stream."""
    if (not x.is_cuda) or len(nets) < 2 or x.shape[0] * x.shape[-1] * x.shape[-2] > ENSEMBLE_STREAMS_MAX_PIXELS:
        return [_split_out(n_(x), num_lands) for n_ in nets]
    cur = torch.cuda.current_stream(x.device)
    key = (x.device.index, len(nets))
    streams = _ENS_STREAMS.get(key)
    if streams is None:
        streams = _ENS_STREAMS[key] = [torch.cuda.Stream(device=x.device) for _ in nets]
    outs = []
    for n_, s in zip(nets, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(_split_out(n_(x), num_lands))
    for s in streams:
        cur.wait_stream(s)
    for o in outs:                                  # allocated on a side stream, consumed on the caller's
        for t in o:
            if t is not None:
                t.record_stream(cur)
    return outs


def _split_out(net_out, num_lands):
    if num_lands > 0 or type(net_out) is tuple:
        return net_out[0], net_out[1]
    return net_out, None


def _squeeze_heats(heats):
    if heats.dim() > 4:
        assert heats.dim() == 5 and heats.shape[2] == 1
        heats = heats.view(heats.shape[0], heats.shape[1], heats.shape[3], heats.shape[4])
    return heats


# ------------------------------------------------------------------------------------------------ validation loops
def test_dataset(ds, net, dev=None, num_lands=0, shard=None):
    """Eval-mode per-image loss; returns (mean, std) over the data set (util.py:116-165).  Leaves the net in eval mode.
    ``shard=(rank, world)`` (data-parallel training): every rank evaluates the images i = rank mod world and the
    per-image losses are summed over ranks, so all ranks return the same numbers -- those of a single-process call on the same
    model.  With BatchNorm the replicas' running statistics must be made equal first (parallel.DataParallel.sync_buffers;
    train.py does it before every validation): they follow each rank's own shards during training."""
    dev = dev if dev is not None else next(net.parameters()).device
    crit = DiceAndHeatMapLoss2D(skip_bg=False) if num_lands > 0 else DiceLoss2D(skip_bg=False)
    rank, world = shard if shard is not None else (0, 1)
    # (the reference reads loss.item() per image, util.py:156: a host round trip with the GPU idle behind every image; the
    # values are kept on the device and read once after the loop -- same numbers, same order)
    losses_dev = torch.zeros(len(ds), dtype=torch.float32, device=dev)
    count = 0
    with torch.no_grad():
        net.eval()
        for i in range(rank, len(ds), world):
            projs, masks, lands, heats = _item(ds, i)
            projs, masks = projs.to(dev), masks.to(dev)
            seg, heat = _split_out(net(projs), num_lands)
            seg = center_crop(seg, masks.shape)
            if num_lands > 0:
                heats = _squeeze_heats(heats).to(dev)
                loss = crit((seg, center_crop(heat, heats.shape)), (masks, heats))
            else:
                loss = crit(seg, masks)
            losses_dev[i] = loss
            count += 1
    losses = losses_dev.cpu()
    assert count == len(range(rank, len(ds), world))
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend() == 'nccl':
            t = losses.to(dev)
            dist.all_reduce(t)
            losses = t.cpu()
        else:
            dist.all_reduce(losses)
    return torch.mean(losses), torch.std(losses)


def _item(ds, i):
    """Item i as the batch of one that ``DataLoader(ds, batch_size=1)`` yields (util.py:123)."""
    if hasattr(ds, '_prepare'):
        return ds._prepare([i])
    it = ds[i]
    return tuple(None if t is None else (t.unsqueeze(0) if torch.is_tensor(t) else torch.as_tensor(t).unsqueeze(0)) for t in it)


def test_dataset_ensemble(ds, nets, dev=None, num_lands=0, dice_only=False):
    """Loss of the averaged (un-normalised) ensemble output per image (util.py:167-241)."""
    dev = dev if dev is not None else next(nets[0].parameters()).device
    use_heat = (not dice_only) and num_lands > 0
    crit = DiceAndHeatMapLoss2D(skip_bg=False) if use_heat else DiceLoss2D(skip_bg=False)
    losses_dev = torch.zeros(len(ds), dtype=torch.float32, device=dev)      # (read once after the loop, see test_dataset)
    count = 0
    with torch.no_grad():
        for n_ in nets:
            n_.eval()
        for i, (projs, masks, lands, heats) in enumerate(_items(ds)):
            projs, masks = projs.to(dev), masks.to(dev)
            outs = forward_nets(nets, projs, num_lands)
            hl = [o[1] for o in outs] if num_lands > 0 else None
            _, avg_heat, avg_seg = ensemble_reduce([o[0] for o in outs], hl, masks.shape, raw_heat=True,
                                                   want_avg_seg=True)
            if use_heat:
                heats = _squeeze_heats(heats).to(dev)
                loss = crit((avg_seg.unsqueeze(0), avg_heat.unsqueeze(0)), (masks, heats))
            else:
                loss = crit(avg_seg.unsqueeze(0), masks)
            losses_dev[i] = loss
            count += 1
    losses = losses_dev.cpu()
    assert count == len(ds)
    return torch.mean(losses), torch.std(losses)


def _items(ds):
    """Batches of one, in order: what ``DataLoader(ds, batch_size=1, shuffle=False)`` yields (util.py:123,298).  The
    GPU-resident data set builds them itself -- that also covers test sets loaded without segmentations, whose item
    tuples hold None (which torch's default collate refuses)."""
    if hasattr(ds, 'batches'):
        return ds.batches(1, shuffle=False)
    return DataLoader(ds, batch_size=1, shuffle=False)


def _create_outputs(h5_f, n_items, orig_shape, num_lands):
    seg_ds = h5_f.create_dataset('nn-segs', (n_items, *orig_shape), dtype='u1', chunks=(1, *orig_shape),
                                 compression='gzip', compression_opts=9)
    heat_ds = None
    if num_lands > 0:
        heat_ds = h5_f.create_dataset('nn-heats', (n_items, num_lands, *orig_shape), chunks=(1, 1, *orig_shape),
                                      compression='gzip', compression_opts=9)
    return seg_ds, heat_ds


def seg_dataset(ds, net, h5_f, dev=None, num_lands=0):
    """Single-net labels ('nn-segs', u1) and raw cropped heat maps ('nn-heats') per image (util.py:243-290)."""
    dev = dev if dev is not None else next(net.parameters()).device
    shape = ds.rob_orig_img_shape
    seg_ds, heat_ds = _create_outputs(h5_f, len(ds), shape, num_lands)
    count = 0
    with torch.no_grad():
        net.eval()
        for i, data in enumerate(_items(ds)):
            seg, heat = _split_out(net(data[0].to(dev)), num_lands)
            labels, _, _ = ensemble_reduce([seg], None, shape)
            seg_ds[i, :, :] = labels.cpu().numpy()
            if heat_ds is not None:
                heat_ds[i, :, :, :] = center_crop(heat, shape)[0].cpu().numpy()
            count += 1
    assert count == len(ds)


def seg_dataset_ensemble(ds, nets, h5_f, dev=None, num_lands=0, times=None):
    """Ensemble labels + averaged min-max-normalised heat maps per image; per-image seconds appended to ``times``
    (timed region as util.py:321-366: H2D copy, all forwards, reduction; excludes the D2H copy and file write)."""
    dev = dev if dev is not None else next(nets[0].parameters()).device
    shape = ds.rob_orig_img_shape
    seg_ds, heat_ds = _create_outputs(h5_f, len(ds), shape, num_lands)
    count = 0
    with torch.no_grad():
        for n_ in nets:
            n_.eval()
        for i, data in enumerate(_items(ds)):
            t0 = time.time()
            projs = data[0].to(dev)
            outs = forward_nets(nets, projs, num_lands)
            hl = [o[1] for o in outs] if heat_ds is not None else None
            labels, heats, _ = ensemble_reduce([o[0] for o in outs], hl, shape)
            torch.cuda.synchronize(dev)
            if times is not None:
                times.append(time.time() - t0)
            seg_ds[i, :, :] = labels.cpu().numpy()
            if heat_ds is not None:
                heat_ds[i, :, :, :] = heats.cpu().numpy()
            count += 1
    assert count == len(ds)


def est_lands(heats, segs=None, label_for_land=None, sigma=2.5, min_ncc=0.9, return_ncc=False):
    """Landmark locations from heat maps, est_lands_csv.py:96-124 on the GPU (dfl_est_lands): heats [B,L,H,W] float,
    segs [B,H,W] integer labels or None, label_for_land[l] = label restricting landmark l (None / negative = none).
    Returns int32 [B,L,2] = (row, col), (-1,-1) where the landmark is not found."""
    from . import _native as nat
    if not heats.is_cuda:
        raise nat.DflError('util.est_lands needs the heat maps on the GPU (no CPU path)')
    h = heats.detach().to(torch.float32).contiguous()
    if h.dim() == 5 and h.shape[2] == 1:
        h = h.view(h.shape[0], h.shape[1], h.shape[3], h.shape[4])
    B, L, H, W = h.shape
    out = torch.empty((B, L, 2), dtype=torch.int32, device=h.device)
    a = nat.EstLandsArgs(heats=h.data_ptr(), rowcol=out.data_ptr(), B=B, L=L, H=H, W=W, sigma=sigma, min_ncc=min_ncc)
    keep = [h]
    if segs is not None and label_for_land is not None:
        sg = segs.detach().to(h.device, torch.uint8).contiguous()
        assert sg.shape == (B, H, W)
        lab = torch.tensor([-1 if v is None else int(v) for v in label_for_land], dtype=torch.int32, device=h.device)
        assert lab.numel() == L
        a.segs, a.label_for_land = sg.data_ptr(), lab.data_ptr()
        keep += [sg, lab]
    ncc = None
    if return_ncc:
        ncc = torch.empty((B, L), dtype=torch.float32, device=h.device)
        a.ncc = ncc.data_ptr()
    nat.call('dfl_est_lands', a, torch.cuda.current_stream(h.device).cuda_stream)
    return (out, ncc) if return_ncc else out


def hard_dice(est_labels, gt_labels, num_classes, return_counts=False):
    """compute_actual_dice_on_test.py:63-93 on the GPU (dfl_hard_dice): [B,H,W] integer label maps -> float64 [B, C-1]
    Dice per image and foreground label (1.0 when a label is absent from both)."""
    from . import _native as nat
    if not est_labels.is_cuda:
        raise nat.DflError('util.hard_dice needs the label maps on the GPU (no CPU path)')
    e = est_labels.detach().to(torch.uint8).contiguous()
    g = gt_labels.detach().to(e.device, torch.uint8).contiguous()
    assert e.shape == g.shape and e.dim() == 3
    B = e.shape[0]
    counts = torch.empty((B, num_classes, 3), dtype=torch.int64, device=e.device)
    dice = torch.empty((B, num_classes - 1), dtype=torch.float64, device=e.device)
    nat.check(nat.lib().dfl_hard_dice(e.data_ptr(), g.data_ptr(), e.shape[1] * e.shape[2], B, num_classes, counts.data_ptr(),
                                      dice.data_ptr(), torch.cuda.current_stream(e.device).cuda_stream), 'dfl_hard_dice')
    return (dice, counts) if return_counts else dice
