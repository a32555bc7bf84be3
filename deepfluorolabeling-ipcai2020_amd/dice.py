"""Soft-Dice and Dice + NCC heat-map losses on the GPU (drop-in for the reference's train_test_code/dice.py).

``DiceLoss2D(skip_bg)(input, target)`` and ``DiceAndHeatMapLoss2D(skip_bg, heatmap_wgt)((seg, heat), (tseg, theat))``
keep the reference's constructor defaults, argument meaning and asserts (dice.py:14-20, 57-67).  The value and the
closed-form gradient come from one call of ``dfl_dice_ncc_loss`` (HIP kernels in csrc/loss.hip); the inputs may be the
strided views ``util.center_crop`` returns -- no copy is made.
"""
import ctypes as C

import torch
import torch.nn.modules.loss

from . import _native as nat

__all__ = ['DiceLoss2D', 'DiceAndHeatMapLoss2D']


def _view4(t, name):
    if not t.is_cuda:
        raise RuntimeError('%s must be a CUDA/HIP tensor: the loss runs on the GPU only (no CPU fallback)' % name)
    if t.dim() != 4:
        raise RuntimeError('%s must be 4-D [B,C,H,W]' % name)
    if t.dtype != torch.float32:
        t = t.float()
    if t.stride(3) != 1 and t.shape[3] != 1:
        t = t.contiguous()
    return t


def _window_of(t, v):
    """(full shape, r0, c0) when `t` (whose kernel view is `v`) is a centre-crop-like window of a contiguous 4-D tensor
    with the same batch and channel counts; None otherwise."""
    b = t._base
    if b is None or v.data_ptr() != t.data_ptr() or b.dim() != 4 or not b.is_contiguous() or b.dtype != torch.float32:
        return None
    if tuple(b.shape[:2]) != tuple(t.shape[:2]) or t.stride() != b.stride():
        return None
    off = t.storage_offset() - b.storage_offset()
    H, W = b.shape[2], b.shape[3]
    r0, c0 = divmod(off, W)
    if off < 0 or r0 + t.shape[2] > H or c0 + t.shape[3] > W:
        return None
    return tuple(b.shape), r0, c0


_ZB_POOL = {}


def _zero_border_buffer(dev, shape, window):
    """A full-size fp32 tensor whose border around `window` = (r0, c0, h, w) is zero: the loss gradient is written into
    its interior.  Buffers are pooled per (device, shape, window) and handed out only while nothing else refers to them
    (a gradient still held by autograd or by the caller keeps its buffer out of circulation).  The library writes the
    interior through a raw pointer, which leaves torch's version counter alone: a counter that HAS moved means some torch
    op wrote into the buffer in place (a gradient hook scaling or clamping it) and may have dirtied the border,
    so the buffer is zeroed again before it goes out."""
    key = (str(dev), shape, window)
    pool = _ZB_POOL.setdefault(key, [])
    use_count = getattr(torch.Tensor, '_use_count', None)
    for buf in pool if use_count is not None else ():
        if use_count(buf) == 1:
            if buf._version != buf._dfl_version:
                buf.zero_()
                buf._dfl_version = buf._version
            return buf
    buf = torch.zeros(shape, dtype=torch.float32, device=dev)
    buf._dfl_zero_border = window
    buf._dfl_version = buf._version
    if len(pool) < 4 and use_count is not None:
        pool.append(buf)
    return buf


class _LossFn(torch.autograd.Function):
    """Value in forward (dfl_dice_ncc_loss stage 1: sums + value, gradient coefficients stay in the scratch), gradient in
    backward (stage 2: one kernel, scaled by the incoming gradient on the device, written -- when the input is a window of
    the network output, i.e. util.center_crop -- straight into a zero-bordered full-size tensor)."""

    @staticmethod
    def forward(ctx, seg, heat, tseg, theat, skip_bg, dice_wgt, heat_wgt):
        lib = nat.lib()
        seg_v, tseg_v = _view4(seg.detach(), 'input'), _view4(tseg.detach(), 'target')
        if seg_v.shape != tseg_v.shape:
            raise RuntimeError('segmentation input %s and target %s differ in shape' % (tuple(seg.shape), tuple(tseg.shape)))
        B, Cc, h, w = seg_v.shape
        a = nat.LossArgs()
        a.seg, a.tseg = seg_v.data_ptr(), tseg_v.data_ptr()
        a.seg_sN, a.seg_sC, a.seg_sH = seg_v.stride(0), seg_v.stride(1), seg_v.stride(2)
        a.tseg_sN, a.tseg_sC, a.tseg_sH = tseg_v.stride(0), tseg_v.stride(1), tseg_v.stride(2)
        L = 0
        heat_v = theat_v = None
        if heat is not None:
            heat_v, theat_v = _view4(heat.detach(), 'heat-map input'), _view4(theat.detach(), 'heat-map target')
            if heat_v.shape != theat_v.shape or heat_v.shape[0] != B or heat_v.shape[2:] != seg_v.shape[2:]:
                raise RuntimeError('heat-map shapes %s / %s do not match' % (tuple(heat.shape), tuple(theat.shape)))
            L = heat_v.shape[1]
            a.heat, a.theat = heat_v.data_ptr(), theat_v.data_ptr()
            a.heat_sN, a.heat_sC, a.heat_sH = heat_v.stride(0), heat_v.stride(1), heat_v.stride(2)
            a.theat_sN, a.theat_sC, a.theat_sH = theat_v.stride(0), theat_v.stride(1), theat_v.stride(2)
        dev = seg_v.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        sums = torch.empty(int(lib.dfl_loss_scratch_doubles(B, Cc, L)), dtype=torch.float64, device=dev)
        a.loss, a.sums = loss.data_ptr(), sums.data_ptr()
        a.B, a.C, a.L, a.h, a.w = B, Cc, L, h, w
        a.skip_bg = 1 if skip_bg else 0
        a.dice_wgt, a.heat_wgt = dice_wgt, heat_wgt
        a.stage = 1
        nat.check(lib.dfl_dice_ncc_loss(C.addressof(a), torch.cuda.current_stream().cuda_stream), 'dfl_dice_ncc_loss')
        ctx.args = a
        ctx.keep = (seg_v, tseg_v, heat_v, theat_v, sums, loss)       # the tensors the gradient stage reads
        ctx.want = (seg.requires_grad, heat is not None and heat.requires_grad)
        ctx.windows = (_window_of(seg, seg_v), _window_of(heat, heat_v) if heat is not None else None)
        ctx.dims = (B, Cc, L, h, w)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = nat.lib()
        a = ctx.args
        B, Cc, L, h, w = ctx.dims
        dev = ctx.keep[0].device
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        a.stage = 2
        a.grad_scale = gs.data_ptr()
        outs = []
        for want, win, nch, which in ((ctx.want[0], ctx.windows[0], Cc, 'dseg'), (ctx.want[1], ctx.windows[1], L, 'dheat')):
            ptr, strides, out = None, (0, 0, 0), None
            if want:
                if win is not None:
                    shape, r0, c0 = win
                    buf = _zero_border_buffer(dev, shape, (r0, c0, h, w))
                    H, W = shape[2], shape[3]
                    ptr = buf.data_ptr() + 4 * (r0 * W + c0)
                    strides = (nch * H * W, H * W, W)
                    out = buf[..., r0:r0 + h, c0:c0 + w]
                else:
                    out = torch.empty((B, nch, h, w), dtype=torch.float32, device=dev)
                    ptr = out.data_ptr()
            setattr(a, which, ptr)
            for name, v in zip(('_sN', '_sC', '_sH'), strides):
                setattr(a, which + name, v)
            outs.append(out)
        if outs[0] is not None or outs[1] is not None:
            nat.check(lib.dfl_dice_ncc_loss(C.addressof(a), torch.cuda.current_stream().cuda_stream), 'dfl_dice_ncc_loss')
        return outs[0], outs[1], None, None, None, None, None


class DiceLoss2D(torch.nn.modules.loss._Loss):
    def __init__(self, skip_bg=True):
        super().__init__()
        self.skip_bg = skip_bg

    def forward(self, input, target):
        return _LossFn.apply(input, None, target, None, self.skip_bg, 1.0, 0.0)


class DiceAndHeatMapLoss2D(torch.nn.modules.loss._Loss):
    def __init__(self, skip_bg=True, heatmap_wgt=0.5):
        super().__init__()
        self.dice_loss = DiceLoss2D(skip_bg=skip_bg)
        assert (heatmap_wgt > 1.0e-8) and (heatmap_wgt < (1 + 1.0e-8))
        self.heatmap_wgt = heatmap_wgt
        self.dice_wgt = 1 - heatmap_wgt

    def forward(self, input, target):
        return _LossFn.apply(input[0], input[1], target[0], target[1], self.dice_loss.skip_bg,
                             float(self.dice_wgt), float(self.heatmap_wgt))
