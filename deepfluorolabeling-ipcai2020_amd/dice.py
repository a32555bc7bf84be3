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


class _LossFn(torch.autograd.Function):
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
        want_seg = seg.requires_grad
        want_heat = heat is not None and heat.requires_grad
        loss = torch.empty((), dtype=torch.float32, device=dev)
        sums = torch.empty(int(lib.dfl_loss_scratch_doubles(B, Cc, L)), dtype=torch.float64, device=dev)
        dseg = torch.empty((B, Cc, h, w), dtype=torch.float32, device=dev) if want_seg else None
        dheat = torch.empty((B, L, h, w), dtype=torch.float32, device=dev) if want_heat else None
        a.loss, a.sums, a.dseg, a.dheat = loss.data_ptr(), sums.data_ptr(), nat.ptr(dseg), nat.ptr(dheat)
        a.B, a.C, a.L, a.h, a.w = B, Cc, L, h, w
        a.skip_bg = 1 if skip_bg else 0
        a.dice_wgt, a.heat_wgt = dice_wgt, heat_wgt
        nat.check(lib.dfl_dice_ncc_loss(C.addressof(a), torch.cuda.current_stream().cuda_stream), 'dfl_dice_ncc_loss')
        ctx.dseg, ctx.dheat = dseg, dheat
        ctx.has_heat = heat is not None
        return loss

    @staticmethod
    def backward(ctx, g):
        dseg = ctx.dseg * g if ctx.dseg is not None else None
        dheat = ctx.dheat * g if ctx.dheat is not None else None
        return dseg, dheat, None, None, None, None, None


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
