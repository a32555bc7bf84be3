"""2-D normalized cross-correlation on the GPU (drop-in for the reference's train_test_code/ncc.py).

``ncc_2d(X, Y)`` reduces the last two dims of ``X`` and ``Y`` exactly as ncc.py:12-38 (sample standard deviation with
N-1, product with N, +1e-8 in the denominator) using the reduction kernels of ``dfl_dice_ncc_loss``.  It is
differentiable in both arguments: the same kernels return the closed-form gradient of the mean of -(ncc + 1) / 2
(SURVEY.md Appendix F); every image's score depends on that image alone, so scaling image l's slice by -2 L gives
d ncc_l / d X_l, which backward multiplies by the incoming gradient of ncc_l.  (The training path does not come through
here: ``dice.DiceAndHeatMapLoss2D`` fuses Dice and NCC into one launch.)
"""
import ctypes as C

import torch

from . import _native as nat

__all__ = ['ncc_2d']


def _launch(Xc, Yc, L, R, Cc, want_grad):
    """ncc [L] of the image pairs (Xc[l], Yc[l]) and, on request, d ncc_l / d Xc[l] as a tensor shaped like Xc."""
    lib = nat.lib()
    a = nat.LossArgs()
    a.heat, a.theat = Xc.data_ptr(), Yc.data_ptr()
    a.heat_sN, a.heat_sC, a.heat_sH = L * R * Cc, R * Cc, Cc
    a.theat_sN, a.theat_sC, a.theat_sH = L * R * Cc, R * Cc, Cc
    out = torch.empty(L, dtype=torch.float32, device=Xc.device)
    loss = torch.empty((), dtype=torch.float32, device=Xc.device)
    sums = torch.empty(int(lib.dfl_loss_scratch_doubles(1, 0, L)), dtype=torch.float64, device=Xc.device)
    a.loss, a.sums, a.ncc_vals = loss.data_ptr(), sums.data_ptr(), out.data_ptr()
    grad = None
    if want_grad:
        grad = torch.empty_like(Xc)
        a.dheat = grad.data_ptr()
    a.B, a.C, a.L, a.h, a.w = 1, 0, L, R, Cc
    a.dice_wgt, a.heat_wgt = 0.0, 1.0
    nat.check(lib.dfl_dice_ncc_loss(C.addressof(a), torch.cuda.current_stream().cuda_stream), 'dfl_dice_ncc_loss')
    if grad is not None:
        grad.mul_(-2.0 * L)                      # loss = mean_l(-(ncc_l + 1) / 2)  ->  d ncc_l / d X_l = -2 L * d loss / d X_l
    return out, grad


class _Ncc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y):
        lead = X.shape[:-2]
        R, Cc = X.shape[-2], X.shape[-1]
        Xc = X.detach().float().contiguous().view(1, -1, R, Cc)
        Yc = Y.detach().float().contiguous().view(1, -1, R, Cc)
        L = Xc.shape[1]
        out, gx = _launch(Xc, Yc, L, R, Cc, ctx.needs_input_grad[0])
        gy = _launch(Yc, Xc, L, R, Cc, True)[1] if ctx.needs_input_grad[1] else None     # the score is symmetric in X and Y
        ctx.save_for_backward(*[t for t in (gx, gy) if t is not None])
        ctx.have = (gx is not None, gy is not None)
        ctx.shape = X.shape
        return out.view(lead)

    @staticmethod
    def backward(ctx, gout):
        saved = list(ctx.saved_tensors)
        gx = saved.pop(0) if ctx.have[0] else None
        gy = saved.pop(0) if ctx.have[1] else None
        w = gout.detach().float().reshape(1, -1, 1, 1)
        return (None if gx is None else (gx * w).view(ctx.shape), None if gy is None else (gy * w).view(ctx.shape))


def ncc_2d(X, Y):
    N = X.shape[-1] * X.shape[-2]
    assert N > 1
    if not X.is_cuda:
        raise RuntimeError('ncc_2d runs on the GPU only (no CPU fallback)')
    if X.shape != Y.shape:
        raise RuntimeError('ncc_2d: shape mismatch')
    return _Ncc.apply(X, Y)
