"""2-D normalized cross-correlation on the GPU (drop-in for the reference's train_test_code/ncc.py).

``ncc_2d(X, Y)`` reduces the last two dims of ``X`` and ``Y`` exactly as ncc.py:12-38 (sample standard deviation with
N-1, product with N, +1e-8 in the denominator) using the reduction kernels of ``dfl_dice_ncc_loss``.  It is the
forward value only; the differentiable use of NCC on the training path is fused into ``dice.DiceAndHeatMapLoss2D``.
"""
import ctypes as C

import torch

from . import _native as nat

__all__ = ['ncc_2d']


def ncc_2d(X, Y):
    N = X.shape[-1] * X.shape[-2]
    assert N > 1
    if X.requires_grad or Y.requires_grad:
        raise NotImplementedError('ncc_2d is forward-only in the HIP path; use dice.DiceAndHeatMapLoss2D for training')
    if not X.is_cuda:
        raise RuntimeError('ncc_2d runs on the GPU only (no CPU fallback)')
    if X.shape != Y.shape:
        raise RuntimeError('ncc_2d: shape mismatch')
    lead = X.shape[:-2]
    R, Cc = X.shape[-2], X.shape[-1]
    Xc = X.detach().float().contiguous().view(1, -1, R, Cc)
    Yc = Y.detach().float().contiguous().view(1, -1, R, Cc)
    L = Xc.shape[1]
    lib = nat.lib()
    a = nat.LossArgs()
    a.heat, a.theat = Xc.data_ptr(), Yc.data_ptr()
    a.heat_sN, a.heat_sC, a.heat_sH = L * R * Cc, R * Cc, Cc
    a.theat_sN, a.theat_sC, a.theat_sH = L * R * Cc, R * Cc, Cc
    out = torch.empty(L, dtype=torch.float32, device=X.device)
    loss = torch.empty((), dtype=torch.float32, device=X.device)
    sums = torch.empty(int(lib.dfl_loss_scratch_doubles(1, 0, L)), dtype=torch.float64, device=X.device)
    a.loss, a.sums, a.ncc_vals = loss.data_ptr(), sums.data_ptr(), out.data_ptr()
    a.B, a.C, a.L, a.h, a.w = 1, 0, L, R, Cc
    nat.check(lib.dfl_dice_ncc_loss(C.addressof(a), torch.cuda.current_stream().cuda_stream), 'dfl_dice_ncc_loss')
    return out.view(lead)
