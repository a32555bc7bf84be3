#!/usr/bin/env python3
"""dfl_head_fwd / dfl_head_bwd alone at the benchmark shape (16 x 192 x 192 pixels, 32 bf16 features, 7 classes, 21 mid, 14 landmarks)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa: E402,F401
from dfl_amd import _native as nat  # noqa: E402

lib = nat.lib()
DEV = 'cuda'
N, H, W, F, NC, NM, L = 16, 192, 192, 32, 7, 21, 14
g = torch.Generator().manual_seed(0)
x = torch.randn(N, H, W, F, generator=g).to(DEV).to(torch.bfloat16)
wseg = (torch.randn(NC, F, generator=g) / 3).to(DEV)
w1 = (torch.randn(NM, F + NC, generator=g) / 3).to(DEV)
w2 = (torch.randn(L, NM, generator=g) / 3).to(DEV)
seg = torch.empty(N, NC, H, W, device=DEV)
heat = torch.empty(N, L, H, W, device=DEV)
dseg, dheat = torch.randn_like(seg), torch.randn_like(heat)
dx = torch.empty_like(x)
M = N * H * W
nb = lib.dfl_head_wgrad_blocks(M)
part = torch.empty(nb * 4096, device=DEV)
dws, dw1, dw2 = torch.empty_like(wseg), torch.empty_like(w1), torch.empty_like(w2)
st = torch.cuda.current_stream().cuda_stream
fa = nat.HeadFwdArgs(x=x.data_ptr(), w_seg=wseg.data_ptr(), w_l1=w1.data_ptr(), w_l2=w2.data_ptr(), seg=seg.data_ptr(), heat=heat.data_ptr(),
                     N=N, H=H, W=W, F=F, ldx=F, NC=NC, NM=NM, L=L, softmax=1, x_bf16=1)
ba = nat.HeadBwdArgs(x=x.data_ptr(), seg=seg.data_ptr(), dseg=dseg.data_ptr(), dheat=dheat.data_ptr(), w_seg=wseg.data_ptr(), w_l1=w1.data_ptr(),
                     w_l2=w2.data_ptr(), dx=dx.data_ptr(), N=N, H=H, W=W, F=F, ldx=F, lddx=F, NC=NC, NM=NM, L=L, softmax=1, x_bf16=1,
                     dw_seg=dws.data_ptr(), dw_l1=dw1.data_ptr(), dw_l2=dw2.data_ptr(), wg_partial=part.data_ptr())
big = torch.empty(512 << 20, device=DEV, dtype=torch.uint8)     # flushes the caches between repetitions


def timed(name, a, reps=20):
    for _ in range(3):
        nat.call(name, a, st)
    tot = 0.0
    for _ in range(reps):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.call(name, a, st)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


print('%s: fwd %.1f us  bwd %.1f us' % (' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('DFL_HEAD')),
                                         timed('dfl_head_fwd', fa), timed('dfl_head_bwd', ba)))
