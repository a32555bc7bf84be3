import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import dfl_amd
from dfl_amd import _native as nat
from conftest import PAPER_CFGS
from oracle import ref_cpu as R
name = sys.argv[1] if len(sys.argv) > 1 else 'paper_mp_l0'
seed, cfg = PAPER_CFGS[name]
lib = nat.lib()
res = {}
for mode in (0, 1):
    lib.dfl_set_math_mode(mode)
    torch.manual_seed(seed)
    net = dfl_amd.UNet(**cfg).to('cuda')
    gen = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(2, 1, 192, 192, generator=gen)
    lab = torch.randint(0, 7, (2, 184, 184), generator=gen)
    tseg = R.one_hot_masks(lab, 7).to('cuda')
    theat = (torch.rand(2, 14, 184, 184, generator=gen) * 0.02).to('cuda')
    net.train()
    out = net(x.to('cuda'))
    seg = out[0] if cfg['num_lands'] > 0 else out
    if cfg['num_lands'] > 0:
        loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(out[1], theat.shape)), (tseg, theat))
    else:
        loss = dfl_amd.DiceLoss2D(skip_bg=False)(dfl_amd.center_crop(seg, tseg.shape), tseg)
    loss.backward()
    res[mode] = ({k: p.grad.detach().cpu().double().clone() for k, p in net.named_parameters() if p.grad is not None}, seg.detach().cpu().double())
print('forward max rel diff %.3e' % float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max()))
tot_n = tot_d = 0.0
for k in res[0][0]:
    a, b = res[0][0][k], res[1][0][k]
    n, d = float((a - b).norm()), float(a.norm())
    tot_n += n * n; tot_d += d * d
    if a.dim() == 4:
        print('%-40s shape %-22s |g| %.3e  rel diff %.3e' % (k, tuple(a.shape), d, n / max(d, 1e-30)))
print('whole gradient rel diff %.3e' % (tot_n / tot_d) ** 0.5)
