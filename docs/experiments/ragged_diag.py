"""Diagnosis: per-tensor gradient error of a ragged-size network (tests/test_gpu_unet.py::test_ragged_sizes_match_oracle)
against the fp64 oracle, per product mode; run under different DFL_* switches to locate a noisy kernel."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dfl_amd
from dfl_amd import _native as nat
from oracle import ref_cpu as R
import noise_floor as NF
H, W = int(sys.argv[1]), int(sys.argv[2])
mp = bool(int(sys.argv[3]))
cfg = dict(n_classes=5, depth=3, wf=4, batch_norm=True, padding=True, max_pool=mp, num_lands=6, do_res=True, block_depth=2)
torch.manual_seed(31 + H)
onet = R.OracleUNet(1, **cfg)
g = torch.Generator().manual_seed(5)
x = torch.randn(3, 1, H, W, generator=g)
with torch.no_grad():
    oseg, oheat = onet(x)
ho, wo = oseg.shape[-2:]
tseg = torch.softmax(torch.randn(3, 5, ho - 2, wo - 2, generator=g), 1)
theat = torch.rand(3, 6, ho - 2, wo - 2, generator=g) * 0.02
o64 = R.OracleUNet(1, **cfg).double(); o64.load_state_dict(onet.state_dict()); o64.train()
def run(net):
    o = net(x.double())
    return R.dice_and_heatmap_loss_2d((R.center_crop(o[0], tseg.shape), R.center_crop(o[1], theat.shape)), (tseg.double(), theat.double()), skip_bg=False, heatmap_wgt=0.5), o[0]
gf = NF.GradientFloor(o64, run, seeds=(1, 2, 3, 4))
lib = nat.lib()
for mode, code in (('fp32', 0), ('bf16x3', 1), ('bf16', 3)):
    nat.check(lib.dfl_set_math_mode(code), 'm')
    net = dfl_amd.UNet(1, **cfg); net.load_state_dict(onet.state_dict()); net = net.to('cuda').train()
    seg, heat = net(x.cuda())
    loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg.cuda(), theat.cuda()))
    loss.backward()
    eps_eff, bars = gf.bars(seg, NF.conv_rel_error(mode))
    rows = []
    for k, p in net.named_parameters():
        if gf.clean[k] is None: continue
        e = NF.rel_l2(p.grad.cpu().numpy(), gf.clean[k].numpy())
        rows.append((e / (gf.spread[k] * eps_eff / gf.EPS_REF + 1e-30), e, k))
    rows.sort(reverse=True)
    print('%s eps_eff %.2e fwd dev %.2e' % (mode, eps_eff, NF.rel_l2(seg.detach().cpu().numpy(), gf.out.numpy())), ' | '.join('%s %.1f (%.1e)' % (k, r, e) for r, e, k in rows[:6]), 'median %.2f' % np.median([r[0] for r in rows]))
nat.check(lib.dfl_set_math_mode(0), 'm')
