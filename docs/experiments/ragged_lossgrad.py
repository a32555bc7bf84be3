import os, sys
os.environ['DFL_WSPLIT'] = '0'; os.environ['DFL_DSPLIT'] = '0'
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dfl_amd
from dfl_amd import _native as nat
from oracle import ref_cpu as R
import noise_floor as NF
H, W = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(n_classes=5, depth=3, wf=4, batch_norm=True, padding=True, max_pool=False, num_lands=6, do_res=True, block_depth=2)
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
o64.zero_grad()
o = o64(x.double())
l64 = R.dice_and_heatmap_loss_2d((R.center_crop(o[0], tseg.shape), R.center_crop(o[1], theat.shape)), (tseg.double(), theat.double()), skip_bg=False, heatmap_wgt=0.5)
l64.backward()
clean = {k: p.grad.clone() for k, p in o64.named_parameters() if p.grad is not None}
seg64, heat64 = o[0].detach(), o[1].detach()
lib = nat.lib()
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
res = {}
for mode in (0, 1):
    nat.check(lib.dfl_set_math_mode(mode), 'm')
    net = dfl_amd.UNet(1, **cfg); net.load_state_dict(onet.state_dict()); net = net.to('cuda').train()
    seg, heat = net(x.cuda())
    sl, hl = seg.detach().clone().requires_grad_(True), heat.detach().clone().requires_grad_(True)
    loss = crit((dfl_amd.center_crop(sl, tseg.shape), dfl_amd.center_crop(hl, theat.shape)), (tseg.cuda(), theat.cuda()))
    ds, dh = torch.autograd.grad(loss, [sl, hl])
    res[mode] = (net, seg, heat, ds, dh, loss.item())
    print('mode %d: loss %.8f (fp64 %.8f)  seg dev %.2e heat dev %.2e (rel L2 vs fp64)  heat std/rms %.3e' % (
        mode, loss.item(), float(l64), NF.rel_l2(seg.detach().cpu().numpy(), seg64.numpy()), NF.rel_l2(heat.detach().cpu().numpy(), heat64.numpy()),
        float(heat64.std() / heat64.pow(2).mean().sqrt())))
print('loss-gradient difference between the modes: dseg %.2e dheat %.2e (rel L2)' % (
    NF.rel_l2(res[1][3].cpu().numpy(), res[0][3].cpu().numpy()), NF.rel_l2(res[1][4].cpu().numpy(), res[0][4].cpu().numpy())))
def report(name, net):
    errs = sorted(((NF.rel_l2(p.grad.cpu().numpy(), clean[k].numpy()), k) for k, p in net.named_parameters() if k in clean), reverse=True)
    print('%-44s worst %s median %.1e' % (name, ' | '.join('%s %.1e' % (k, e) for e, k in errs[:3]), np.median([e for e, _ in errs])))
# mode-1 forward state, backward in mode 1, with the loss gradients of the mode-0 forward
nat.check(lib.dfl_set_math_mode(1), 'm')
net1, seg1, heat1 = res[1][0], res[1][1], res[1][2]
torch.autograd.backward([seg1, heat1], [res[0][3], res[0][4]])
report('fwd bf16x3 + loss grads from fp32 forward', net1)
nat.check(lib.dfl_set_math_mode(0), 'm')
net0, seg0, heat0 = res[0][0], res[0][1], res[0][2]
torch.autograd.backward([seg0, heat0], [res[1][3], res[1][4]])
report('fwd fp32 + loss grads from bf16x3 forward', net0)
