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
o = o64(x.double())
R.dice_and_heatmap_loss_2d((R.center_crop(o[0], tseg.shape), R.center_crop(o[1], theat.shape)), (tseg.double(), theat.double()), skip_bg=False, heatmap_wgt=0.5).backward()
clean = {k: p.grad.clone() for k, p in o64.named_parameters() if p.grad is not None}
lib = nat.lib()
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)

def fwd(mode):
    nat.check(lib.dfl_set_math_mode(mode), 'm')
    net = dfl_amd.UNet(1, **cfg); net.load_state_dict(onet.state_dict()); net = net.to('cuda').train()
    seg, heat = net(x.cuda())
    torch.cuda.synchronize()
    plan = [p for ps in net._plans.values() for p in ps][0]
    return net, plan, seg, heat

def kinds(plan):
    """buffer index -> set of writer kinds among forward ops"""
    out = {}
    for i, t in enumerate(plan._keep):
        if t.dtype != torch.float32: continue
        base, end = t.data_ptr(), t.data_ptr() + 4 * t.numel()
        ks = set()
        for j, st in enumerate(plan.fwd.structs):
            for f, tag in (('y', 'y'), ('stat_partials', 'stat'), ('scale', 'bn'), ('shift', 'bn'), ('save_mean', 'bn'), ('save_invstd', 'bn')):
                ptr = getattr(st, f, None)
                if ptr and base <= ptr < end:
                    ks.add(tag + ('%d' % j if tag == 'y' else ''))
        if ks: out[i] = ks
    return out

def trial(name, select):
    net1, p1, seg1, heat1 = fwd(1)
    net0, p0, seg0, heat0 = fwd(0)
    kk = kinds(p0)
    n = 0
    for i, ks in kk.items():
        if select(ks):
            p0._keep[i].copy_(p1._keep[i]); n += 1
    nat.check(lib.dfl_set_math_mode(0), 'm')
    loss = crit((dfl_amd.center_crop(seg0, tseg.shape), dfl_amd.center_crop(heat0, theat.shape)), (tseg.cuda(), theat.cuda()))
    loss.backward()
    errs = sorted(((NF.rel_l2(p.grad.cpu().numpy(), clean[k].numpy()), k) for k, p in net0.named_parameters() if k in clean), reverse=True)
    print('%-40s (%2d buffers) worst %s median %.1e' % (name, n, ' | '.join('%s %.1e' % (k, e) for e, k in errs[:2]), np.median([e for e, _ in errs])))

trial('nothing swapped', lambda ks: False)
trial('everything from the bf16x3 forward', lambda ks: True)
trial('only BatchNorm scale/shift/mean/invstd', lambda ks: 'bn' in ks)
trial('only statistics partials', lambda ks: 'stat' in ks)
trial('only activations', lambda ks: any(k.startswith('y') for k in ks))
net, plan, _, _ = fwd(0)
ys = sorted({int(k[1:]) for ks in kinds(plan).values() for k in ks if k.startswith('y')})
for j in ys:
    st = plan.fwd.structs[j]
    d = type(st).__name__ + (' %dx%d Cin%d->%d k%d aff%d add%d sp%d' % (st.Hin, st.Win, st.Cin, st.Ntot, st.KH, bool(st.in_scale), bool(st.add), st.splits) if isinstance(st, nat.ConvArgs) else '')
    trial('only output of op %d %s' % (j, d), lambda ks, j=j: ('y%d' % j) in ks)
