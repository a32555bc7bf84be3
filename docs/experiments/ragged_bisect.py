"""Bisect which kernels of the bf16x3 mode lose precision on ragged sizes: the plan is recorded in mode 1 WITHOUT pre-split
operands (DFL_WSPLIT=0 DFL_DSPLIT=0), so the product mode can be switched per op at run time."""
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
def run(net):
    o = net(x.double())
    return R.dice_and_heatmap_loss_2d((R.center_crop(o[0], tseg.shape), R.center_crop(o[1], theat.shape)), (tseg.double(), theat.double()), skip_bg=False, heatmap_wgt=0.5), o[0]
gf = NF.GradientFloor(o64, run, seeds=(1, 2))
lib = nat.lib()

def trial(name, fwd_mode, bwd_mode_of):
    nat.check(lib.dfl_set_math_mode(1), 'm')
    net = dfl_amd.UNet(1, **cfg); net.load_state_dict(onet.state_dict()); net = net.to('cuda').train()
    def fwd_runner(plan, xx):
        return orig_fwd(plan, xx)
    def bwd_runner(plan, stream):
        for i, st in enumerate(plan.bwd.structs):
            nat.check(lib.dfl_set_math_mode(bwd_mode_of(st)), 'm')
            plan.bwd.run(stream, i, 1)
        nat.check(lib.dfl_set_math_mode(1), 'm')
    net._backward_runner = bwd_runner
    # forward: record in mode 1, run ops in fwd_mode
    orig = net._run_forward
    def run_forward(plan, xx):
        stream = torch.cuda.current_stream().cuda_stream
        net._ensure_packed(plan, stream)
        plan.x_in.copy_(xx.reshape(-1))
        seg, heat = plan.new_outputs()
        plan.head_fwd.seg = seg.data_ptr(); plan.head_fwd.heat = nat.ptr(heat)
        nat.check(lib.dfl_set_math_mode(fwd_mode), 'm')
        plan.fwd.run(stream)
        nat.check(lib.dfl_set_math_mode(1), 'm')
        return seg, heat
    net._run_forward = run_forward
    seg, heat = net(x.cuda())
    loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg.cuda(), theat.cuda()))
    loss.backward()
    errs = sorted(((NF.rel_l2(p.grad.cpu().numpy(), gf.clean[k].numpy()), k) for k, p in net.named_parameters() if gf.clean[k] is not None), reverse=True)
    print('%-34s worst %s  median %.1e' % (name, ' | '.join('%s %.1e' % (k, e) for e, k in errs[:3]), np.median([e for e, _ in errs])))

C, Wg = nat.ConvArgs, nat.WgradArgs
trial('all fp32', 0, lambda st: 0)
trial('all bf16x3', 1, lambda st: 1)
trial('fwd bf16x3, bwd fp32', 1, lambda st: 0)
trial('fwd fp32, bwd bf16x3', 0, lambda st: 1)
trial('bwd: only convs bf16x3', 0, lambda st: 1 if isinstance(st, C) else 0)
trial('bwd: only wgrads bf16x3', 0, lambda st: 1 if isinstance(st, Wg) else 0)
trial('bwd: only 3x3 convs bf16x3', 0, lambda st: 1 if isinstance(st, C) and st.KH == 3 else 0)
trial('bwd: only 1x1 convs bf16x3', 0, lambda st: 1 if isinstance(st, C) and st.KH == 1 and not st.scatter2x2 else 0)
trial('bwd: only 2x2/scatter convs bf16x3', 0, lambda st: 1 if isinstance(st, C) and (st.KH == 2 or st.scatter2x2) else 0)
trial('bwd: convs with stats bf16x3', 0, lambda st: 1 if isinstance(st, C) and st.stat_partials else 0)
trial('bwd: convs w/o stats bf16x3', 0, lambda st: 1 if isinstance(st, C) and not st.stat_partials else 0)
nat.check(lib.dfl_set_math_mode(0), 'm')
