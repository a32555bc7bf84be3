import os, sys
os.environ['DFL_WSPLIT'] = '0'; os.environ['DFL_DSPLIT'] = '0'
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dfl_amd
from dfl_amd import _native as nat
from oracle import ref_cpu as R
H, W = 37, 41
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
lib = nat.lib()
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
out = {}
for mode in (0, 1):
    nat.check(lib.dfl_set_math_mode(mode), 'm')
    net = dfl_amd.UNet(1, **cfg); net.load_state_dict(onet.state_dict()); net = net.to('cuda').train()
    seg, heat = net(x.cuda())
    plan = [p for ps in net._plans.values() for p in ps][0]
    st = plan.fwd.structs[27]
    bn = plan.fwd.structs[28]
    def grab(ptr, n):
        for t in plan._keep:
            if t.dtype == torch.float32 and t.data_ptr() <= ptr < t.data_ptr() + 4 * t.numel():
                o = (ptr - t.data_ptr()) // 4
                return t[o:o + n].clone().cpu().numpy()
    M = st.N * st.Hout * st.Wout
    out['r%d' % mode] = grab(st.y, M * st.Ntot).reshape(M, st.Ntot)
    out['xin%d' % mode] = grab(st.x, M * st.Cin).reshape(M, st.Cin)
    out['mean%d' % mode] = grab(bn.save_mean, st.Ntot)
    out['invstd%d' % mode] = grab(bn.save_invstd, st.Ntot)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg.cuda(), theat.cuda()))
    loss.backward()
    torch.cuda.synchronize()
    hb = plan.head_bwd
    out['dfeat%d' % mode] = grab(hb.dx, M * st.Ntot).reshape(M, st.Ntot)
    out['gbias%d' % mode] = dict(net.named_parameters())['up_path.1.conv_block.block.3.bias'].grad.cpu().numpy()
    out['ggamma%d' % mode] = dict(net.named_parameters())['up_path.1.conv_block.block.5.weight'].grad.cpu().numpy()
out['shape'] = np.array([st.N, st.Hout, st.Wout, st.Ntot])
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'ragged_dump.npz'), **out)
print('ok')
