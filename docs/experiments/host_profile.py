"""cProfile of the host side of training steps (paper network, batch 16, bf16 storage): where the enqueue time goes."""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
import torch
import dfl_amd
from dfl_amd import _native as nat
nat.check(nat.lib().dfl_set_math_mode(4))
torch.manual_seed(0)
net = dfl_amd.UNet(1, n_classes=7, depth=6, wf=5, padding=True, batch_norm=True, up_mode='upconv', num_lands=14, max_pool=False).cuda().train()
opt = dfl_amd.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
x = torch.randn(16, 1, 192, 192, device='cuda')
tseg = torch.zeros(16, 7, 184, 184, device='cuda'); tseg[:, 0] = 1
theat = torch.rand(16, 14, 184, 184, device='cuda')
def step():
    opt.zero_grad()
    seg, heat = net(x)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
    loss.backward()
    opt.step()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
for _ in range(20):
    torch.cuda.synchronize()
    pr.enable(); step(); pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
