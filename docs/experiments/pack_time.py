"""Time of one weight re-layout (plan.pack) and of the optimizer step of the paper network (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dfl_amd, bench
from dfl_amd import _native as nat
nat.check(nat.lib().dfl_set_math_mode(4), 'm')
dev = torch.device('cuda:0')
torch.manual_seed(1)
net = dfl_amd.UNet(**bench.PAPER).to(dev).train()
x, tseg, theat = bench.synth_batch(16, 1, dev)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
for _ in range(3):
    opt.zero_grad(); s, h = net(x)
    crit((dfl_amd.center_crop(s, tseg.shape), dfl_amd.center_crop(h, theat.shape)), (tseg, theat)).backward(); opt.step()
plan = net._last_train_plan()
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (('pack', lambda: plan.pack.run(st)), ('sgd', lambda: opt.step())):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.1f us' % (name, e0.elapsed_time(e1) / 20 * 1e3))
print('pack jobs: %d, bytes of packed layouts: %.1f MB' % (len(plan._pack_jobs), sum(j[1].numel() * j[1].element_size() for j in plan._pack_jobs) / 1e6))
