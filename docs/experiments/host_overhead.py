import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import dfl_amd
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
dev = torch.device('cuda:0')
net = dfl_amd.UNet(**bench.PAPER).to(dev)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
x, tseg, theat = bench.synth_batch(16, 1, dev)
net.train()
T = {k: 0.0 for k in ('zero', 'fwd', 'loss', 'bwd', 'opt', 'item')}
N = 30
for it in range(N + 5):
    if it == 5:
        T = {k: 0.0 for k in T}
    t = time.perf_counter(); opt.zero_grad(); t1 = time.perf_counter(); T['zero'] += t1 - t
    seg, heat = net(x); t2 = time.perf_counter(); T['fwd'] += t2 - t1
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat)); t3 = time.perf_counter(); T['loss'] += t3 - t2
    loss.backward(); t4 = time.perf_counter(); T['bwd'] += t4 - t3
    opt.step(); t5 = time.perf_counter(); T['opt'] += t5 - t4
    loss.item(); t6 = time.perf_counter(); T['item'] += t6 - t5
print({k: round(v / N * 1e6, 1) for k, v in T.items()}, 'us per step (host wall time per phase)')
