"""Which torch-side kernels / copies does one training step of bench.py's loop enqueue beside the library's launches?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import dfl_amd
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda', 0)
dfl_amd._native.lib().dfl_set_math_mode(1)
torch.manual_seed(1234)
net = dfl_amd.UNet(**bench.PAPER).to(dev)
crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
opt = dfl_amd.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
x, tseg, theat = bench.synth_batch(16, 4321, dev)
net.train()

def step():
    opt.zero_grad()
    seg, heat = net(x)
    loss = crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat))
    loss.backward()
    opt.step()
    return loss.item()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=70))
