"""Diagnosis of tests/test_gpu_fullsize.py::test_config3_batch8_properties: how far do the outputs of a batch-8 768x768 training forward
move when the batch is permuted, and where?  (DFL_TUNE=0: cost-model geometries, no unrolled 3x3 form.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import dfl_amd
from dfl_amd import _native as nat
from conftest import PAPER_CFGS
lib = nat.lib(); nat.check(lib.dfl_set_math_mode(4), 'm')
_, cfg = PAPER_CFGS['paper_sc_l14']
B, P = 8, int(os.environ.get('SIZE', '768'))
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 1, P, P, generator=g)
perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
def run(order):
    torch.manual_seed(4242)
    net = dfl_amd.UNet(**cfg).to('cuda').train()
    with torch.no_grad():
        seg, heat = net(x[order].to('cuda'))
    return seg.float(), heat.float()
s0, h0 = run(torch.arange(B)); s0b, _ = run(torch.arange(B)); s1, h1 = run(perm)
d = (s1 - s0[perm.cuda()]).abs()
print('tune', os.environ.get('DFL_TUNE', '1'), 'size', P, 'repeat max diff %.3e; permuted: max %.3e, frac > 1e-3 %.2e, > 1e-2 %.2e, > 5e-2 %.2e; per image max %s; heat max diff %.3e (max %.2e)' % (
    float((s0b - s0).abs().max()), float(d.max()), float((d > 1e-3).float().mean()), float((d > 1e-2).float().mean()), float((d > 5e-2).float().mean()),
    [round(float(d[i].max()), 3) for i in range(B)], float((h1 - h0[perm.cuda()]).abs().max()), float(h0.abs().max())))
