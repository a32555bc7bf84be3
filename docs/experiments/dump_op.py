import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import dfl_amd
from dfl_amd import _native as nat
from conftest import PAPER_CFGS
lib = nat.lib(); nat.check(lib.dfl_set_math_mode(4), 'm')
_, cfg = PAPER_CFGS['paper_sc_l14']
x = torch.randn(8, 1, 768, 768).cuda()
net = dfl_amd.UNet(**cfg).to('cuda').train()
seg, heat = net(x)
plan = [p for ps in net._plans.values() for p in ps][0]
for i, st in enumerate(plan.fwd.structs):
    if isinstance(st, nat.ConvArgs) and st.KH == 1 and st.Cin == 64 and st.Ntot == 128 and st.Hin == 192:
        print(i, {f: getattr(st, f) for f, _ in st._fields_ if f not in ('reserved3',)})
        print('config', lib.dfl_conv_config(C.addressof(st)), 'grid_m', lib.dfl_conv_grid_m(C.addressof(st)))
