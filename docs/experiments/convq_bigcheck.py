import sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
import torch.nn.functional as F
from dfl_amd import _native as nat
import test_gpu_bf16 as T
lib = nat.lib(); nat.check(lib.dfl_set_math_mode(4),'m')
QCFG = {40: (1, 4, 1), 41: (1, 4, 2), 42: (2, 4, 1), 43: (2, 2, 1), 44: (4, 2, 1), 45: (2, 2, 2), 46: (4, 1, 1), 47: (8, 1, 1), 48: (4, 1, 2)}
def run(case, tiles):
    N,Cin,Cout,H,W = case
    g = torch.Generator().manual_seed(3)
    x = T.rb(torch.randn(N,Cin,H,W,generator=g))
    w = T.rb(torch.randn(Cout,Cin,3,3,generator=g)/(Cin*9)**0.5)
    wp = T.pack16(w,1)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    base, st0 = T.conv_bf16(x, wp, Cout,3,3,1,1,H,W, in_aff=(sc,sh), relu=1, stats=True)
    cands = T._candidates(N,Cin,Cout,H,W,3,1,1)
    for tile in tiles:
        for sp in (1,2):
            geom=(tile,1,8*QCFG[tile][0],12,sp)
            if geom not in cands: continue
            gv=(C.c_int32*5)(*geom); nat.check(lib.dfl_conv_force_geometry(C.addressof(gv)),'f')
            try:
                ys=[T.conv_bf16(x, wp, Cout,3,3,1,1,H,W, in_aff=(sc,sh), relu=1, stats=True, force_splits=sp) for _ in range(2)]
            finally:
                lib.dfl_conv_force_geometry(None)
            y, st = ys[0]
            d=(y.double()-base.double()).abs()
            print(case, geom, 'max diff vs convp %.3e (max |y| %.2f), frac differing %.2e, repeat equal %s, stats rel diff %.2e' % (
                float(d.max()), float(base.abs().max()), float((d>0).double().mean()), bool(torch.equal(ys[0][0], ys[1][0])),
                float((st[0]-st0[0]).abs().max()/st0[0].abs().max())), flush=True)
cases=[((8,64,128,192,192),(40,41,42)), ((8,128,256,96,96),(40,41,42)), ((8,512,256,96,96),(41,42)), ((8,512,512,48,48),(41,42)), ((8,512,1024,48,48),(41,42)),((8,1024,1024,24,24),(41,)),((8,1024,512,48,48),(41,)),
       ((8,64,64,384,384),(43,44,45)),((8,32,32,768,768),(46,47)),((8,64,32,768,768),(46,48))]
for c,t in cases: run(c,t)
