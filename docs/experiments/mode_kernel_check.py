import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import dfl_amd
from dfl_amd import _native as nat
import test_gpu_kernels as T
lib = nat.lib()
g = torch.Generator().manual_seed(1)
for (N, H, Ci, Co) in [(2, 12, 256, 512), (2, 12, 512, 512), (2, 6, 512, 1024), (2, 24, 128, 256), (2, 12, 512, 256), (16, 12, 256, 512)]:
    x = torch.randn(N, Ci, H, H, generator=g)
    d = torch.randn(N, Co, H, H, generator=g) * 1e-3
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    y = F.conv2d(xr, wr, padding=1)
    y.backward(d.double())
    for mode in (0, 1):
        lib.dfl_set_math_mode(mode)
        T.TOLK[0] = 1.0
        dw, s = T.wgrad_call(x, d, 3, 3, 1, 1, H, H)
        wd = T.pack(w, 2, flip=1)
        dx = T.conv_call(d, wd, Ci, 3, 3, 1, 1, H, H)
        e_dw = float((dw.double() - wr.grad).norm() / wr.grad.norm()) if dw is not None else -1
        e_dx = float((T.nchw(dx).double() - xr.grad).norm() / xr.grad.norm())
        print('N%d %dx%d %d->%d mode %d: wgrad rel %.3e (splits %s)  dgrad rel %.3e' % (N, H, H, Ci, Co, mode, e_dw, s, e_dx))
lib.dfl_set_math_mode(0)
