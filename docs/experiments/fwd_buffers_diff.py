"""Which buffers of the recorded forward differ between fp32 and bf16x3 products beyond rounding level (ragged sizes)."""
import os, sys
os.environ['DFL_WSPLIT'] = '0'; os.environ['DFL_DSPLIT'] = '0'
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd
from dfl_amd import _native as nat
H, W = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(n_classes=5, depth=3, wf=4, batch_norm=True, padding=True, max_pool=False, num_lands=6, do_res=True, block_depth=2)
torch.manual_seed(31 + H)
net0 = dfl_amd.UNet(1, **cfg)
sd = {k: v.clone() for k, v in net0.state_dict().items()}
x = torch.randn(3, 1, H, W, generator=torch.Generator().manual_seed(5)).cuda()
lib = nat.lib()
plans = {}
for mode in (0, 1):
    nat.check(lib.dfl_set_math_mode(mode), 'm')
    net = dfl_amd.UNet(1, **cfg); net.load_state_dict(sd); net = net.cuda().train()
    seg, heat = net(x)
    torch.cuda.synchronize()
    plan = [p for ps in net._plans.values() for p in ps][0]
    plans[mode] = (net, plan, seg, heat)
nat.check(lib.dfl_set_math_mode(0), 'm')
p0, p1 = plans[0][1], plans[1][1]
print('buffers', len(p0._keep), len(p1._keep))
# label buffers by the forward op that writes them
def writers(plan):
    w = {}
    for i, st in enumerate(plan.fwd.structs):
        for f in ('y', 'stat_partials', 'scale', 'shift', 'save_mean', 'save_invstd', 'partial'):
            ptr = getattr(st, f, None)
            if ptr:
                desc = type(st).__name__
                if isinstance(st, nat.ConvArgs):
                    desc += ' %dx%d Cin%d->%d k%d s%d aff%d add%d sp%d' % (st.Hin, st.Win, st.Cin, st.Ntot, st.KH, st.stride, bool(st.in_scale), bool(st.add), st.splits)
                w.setdefault(ptr, []).append('op%d.%s %s' % (i, f, desc))
    return w
w0 = writers(p0)
for i, (a, b) in enumerate(zip(p0._keep, p1._keep)):
    if a.dtype != torch.float32 or a.numel() != b.numel() or a.numel() < 2:
        continue
    base = a.data_ptr()
    who = [d for ptr, ds in w0.items() if base <= ptr < base + 4 * a.numel() for d in ds]
    if not who:
        continue
    fa, fb = a.double(), b.double()
    ok = torch.isfinite(fa) & torch.isfinite(fb)
    den = float(fa[ok].pow(2).sum().sqrt())
    rel = float((fa[ok] - fb[ok]).pow(2).sum().sqrt()) / max(den, 1e-30)
    mx = float((fa[ok] - fb[ok]).abs().max()) / max(float(fa[ok].abs().max()), 1e-30)
    flag = '  <<<<<<' if rel > 2e-4 else ''
    print('buf%03d n=%8d relL2 %.2e maxrel %.2e  %s%s' % (i, a.numel(), rel, mx, '; '.join(who)[:150], flag))
