"""Layer-by-layer: stored ReLU outputs of a bf16-storage HIP run against oracle/bf16_emu.py (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import noise_floor as NF
import problems as PR
from oracle import bf16_emu as E
from gpu_common import hip_net, hip_step, math_mode_set

key = sys.argv[1] if len(sys.argv) > 1 else 'ragged__37x41__mp0'
pr = PR.REGISTRY[key]()
gc = NF.GradientCheck(pr)
with math_mode_set('bf16s'):
    net = hip_net(pr)
    out, seg, loss = hip_step(pr, net)
    plan = NF.train_plan(net)
    ch = NF.hip_choices(plan)
    for forced in (False, True):
        emu = E.Bf16Emulation(gc.net, dict(pr.cfg), choices=ch if forced else None)
        res = emu.run(pr.x, pr.loss_of)
        print('forced' if forced else 'natural')
        for name, a in plan.relu_out.items():
            h = plan.act_nchw(a).double().cpu()
            e = emu.acts[name]
            diff = (h - e)
            nz = int((diff != 0).sum())
            print('  %-40s rel_l2 %.3e  differing %d of %d (%.2e)  max|diff|/rms %.2e' % (
                name, NF.rel_l2(h.numpy(), e.numpy()), nz, h.numel(), nz / h.numel(), float(diff.abs().max()) / NF._rms(e)))
        print('  seg rel_l2 %.3e' % NF.rel_l2(seg.detach().double().cpu().numpy(), res['seg'].numpy()))
