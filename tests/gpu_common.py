"""Helpers shared by the GPU parity tests (test infrastructure; imports oracle/)."""
import contextlib

import numpy as np
import torch

import dfl_amd
from dfl_amd import _native as nat
from oracle import ref_cpu as R

DEV = 'cuda'
MODE_ID = {'fp32': 0, 'bf16x3': 1, 'bf16x6': 2, 'bf16': 3, 'bf16s': 4}


@contextlib.contextmanager
def math_mode_set(name):
    """Run a block in product arithmetic `name` (include/dfl_hip.h: dfl_set_math_mode), then restore the previous mode."""
    lib = nat.lib()
    prev = lib.dfl_get_math_mode()
    nat.check(lib.dfl_set_math_mode(MODE_ID[name]), 'dfl_set_math_mode')
    try:
        yield
    finally:
        nat.check(lib.dfl_set_math_mode(prev), 'dfl_set_math_mode')


def _t(a):
    return torch.from_numpy(np.asarray(a))


def oracle64(cfg, state_dict):
    o = R.OracleUNet(**cfg).double()
    o.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state_dict.items()})
    return o.train()


def load_net(g, cfg, prefix='sd0/'):
    net = dfl_amd.UNet(**cfg)
    sd = {k[len(prefix):]: _t(v) for k, v in g.items() if k.startswith(prefix)}
    assert list(sd.keys()) == list(net.state_dict().keys())
    net.load_state_dict(sd)
    return net.to(DEV)


def hip_net(problem):
    """dfl_amd.UNet carrying the problem's weights, on the GPU, in training mode."""
    cfg = problem.cfg
    net = dfl_amd.UNet(**cfg) if 'in_channels' in cfg else dfl_amd.UNet(1, **cfg)
    assert list(net.state_dict().keys()) == list(problem.sd.keys())
    net.load_state_dict(problem.sd)
    return net.to(DEV).train()


def hip_step(problem, net):
    """Forward + loss + backward of the HIP path on the problem, wired as train.py:405-422.  Returns (out, seg, loss)."""
    out = net(problem.x.to(DEV))
    seg = out[0] if isinstance(out, tuple) else out
    tseg = problem.tseg.to(DEV)
    if problem.theat is not None:
        theat = problem.theat.to(DEV)
        loss = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)(
            (dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(out[1], theat.shape)), (tseg, theat))
    else:
        loss = dfl_amd.DiceLoss2D(skip_bg=problem.skip_bg)(dfl_amd.center_crop(seg, tseg.shape), tseg)
    loss.backward()
    return out, seg, loss


def label_mask(seg64, hip_seg=None):
    """Pixels where arg-max labels may legitimately differ from the fp64 reference: top-2 margin below 1e-5 (SURVEY
    section 7: the reference's own fp32 run flips there), or -- when the HIP soft-max is given -- below 2.5 x its largest
    deviation from fp64 (a label can only flip where the margin is under twice the deviation; the deviation itself is
    held to the 1e-4 forward bar).  Everything outside the mask must match bit for bit."""
    top2 = seg64.topk(2, dim=1)[0]
    margin = (top2[:, 0] - top2[:, 1])
    thr = 1e-5
    if hip_seg is not None:
        thr = max(thr, 2.5 * float((hip_seg.detach().double().cpu() - seg64).abs().max()))
    assert thr < 2.5e-4, 'forward deviation %.3e is outside the 1e-4 bar' % (thr / 2.5)
    return margin < thr


def rel_close(actual, ref, rtol, what):
    scale = max(float(np.abs(ref).max()), 1e-6)
    err = float(np.abs(actual - ref).max())
    assert err <= rtol * scale, '%s: max abs err %.3e vs scale %.3e (rel %.3e > %.1e)' % (what, err, scale, err / scale, rtol)
