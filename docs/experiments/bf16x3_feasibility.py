#!/usr/bin/env python3
"""Feasibility study (CPU): how far do logits/heat-maps/gradients of the paper U-Net move when every convolution is
computed as a 2-way (bf16x3) or 3-way (bf16x6) split-bf16 product with fp32 accumulation instead of fp32 products?
Compared against an fp64 run of the same network.  Test infrastructure only (uses the oracle)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import ref_cpu  # noqa: E402

MODE = {'n': 0}
_conv2d, _convT = F.conv2d, F.conv_transpose2d


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        h = r.bfloat16().float()
        parts.append(h)
        r = r - h
    return parts


def emu(fn, x, w, b, *a, **k):
    n = MODE['n']
    if n == 0 or x.dtype != torch.float32:
        return fn(x, w, b, *a, **k)
    xs, ws = split(x, n), split(w, n)
    out = None
    # keep terms whose order i+j < n  (n=2: hh, hl, lh; n=3: 6 terms)
    terms = sorted([(i, j) for i in range(n) for j in range(n) if i + j < n], key=lambda t: -(t[0] + t[1]))
    for i, j in terms:          # small terms first
        t = fn(xs[i], ws[j], None, *a, **k)
        out = t if out is None else out + t
    if b is not None:
        out = out + b.view(1, -1, 1, 1)
    return out


F.conv2d = lambda x, w, b=None, *a, **k: emu(_conv2d, x, w, b, *a, **k)
F.conv_transpose2d = lambda x, w, b=None, *a, **k: emu(_convT, x, w, b, *a, **k)
torch.conv2d = F.conv2d
torch.conv_transpose2d = F.conv_transpose2d


def run(net, x, tseg, theat, dtype):
    net = net.to(dtype)
    net.zero_grad()
    seg, heat = net(x.to(dtype))
    seg_c, heat_c = ref_cpu.center_crop(seg, tseg.shape), ref_cpu.center_crop(heat, theat.shape)
    loss = ref_cpu.dice_and_heatmap_loss_2d((seg_c, heat_c), (tseg.to(dtype), theat.to(dtype)))
    loss.backward()
    g = torch.cat([p.grad.flatten().double() for p in net.parameters() if p.grad is not None])
    return seg.detach().double(), heat.detach().double(), float(loss), g


def main():
    torch.manual_seed(7)
    torch.set_num_threads(32)
    B = 2
    net = ref_cpu.OracleUNet(1, 7, depth=6, wf=5, padding=True, batch_norm=True, max_pool=False, num_lands=14)
    net.train()
    x = torch.randn(B, 1, 192, 192)
    lab = torch.randint(0, 7, (B, 184, 184))
    tseg = torch.stack([(lab == c) for c in range(7)], 1).float()
    theat = torch.rand(B, 14, 184, 184) * 0.02
    import copy
    ref = run(copy.deepcopy(net), x, tseg, theat, torch.float64)
    for name, n in (('fp32', 0), ('bf16x3', 2), ('bf16x6', 3)):
        MODE['n'] = n
        r = run(copy.deepcopy(net), x, tseg, theat, torch.float32)
        MODE['n'] = 0
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        print('%-7s seg max-rel %.3e  heat max-rel %.3e  loss diff %.3e  grad rel-L2 %.3e' % (
            name, rel(r[0], ref[0]), rel(r[1], ref[1]), abs(r[2] - ref[2]), float((r[3] - ref[3]).norm() / ref[3].norm())))


if __name__ == '__main__':
    main()
