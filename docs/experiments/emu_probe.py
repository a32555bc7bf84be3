"""Measure the bf16-storage HIP path against oracle/bf16_emu.py (GPU box): forward / loss / gradient residuals, flips."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import noise_floor as NF
import problems as PR
from gpu_common import hip_net, hip_step, math_mode_set

torch.set_num_threads(min(64, os.cpu_count() or 32))
keys = sys.argv[1:] or ['ragged__37x41__mp0', 'ragged__50x70__mp1', 'ragged__64x96__mp0', 'paper__paper_sc_l14__b2', 'paper__paper_mp_l0__b2', 'paper__paper_sc_l14__b16']
for key in keys:
    pr = PR.REGISTRY[key]()
    gc = NF.GradientCheck(pr)
    with math_mode_set('bf16s'):
        net = hip_net(pr)
        out, seg, loss = hip_step(pr, net)
        t0 = time.time()
        res = NF.check_bf16_storage(gc, net, seg, loss.item(), out[1] if isinstance(out, tuple) else None, measure_only=True)
    errs = sorted(((e / b, k, e, b) for k, (kind, e, b) in res['errs'].items()), reverse=True)
    print('%s: d_fwd %.3e d_heat %.3e d_loss %.3e | whole %.3e (bar %.2e) worst %.2f @ %s | flips %d relu %d pool of %d (%.2e), margin %.2e | emu %.1fs' % (
        key, res['d_fwd'], res.get('d_heat', 0), res.get('d_loss', 0), res['whole'], res['bars']['*'], res['worst'], res['worst_k'],
        res['info']['relu_flips'], res['info']['pool_flips'], res['info']['relu_total'], res['flip_frac'], res['info']['max_margin'], time.time() - t0), flush=True)
    for r_, k, e, b in errs[:6]:
        print('    %-45s err %.3e bar %.3e ratio %.2f' % (k, e, b, r_))
    # raw relative errors (no bars): distribution
    rel = sorted((e for k, (kind, e, b) in res['errs'].items() if kind == 'rel'), reverse=True)
    print('    raw rel errors: max %.3e median %.3e' % (rel[0], rel[len(rel) // 2]), flush=True)
