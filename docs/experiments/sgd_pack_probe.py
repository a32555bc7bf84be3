"""Which parameters / steps differ between the update inside the tiled re-layout (dfl_sgd_pack_tiled) and update + re-layout."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
import dfl_amd
from dfl_amd import _native as nat
import problems as PR
from gpu_common import hip_net, hip_step
nat.check(nat.lib().dfl_set_math_mode(4))
pr = PR.REGISTRY[sys.argv[1] if len(sys.argv) > 1 else 'paper__paper_sc_l14__b2']()
res = []
for fuse in (True, False):
    net = hip_net(pr)
    opt = dfl_amd.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
    opt.FUSE_PACK = fuse
    log = []
    exp = {}
    for step in range(3):
        opt.zero_grad()
        out, seg, loss = hip_step(pr, net)
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        if step == 0:
            for k, q in net.named_parameters():
                if k.endswith('up.weight') or k == 'down_path.1.block.0.weight':
                    w = q.detach().double(); g = q.grad.double() + 1e-3 * w; want = (w - 0.05 * (g + 0.9 * g)).float()
                    old = q.detach().clone()
                    exp[k] = (want, old)
        opt.step()
        if step == 0:
            for k, q in net.named_parameters():
                if k in exp:
                    want, old = exp[k]
                    print(fuse, k, tuple(q.shape), 'err vs formula %.3e' % float((q.detach() - want).abs().max()), 'update size %.3e' % float((q.detach() - old).abs().max()),
                          'grad ptr delta', q.grad.data_ptr() - q.data_ptr(), 'buf delta', opt.state[q]['momentum_buffer'].data_ptr() - q.data_ptr(), 'wd part %.3e' % float(0.05*1.9*1e-3*old.abs().max()))
        torch.cuda.synchronize()
        log.append((seg.detach().clone(), grads, {k: p.detach().clone() for k, p in net.named_parameters()}))
    res.append(log)
for step in range(3):
    a, b = res[0][step], res[1][step]
    print('step', step, 'seg equal', torch.equal(a[0], b[0]))
    bad = [k for k in b[1] if not torch.equal(a[1][k], b[1][k])]
    print('  grads differing:', len(bad), bad[:6])
    bad = [(k, float((a[2][k] - b[2][k]).abs().max())) for k in b[2] if not torch.equal(a[2][k], b[2][k])]
    print('  params differing:', len(bad), bad[:8])
