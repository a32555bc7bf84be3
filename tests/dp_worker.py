"""Worker of tests/test_gpu_parallel.py: one rank of a data-parallel run on one GPU -- 2 processes sharing it over gloo
(DP_BACKEND unset), or ONE process in a one-rank RCCL group (DP_BACKEND=nccl: RCCL refuses two ranks on one device, so
this is how the nccl code path -- init, ReduceOp.AVG probe, broadcast, bucketed all-reduce on the comm stream -- runs
on a 1-GPU box).

Checks, on every rank: (1) parameters equal rank 0's after DataParallel construction; (2) after backward the gradients
equal the mean over ranks of the gradients each rank computes alone on its shard; (3) overlapped and in-line
communication give identical gradients; (4) after one dfl_amd.SGD step the parameters are identical on all ranks.
Prints 'DP_OK rank=<r>' on success.
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import dfl_amd  # noqa: E402
from dfl_amd.parallel import DataParallel, init_process_group_from_env  # noqa: E402


def grads_of(net):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()]).clone()


def main():
    backend = os.environ.get('DP_BACKEND', 'gloo')
    rank, world, local = init_process_group_from_env(backend, force=True)
    assert world == (1 if backend == 'nccl' else 2)
    assert dist.get_backend() == backend
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    cfg = dict(n_classes=4, depth=3, wf=3, batch_norm=True, padding=True, max_pool=False, num_lands=3)
    torch.manual_seed(100 + rank)                      # different initial weights per rank: the broadcast must fix that
    net = dfl_amd.UNet(1, **cfg).to(dev)
    g = torch.Generator().manual_seed(7 + rank)        # different shard per rank
    x = torch.randn(2, 1, 32, 32, generator=g).to(dev)
    tseg = torch.softmax(torch.randn(2, 4, 32, 32, generator=g), 1).to(dev)
    theat = (torch.rand(2, 3, 32, 32, generator=g) * 0.02).to(dev)
    crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)

    def fwd_bwd():
        net.zero_grad()
        seg, heat = net(x)
        loss = crit((seg, heat), (tseg, theat))
        loss.backward()
        torch.cuda.synchronize()
        return grads_of(net)

    dp = DataParallel(net, bucket_mb=0.02, overlap=True, force_collectives=True)     # tiny buckets: several segments even for this toy net
    assert dp.active and (backend != 'nccl' or dp._avg_in_collective), 'RCCL should average inside the collective'
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat, ref), 'parameters differ from rank 0 after DataParallel()'
    net.train()
    # (2) local gradients with communication switched off, then their mean over ranks
    dp.active = False
    local_g = fwd_bwd()
    dp.active = True
    mean_g = local_g.clone()
    dist.all_reduce(mean_g)
    mean_g /= world
    got = fwd_bwd()
    assert len(dp._segments(next(p for ps in net._plans.values() for p in ps if p.need_grad))) > 2
    err = float((got - mean_g).abs().max()) / max(float(mean_g.abs().max()), 1e-12)
    assert err < 1e-6, 'averaged gradients differ from the mean of the local ones: %g' % err
    # (3) in-line communication gives the same numbers
    dp.overlap = False
    got2 = fwd_bwd()
    assert torch.equal(got, got2), 'overlapped and in-line reductions differ'
    dp.overlap = True
    # (4) one optimizer step keeps the replicas identical
    opt = dfl_amd.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    fwd_bwd()
    opt.step()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat, ref), 'replicas diverged after the optimizer step'
    print('DP_OK rank=%d' % rank, flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
