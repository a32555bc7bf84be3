"""Data-parallel plumbing on CPU: bucket construction over the real backward program of a plan built on the CPU
(addresses are never dereferenced here) and a world_size-2 gloo run of DataParallel._run_backward over a stand-in
plan whose "backward" writes rank-dependent gradients -- checks that every live gradient range is averaged exactly once,
the dead parameter is skipped, and the segment order matches gradient readiness."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dfl_amd
from dfl_amd.parallel import DataParallel, bucket_ranges, merge_ranges
from dfl_amd.plan import UNetPlan

CFG = dict(n_classes=7, depth=4, wf=3, batch_norm=True, padding=True, max_pool=False, num_lands=14, do_res=True, block_depth=2)


def _plan():
    net = dfl_amd.UNet(**CFG)
    P, B = net._state()
    return net, UNetPlan(net._cfg, P, B, 2, 32, 32, True, True, torch.device('cpu'))


def test_merge_ranges():
    assert merge_ranges([(0, 10), (12, 20), (40, 50), (20, 30)]) == [(0, 30), (40, 50)]


def test_buckets_cover_every_live_gradient_once():
    net, plan = _plan()
    sizes = {k: plan.P[k].numel() for k in plan.grad_names}
    ready = plan.grad_ready_op
    live = [k for k in plan.grad_names if k not in plan.dead_params]
    assert set(ready) == set(live)                       # every live gradient is written by some backward op
    assert plan.dead_params == {'downsample_convs.3.weight', 'downsample_convs.3.bias'}
    order = sorted(live, key=lambda k: ready[k])
    buckets = bucket_ranges(plan.grad_offsets, sizes, order, plan.dead_params, bucket_elems=20000)
    assert len(buckets) > 3
    covered = torch.zeros(plan.grad_flat.numel(), dtype=torch.int32)
    for _, ranges in buckets:
        for s, e in ranges:
            covered[s:e] += 1
    for k in live:
        s = plan.grad_offsets[k]
        assert int(covered[s:s + sizes[k]].min()) == 1 and int(covered[s:s + sizes[k]].max()) == 1, k
    for k in plan.dead_params:
        s = plan.grad_offsets[k]
        assert int(covered[s:s + sizes[k]].max()) == 0
    # the heads / last decoder block are final first (small sums wait for a batched flush), the first encoder block last
    assert order[0].startswith(('seg_conv', 'lands_1x1', 'up_path.%d.' % (net.depth - 2)))
    assert max(ready.values()) == max(v for k, v in ready.items() if k.startswith('down_path.0.'))


class _FakeProgram:
    def __init__(self, plan, rank):
        self.plan, self.rank, self.calls = plan, rank, []

    def __len__(self):
        return self.plan.n_ops

    def run(self, stream, start=0, count=None):
        count = self.plan.n_ops - start if count is None else count
        self.calls.append((start, count))
        for name, op in self.plan.grad_ready_op.items():
            if start <= op < start + count:
                s = self.plan.grad_offsets[name]
                n = self.plan.P[name].numel()
                self.plan.grad_flat[s:s + n] = float(self.rank + 1) * (1 + op)


class _FakePlan:
    def __init__(self, real, rank):
        self.P, self.grad_names, self.grad_offsets = real.P, real.grad_names, real.grad_offsets
        self.dead_params, self.grad_ready_op = real.dead_params, real.grad_ready_op
        self.n_ops = len(real.bwd)
        self.grad_flat = torch.full((real.grad_flat.numel(),), -7.0)
        self.bwd = _FakeProgram(self, rank)


def _worker(rank, world, port, compress=None):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)                       # different initial weights per rank: broadcast must fix that
        net, real = _plan()
        dp = DataParallel(net, bucket_mb=0.05, compress=compress)
        w0 = [p.detach().clone() for p in net.parameters()]
        gathered = [torch.zeros_like(w0[0]) for _ in range(world)]
        dist.all_gather(gathered, w0[0])
        assert all(torch.equal(gathered[0], g) for g in gathered[1:])
        plan = _FakePlan(real, rank)
        dp._run_backward(plan, None)
        # the fake backward ran segment by segment over the whole program, in order
        calls = plan.bwd.calls
        assert calls[0][0] == 0 and sum(c for _, c in calls) == plan.n_ops
        assert all(calls[i][0] + calls[i][1] == calls[i + 1][0] for i in range(len(calls) - 1))
        mean_factor = sum(r + 1 for r in range(world)) / world
        for name, op in plan.grad_ready_op.items():
            s = plan.grad_offsets[name]
            n = plan.P[name].numel()
            got = plan.grad_flat[s:s + n]
            # bf16 buckets: every rank's value and the sum are rounded to bf16 (2^-9 each)
            assert torch.allclose(got, torch.full_like(got, mean_factor * (1 + op)), rtol=1e-2 if compress else 1e-5), name
        for name in plan.dead_params:
            s = plan.grad_offsets[name]
            assert float(plan.grad_flat[s]) == -7.0     # untouched
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,compress', [(2, None), (4, None), (2, 'bf16')])
def test_data_parallel_gloo(world, compress):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, compress), nprocs=world, join=True)


def test_bucket_of_several_slices_is_one_collective(monkeypatch):
    """A bucket whose arena slices do not touch is gathered into one staging buffer: one all_reduce per bucket."""
    net, real = _plan()
    dp = DataParallel(net)
    dp.world, dp.active = 2, True
    calls = []
    monkeypatch.setattr(dist, 'all_reduce', lambda t, op=None, group=None: calls.append(t.numel()))
    plan = _FakePlan(real, 0)
    plan.grad_flat[:] = torch.arange(plan.grad_flat.numel(), dtype=torch.float32)
    before = plan.grad_flat.clone()
    dp._reduce_bucket(plan, 0, [(0, 100), (200, 260), (1000, 1004)], 0.5)
    assert calls == [164]
    touched = torch.zeros_like(before, dtype=torch.bool)
    for s_, e_ in ((0, 100), (200, 260), (1000, 1004)):
        touched[s_:e_] = True
    assert torch.equal(plan.grad_flat[~touched], before[~touched])
    assert torch.allclose(plan.grad_flat[touched], before[touched] * 0.5)     # the (mocked) sum, then the mean
