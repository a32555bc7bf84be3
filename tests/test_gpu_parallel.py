"""Data-parallel path on the GPU: two ranks (gloo transport, sharing GPU 0) and a one-rank RCCL group run tests/dp_worker.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_average_gradients_and_stay_in_sync():
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'tests', 'dp_worker.py')]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert 'DP_OK rank=0' in out and 'DP_OK rank=1' in out, out[-3000:]


def test_one_rank_rccl_group_runs_the_nccl_path():
    """RCCL init, ReduceOp.AVG, broadcast and the overlapped bucketed all-reduce on the comm stream, on the one GPU of
    the test box (the 8-GPU run is the driver's): results must equal the purely local ones."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['DP_BACKEND'] = 'nccl'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29534', os.path.join(ROOT, 'tests', 'dp_worker.py')]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert 'DP_OK rank=0' in out, out[-3000:]


def test_bench_under_torchrun_one_rank_rccl():
    """bench.py exactly as the driver launches it for N > 1, with one rank (--force-dp keeps the collective path on)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29535', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline', '--no-profile', '--no-fp32-reference', '--force-dp', '--prewarm-seconds', '0']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    import json
    line = [l for l in p.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 100 and d['config']['parallelism'] == 'dp1' and d['rccl_ranks'] == 1


def test_gradient_readiness_leaves_room_for_overlap():
    """Data-parallel overlap is only possible if gradients become FINAL early in the backward pass: on the real program of the
    benchmarked step (paper preset, batch 16, bf16 storage) -- its per-op hipEvent times, the op that last writes each gradient
    (plan.grad_ready_op, batched sums included) and the bucket cuts DataParallel makes -- at least 75 % of the gradient bytes
    must be final before the last 25 % of the backward kernel time, and every bucket but the last must have kernel time
    left behind it to hide its all-reduce.  Prints the per-bucket table DESIGN.md section 6 quotes."""
    import torch
    import dfl_amd
    from dfl_amd import _native as nat
    from dfl_amd.parallel import DataParallel
    import bench
    lib = nat.lib()
    prev = lib.dfl_get_math_mode()
    nat.check(lib.dfl_set_math_mode(4), 'dfl_set_math_mode')
    try:
        torch.manual_seed(1)
        net = dfl_amd.UNet(**bench.PAPER).to('cuda').train()
        x, tseg, theat = bench.synth_batch(16, 7, 'cuda')
        crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
        seg, heat = net(x)
        crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat)).backward()
        torch.cuda.synchronize()
        plan = net._last_train_plan()
        seg, heat = net(x)                                   # fresh activations for the timed replay
        hold = (torch.randn_like(seg) * 1e-6, torch.randn_like(heat) * 1e-6)
        plan.bind_grads(seg, hold[0], hold[1])
        stream = torch.cuda.current_stream().cuda_stream
        plan.bwd.run(stream)
        ms = plan.bwd.run_timed(stream)
        plan.busy = False
    finally:
        nat.check(lib.dfl_set_math_mode(prev), 'dfl_set_math_mode')
    total = sum(ms)
    done_at, acc = [], 0.0
    for t in ms:
        acc += t
        done_at.append(acc)                                  # time at which op i has finished
    dp = DataParallel.__new__(DataParallel)                  # the bucket logic only: no process group on this box
    dp.bucket_elems = int(32.0 * (1 << 20) / 4)
    segs = dp._segments(plan)
    nbytes = {k: 4 * plan.P[k].numel() for k in plan.grad_names if k not in plan.dead_params}
    all_bytes = float(sum(nbytes.values()))
    early = sum(b for k, b in nbytes.items() if done_at[plan.grad_ready_op[k]] <= 0.75 * total)
    print('backward kernel time %.3f ms, %d ops; gradient bytes final before 75 %% of it: %.1f %%' % (total, len(ms), 100 * early / all_bytes))
    ring_ms_per_mb = 2.0 * 7 / 8 / 153e3 * 1e3 * 1.048576       # 8-GPU ring all-reduce, one xGMI link of ~153 GB/s per hop
    for k, (op0, cnt, ranges) in enumerate(segs):
        if not ranges:
            continue
        mb = sum(e - s for s, e in ranges) * 4 / 2 ** 20
        t_ready = done_at[op0 + cnt - 1]
        print('  bucket %d: %6.1f MB ready at %.3f ms (%.0f %% of backward), %.3f ms of kernels behind it, ring all-reduce ~%.2f ms' % (
            k, mb, t_ready, 100 * t_ready / total, total - t_ready, mb * ring_ms_per_mb))
    assert early / all_bytes >= 0.75, 'only %.1f %% of the gradient bytes are final before the last quarter of backward' % (100 * early / all_bytes)
    ready = [done_at[op0 + cnt - 1] for op0, cnt, ranges in segs if ranges]
    assert all(t < 0.9 * total for t in ready[:-1]), 'a bucket other than the last becomes ready in the last 10 % of backward'


@pytest.mark.parametrize('batch', [2])
def test_every_gradient_is_final_behind_the_op_the_plan_names(batch):
    """plan.grad_ready_op is what DataParallel cuts its buckets by: behind op grad_ready_op[name] the gradient `name` must hold
    its final value (the zero fill inserted in front of a bf16-storage backward program had moved every batched
    sum's index by one, so a bucket could be all-reduced before its last gradient was written).  The backward program of a
    bf16-storage paper plan is replayed piecewise over a NaN-filled gradient arena and every gradient is compared, bit for
    bit, with the full replay right behind the op that is said to complete it -- and must NOT be final one op earlier."""
    import torch
    import dfl_amd
    from dfl_amd import _native as nat
    import bench
    lib = nat.lib()
    prev = lib.dfl_get_math_mode()
    nat.check(lib.dfl_set_math_mode(4), 'dfl_set_math_mode')
    try:
        torch.manual_seed(3)
        net = dfl_amd.UNet(**bench.PAPER).to('cuda').train()
        x, tseg, theat = bench.synth_batch(batch, 11, 'cuda')
        crit = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=0.5)
        seg, heat = net(x)
        crit((dfl_amd.center_crop(seg, tseg.shape), dfl_amd.center_crop(heat, theat.shape)), (tseg, theat)).backward()
        torch.cuda.synchronize()
        plan = net._last_train_plan()
        seg, heat = net(x)
        hold = (torch.randn_like(seg) * 1e-3, torch.randn_like(heat) * 1e-3)
        plan.bind_grads(seg, hold[0], hold[1])
        stream = torch.cuda.current_stream().cuda_stream
        plan.bwd.run(stream)
        torch.cuda.synchronize()
        ref = plan.grad_flat.clone()
        ready = {k: v for k, v in plan.grad_ready_op.items() if k not in plan.dead_params}
        assert set(ready) == set(k for k in plan.grad_names if k not in plan.dead_params)
        cuts = sorted(set(ready.values()))
        plan.grad_flat.fill_(float('nan'))
        offs = {k: (plan.G[k].data_ptr() - plan.grad_flat.data_ptr()) // 4 for k in ready}
        done, early = 0, []
        for c in cuts:
            if c > done:                                       # ops [done, c): everything in front of the completing op
                plan.bwd.run(stream, done, c - done)
            torch.cuda.synchronize()
            for k in [k for k, v in ready.items() if v == c]:
                o, n = offs[k], plan.P[k].numel()
                if torch.equal(plan.grad_flat[o:o + n], ref[o:o + n]):
                    early.append(k)
            plan.bwd.run(stream, c, 1)
            done = c + 1
            torch.cuda.synchronize()
            for k in [k for k, v in ready.items() if v == c]:
                o, n = offs[k], plan.P[k].numel()
                assert torch.equal(plan.grad_flat[o:o + n], ref[o:o + n]), '%s is not final behind op %d (%s)' % (
                    k, c, type(plan.bwd.structs[c]).__name__)
        if done < len(plan.bwd.structs):
            plan.bwd.run(stream, done, len(plan.bwd.structs) - done)
        torch.cuda.synchronize()
        plan.busy = False
        assert not early, 'final before the op that is said to write them: %s' % early[:5]
    finally:
        nat.check(lib.dfl_set_math_mode(prev), 'dfl_set_math_mode')


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver types it for N = 1 (no torchrun in front): bench.py re-executes itself
    under torch.distributed.run with one rank per GPU.  Two gloo ranks share the one GPU of the test box."""
    import json
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline', '--no-profile', '--no-fp32-reference', '--no-fwd', '--no-configs3', '--prewarm-seconds', '0']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    # (the value is whatever two ranks sharing one GPU and all-reducing 152 MB through gloo's host buffers reach: not a measurement)
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['value'] > 0 and d['scaling'] == 'weak' and d['steps'] == 3
