"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl")
gradient all-reduce over xGMI.

The reference has no distributed code; this is the north-star extension: every rank runs the same
U-Net on its shard of the minibatch and the parameter gradients are averaged.  Both losses are means over per-image
terms, so the averaged gradients equal those of the global batch; BatchNorm statistics stay per replica (= the
reference run at the per-replica batch size).

Mechanics: the backward program writes every parameter gradient into one flat fp32 arena (plan.grad_flat).  The
program is cut into segments at bucket boundaries; after each segment an event is recorded and the bucket's slice of
the arena is all-reduced on a side stream while the next segment's kernels run, i.e. communication of the decoder /
deep-encoder gradients overlaps the remaining backward convolutions.  The parameter created but never used by the
reference (downsample_convs[depth-1]) has no gradient and is skipped.
"""
import torch
import torch.distributed as dist

__all__ = ['DataParallel', 'init_process_group_from_env']


def init_process_group_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    ``force``: also with WORLD_SIZE 1 (a one-rank RCCL group: the self-test of the collective path on a 1-GPU box)."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def bucket_ranges(offsets, sizes, order, dead, bucket_elems):
    """Greedy contiguous buckets over the flat arena.  ``order``: parameter names in the order their gradients become
    final during backward.  Returns a list of (ready_after_name, [(start, stop), ...]) -- the element ranges to
    reduce once ``ready_after_name``'s gradient has been written."""
    buckets, cur, cur_n, last = [], [], 0, None
    for name in order:
        if name in dead:
            continue
        s = offsets[name]
        cur.append((s, s + sizes[name]))
        cur_n += sizes[name]
        last = name
        if cur_n >= bucket_elems:
            buckets.append((last, merge_ranges(cur)))
            cur, cur_n = [], 0
    if cur:
        buckets.append((last, merge_ranges(cur)))
    return buckets


def merge_ranges(ranges, align=4):
    """Merge element ranges that touch (allowing for the <=3-element alignment gaps of the arena)."""
    out = []
    for s, e in sorted(ranges):
        if out and s - out[-1][1] < align:
            out[-1] = (out[-1][0], max(out[-1][1], e))
        else:
            out.append((s, e))
    return out


class DataParallel:
    """Wraps a dfl_amd.UNet for data-parallel training.  Usage::

        net = UNet(...).to(dev); dp = DataParallel(net)      # broadcasts rank 0's parameters and buffers
        out = net(x_shard); loss.backward()                   # gradients arrive averaged over ranks
    """

    def __init__(self, net, process_group=None, bucket_mb=32.0, overlap=True, force_collectives=False, compress=None):
        """compress: None -- buckets travel as the fp32 values of the gradient arena; 'bf16' -- a bucket is rounded to bf16
        into a staging buffer, summed over the ranks in bf16 and widened back (half the xGMI bytes per step; the ring
        collective is per-link bound, so this is the lever when communication does not hide behind backward.  The sum of
        `world` bf16 values carries a 2^-9 relative error: an option for throughput runs, off by default)."""
        if compress not in (None, 'bf16'):
            raise ValueError("compress must be None or 'bf16'")
        self.compress = compress
        self.net = net
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # collectives run when there is more than one rank -- or on request in a one-rank group (self-test: the same
        # broadcast / bucketed all-reduce / stream choreography, numerically the identity)
        self.active = dist.is_initialized() and (self.world > 1 or force_collectives)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.overlap = overlap
        self.comm_stream = None
        # RCCL averages inside the collective (ReduceOp.AVG); gloo (CPU tests, single-GPU self-test) sums, then we scale
        self._avg_in_collective = False
        if self.active and dist.get_backend(process_group) == 'nccl':
            try:                                   # probe once (collective: every rank takes the same branch)
                probe = torch.ones(1, device=next(net.parameters()).device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                self._avg_in_collective = abs(float(probe.item()) - 1.0) < 1e-6
            except Exception:
                self._avg_in_collective = False
        if self.active:
            with torch.no_grad():
                for t in list(net.parameters()) + list(net.buffers()):
                    dist.broadcast(t, src=0, group=process_group)
        net._backward_runner = self._run_backward
        net.dp = self

    def sync_buffers(self, src=0):
        """Broadcast rank `src`'s buffers (BatchNorm running statistics, batch counters).  The statistics follow each replica's
        own shards during training and drift apart; before anything that READS them on several ranks and must agree with what
        rank 0 saves -- sharded validation, checkpoints -- every rank takes rank `src`'s."""
        if not self.active:
            return
        with torch.no_grad():
            for b in self.net.buffers():
                dist.broadcast(b, src=src, group=self.group)
        # collectives write the buffers without a version bump: inference plans keep scale / shift derived from them in
        # their pack program and must derive them again
        if hasattr(self.net, '_bn_epoch'):
            self.net._bn_epoch += 1

    # -------------------------------------------------------------------------------------------------------
    def _segments(self, plan):
        """[(op_start, op_count, [(start, stop), ...])]: run ops, then reduce those arena ranges."""
        seg = getattr(plan, '_dp_segments', None)      # kept ON the plan: dies with it (a dict keyed by id(plan) could hand
        if seg is not None and seg[0] == self.bucket_elems:   # a recycled id the cut points of a dead plan)
            return seg[1]
        ready = plan.grad_ready_op            # name -> index of the last op that writes this gradient
        sizes = {k: plan.P[k].numel() for k in plan.grad_names}
        order = sorted((k for k in plan.grad_names if k not in plan.dead_params), key=lambda k: ready[k])
        buckets = bucket_ranges(plan.grad_offsets, sizes, order, plan.dead_params, self.bucket_elems)
        seg, start = [], 0
        for last_name, ranges in buckets:
            stop = ready[last_name] + 1
            seg.append((start, stop - start, ranges))
            start = stop
        n_ops = len(plan.bwd)
        if start < n_ops:
            seg.append((start, n_ops - start, []))
        plan._dp_segments = (self.bucket_elems, seg)
        return seg

    def _run_ops(self, plan, stream, start=0, count=None):
        """A segment of the backward program; on the GPU through the network's replay (hipGraph per segment, unet.run_program)."""
        run = getattr(self.net, 'run_program', None)
        if run is not None and plan.grad_flat.is_cuda:
            run(plan, plan.bwd, stream, start, count)
        else:
            plan.bwd.run(stream, start, count)

    def _reduce(self, view, inv):
        if self._avg_in_collective:
            dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(view, group=self.group)
            view.mul_(inv)

    def _staging(self, plan, key, n, dtype):
        """Staging buffers live on the plan (one per bucket and dtype): allocated once, reused every step."""
        bufs = plan.__dict__.setdefault('_dp_staging', {})
        buf = bufs.get((key, dtype))
        if buf is None or buf.numel() < n:
            buf = torch.empty(n, dtype=dtype, device=plan.grad_flat.device)
            bufs[(key, dtype)] = buf
        return buf[:n]

    def _reduce_bucket(self, plan, key, ranges, inv):
        """ONE collective per bucket.  A bucket that is one contiguous slice of the arena is reduced in place; several
        slices (the greedy cut met an alignment gap or a dead parameter) are gathered into a staging buffer first.  With
        compress='bf16' the staging buffer is bf16."""
        flat = plan.grad_flat
        if self.compress is None and len(ranges) == 1:
            s, e = ranges[0]
            self._reduce(flat[s:e], inv)
            return
        n = sum(e - s for s, e in ranges)
        dtype = torch.bfloat16 if self.compress == 'bf16' else flat.dtype
        buf = self._staging(plan, key, n, dtype)
        o = 0
        for s, e in ranges:
            buf[o:o + e - s].copy_(flat[s:e])
            o += e - s
        if dtype == torch.bfloat16:
            dist.all_reduce(buf, group=self.group)          # sum in bf16; the mean is taken in fp32 below
            o = 0
            for s, e in ranges:
                flat[s:e].copy_(buf[o:o + e - s])
                flat[s:e].mul_(inv)
                o += e - s
            return
        self._reduce(buf, inv)
        o = 0
        for s, e in ranges:
            flat[s:e].copy_(buf[o:o + e - s])
            o += e - s

    def _run_backward(self, plan, stream):
        if not self.active:
            self._run_ops(plan, stream)
            getattr(plan, 'unfold_tail_grads', lambda: None)()
            return
        flat = plan.grad_flat
        inv = 1.0 / self.world
        if not self.overlap or not flat.is_cuda:
            # same segment walk, communication in line (also the path the CPU/gloo tests exercise)
            segs = self._segments(plan)
            for k, (op_start, op_count, ranges) in enumerate(segs):
                self._run_ops(plan, stream, op_start, op_count)
                if op_start + op_count == len(plan.bwd):
                    getattr(plan, 'unfold_tail_grads', lambda: None)()      # gradients made after the last op (plan.py)
                if ranges:
                    self._reduce_bucket(plan, k, ranges, inv)
            return
        cur = torch.cuda.current_stream()
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream()
        comm = self.comm_stream
        events = plan.__dict__.setdefault('_dp_events', {})   # one event per bucket, created once and re-recorded every step
        for k, (op_start, op_count, ranges) in enumerate(self._segments(plan)):
            self._run_ops(plan, stream, op_start, op_count)
            if op_start + op_count == len(plan.bwd):
                getattr(plan, 'unfold_tail_grads', lambda: None)()
            if ranges:
                ev = events.get(k)
                if ev is None:
                    ev = events[k] = torch.cuda.Event()
                ev.record(cur)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    self._reduce_bucket(plan, k, ranges, inv)
        cur.wait_stream(comm)
