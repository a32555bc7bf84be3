"""SGD with momentum / Nesterov / weight decay, torch.optim.SGD semantics, on the flat parameter arena.

Reference: train.py:287-290 builds ``optim.SGD(net.parameters(), lr, momentum, weight_decay, nesterov)`` and calls
``optimizer.step()`` once per batch (train.py:423).  torch runs that as several multi-tensor kernels over ~140 tensors;
here parameters (UNet._flatten_parameters), gradients (plan.grad_flat) and momentum buffers share one layout, so a step
is one dfl_sgd_step launch per contiguous run of live parameters (two runs for the paper network: the never-used
``downsample_convs[depth-1]`` has no gradient and, exactly as in torch, is skipped -- no weight decay either).
Parameters whose tensors do not line up (foreign modules, accumulated gradients) are updated tensor by tensor with
the same kernel.  CPU tensors are refused: there is no fallback path.
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch
from torch.optim.optimizer import Optimizer, required

from . import _native as nat


class SGD(Optimizer):
    def __init__(self, params, lr=required, momentum=0, dampening=0, weight_decay=0, nesterov=False):
        if lr is not required and lr < 0.0:
            raise ValueError('Invalid learning rate: {}'.format(lr))
        if momentum < 0.0:
            raise ValueError('Invalid momentum value: {}'.format(momentum))
        if weight_decay < 0.0:
            raise ValueError('Invalid weight_decay value: {}'.format(weight_decay))
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        if dampening != 0:
            raise NotImplementedError('dampening != 0 is not implemented in the HIP path (no reference CLI selects it)')
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        self.grad_scale = 1.0          # parallel.DataParallel leaves SUMMED gradients when asked to; see there
        self._lib = nat.lib()
        self._fused_cache = {}         # dfl_sgd_pack_tiled job lists, see _fused_pack

    def _momentum_buffers(self, group):
        """Zero momentum buffers for parameters that have none (torch's first step, buf = g, is mom*0 + g).  When the
        whole group is new and its parameters share one arena, the buffers get one arena with the same layout."""
        ps = [p for p in group['params'] if p.grad is not None and 'momentum_buffer' not in self.state.get(p, {})]
        if not ps:
            return
        fresh = all('momentum_buffer' not in self.state.get(p, {}) for p in group['params'])
        base = min(p.data_ptr() for p in group['params'])
        end = max(p.data_ptr() + 4 * p.numel() for p in group['params'])
        span = (end - base) // 4
        total = sum(p.numel() for p in group['params'])
        if fresh and len({p.device for p in group['params']}) == 1 and span <= total + 4 * len(group['params']):
            flat = torch.zeros(span, dtype=torch.float32, device=ps[0].device)
            for p in ps:                      # like torch: only parameters that received a gradient get state
                o = (p.data_ptr() - base) // 4
                self.state[p]['momentum_buffer'] = flat[o:o + p.numel()].view(p.shape)
        else:
            for p in ps:
                self.state[p]['momentum_buffer'] = torch.zeros_like(p.data)

    def load_state_dict(self, state_dict):
        """torch restores every momentum buffer as a tensor of its own; put them back into one arena laid out like the
        parameters, so that a resumed run keeps the one-launch-per-run update (otherwise: ~135 launches per step)."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            ps = group['params']
            have = [p for p in ps if 'momentum_buffer' in self.state.get(p, {}) and self.state[p]['momentum_buffer'] is not None]
            if not have or not all(p.is_cuda and p.dtype == torch.float32 for p in ps) or len({p.device for p in ps}) != 1:
                continue
            base = min(p.data_ptr() for p in ps)
            end = max(p.data_ptr() + 4 * p.numel() for p in ps)
            span = (end - base) // 4
            if span > sum(p.numel() for p in ps) + 4 * len(ps):
                continue                      # the parameters do not share an arena: nothing to line up with
            flat = torch.zeros(span, dtype=torch.float32, device=ps[0].device)
            with torch.no_grad():
                for p in have:
                    o = (p.data_ptr() - base) // 4
                    view = flat[o:o + p.numel()].view(p.shape)
                    view.copy_(self.state[p]['momentum_buffer'])
                    self.state[p]['momentum_buffer'] = view

    FUSE_PACK = os.environ.get('DFL_SGD_PACK', '1') != '0'

    def _fused_pack(self, net, live, runs, mom):
        """(plan, device job list, jobs, tiles, gradient delta, buffer delta) for dfl_sgd_pack_tiled, or None when the update
        cannot run inside the tiled weight re-layout of the network's training plan: every tiled parameter must be updated by
        this step, and parameters, gradients and momentum buffers must share one layout (one delta for all runs)."""
        plan = net.plan_for_fused_update()
        if plan is None:
            return None
        gd = {gp - pp for pp, gp, bp, n in runs}
        bd = {bp - pp for pp, gp, bp, n in runs} if mom != 0 else {0}
        if len(gd) != 1 or len(bd) != 1:
            return None
        gdelta, bdelta = gd.pop(), bd.pop()
        if gdelta % 16 or bdelta % 16:
            return None
        key = (id(plan), gdelta, bdelta, tuple(p.data_ptr() for p in live))
        hit = self._fused_cache.get(key)
        if hit is not None and hit[0]() is plan:
            return (plan,) + hit[1:]
        raw, ntiled, tiles, tiled_src = plan._tiled_host
        ptrs = {p.data_ptr(): p.numel() for p in live}
        if any(ptrs.get(s) != n for s, n in tiled_src.items()):
            return None                      # a tiled parameter without a gradient (or a view of one): not this path
        plain, cur = [], None                # what has no tiled layout, adjacent slices merged (alignment padding absorbed)
        flat = getattr(net, '_param_flat', None)
        lo = flat.data_ptr() if flat is not None else 0
        hi = lo + 4 * flat.numel() if flat is not None else 0

        def in_arena(ptr):                   # only slices of the network's own flat arena are merged across a gap
            return lo <= ptr < hi
        for p in live:
            pp, n = p.data_ptr(), p.numel()
            if pp in tiled_src:
                cur = None
                continue
            if cur is not None and 0 <= pp - cur[0] - 4 * cur[1] <= 12 and in_arena(cur[0]) and in_arena(pp):
                cur[1] = (pp - cur[0]) // 4 + n          # (the gap is the arena's alignment padding: nobody else's memory)
            else:
                cur = [pp, n]
                plain.append(cur)
        arr = (nat.PackJob * (ntiled + len(plain)))()
        C.memmove(arr, raw, len(raw))
        for i, (pp, n) in enumerate(plain):
            a = arr[ntiled + i]
            a.src, a.A, a.B, a.C, a.kind, a.first_tile = pp, n, 1, 1, nat.PACK_PLAIN, tiles
            tiles += -(-n // nat.SGD_PLAIN_TILE)
        jobs_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(live[0].device)
        if len(self._fused_cache) >= 4:
            self._fused_cache.clear()
        self._fused_cache[key] = (weakref.ref(plan), jobs_dev, len(arr), tiles, gdelta // 4, bdelta // 4)
        return (plan,) + self._fused_cache[key][1:]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = self._lib
        for group in self.param_groups:
            lr, mom, wd, nest = group['lr'], group['momentum'], group['weight_decay'], group['nesterov']
            live = [p for p in group['params'] if p.grad is not None]
            if not live:
                continue
            for p in live:
                if not p.is_cuda or not p.grad.is_cuda:
                    raise nat.DflError('sgd.SGD needs parameters and gradients on the GPU (no CPU path)')
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise nat.DflError('sgd.SGD needs contiguous float32 parameters and gradients')
            if mom != 0:
                self._momentum_buffers(group)
            stream = torch.cuda.current_stream(live[0].device).cuda_stream
            # contiguous runs: parameter, gradient and buffer addresses all advance by the same number of bytes
            runs, cur = [], None
            for p in live:
                pp, gp, n = p.data_ptr(), p.grad.data_ptr(), p.numel()
                bp = self.state[p]['momentum_buffer'].data_ptr() if mom != 0 else 0
                if cur is not None and pp - cur[0] == gp - cur[1] and (mom == 0 or pp - cur[0] == bp - cur[2]) \
                        and 0 <= pp - cur[0] - 4 * cur[3] <= 12:
                    cur[3] = (pp - cur[0]) // 4 + n          # absorbs the alignment padding between slices
                else:
                    cur = [pp, gp, bp, n]
                    runs.append(cur)
            from .unet import owner_of
            net = owner_of(live[0])
            fused = self._fused_pack(net, live, runs, mom) if (net is not None and self.FUSE_PACK and len(self.param_groups) == 1) else None
            if fused is not None:
                # one pass over the weights: the workgroups of the tiled re-layout update their tile first (dfl_sgd_pack_tiled)
                plan, jobs_dev, njobs, tiles, gdelta, bdelta = fused
                a = nat.SgdPackArgs(jobs_dev=jobs_dev.data_ptr(), grad_delta=gdelta, buf_delta=bdelta, njobs=njobs, total_tiles=tiles,
                                    lr=lr, momentum=mom, weight_decay=wd, grad_scale=self.grad_scale, nesterov=int(nest))
                nat.check(lib.dfl_sgd_pack_tiled(C.addressof(a), stream), 'dfl_sgd_pack_tiled')
                torch.autograd.graph.increment_version(live)
                net.after_fused_update(plan, stream)
                continue
            for pp, gp, bp, n in runs:
                nat.check(lib.dfl_sgd_step(pp, gp, bp or None, n, lr, mom, wd, self.grad_scale, int(nest), 0, stream),
                          'dfl_sgd_step')
            # the kernel wrote behind autograd's back: bump the version counters like an in-place torch op would (the
            # network re-packs its weights when they move, and autograd must see saved tensors as modified)
            torch.autograd.graph.increment_version(live)
            if net is not None:
                net.prepack()            # next step's weight re-layout starts now, behind the update kernels
        return loss
