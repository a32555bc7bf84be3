"""U-Net with residual Conv3x3 -> ReLU -> BatchNorm blocks, two heads -- MI355X (gfx950) implementation.

Drop-in for the reference's ``train_test_code/unet.py``: same constructor (unet.py:41-45), same module tree, hence the
same ``state_dict`` keys / parameter order / seeded initialisation, same forward
contract (``seg`` or ``(seg, heat_maps)``, unet.py:183-193) and ordinary autograd outputs (``loss.backward()`` fills
``.grad`` for ``torch.optim``).  What differs is everything underneath: ``forward`` does not call ``torch.nn`` -- it
replays a recorded program of hand-written HIP kernels from ``libdfl_hip.so`` (see ``plan.py``, ``include/dfl_hip.h``);
the sub-modules below only own the parameters.  There is no CPU or eager fallback: tensors must live on the GPU.
"""
import os
import weakref

import torch
from torch import nn

from . import _native as nat
from .plan import UNetPlan, PlanError

__all__ = ['UNet', 'UNetConvBlock', 'UNetUpBlock']


def _no_forward(name):
    def forward(self, *a, **k):
        raise RuntimeError('%s is a parameter container; the computation runs in UNet.forward (HIP program)' % name)
    return forward


class UNetConvBlock(nn.Module):
    """Parameters of one block (reference: unet.py:196-224): optional 1x1 residual conv, then block_depth x
    [Conv3x3, ReLU, (BatchNorm)] kept in a Sequential so the state_dict indices match (0,2,3,5 / 0,2)."""

    def __init__(self, in_size, out_size, padding, batch_norm, pad_mode, do_res, block_depth):
        super().__init__()
        if block_depth <= 0:
            raise AssertionError('block_depth must be positive')
        self.do_res = do_res
        if do_res:                                   # created first: RNG order of the reference
            self.res_conv1x1 = nn.Conv2d(in_size, out_size, kernel_size=1, padding=0)
        mods = []
        cin = in_size
        for _ in range(block_depth):
            mods += [nn.Conv2d(cin, out_size, kernel_size=3, padding=int(padding), padding_mode=pad_mode), nn.ReLU()]
            if batch_norm:
                mods.append(nn.BatchNorm2d(out_size))
            cin = out_size
        self.block = nn.Sequential(*mods)

    forward = _no_forward('UNetConvBlock')


class UNetUpBlock(nn.Module):
    """Parameters of one decoder stage (reference: unet.py:236-246)."""

    def __init__(self, in_size, out_size, up_mode, padding, batch_norm, pad_mode, do_res, block_depth):
        super().__init__()
        if up_mode == 'upconv':
            self.up = nn.ConvTranspose2d(in_size, out_size, kernel_size=2, stride=2)
        else:                                       # reference unet.py:242-244; keys up.1.weight / up.1.bias
            self.up = nn.Sequential(nn.Upsample(mode='bilinear', scale_factor=2), nn.Conv2d(in_size, out_size, kernel_size=1))
        self.conv_block = UNetConvBlock(in_size, out_size, padding, batch_norm, pad_mode, do_res, block_depth)

    forward = _no_forward('UNetUpBlock')


_NETS = weakref.WeakValueDictionary()     # id(net) -> net, for parameters to find their network (plain ints pickle)


def owner_of(param):
    """The UNet a parameter belongs to, or None."""
    net = _NETS.get(getattr(param, '_dfl_net_id', None))
    return net if net is not None and any(param is q for q in net.parameters()) else None


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, plan, x, *params):
        ctx.x_meta = (x.shape, x.dtype) if plan.input_grad else None
        seg, heat = net._run_forward(plan, x.detach().to(torch.float32).contiguous())
        ctx.net, ctx.plan, ctx.gen = net, plan, plan.generation
        ctx.save_for_backward(seg)
        ctx.n_params = len(params)
        return (seg, heat) if heat is not None else seg

    @staticmethod
    def backward(ctx, *gouts):
        plan, net = ctx.plan, ctx.net
        if plan.generation != ctx.gen or not plan.busy:
            raise RuntimeError('the activations of this forward pass were overwritten by a later forward of the same '
                               'network; run backward before the next-but-one forward')
        (seg,) = ctx.saved_tensors
        dseg = gouts[0]
        dheat = gouts[1] if len(gouts) > 1 else None
        if dseg is None:
            dseg = torch.zeros_like(seg)
        dseg = dseg.contiguous()
        if dheat is not None:
            dheat = dheat.contiguous()
        plan.bind_grads(seg, dseg, dheat)
        stream = torch.cuda.current_stream().cuda_stream
        params = net._param_list
        # a .grad that still aliases this plan's arena (no zero_grad since the last backward) would be overwritten by
        # the program: detach it into its own storage first so that accumulation adds old + new
        lo = plan.grad_flat.data_ptr()
        hi = lo + 4 * plan.grad_flat.numel()
        for p in params:
            if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                p.grad = p.grad.clone()
        net._backward_runner(plan, stream)
        plan.busy = False
        grads = plan.grads()
        dx = None
        if ctx.x_meta is not None:                   # NHWC gradient of the plan -> the caller's [B, C, H, W]
            shape, dtype = ctx.x_meta
            d = plan.dx_in
            dx = d.t.view(d.N, d.H, d.W, d.ld)[..., :d.C].permute(0, 3, 1, 2).to(dtype).contiguous().view(shape)
        # Fast hand-off: the gradients are views of the plan's flat arena.  Returning them makes autograd's
        # AccumulateGrad clone every one (135 small copies per step), so when no parameter has hooks and every .grad
        # is empty (the zero_grad() -> backward() pattern of train.py:405-422) the views are installed as .grad
        # directly and autograd gets None.  Otherwise the views are returned and autograd accumulates them.
        if net.direct_grad and all(p.grad is None and not p._backward_hooks for p in params):
            for g, p in zip(grads, params):
                if g is not None and p.requires_grad:
                    p.grad = g
            return (None, None, dx) + (None,) * len(params)
        out = []
        for g, p in zip(grads, params):
            out.append(g if p.requires_grad else None)
        return (None, None, dx) + tuple(out)


class UNet(nn.Module):
    def __init__(self, in_channels=1, n_classes=2, depth=5, wf=6, padding=False, pad_mode='zeros', batch_norm=False,
                 up_mode='upconv', max_pool=True, num_lands=0, do_res=True, block_depth=2, lands_block_depth=0,
                 lands_num_1x1=2, do_soft_max=True):
        super().__init__()
        if up_mode not in ('upconv', 'upsample'):
            raise AssertionError("up_mode must be 'upconv' or 'upsample'")
        if pad_mode not in ('zeros', 'circular'):
            raise NotImplementedError("pad_mode='%s' is not implemented in the HIP path (the reference's flag values: 'zeros', "
                                      "'circular')" % pad_mode)
        if num_lands > 0 and lands_num_1x1 < 1:
            raise AssertionError('lands_num_1x1 must be positive')
        self.padding, self.pad_mode, self.depth = padding, pad_mode, depth
        self.do_max_pool, self.num_lands, self.do_soft_max = max_pool, num_lands, do_soft_max
        self._cfg = dict(in_channels=in_channels, n_classes=n_classes, depth=depth, wf=wf, padding=bool(padding),
                         batch_norm=bool(batch_norm), max_pool=bool(max_pool), num_lands=num_lands, do_res=bool(do_res),
                         block_depth=block_depth, lands_num_1x1=lands_num_1x1, lands_block_depth=lands_block_depth,
                         do_soft_max=bool(do_soft_max), pad_mode=pad_mode, up_mode=up_mode)

        # registration order: downsample_convs is assigned before down_path (reference unet.py:80-85)
        self.downsample_convs = None if max_pool else nn.ModuleList()
        self.down_path = nn.ModuleList()
        ch = in_channels
        for lvl in range(depth):
            width = 2 ** (wf + lvl)
            self.down_path.append(UNetConvBlock(ch, width, padding, batch_norm, pad_mode, do_res, block_depth))
            ch = width
            if not max_pool:
                # one per level, including the unused last one (kept for checkpoint compatibility)
                self.downsample_convs.append(nn.Conv2d(ch, ch, kernel_size=2, stride=2))
        self.up_path = nn.ModuleList()
        for lvl in reversed(range(depth - 1)):
            width = 2 ** (wf + lvl)
            self.up_path.append(UNetUpBlock(ch, width, up_mode, padding, batch_norm, pad_mode, do_res, block_depth))
            ch = width
        self.seg_conv = nn.Conv2d(ch, n_classes, kernel_size=1, bias=False)
        if do_soft_max:
            self.soft_max = nn.Softmax2d()           # stateless; present so attribute access matches the reference
        if num_lands > 0:
            self.lands_block = None
            lands_ch = ch
            if lands_block_depth > 0:             # reference unet.py:118-137: bias-only 3x3 convolutions, no non-linearity
                lands_ch = ch // 2
                convs = [nn.Conv2d(ch, lands_ch, kernel_size=3, padding=int(padding), padding_mode=pad_mode)]
                for _ in range(lands_block_depth - 1):
                    convs.append(nn.Conv2d(lands_ch, lands_ch, kernel_size=3, padding=int(padding), padding_mode=pad_mode))
                self.lands_block = nn.Sequential(*convs)
            mid = num_lands + n_classes if lands_num_1x1 > 1 else num_lands
            heads = [nn.Conv2d(lands_ch + n_classes, mid, kernel_size=1, bias=False)]
            feats = mid
            for _ in range(lands_num_1x1 - 1):       # reference unet.py:152-157: mid -> L, then L -> L ...
                heads.append(nn.Conv2d(feats, num_lands, kernel_size=1, bias=False))
                feats = num_lands
            self.lands_1x1 = nn.Sequential(*heads)

        self._plans = {}
        self._param_list = None
        self._pack_version = None
        self._bn_epoch = 0
        self._last_train_plan = None
        self._backward_runner = self._run_backward
        self.dp = None                                # set by parallel.DataParallel
        self.direct_grad = True                       # install gradient views as .grad without autograd copies
        # inference forwards replay one hipGraph per recorded plan (DFL_HIPGRAPH=0: launch the ops one by one)
        self.use_graphs = os.environ.get('DFL_HIPGRAPH', '1') != '0'
        # ... and so do training forwards / backwards (DFL_TRAIN_GRAPH=0: op by op)
        self.train_graphs = os.environ.get('DFL_TRAIN_GRAPH', '1') != '0'
        self._flatten_parameters()

    # ---------------------------------------------------------------------------------------------- plumbing
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten_parameters()
        self.invalidate_plans()
        return out

    def _flatten_parameters(self):
        """Keep every parameter as a view of ONE arena, in parameters() order, each slice padded to 4 floats -- the
        layout of the gradient arena the backward program fills (plan.grad_flat).  sgd.SGD then updates the whole
        network with one dfl_sgd_step per contiguous run instead of one launch per tensor."""
        ps = list(self.parameters())
        if not ps or any(p.dtype != torch.float32 for p in ps) or len({p.device for p in ps}) != 1:
            self._param_flat = None
            return
        tot = sum((p.numel() + 3) // 4 * 4 for p in ps)
        flat = torch.zeros(tot, dtype=torch.float32, device=ps[0].device)
        off = 0
        with torch.no_grad():
            for p in ps:
                n = p.numel()
                v = flat[off:off + n].view(p.shape)
                v.copy_(p.data)
                p.data = v
                off += (n + 3) // 4 * 4
        self._param_flat = flat
        self._param_ptrs = tuple(p.data_ptr() for p in ps)
        self.invalidate_plans() if hasattr(self, '_plans') else None
        _NETS[id(self)] = self
        for p in ps:
            p._dfl_net_id = id(self)   # lets sgd.SGD start the next step's weight re-layout right behind its update

    def __getstate__(self):
        """Recorded programs hold raw device addresses and ctypes structures: they are rebuilt on demand, not pickled."""
        d = self.__dict__.copy()
        d['_plans'], d['_param_list'], d['_pack_version'], d['_last_train_plan'] = {}, None, None, None
        d['_backward_runner'], d['dp'] = None, None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._backward_runner = self._run_backward
        self._flatten_parameters()

    def invalidate_plans(self):
        """Forget recorded programs (they hold raw parameter addresses)."""
        self._plans = {}
        self._param_list = None
        self._pack_version = None

    def mark_weights_dirty(self):
        """Tell the network that convolution weights were edited behind autograd's back.  The 3x3 / 2x2 / transposed
        convolution kernels read RE-LAID-OUT copies of the weights, rebuilt when a weight's autograd version counter moves
        (optimizer steps, load_state_dict, any in-place torch op).  Writes that do not bump the counter -- ``p.data.fill_()``,
        raw-pointer writes, foreign kernels -- need this call before the next forward; ``p.data = tensor`` rebinding is
        detected by itself (the parameters are moved back into the arena)."""
        self._pack_version = None

    def _state(self):
        if self._param_list is not None and self._param_flat is not None and \
                tuple(p.data_ptr() for p in self._param_list) != self._param_ptrs:
            self._flatten_parameters()       # a parameter was re-bound (p.data = ...): recorded addresses are stale
        if self._param_list is None:
            self._param_names = [k for k, _ in self.named_parameters()]
            self._param_list = [p for _, p in self.named_parameters()]
            self._param_dict = dict(zip(self._param_names, self._param_list))
            self._buffer_dict = dict(self.named_buffers())
            self._weight_params = [p for p in self._param_list if p.dim() == 4]
        return self._param_dict, self._buffer_dict

    def _get_plan(self, x, need_grad, input_grad=False):
        P, B = self._state()
        N, _, H, W = x.shape
        key = (N, H, W, self.training, need_grad, nat.lib().dfl_get_math_mode(), bool(input_grad))   # operand formats depend on the mode
        plans = self._plans.setdefault(key, [])
        for p in plans:
            if not p.busy:
                return p
        if len(plans) >= 2:                          # both in flight: recycle the older one
            p = plans.pop(0)
            p.busy = False
            p.generation += 1
            plans.append(p)
            return p
        for p_ in self._param_list:
            if p_.device != x.device or p_.dtype != torch.float32 or not p_.is_contiguous():
                raise RuntimeError('UNet parameters must be contiguous float32 tensors on %s' % x.device)
        try:
            plan = UNetPlan(self._cfg, P, B, N, H, W, self.training, need_grad, x.device, input_grad=input_grad)
        except PlanError as e:
            raise RuntimeError(str(e))
        plans.append(plan)
        return plan

    def prepack(self):
        """Enqueue the weight re-layout of the plan the last training forward used (no-op when nothing changed).  Called
        by sgd.SGD.step(): the GPU then packs while the host is still in loss.item() / zero_grad() / forward()'s
        bookkeeping instead of idling until the next forward's first launch (~0.14 ms per step)."""
        plan = self._last_train_plan() if self._last_train_plan is not None else None
        if plan is None or self._param_list is None or not any(plan is q for ps in self._plans.values() for q in ps):
            return
        if plan.math != nat.lib().dfl_get_math_mode():
            return
        self._ensure_packed(plan, torch.cuda.current_stream(plan.dev).cuda_stream)

    def plan_for_fused_update(self):
        """The training plan whose tiled weight re-layout sgd.SGD.step() may run inside its update kernel (dfl_sgd_pack_tiled), or
        None: the same conditions as prepack() + a tiled pack list without aliases."""
        plan = self._last_train_plan() if self._last_train_plan is not None else None
        if plan is None or self._param_list is None or not any(plan is q for ps in self._plans.values() for q in ps):
            return None
        if plan.math != nat.lib().dfl_get_math_mode() or getattr(plan, '_tiled_host', None) is None:
            return None
        return plan

    def after_fused_update(self, plan, stream):
        """The tiled layouts of `plan` were written with the update: run what else its pack program holds and mark it packed."""
        plan.run_pack_rest(stream)
        plan.fold_tail()
        self._pack_version = (id(plan), sum(p._version for p in self._weight_params))

    def _ensure_packed(self, plan, stream):
        ver = sum(p._version for p in self._weight_params)
        if not plan.training and not plan.need_grad:
            # inference plans also prepare the eval-mode BatchNorm scale / shift there: every parameter, every buffer (in-place
            # torch edits, load_state_dict) and the training forwards of this network (they update the running statistics
            # through raw pointers) count
            ver = (ver, sum(p._version for p in self._param_list), sum(b._version for b in self._buffer_dict.values()), self._bn_epoch)
        if self._pack_version != (id(plan), ver):
            plan.pack.run(stream)                    # weight re-layout for this plan's kernels
            plan.fold_tail()                         # (lands_num_1x1 > 2: product of the trailing 1x1 convolutions)
            self._pack_version = (id(plan), ver)

    def _run_forward(self, plan, x):
        stream = torch.cuda.current_stream().cuda_stream
        self._ensure_packed(plan, stream)
        cin = self._cfg['in_channels']
        if cin == 1:
            plan.x_in.copy_(x.reshape(-1))
        else:
            plan.x_in.copy_(x.permute(0, 2, 3, 1).reshape(-1))
        seg, heat = plan.new_outputs()
        plan.bind_outputs(seg, heat)
        self.run_program(plan, plan.fwd, stream)
        return seg, heat

    def graphs_on(self, plan):
        return self.use_graphs and (self.train_graphs or not plan.training)

    def run_program(self, plan, prog, stream, start=0, count=None):
        """Replay (part of) a recorded program.  With graphs on, every run of main-stream kernels with fixed argument blocks
        is ONE hipGraph launch (captured on first use): the inference forward up to the heads (BASELINE configs[4]; the
        per-image loops of util.test_dataset / seg_dataset_ensemble replay ~100 small launches per net) and -- round 3 --
        the training forward and backward (~250 launches per step: on a slow host the GPU waited for them).
        The head ops read / write caller-owned tensors whose addresses change per call, event waits order the side-stream
        weight packing: both stay plain launches in program order (Program.graph_chunks)."""
        if self.graphs_on(plan):
            chunks = prog.graph_chunks(stream, start, count)
            if prog is plan.fwd and plan.graph is None:
                gs = [g for _, _, g in chunks if g is not None]
                plan.graph = max(gs, key=lambda g: g.nodes) if gs else None
            prog.run_graphed(stream, start, count)
        else:
            prog.run(stream, start, count)

    def _run_backward(self, plan, stream):
        self.run_program(plan, plan.bwd, stream)
        plan.unfold_tail_grads()

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('deepfluorolabeling-ipcai2020_amd.UNet runs on the GPU only (HIP kernels, no CPU '
                               'fallback); move the network and its input to the device')
        if x.dim() != 4 or x.shape[1] != self._cfg['in_channels']:
            raise RuntimeError('expected input of shape [B, %d, H, W]' % self._cfg['in_channels'])
        x_src = x if (x.requires_grad and torch.is_grad_enabled()) else None     # d(loss)/d(input) wanted (nn.Module semantics)
        x = x.detach().to(torch.float32).contiguous()
        self._state()
        need_grad = torch.is_grad_enabled() and (x_src is not None or any(p.requires_grad for p in self._param_list))
        plan = self._get_plan(x, need_grad, input_grad=x_src is not None)
        if plan.training:
            self._bn_epoch += 1                      # running statistics move: inference plans re-derive their scale / shift
        # first GPU work of a training step: enqueue it before the autograd bookkeeping below (the GPU sits idle between
        # the previous step's loss.item() and this launch)
        self._ensure_packed(plan, torch.cuda.current_stream().cuda_stream)
        if not need_grad:
            seg, heat = self._run_forward(plan, x)
            return (seg, heat) if heat is not None else seg
        plan.busy = True
        plan.generation += 1
        self._last_train_plan = weakref.ref(plan)
        return _UNetFn.apply(self, plan, x if x_src is None else x_src, *self._param_list)
