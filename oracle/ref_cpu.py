"""CPU oracle for the U-Net hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (PyTorch CPU ops, fp32 or fp64) of the
algorithm that rg2/DeepFluoroLabeling-IPCAI2020 runs on its hot path.  It exists so that
the HIP path in ``deepfluorolabeling-ipcai2020_amd/`` can be checked against something
that does not depend on the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product never does.

Parity pinning: the reference ships no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, generated in the build
container by ``tools/gen_golden.py`` (which imports ``/root/reference/train_test_code``)
and committed under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every
function here against those fixtures.

Every function cites the reference lines it restates (paths relative to the reference
repo root).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1.0e-5       # nn.BatchNorm2d default (reference: train_test_code/unet.py:215,222)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# util.center_crop  (train_test_code/util.py:92-114)
# --------------------------------------------------------------------------------------
def center_crop(img, dst_shape):
    """Centre slice of the last two dims; returns ``img`` itself when sizes match.

    start = int((src - dst) / 2) (truncation toward zero), as util.py:100-104.
    """
    sr, sc = img.shape[-2], img.shape[-1]
    dr, dc = dst_shape[-2], dst_shape[-1]
    if sr == dr and sc == dc:
        return img
    r0 = int((sr - dr) / 2)
    c0 = int((sc - dc) / 2)
    assert img.dim() in (2, 3, 4)
    return img[..., r0:r0 + dr, c0:c0 + dc]


# --------------------------------------------------------------------------------------
# Model (train_test_code/unet.py:40-260)
# --------------------------------------------------------------------------------------
class _Block(nn.Module):
    """UNetConvBlock (unet.py:196-233): [conv3x3 -> ReLU -> BN] x block_depth, + 1x1 residual."""

    def __init__(self, cin, cout, padding, batch_norm, pad_mode, do_res, block_depth):
        super().__init__()
        assert block_depth > 0
        self.do_res = do_res
        self.batch_norm = batch_norm
        self.block_depth = block_depth
        # creation order matters for RNG parity: residual conv first (unet.py:206-207)
        if do_res:
            self.res_conv1x1 = nn.Conv2d(cin, cout, kernel_size=1, padding=0)
        layers = []
        c = cin
        for _ in range(block_depth):
            layers.append(nn.Conv2d(c, cout, kernel_size=3, padding=int(padding),
                                    padding_mode=pad_mode))
            layers.append(nn.ReLU())
            if batch_norm:
                layers.append(nn.BatchNorm2d(cout))
            c = cout
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        out = x
        for m in self.block:
            out = m(out)
        if self.do_res:
            out = out + self.res_conv1x1(x)      # reference does it in place (unet.py:231)
        return out


class _Up(nn.Module):
    """UNetUpBlock (unet.py:236-260)."""

    def __init__(self, cin, cout, up_mode, padding, batch_norm, pad_mode, do_res, block_depth):
        super().__init__()
        if up_mode == 'upconv':
            self.up = nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2)
        else:
            self.up = nn.Sequential(nn.Upsample(mode='bilinear', scale_factor=2),
                                    nn.Conv2d(cin, cout, kernel_size=1))
        self.conv_block = _Block(cin, cout, padding, batch_norm, pad_mode, do_res, block_depth)

    def forward(self, x, bridge):
        up = self.up(x)
        th, tw = up.shape[2], up.shape[3]
        dy = (bridge.shape[2] - th) // 2        # unet.py:248-252 (floor division here)
        dx = (bridge.shape[3] - tw) // 2
        crop = bridge[:, :, dy:dy + th, dx:dx + tw]
        return self.conv_block(torch.cat([up, crop], 1))


class OracleUNet(nn.Module):
    """Same constructor, module tree, state_dict keys/order and RNG consumption order as the
    reference ``UNet`` (unet.py:40-159); forward as unet.py:161-193."""

    def __init__(self, in_channels=1, n_classes=2, depth=5, wf=6, padding=False, pad_mode='zeros',
                 batch_norm=False, up_mode='upconv', max_pool=True, num_lands=0, do_res=True,
                 block_depth=2, lands_block_depth=0, lands_num_1x1=2, do_soft_max=True):
        super().__init__()
        assert up_mode in ('upconv', 'upsample')
        self.padding = padding
        self.depth = depth
        self.do_max_pool = max_pool
        self.num_lands = num_lands
        self.do_soft_max = do_soft_max
        self.pool_override = None      # tests: callable(level, x) -> pooled x replacing F.max_pool2d (forced arg-max choices)
        # attribute assigned before down_path => registered first
        self.downsample_convs = None
        if not max_pool:
            self.downsample_convs = nn.ModuleList()
        self.down_path = nn.ModuleList()
        prev = in_channels
        for i in range(depth):
            c = 2 ** (wf + i)
            self.down_path.append(_Block(prev, c, padding, batch_norm, pad_mode, do_res, block_depth))
            prev = c
            if not max_pool:
                # note: the last one is created but never used in forward
                self.downsample_convs.append(nn.Conv2d(prev, prev, kernel_size=2, stride=2))
        self.up_path = nn.ModuleList()
        for i in reversed(range(depth - 1)):
            c = 2 ** (wf + i)
            self.up_path.append(_Up(prev, c, up_mode, padding, batch_norm, pad_mode, do_res, block_depth))
            prev = c
        self.seg_conv = nn.Conv2d(prev, n_classes, kernel_size=1, bias=False)
        if num_lands > 0:
            self.lands_block = None
            chan = prev
            if lands_block_depth > 0:
                chan = prev // 2
                lb = [nn.Conv2d(prev, chan, kernel_size=3, padding=int(padding), padding_mode=pad_mode)]
                for _ in range(lands_block_depth - 1):
                    lb.append(nn.Conv2d(chan, chan, kernel_size=3, padding=int(padding),
                                        padding_mode=pad_mode))
                self.lands_block = nn.Sequential(*lb)
            assert lands_num_1x1 > 0
            out_feat = num_lands + n_classes if lands_num_1x1 > 1 else num_lands
            l1 = [nn.Conv2d(chan + n_classes, out_feat, kernel_size=1, bias=False)]
            for _ in range(lands_num_1x1 - 1):
                l1.append(nn.Conv2d(out_feat, num_lands, kernel_size=1, bias=False))
                out_feat = num_lands
            self.lands_1x1 = nn.Sequential(*l1)

    def forward(self, x, taps=None):
        """``taps``: optional dict that receives named intermediate activations."""
        bridges = []
        for i, down in enumerate(self.down_path):
            x = down(x)
            if taps is not None:
                taps['down%d' % i] = x
            if i != len(self.down_path) - 1:
                bridges.append(x)
                if self.do_max_pool:
                    x = F.max_pool2d(x, 2) if self.pool_override is None else self.pool_override(i, x)
                else:
                    x = self.downsample_convs[i](x)
        for i, up in enumerate(self.up_path):
            x = up(x, bridges[-i - 1])
            if taps is not None:
                taps['up%d' % i] = x
        logits = self.seg_conv(x)
        if taps is not None:
            taps['logits'] = logits
        seg = torch.softmax(logits, dim=-3) if self.do_soft_max else logits   # nn.Softmax2d
        if self.num_lands > 0:
            if self.lands_block is not None:
                x = self.lands_block(x)
            x = torch.cat((x, center_crop(logits, x.shape)), dim=1)
            return seg, self.lands_1x1(x)
        return seg


# --------------------------------------------------------------------------------------
# Losses (train_test_code/dice.py:14-86, train_test_code/ncc.py:12-38)
# --------------------------------------------------------------------------------------
def dice_loss_2d(inp, target, skip_bg=True):
    """DiceLoss2D.forward (dice.py:20-55).  Note the numerator is -2*sum(t*s)+eps (dice.py:29,40)."""
    eps = 1.0e-4
    if skip_bg:
        inp, target = inp[:, 1:], target[:, 1:]
    ncls = inp.shape[1]
    num = -2 * (target * inp).sum(dim=(2, 3)) + eps
    den = (target * target).sum(dim=(2, 3)) + (inp * inp).sum(dim=(2, 3)) + eps
    return ((num / den).sum(dim=1) / ncls).mean()


def ncc_2d(X, Y):
    """ncc.ncc_2d (ncc.py:12-38): sd uses N-1, the product uses N => ncc(X,X) = (N-1)/N."""
    N = X.shape[-1] * X.shape[-2]
    assert N > 1
    xz = X - X.mean(dim=(-2, -1), keepdim=True)
    yz = Y - Y.mean(dim=(-2, -1), keepdim=True)
    xs = torch.sqrt((xz * xz).sum(dim=(-2, -1)) / (N - 1))
    ys = torch.sqrt((yz * yz).sum(dim=(-2, -1)) / (N - 1))
    return (xz * yz).sum(dim=(-2, -1)) / (N * (xs * ys) + 1.0e-8)


def dice_and_heatmap_loss_2d(inp, target, skip_bg=True, heatmap_wgt=0.5):
    """DiceAndHeatMapLoss2D.forward (dice.py:67-86)."""
    assert 1.0e-8 < heatmap_wgt < 1 + 1.0e-8
    ncc_l = (ncc_2d(inp[1], target[1]) + 1) * -0.5
    return (1 - heatmap_wgt) * dice_loss_2d(inp[0], target[0], skip_bg) + heatmap_wgt * ncc_l.mean()


# --------------------------------------------------------------------------------------
# Ensemble inference arithmetic (train_test_code/util.py:318-373)
# --------------------------------------------------------------------------------------
def ensemble_reduce(seg_list, heat_list, orig_shape):
    """Per image: mean of cropped softmax maps -> first-max argmax over channels (uint8);
    heat maps min-max normalised per net over the whole cropped tensor, then averaged."""
    n = len(seg_list)
    avg = None
    for s in seg_list:
        s = center_crop(s, orig_shape)
        avg = s.clone() if avg is None else avg + s
    avg = avg / n
    labels = torch.max(avg, dim=1)[1].to(torch.uint8)
    heats = None
    if heat_list:
        for h in heat_list:
            h = center_crop(h, orig_shape)
            lo, hi = h.min().item(), h.max().item()
            h = (h - lo) / (hi - lo)
            heats = h.clone() if heats is None else heats + h
        heats = heats / n
    return labels, heats, avg


# --------------------------------------------------------------------------------------
# Validation loops (train_test_code/util.py:116-165 and :167-241)
# --------------------------------------------------------------------------------------
def validation_loss(net, items, num_lands):
    """util.test_dataset: eval mode, batch 1, per-image Dice (+ NCC with the FIXED weight 0.5, util.py:126-129 -- SURVEY D11)
    on the centre-cropped outputs; returns (mean, std) with torch.std's unbiased estimator.  items: (proj [1,Hp,Wp],
    mask [C,H,W], lands, heat [L,1,H,W]) tuples."""
    losses = torch.zeros(len(items))
    with torch.no_grad():
        net.eval()
        for i, (p, m, _, h) in enumerate(items):
            out = net(p.unsqueeze(0))
            m = m.unsqueeze(0)
            if num_lands > 0:
                h = h.view(1, h.shape[0], h.shape[-2], h.shape[-1])
                losses[i] = dice_and_heatmap_loss_2d((center_crop(out[0], m.shape), center_crop(out[1], h.shape)), (m, h),
                                                     skip_bg=False, heatmap_wgt=0.5).item()
            else:
                seg = out[0] if isinstance(out, tuple) else out
                losses[i] = dice_loss_2d(center_crop(seg, m.shape), m, skip_bg=False).item()
    return torch.mean(losses), torch.std(losses)


def validation_loss_ensemble(nets, items, num_lands, dice_only=False):
    """util.test_dataset_ensemble: the loss of the AVERAGED cropped soft-max (and averaged RAW heat maps -- no min-max
    normalisation here, unlike seg_dataset_ensemble) per image; Dice only when dice_only or there are no landmarks."""
    losses = torch.zeros(len(items))
    use_heat = (not dice_only) and num_lands > 0
    with torch.no_grad():
        for n in nets:
            n.eval()
        for i, (p, m, _, h) in enumerate(items):
            m = m.unsqueeze(0)
            outs = [n(p.unsqueeze(0)) for n in nets]
            segs = [o[0] if isinstance(o, tuple) else o for o in outs]
            avg_seg = sum(center_crop(s_, m.shape) for s_ in segs) / len(nets)
            if use_heat:
                h = h.view(1, h.shape[0], h.shape[-2], h.shape[-1])
                avg_heat = sum(center_crop(o[1], h.shape) for o in outs) / len(nets)
                losses[i] = dice_and_heatmap_loss_2d((avg_seg, avg_heat), (m, h), skip_bg=False, heatmap_wgt=0.5).item()
            else:
                losses[i] = dice_loss_2d(avg_seg, m, skip_bg=False).item()
    return torch.mean(losses), torch.std(losses)


# --------------------------------------------------------------------------------------
# SGDR schedule (train_test_code/warm_restarts_lr.py:14-63) as a pure function trace
# --------------------------------------------------------------------------------------
def warm_restart_lr_trace(base_lr, period, growth, lr_min, n_epochs, intra_steps):
    """LR after every intra-epoch step (ratio = (k+1)/intra_steps) and after each epoch step."""
    out = []
    last_epoch, last_restart, next_restart, cur_period = 0, 0, period, period

    def lr(ratio):
        return lr_min + (base_lr - lr_min) / 2 * (1 + math.cos(
            math.pi * (last_epoch - last_restart + ratio) / cur_period))

    for _ in range(n_epochs):
        for k in range(intra_steps):
            out.append(lr((k + 1) / intra_steps))
        last_epoch += 1
        ratio0_lr = lr(0.0)               # step(): ratio reset, LR set before restart bookkeeping
        if last_epoch >= next_restart:
            last_restart = next_restart
            cur_period *= growth
            next_restart += cur_period
        out.append(ratio0_lr)
    return out


# --------------------------------------------------------------------------------------
# Deterministic part of the loader (train_test_code/dataset.py:26-40, 287-328, 405-452)
# --------------------------------------------------------------------------------------
def calc_pad_amount(padded_dim, cur_dim):
    """dataset.py:26-40 -- odd differences round up."""
    assert padded_dim > cur_dim
    pad = (padded_dim - cur_dim) / 2
    return int(pad) + 1 if pad != int(pad) else int(pad)


def preprocess_proj(p, extra_pad):
    """Reflect-pad [1,H,W] by extra_pad, then (p-mean)/std with the unbiased std (dataset.py:287-293)."""
    if extra_pad > 0:
        p = F.pad(p.unsqueeze(0), (extra_pad,) * 4, mode='reflect').squeeze(0)
    return (p - p.mean()) / p.std()


def gaussian_heatmaps(lands, H, W, sigma=2.5):
    """[2,L] landmarks (row 0 = column/x, row 1 = row/y) -> [L,1,H,W]; inf landmarks give zero maps
    (dataset.py:302-325)."""
    L = lands.shape[-1]
    h = torch.zeros(L, 1, H, W)
    Y, X = torch.meshgrid(torch.arange(0, H), torch.arange(0, W), indexing='ij')
    Y, X = Y.float(), X.float()
    for l in range(L):
        mx, my = lands[0, l], lands[1, l]
        if not math.isinf(mx) and not math.isinf(my):
            h[l, 0] = torch.exp(((X - mx).pow(2) + (Y - my).pow(2)) / (sigma * sigma * -2)) \
                / (2 * math.pi * sigma * sigma)
    return h


def one_hot_masks(segs, num_classes):
    """[N,H,W] integer labels -> float [N,C,H,W] (dataset.py:448-452)."""
    return torch.stack([(segs == c) for c in range(num_classes)], dim=1).float()


def mark_oob_landmarks(lands, H, W):
    """Landmarks outside [0,W-1]x[0,H-1] -> inf (dataset.py:421-429). lands: [N,2,L]."""
    lands = lands.clone()
    x, y = lands[:, 0], lands[:, 1]
    oob = (x < 0) | (x > W - 1) | (y < 0) | (y > H - 1)
    x[oob] = math.inf
    y[oob] = math.inf
    return lands


# --------------------------------------------------------------------------------------
# Landmark extraction from heat maps (train_test_code/est_lands_csv.py:85-124)
# --------------------------------------------------------------------------------------
def gaussian_template(rows, cols, sigma):
    """util.get_gaussian_2d_heatmap (util.py:38-51) with the peak at the centre."""
    Y, X = torch.meshgrid(torch.arange(0, rows), torch.arange(0, cols), indexing='ij')
    return torch.exp(((X.float() - cols // 2).pow(2) + (Y.float() - rows // 2).pow(2)) / (sigma * sigma * -2)) \
        / (2 * math.pi * sigma * sigma)


def est_landmarks(heats, segs=None, label_for_land=None, sigma=2.5, min_ncc=0.9, return_ncc=False):
    """heats [B,L,H,W]; segs [B,H,W] labels or None; label_for_land[l] = label whose pixels may hold landmark l (None
    / negative: anywhere).  Returns int [B,L,2] (row, col), (-1,-1) when nothing is found (est_lands_csv.py:96-124:
    'rule_3': masked arg-max, then the 25x25 window of the reflect-padded map must correlate >= 0.9 with the template)."""
    B, L, H, W = heats.shape
    tmpl = gaussian_template(25, 25, sigma)
    out = torch.full((B, L, 2), -1, dtype=torch.int64)
    nccs = torch.zeros(B, L)
    for i in range(B):
        for l in range(L):
            cur = heats[i, l]
            pad = F.pad(cur[None, None], (12, 12, 12, 12), mode='reflect')[0, 0]
            lab = None if (segs is None or label_for_land is None) else label_for_land[l]
            if lab is None or lab < 0:
                idx = int(torch.argmax(cur))
            else:
                tmp = cur.clone()
                tmp[segs[i] != lab] = -math.inf
                idx = int(torch.argmax(tmp))
                if tmp.view(-1)[idx] == -math.inf:
                    continue
            r, c = idx // W, idx % W
            v = float(ncc_2d(tmpl, pad[r:r + 25, c:c + 25]))
            nccs[i, l] = v
            if not v < min_ncc:
                out[i, l, 0], out[i, l, 1] = r, c
    return (out, nccs) if return_ncc else out


# --------------------------------------------------------------------------------------
# Hard Dice (train_test_code/compute_actual_dice_on_test.py:63-93)
# --------------------------------------------------------------------------------------
def hard_dice(pred_labels, gt_labels, num_classes):
    """2|A&B| / (|A|+|B|) for labels 1..C-1 (background excluded); 1.0 when both are empty."""
    out = []
    for c in range(1, num_classes):
        a, b = pred_labels == c, gt_labels == c
        den = int(a.sum()) + int(b.sum())
        out.append(2.0 * int((a & b).sum()) / den if den > 0 else 1.0)
    return out


# --------------------------------------------------------------------------------------
# One training step exactly as train.py:405-430, used for the CPU baseline and trajectory goldens
# --------------------------------------------------------------------------------------
def train_step(net, optimizer, projs, masks, heats=None, heat_coeff=0.5):
    optimizer.zero_grad()
    out = net(projs)
    if heats is not None:
        seg = center_crop(out[0], masks.shape)
        hm = center_crop(out[1], heats.shape)
        loss = dice_and_heatmap_loss_2d((seg, hm), (masks, heats), skip_bg=False, heatmap_wgt=heat_coeff)
    else:
        seg = center_crop(out, masks.shape)
        loss = dice_loss_2d(seg, masks, skip_bg=False)
    loss.backward()
    optimizer.step()
    return loss.item()
