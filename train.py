#!/usr/bin/env python3
"""Training entry point for the MI355X path, command-line compatible with the reference's train_test_code/train.py.

What is kept from the reference (it is the contract a user of the reference relies on): every flag and default of its
parser (train.py:29-100), the rule that a checkpoint file found at --checkpoint-net overrides the command line
(:189-270), the two loss logs with one '{:.6f}' line per value (util.py:72-74), the checkpoint dictionary -- 37 keys in
the order of :463-513, so checkpoints move between the two implementations in both directions --, the best-validation
and pre-warm-restart copies and the three stop rules (:556-577).

What is this build's own: the structure below (Settings / Snapshots / Trainer instead of one long script), the
GPU-resident loader (dfl_amd.dataset: batches are built on the device by dfl_prep_batch, no DataLoader), the one-launch
optimizer (dfl_amd.SGD) and DATA-PARALLEL training: started under torchrun (one process per GPU, RCCL) every rank takes
a contiguous slice of each global minibatch, gradients are averaged by dfl_amd.DataParallel while backward is still
running, rank 0 alone writes logs and checkpoints.  --batch-size is the batch PER GPU (BatchNorm statistics stay per
replica, i.e. what the reference computes at that batch size); the global batch is --batch-size x world size.

    python train.py data.h5 --train-pats 1,2,3 --valid-pats 4 --num-classes 7 --unet-img-dim 192 --batch-size 16 \
        --unet-num-lvls 6 --unet-init-feats-exp 5 --unet-batch-norm --unet-padding --unet-no-max-pool --use-lands \
        --nesterov --wgt-decay 1e-4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py data.h5 ...

Extensions (not in the reference): --math, --seed, --dist-backend.  Refused loudly: --no-gpu (there is no CPU path) and
--data-aug (PIL-based random augmentation is outside the HIP path).  The data file is the
reference's HDF5 layout (read by the dependency-free dfl_amd.h5lite) or an .npz with the same dataset names.
"""
import argparse
import os
import random
import shutil
import sys
import time

import torch
from torch import optim

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dfl_amd  # noqa: E402
from dfl_amd import dataset, util  # noqa: E402
from dfl_amd._native import DflError  # noqa: E402

# The checkpoint dictionary of the reference (train.py:465-510), in its order.  The first group is run state, the rest
# are the settings a checkpoint pins (Settings below), the last two the train/validation split.
CHECKPOINT_KEYS = ['epoch', 'model-state-dict', 'optim-type', 'optimizer-state-dict', 'scheduler-state-dict', 'loss',
                   'best-valid-loss', 'save-best-valid', 'num-classes', 'depth', 'init-feats-exp', 'batch-norm', 'padding',
                   'no-max-pool', 'pad-img-size', 'batch-size', 'data-aug', 'opt-nesterov', 'opt-momentum',
                   'opt-wgt-decay', 'num-lands', 'heat-coeff', 'use-dice-valid', 'unet-use-res', 'unet-block-depth',
                   'lrs-meth', 'lrs-num-epochs', 'lrs-growth-factor', 'lrs-max-num-restarts',
                   'lrs-save-restart-net-prefix', 'lrs-save-after-n-restarts', 'lrs-num-restarts', 'lrs-patience',
                   'lrs-cooldown', 'checkpoint-freq', 'train-idx', 'valid-idx']
RUN_STATE_KEYS = ('epoch', 'model-state-dict', 'optimizer-state-dict', 'scheduler-state-dict', 'loss', 'best-valid-loss',
                  'train-idx', 'valid-idx')
MATH_MODES = {'fp32': 0, 'bf16x3': 1, 'bf16x6': 2, 'bf16': 3, 'bf16s': 4}


def build_parser():
    p = argparse.ArgumentParser(description='Train the segmentation / landmark U-Net on one MI355X, or data-parallel on '
                                            'several (launch with torch.distributed.run).',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    a = p.add_argument
    a('input_data_file_path', type=str, help='pre-processed data file (HDF5 layout of the reference, or .npz)')
    a('--train-pats', type=str, help='specimen numbers to train on, comma separated')
    a('--valid-pats', type=str, help='specimen numbers to validate on, comma separated')
    a('--num-classes', type=int, help='segmentation labels including background')
    a('--batch-size', type=int, default=1, help='images per minibatch (per GPU when data parallel)')
    a('--unet-img-dim', type=int, default=364, help='side the projections are reflect-padded to before the network')
    a('--checkpoint-net', type=str, default='zz_checkpoint.pt', help='checkpoint file; resumed from when it exists')
    a('--best-net', type=str, default='zz_best_valid.pt', help='copy of the checkpoint with the lowest validation loss')
    a('--checkpoint-freq', type=int, default=1, help='epochs between checkpoints')
    a('--no-save-best-valid', action='store_true', help='keep no best-validation copy')
    a('--optim', type=str, default='sgd', help='sgd | adam | rmsprop')
    a('--lr-sched', type=str, default='cos', help="cos (cosine annealing with warm restarts) | plateau | none")
    a('--init-lr', type=float, default=1.0e-2, help='initial learning rate')
    a('--lr-patience', type=int, default=20, help='plateau schedule: epochs without improvement before a cut')
    a('--lr-cooldown', type=int, default=20, help='plateau schedule: epochs to wait after a cut')
    a('--nesterov', action='store_true', help='Nesterov momentum (SGD)')
    a('--momentum', type=float, default=0.9, help='momentum (SGD, RMSprop)')
    a('--wgt-decay', type=float, default=0, help='weight decay')
    a('--cos-anneal-epochs', type=int, default=10, help='length of the first cosine period in epochs')
    a('--cos-growth', type=int, default=2, help='factor by which each cosine period is longer than the previous one')
    a('--save-restart-net', type=str, help='keep the network from just before every warm restart as <PREFIX>_XX.pt')
    a('--save-after-n-restarts', type=int, default=0, help='... but only from this restart on')
    a('--max-num-restarts', type=int, default=-1, help='stop after this many warm restarts (> 0 replaces --max-num-epochs)')
    a('--max-num-epochs', type=int, default=200, help='stop after this many epochs')
    a('--train-loss-txt', type=str, default='train_iter_loss.txt', help='per-iteration training loss log')
    a('--valid-loss-txt', type=str, default='valid_loss.txt', help='per-epoch validation loss log')
    a('--no-gpu', action='store_true', help='(refused: this implementation has no CPU path)')
    a('--max-hours', type=float, default=-1.0, help='stop before an epoch that would run past this many hours (> 0)')
    a('--unet-num-lvls', type=int, default=5, help='U-Net depth')
    a('--unet-init-feats-exp', type=int, default=4, help='log2 of the channel count of the first level')
    a('--unet-batch-norm', action='store_true', help='BatchNorm after every 3x3 convolution + ReLU')
    a('--unet-padding', action='store_true', help='zero-pad the 3x3 convolutions (output size = input size)')
    a('--unet-no-max-pool', action='store_true', help='down-sample with learned 2x2 stride-2 convolutions')
    a('--unet-block-depth', type=int, default=2, help='3x3 convolutions per block')
    a('--data-aug', action='store_true', help='(refused: random augmentation is outside the HIP path)')
    a('--use-lands', action='store_true', help='also learn the landmark heat maps (count read from the data file)')
    a('--heat-coeff', type=float, default=0.5, help='weight of the heat-map loss; Dice gets one minus this')
    a('--dice-valid', action='store_true', help='validate on Dice alone even when training with the heat-map loss')
    a('--unet-no-res', action='store_true', help='no 1x1 residual branch in the blocks')
    a('--train-valid-split', type=float, default=-1.0,
      help='fraction in [0,1] of the training specimens\' images to train on, the rest validates (replaces --valid-pats)')
    return p


EXTENSION_FLAGS = {'math': None, 'seed': None, 'dist_backend': None, 'sync_loss': False, 'grad_compress': None}


def build_full_parser():
    """The reference's flags plus this build's extensions."""
    p = build_parser()
    p.add_argument('--math', type=str, default=None, choices=sorted(MATH_MODES),
                   help='product arithmetic of the convolution GEMMs (default: the library default, DFL_MATH or fp32)')
    p.add_argument('--seed', type=int, default=None, help='seed of initialisation and shuffling (default: unseeded)')
    p.add_argument('--dist-backend', type=str, default=None, help="torch.distributed backend when launched with several "
                                                                   "ranks (default nccl = RCCL)")
    p.add_argument('--grad-compress', type=str, default=None, choices=['bf16'],
                   help='data parallel: all-reduce the gradient buckets as bf16 (half the xGMI bytes; default: fp32)')
    p.add_argument('--sync-loss', action='store_true', help='read every minibatch loss right after its optimizer step, as the '
                                                             'reference does (default: one step late, the GPU never waits for the host)')
    return p


# ----------------------------------------------------------------------------------------------------- settings
class Settings(dict):
    """The part of a checkpoint that pins how a run is configured (every checkpoint key that is not run state).  Built
    from the command line; when a checkpoint exists its values win (train.py:196-270 of the reference)."""

    LABELS = (('num-classes', 'classes'), ('optim-type', 'optimizer'), ('depth', 'U-Net levels'),
              ('init-feats-exp', 'first-level channels (log2)'), ('batch-norm', 'BatchNorm'), ('padding', 'padded convolutions'),
              ('no-max-pool', 'strided-convolution down-sampling'), ('pad-img-size', 'padded image side'),
              ('batch-size', 'batch per GPU'), ('data-aug', 'augmentation'), ('num-lands', 'landmarks'),
              ('use-dice-valid', 'Dice-only validation'), ('unet-use-res', 'residual blocks'),
              ('unet-block-depth', 'convolutions per block'), ('opt-nesterov', 'Nesterov'), ('opt-momentum', 'momentum'),
              ('opt-wgt-decay', 'weight decay'), ('lrs-meth', 'LR schedule'), ('lrs-num-epochs', 'first cosine period'),
              ('lrs-growth-factor', 'cosine period growth'), ('lrs-max-num-restarts', 'restart limit'),
              ('lrs-save-restart-net-prefix', 'pre-restart copies'), ('lrs-save-after-n-restarts', '... from restart'),
              ('lrs-num-restarts', 'restarts so far'), ('lrs-patience', 'plateau patience'),
              ('lrs-cooldown', 'plateau cooldown'), ('checkpoint-freq', 'checkpoint every (epochs)'))

    @classmethod
    def from_args(cls, a):
        return cls({'save-best-valid': not a.no_save_best_valid, 'num-classes': a.num_classes, 'optim-type': a.optim,
                    'depth': a.unet_num_lvls, 'init-feats-exp': a.unet_init_feats_exp, 'batch-norm': a.unet_batch_norm,
                    'padding': a.unet_padding, 'no-max-pool': a.unet_no_max_pool, 'pad-img-size': a.unet_img_dim,
                    'batch-size': a.batch_size, 'data-aug': a.data_aug, 'num-lands': 0, 'heat-coeff': a.heat_coeff,
                    'use-dice-valid': a.dice_valid, 'unet-use-res': not a.unet_no_res,
                    'unet-block-depth': a.unet_block_depth, 'opt-nesterov': a.nesterov, 'opt-momentum': a.momentum,
                    'opt-wgt-decay': a.wgt_decay, 'lrs-meth': a.lr_sched.lower(), 'lrs-num-epochs': a.cos_anneal_epochs,
                    'lrs-growth-factor': a.cos_growth, 'lrs-max-num-restarts': a.max_num_restarts,
                    'lrs-save-restart-net-prefix': a.save_restart_net,
                    'lrs-save-after-n-restarts': a.save_after_n_restarts, 'lrs-num-restarts': 0,
                    'lrs-patience': a.lr_patience, 'lrs-cooldown': a.lr_cooldown, 'checkpoint-freq': a.checkpoint_freq})

    def adopt(self, checkpoint):
        for k in self:
            self[k] = checkpoint[k]

    def describe(self, out=print):
        for key, label in self.LABELS:
            out('  {:<36} {}'.format(label + ':', self[key]))


# ----------------------------------------------------------------------------------------------------- snapshots
class Snapshots:
    """Which files an epoch leaves behind.  A network is serialised at most once per epoch; further destinations of the
    same epoch (best-validation copy, pre-restart copy, final checkpoint) are file copies of it."""

    def __init__(self, writer, enabled=True):
        self._write, self.enabled = writer, enabled
        self._this_epoch = None

    def new_epoch(self):
        self._this_epoch = None

    def put(self, path):
        if not self.enabled:
            return
        if self._this_epoch is None:
            tmp = path + '.tmp'
            self._write(tmp)
            shutil.move(tmp, path)                  # never leave a half-written checkpoint under the real name
            self._this_epoch = path
        elif self._this_epoch != path:
            shutil.copy(self._this_epoch, path)


# ----------------------------------------------------------------------------------------------------- trainer
class Trainer:
    def __init__(self, args):
        self.args = args
        self.rank, self.world, self.local = dfl_amd.parallel.init_process_group_from_env(args.dist_backend)
        self.main = self.rank == 0
        self.say = print if self.main else (lambda *a, **k: None)
        if args.no_gpu:
            raise DflError('--no-gpu: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')
        self.dev = dfl_amd.get_device()
        if self.dev.type != 'cuda':
            raise DflError('no GPU visible: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')
        if self.world > 1:
            self.dev = torch.device('cuda', self.local % torch.cuda.device_count())
            torch.cuda.set_device(self.dev)
        from dfl_amd import _native as nat
        if args.math is not None:
            nat.check(nat.lib().dfl_set_math_mode(MATH_MODES[args.math]), 'dfl_set_math_mode')
        mode = nat.lib().dfl_get_math_mode()
        names = {v: k for k, v in MATH_MODES.items()}
        # (the library default is fp32 products, the 1e-4 parity mode; BASELINE configs[1] is --math bf16s, 2.4x the speed)
        self.say('arithmetic: {}{}'.format(names.get(mode, mode), '' if args.math is not None or mode != 0 else
                                            ' (library default; --math bf16s = bf16 tensors in HBM, see DESIGN.md section 4b)'))
        self._seed(args.seed)

        self.cfg = Settings.from_args(args)
        if args.use_lands:
            self.cfg['num-lands'] = dataset.get_num_lands_from_dataset(args.input_data_file_path)
            self.say('num. lands read from file: {}'.format(self.cfg['num-lands']))
            if self.cfg['num-lands'] <= 0:
                raise ValueError('--use-lands: the data file names no landmarks')
        self.train_idx = self.valid_idx = None
        self.best_valid_loss = None
        self.epoch = 0
        self.last_loss = None
        resume = None
        self.resumed = os.path.exists(args.checkpoint_net)
        if self.resumed:
            self.say('loading state from checkpoint...')
            resume = torch.load(args.checkpoint_net, map_location='cpu', weights_only=False)
            self.cfg.adopt(resume)
            self.say('settings taken from the checkpoint (they override the command line):')
            self.cfg.describe(self.say)
            if args.train_valid_split >= 0:
                self.train_idx, self.valid_idx = resume['train-idx'], resume['valid-idx']
                if self.train_idx is None or self.valid_idx is None:
                    raise ValueError('--train-valid-split: the checkpoint holds no train/validation split')
        c = self.cfg
        if c['data-aug']:
            raise NotImplementedError('--data-aug: random data augmentation is outside the HIP path (DESIGN.md section 7)')
        self._load_data()
        self._build_model(resume)
        self._build_optimizer(resume)
        if resume is not None:
            self.best_valid_loss = resume['best-valid-loss']
            self.epoch = resume['epoch']
        del resume
        self.snapshots = Snapshots(self._write_checkpoint, enabled=self.main)
        fresh = not self.resumed
        self.train_log = util.RunningFloatWriter(args.train_loss_txt, new_file=fresh) if self.main else None
        self.valid_log = util.RunningFloatWriter(args.valid_loss_txt, new_file=fresh) if self.main else None

    # ------------------------------------------------------------------------------------------- construction
    def _seed(self, seed):
        """Every rank must shuffle alike (each takes its slice of the same permutation): one seed for all of them."""
        if self.world > 1 and seed is None:
            import torch.distributed as dist
            box = [random.SystemRandom().randrange(1 << 31) if self.main else None]
            dist.broadcast_object_list(box, src=0)
            seed = box[0]
        if seed is not None:
            random.seed(seed)
            torch.manual_seed(seed)

    def _load_data(self):
        a, c = self.args, self.cfg
        if a.train_pats is None:
            raise ValueError('--train-pats is required')
        train_pats = [int(i) for i in a.train_pats.split(',')]
        split = a.train_valid_split if a.train_valid_split >= 0 else None
        if split is None and a.valid_pats is None:
            raise ValueError('give --valid-pats or --train-valid-split')
        self.say('initializing training dataset/dataloader')
        got = dataset.get_dataset(a.input_data_file_path, train_pats, num_classes=c['num-classes'],
                                  pad_img_dim=c['pad-img-size'], data_aug=False, train_valid_split=split,
                                  train_valid_idx=(self.train_idx, self.valid_idx), dup_data_w_left_right_flip=False,
                                  device=self.dev)
        if split is not None:
            self.train_ds, self.valid_ds, self.train_idx, self.valid_idx = got
        else:
            self.train_ds = got
            self.valid_ds = dataset.get_dataset(a.input_data_file_path, [int(i) for i in a.valid_pats.split(',')],
                                                num_classes=c['num-classes'], pad_img_dim=c['pad-img-size'], device=self.dev)
        self.say('Length of training dataset: {}'.format(len(self.train_ds)))
        self.say('Length of validation dataset: {}'.format(len(self.valid_ds)))

    def _build_model(self, resume):
        c = self.cfg
        self.say('creating network')
        self.net = dfl_amd.UNet(n_classes=c['num-classes'], depth=c['depth'], wf=c['init-feats-exp'],
                                batch_norm=c['batch-norm'], padding=c['padding'], max_pool=not c['no-max-pool'],
                                num_lands=c['num-lands'], do_res=c['unet-use-res'], block_depth=c['unet-block-depth'])
        if resume is not None:
            self.net.load_state_dict(resume['model-state-dict'])
        self.net.to(self.dev)
        self.dp = dfl_amd.DataParallel(self.net, compress=self.args.grad_compress) if self.world > 1 else None   # broadcasts rank 0's weights
        if c['num-lands'] > 0:
            self.say('loss: Dice + heat-map NCC (weight {})'.format(c['heat-coeff']))
            self.criterion = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=c['heat-coeff'])
        else:
            self.say('loss: Dice')
            self.criterion = dfl_amd.DiceLoss2D(skip_bg=False)

    def _build_optimizer(self, resume):
        a, c = self.args, self.cfg
        kind, meth = c['optim-type'], c['lrs-meth']
        if meth not in ('cos', 'plateau', 'none'):
            raise ValueError('unknown --lr-sched {!r}'.format(meth))
        self.sched = None
        if kind == 'sgd':
            self.optimizer = dfl_amd.SGD(self.net.parameters(), lr=a.init_lr, momentum=c['opt-momentum'],
                                         weight_decay=c['opt-wgt-decay'], nesterov=c['opt-nesterov'])
            if meth == 'cos':
                self.sched = dfl_amd.WarmRestartLR(self.optimizer, init_run_period_epochs=c['lrs-num-epochs'],
                                                   growth_factor=c['lrs-growth-factor'])
            elif meth == 'plateau':
                self.sched = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='min', factor=0.1,
                                                                  patience=c['lrs-patience'], cooldown=c['lrs-cooldown'])
        elif kind in ('adam', 'rmsprop'):
            if meth != 'none':
                raise ValueError("--optim {} takes --lr-sched none".format(kind))
            if kind == 'adam':
                self.optimizer = optim.Adam(self.net.parameters(), lr=a.init_lr, weight_decay=c['opt-wgt-decay'])
            else:
                self.optimizer = optim.RMSprop(self.net.parameters(), lr=a.init_lr, weight_decay=c['opt-wgt-decay'],
                                               momentum=c['opt-momentum'])
        else:
            raise ValueError('unknown --optim {!r}'.format(kind))
        self.say('optimizer: {}, LR schedule: {}'.format(kind, meth))
        if resume is not None:
            self.optimizer.load_state_dict(resume['optimizer-state-dict'])
            if self.sched is not None:
                self.sched.load_state_dict(resume['scheduler-state-dict'])

    # ------------------------------------------------------------------------------------------- one epoch
    def _flush_rank_losses(self):
        """Data parallel: the per-rank minibatch losses of the last report window, kept on the device, are averaged over the
        ranks with ONE collective and read with one copy -- not one all-reduce + host synchronisation per step."""
        if not self._rank_losses:
            return
        import torch.distributed as dist
        t = torch.stack([v.detach().reshape(()).to(torch.float64) for v in self._rank_losses])
        self._rank_losses = []
        if dist.get_backend() != 'nccl':
            t = t.cpu()
        dist.all_reduce(t)
        for value in (t / self.world).tolist():          # the loss of each global minibatch (mean of equal shards)
            self._account(value)

    def train_epoch(self):
        c, net, opt = self.cfg, self.net, self.optimizer
        n_lands = c['num-lands']
        n_images = len(self.train_ds)
        cosine = self.sched is not None and c['lrs-meth'] == 'cos'
        report_every = int(0.05 * n_images)              # running average printed every 5 % of an epoch's images
        net.train()
        seen = steps = 0
        self._acc = [0.0, 0, 0.0, 0, report_every]       # total, steps, window sum, window count, window length
        late = util.LateScalars(depth=0 if self.args.sync_loss else 1)   # loss values are read one step late (util.LateScalars)
        self._rank_losses = []
        window = max(report_every // max(c['batch-size'] * self.world, 1), 1) if self.world > 1 else 0
        for projs, masks, _, heats in self.train_ds.batches(c['batch-size'], shuffle=True, shard=(self.rank, self.world)):
            opt.zero_grad()
            out = net(projs)
            if n_lands > 0:
                heats = util._squeeze_heats(heats)
                loss = self.criterion((dfl_amd.center_crop(out[0], masks.shape), dfl_amd.center_crop(out[1], heats.shape)),
                                      (masks, heats))
            else:
                loss = self.criterion(dfl_amd.center_crop(out, masks.shape), masks)
            loss.backward()                              # data parallel: gradients leave this call averaged over ranks
            opt.step()
            seen += projs.shape[0] * self.world
            if cosine:
                self.sched.intra_epoch_step(seen / n_images)
            self.last_loss = loss.detach()
            steps += 1
            if self.world > 1:
                self._rank_losses.append(self.last_loss)     # (detached: only the value is used)
                if len(self._rank_losses) >= window or self.args.sync_loss:
                    self._flush_rank_losses()
                continue
            value = late.push(loss)
            if value is not None:
                self._account(value)
        for value in late.flush():
            self._account(value)
        if self.world > 1:
            self._flush_rank_losses()
        if steps == 0:
            raise ValueError('the training set ({} images) yields no minibatch of {} x {} ranks'.format(
                n_images, c['batch-size'], self.world))
        return self._acc[0] / self._acc[1]

    def _account(self, value):
        """One minibatch loss (train.py:430-441): log line, epoch mean, running average every 5 % of the images."""
        if self.main:
            self.train_log.write(value)
        acc = self._acc
        acc[0] += value
        acc[1] += 1
        acc[2] += value
        acc[3] += 1
        if acc[3] == acc[4]:
            self.say('    Running Avg. Loss: {:.6f}'.format(acc[2] / acc[3]))
            acc[2], acc[3] = 0.0, 0

    def validate(self):
        c = self.cfg
        if self.dp is not None:
            # BatchNorm running statistics follow each replica's own shards and drift apart; validation is sharded over the
            # ranks and rank 0 writes the checkpoint, so every rank scores its images with rank 0's statistics -- the logged
            # validation loss, the best-validation choice and the plateau scheduler then describe the model that is saved
            self.dp.sync_buffers(src=0)
        mean, std = util.test_dataset(self.valid_ds, self.net, dev=self.dev,
                                      num_lands=0 if c['use-dice-valid'] else c['num-lands'],
                                      shard=(self.rank, self.world))
        return float(mean), float(std)

    # ------------------------------------------------------------------------------------------- checkpoints
    def _write_checkpoint(self, path):
        state = dict(self.cfg)
        state.update({'epoch': self.epoch, 'model-state-dict': self.net.state_dict(),
                      'optimizer-state-dict': self.optimizer.state_dict(),
                      'scheduler-state-dict': self.sched.state_dict() if self.sched is not None else None,
                      'loss': self.last_loss, 'best-valid-loss': self.best_valid_loss,
                      'train-idx': self.train_idx, 'valid-idx': self.valid_idx})
        torch.save({k: state[k] for k in CHECKPOINT_KEYS}, path)

    def _finish_epoch(self, valid_loss):
        """Scheduler step, bookkeeping and the files of this epoch.  Returns True when a warm restart follows."""
        a, c, snap = self.args, self.cfg, self.snapshots
        restarting = False
        if self.sched is not None:
            if c['lrs-meth'] == 'plateau':
                self.sched.step(valid_loss)
            else:
                self.sched.step()
                restarting = bool(self.sched.just_restarted)
        if restarting:
            self.say('  warm restart: the next epoch starts a new cosine period')
            c['lrs-num-restarts'] += 1
        self.epoch += 1
        improved = self.best_valid_loss is None or valid_loss < self.best_valid_loss
        if improved:
            self.best_valid_loss = valid_loss
        snap.new_epoch()
        if self.epoch % c['checkpoint-freq'] == 0:
            self.say('  Saving checkpoint')
            snap.put(a.checkpoint_net)
        if improved and c['save-best-valid']:
            self.say('  Saving best validation (loss: {:.6f})'.format(self.best_valid_loss))
            snap.put(a.best_net)
        prefix = c['lrs-save-restart-net-prefix']
        if restarting and prefix is not None and c['lrs-num-restarts'] >= c['lrs-save-after-n-restarts']:
            path = '{}_{:02d}.pt'.format(prefix, c['lrs-num-restarts'] - 1)
            self.say('  Saving network before restart {} to {}'.format(c['lrs-num-restarts'], path))
            snap.put(path)

    def _stop_reason(self, hours_so_far, hours_per_epoch):
        a, c = self.args, self.cfg
        if a.max_hours > 0 and hours_so_far + hours_per_epoch > a.max_hours:
            return 'another epoch would not fit into the time limit'
        if c['lrs-max-num-restarts'] > 0:
            if c['lrs-num-restarts'] >= c['lrs-max-num-restarts']:
                return 'maximum number of restarts performed'
        elif self.epoch >= a.max_num_epochs:
            return 'maximum number of epochs performed'
        return None

    # ------------------------------------------------------------------------------------------- driver
    def run(self):
        a = self.args
        self.say('Start Training...' + (' ({} ranks, global batch {})'.format(self.world, self.world * self.cfg['batch-size'])
                                        if self.world > 1 else ''))
        hours = 0.0
        done = 0
        while True:
            t0 = time.time()
            self.say('Epoch: {:03d}'.format(self.epoch))
            train_loss = self.train_epoch()
            self.say('  Running validation')
            valid_loss, valid_std = self.validate()
            if self.main:
                self.valid_log.write(valid_loss)
            self.say('  Avg. Training Loss: {:.6f}'.format(train_loss))
            self.say('  Validation Loss: {:.6f} +/- {:.6f}'.format(valid_loss, valid_std))
            self._finish_epoch(valid_loss)
            spent = (time.time() - t0) / 3600.0
            hours += spent
            done += 1
            self.say('  This epoch took {:.4f} hours (average {:.4f})'.format(spent, hours / done))
            why = self._stop_reason(hours, hours / done)
            if self.world > 1:                      # wall-clock decisions must not differ between ranks: rank 0 decides
                import torch.distributed as dist
                box = [why]
                dist.broadcast_object_list(box, src=0)
                why = box[0]
            if why is not None:
                self.say('  Exiting - {}!'.format(why))
                self.snapshots.put(a.checkpoint_net)      # the final state always ends up in the checkpoint file
                break
        if self.main:
            self.train_log.close()
            self.valid_log.close()
        self.say('Training Hours: {:.4f}'.format(hours))
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def main(argv=None):
    Trainer(build_full_parser().parse_args(argv)).run()


if __name__ == '__main__':
    main()
