#!/usr/bin/env python3
"""Training entry point: the command line, epoch loop, loss logs and checkpoint files of the reference's
train_test_code/train.py (flags :29-100, resume :189-270, step body :393-430, checkpoint dictionary :463-513, exit
rules :556-577), driving the MI355X path -- dfl_amd.UNet / losses / SGD / WarmRestartLR and the GPU-resident loader.

    python train.py data.h5 --train-pats 1,2,3 --valid-pats 4 --num-classes 7 --unet-img-dim 192 --batch-size 16 \
        --unet-num-lvls 6 --unet-init-feats-exp 5 --unet-batch-norm --unet-padding --unet-no-max-pool --use-lands \
        --nesterov --wgt-decay 1e-4

Checkpoints written here load in the reference's scripts and vice versa: same dictionary keys, same state_dict names.
Differences, all loud: --no-gpu and --data-aug are refused (the HIP path has neither a CPU fallback nor the PIL-based
random augmentation); the data file may be the reference's HDF5 (read through h5py when it is
installed) or an .npz with the same dataset names; batches come from the loader's own GPU-side batch builder instead
of a torch DataLoader over host tensors (same shuffling granularity: one random permutation per epoch).
"""
import argparse
import os
import shutil
import sys
import time

import torch
from torch import optim

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dfl_amd  # noqa: E402
from dfl_amd import dataset, util  # noqa: E402
from dfl_amd._native import DflError  # noqa: E402

# keys of the checkpoint dictionary, in the reference's order (train.py:465-510)
CHECKPOINT_KEYS = ['epoch', 'model-state-dict', 'optim-type', 'optimizer-state-dict', 'scheduler-state-dict', 'loss',
                   'best-valid-loss', 'save-best-valid', 'num-classes', 'depth', 'init-feats-exp', 'batch-norm', 'padding',
                   'no-max-pool', 'pad-img-size', 'batch-size', 'data-aug', 'opt-nesterov', 'opt-momentum',
                   'opt-wgt-decay', 'num-lands', 'heat-coeff', 'use-dice-valid', 'unet-use-res', 'unet-block-depth',
                   'lrs-meth', 'lrs-num-epochs', 'lrs-growth-factor', 'lrs-max-num-restarts',
                   'lrs-save-restart-net-prefix', 'lrs-save-after-n-restarts', 'lrs-num-restarts', 'lrs-patience',
                   'lrs-cooldown', 'checkpoint-freq', 'train-idx', 'valid-idx']


def build_parser():
    p = argparse.ArgumentParser(description='Training.', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('input_data_file_path', type=str, help='Path to the datafile containing projections and segmentations')
    p.add_argument('--train-pats', type=str, help='comma delimited list of patient IDs used for training')
    p.add_argument('--valid-pats', type=str, help='comma delimited list of patient IDs used for validation')
    p.add_argument('--num-classes', type=int, help='The number of label classes to be identified')
    p.add_argument('--batch-size', type=int, default=1, help='Number of images each minibatch')
    p.add_argument('--unet-img-dim', type=int, default=364,
                   help='Dimension to adjust input images to before inputting into U-Net')
    p.add_argument('--checkpoint-net', type=str, default='zz_checkpoint.pt', help='Path to network saved as checkpoint')
    p.add_argument('--best-net', type=str, default='zz_best_valid.pt',
                   help='Path to network saved with best score on the validation data')
    p.add_argument('--checkpoint-freq', type=int, default=1,
                   help='Frequency (in terms of epochs) at which to save the network checkpoint to disk.')
    p.add_argument('--no-save-best-valid', action='store_true', help='Do not save best validation netowrk to disk.')
    p.add_argument('--optim', type=str, default='sgd', help='Optimization strategy to use.')
    p.add_argument('--lr-sched', type=str, default='cos',
                   help="Learning rate scheduling method. 'cos' --> Cosine annealing with warm restarts, 'none' --> fixed "
                        "LR (at initial), 'plateau' --> reduce learning rate when validation score plateaus")
    p.add_argument('--init-lr', type=float, default=1.0e-2, help='Initial learning rate for SGN using cosine annealing')
    p.add_argument('--lr-patience', type=int, default=20, help='Patience, in # epochs, when using LR plateau decay')
    p.add_argument('--lr-cooldown', type=int, default=20, help='Cooldown, in # epochs, when using LR plateau decay')
    p.add_argument('--nesterov', action='store_true', help='Use Nesterov momentum in SGD')
    p.add_argument('--momentum', type=float, default=0.9, help='SGD momentum term')
    p.add_argument('--wgt-decay', type=float, default=0, help='SGD weight decay term')
    p.add_argument('--cos-anneal-epochs', type=int, default=10,
                   help='Number of epochs in the cosine annealing LR scheduling. When using warm restarts with a growth '
                        'factor, this is the initial period.')
    p.add_argument('--cos-growth', type=int, default=2, help='Growth factor to use with warm restarts.')
    p.add_argument('--save-restart-net', type=str,
                   help='Prefix used to save networks before warm restart, file path will be <PREFIX>_XX.pt, where XX is '
                        'the restart index')
    p.add_argument('--save-after-n-restarts', type=int, default=0,
                   help='Save networks prior to warm restart only after this number of restarts have been performed.')
    p.add_argument('--max-num-restarts', type=int, default=-1,
                   help='Maximum number of warm restarts; disabled when <= 0, otherwise overrides --max-num-epochs')
    p.add_argument('--max-num-epochs', type=int, default=200, help='Maximum number of epochs')
    p.add_argument('--train-loss-txt', type=str, default='train_iter_loss.txt', help='output file for training loss')
    p.add_argument('--valid-loss-txt', type=str, default='valid_loss.txt', help='output file for validation loss')
    p.add_argument('--no-gpu', action='store_true', help='Only use CPU - do not use GPU even if it is available')
    p.add_argument('--max-hours', type=float, default=-1.0,
                   help='Maximum number of hours to run for; terminates when the program does not expect to be able to '
                        'complete another epoch. A non-positive value indicates no maximum limit.')
    p.add_argument('--unet-num-lvls', type=int, default=5, help='Number of levels in the U-Net')
    p.add_argument('--unet-init-feats-exp', type=int, default=4,
                   help='Number of initial features used in the U-Net, two raised to this power.')
    p.add_argument('--unet-batch-norm', action='store_true', help='Use Batch Normalization in U-Net')
    p.add_argument('--unet-padding', action='store_true', help='Add padding to preserve image sizes for U-Net')
    p.add_argument('--unet-no-max-pool', action='store_true', help='Learn downsampling weights instead of max-pooling')
    p.add_argument('--unet-block-depth', type=int, default=2, help='Depth of the blocks of convolutions at each level')
    p.add_argument('--data-aug', action='store_true', help='Randomly augment the data')
    p.add_argument('--use-lands', action='store_true', help='Learn landmark heatmaps')
    p.add_argument('--heat-coeff', type=float, default=0.5,
                   help='Weighting applied to heatmap loss - dice gets one minus this.')
    p.add_argument('--dice-valid', action='store_true',
                   help='Use only dice validation loss even when training with dice + heatmap loss')
    p.add_argument('--unet-no-res', action='store_true', help='Do not use residual connections in U-Net blocks')
    p.add_argument('--train-valid-split', type=float, default=-1.0,
                   help='Ratio of training data to keep as training, one minus this is used for validation. Enabled when '
                        'a value in [0,1] is provided, and overrides the valid-pats flag.')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    data_file_path = args.input_data_file_path
    assert args.train_pats is not None
    train_pats = [int(i) for i in args.train_pats.split(',')]
    assert len(train_pats) > 0
    valid_pats = None
    if args.train_valid_split < 0:
        assert args.valid_pats is not None
        valid_pats = [int(i) for i in args.valid_pats.split(',')]
        assert len(valid_pats) > 0
    if args.no_gpu:
        raise DflError('--no-gpu: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')
    dev = dfl_amd.get_device()
    if dev.type != 'cuda':
        raise DflError('no GPU visible: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')

    # run configuration: the command line, overridden by a checkpoint when one exists (train.py:189-270)
    c = {'save-best-valid': not args.no_save_best_valid, 'num-classes': args.num_classes, 'optim-type': args.optim,
         'depth': args.unet_num_lvls, 'init-feats-exp': args.unet_init_feats_exp, 'batch-norm': args.unet_batch_norm,
         'padding': args.unet_padding, 'no-max-pool': args.unet_no_max_pool, 'pad-img-size': args.unet_img_dim,
         'batch-size': args.batch_size, 'data-aug': args.data_aug, 'num-lands': 0, 'heat-coeff': args.heat_coeff,
         'use-dice-valid': args.dice_valid, 'unet-use-res': not args.unet_no_res,
         'unet-block-depth': args.unet_block_depth, 'opt-nesterov': args.nesterov, 'opt-momentum': args.momentum,
         'opt-wgt-decay': args.wgt_decay, 'lrs-meth': args.lr_sched.lower(), 'lrs-num-epochs': args.cos_anneal_epochs,
         'lrs-growth-factor': args.cos_growth, 'lrs-max-num-restarts': args.max_num_restarts,
         'lrs-save-restart-net-prefix': args.save_restart_net, 'lrs-save-after-n-restarts': args.save_after_n_restarts,
         'lrs-num-restarts': 0, 'lrs-patience': args.lr_patience, 'lrs-cooldown': args.lr_cooldown,
         'checkpoint-freq': args.checkpoint_freq}
    if args.use_lands:
        c['num-lands'] = dataset.get_num_lands_from_dataset(data_file_path)
        print('num. lands read from file: {}'.format(c['num-lands']))
        assert c['num-lands'] > 0
    train_valid_split = args.train_valid_split
    train_idx = valid_idx = None
    prev_state = None
    load_from_checkpoint = os.path.exists(args.checkpoint_net)
    if load_from_checkpoint:
        print('loading state from checkpoint...')
        prev_state = torch.load(args.checkpoint_net, map_location='cpu', weights_only=False)
        print('loading unet params from checkpoint state dict...')
        for k in c:
            c[k] = prev_state[k]
        for label, k in (('num. classes', 'num-classes'), ('optim. type', 'optim-type'), ('depth', 'depth'),
                         ('init. feats. exp.', 'init-feats-exp'), ('batch norm.', 'batch-norm'),
                         ('unet do pad img.', 'padding'), ('no max pool', 'no-max-pool'),
                         ('reflect pad img. dim.', 'pad-img-size'), ('batch size', 'batch-size'), ('data aug.', 'data-aug'),
                         ('num. landmarks', 'num-lands'), ('use dice for valid.', 'use-dice-valid'),
                         ('unet use res.', 'unet-use-res'), ('unet block depth', 'unet-block-depth'),
                         ('nesterov', 'opt-nesterov'), ('momentum', 'opt-momentum'), ('weight decay', 'opt-wgt-decay'),
                         ('LR Sched. Method', 'lrs-meth'), ('LR Sched. Num. Epochs', 'lrs-num-epochs'),
                         ('LR Sched. Growth Factor', 'lrs-growth-factor'),
                         ('LR Sched. Max. Num. Restarts', 'lrs-max-num-restarts'),
                         ('LR Sched. Save After Restart Prefix', 'lrs-save-restart-net-prefix'),
                         ('LR Sched. Save After N Restarts', 'lrs-save-after-n-restarts'),
                         ('LR Sched. Cur. Num. Restarts', 'lrs-num-restarts'), ('LR Plateau Patience', 'lrs-patience'),
                         ('LR Plateau Cooldown', 'lrs-cooldown')):
            print('{:>38}: {}'.format(label, c[k]))
        print('Checkpoint Freq.: {} epochs'.format(c['checkpoint-freq']))
        if train_valid_split >= 0:
            print('loading previous train/valid split inds.')
            train_idx, valid_idx = prev_state['train-idx'], prev_state['valid-idx']
            assert train_idx is not None and valid_idx is not None
    num_lands = c['num-lands']
    lrs_is_cos, lrs_none, lrs_plateau = c['lrs-meth'] == 'cos', c['lrs-meth'] == 'none', c['lrs-meth'] == 'plateau'
    enforce_max_num_restarts = c['lrs-max-num-restarts'] > 0
    enforce_max_hours = args.max_hours > 0
    if c['data-aug']:
        raise NotImplementedError('--data-aug: random data augmentation is outside the HIP path (DESIGN.md section 7)')

    print('initializing training dataset/dataloader')
    train_ds = dataset.get_dataset(data_file_path, train_pats, num_classes=c['num-classes'], pad_img_dim=c['pad-img-size'],
                                   data_aug=False, train_valid_split=train_valid_split if train_valid_split >= 0 else None,
                                   train_valid_idx=(train_idx, valid_idx), dup_data_w_left_right_flip=False, device=dev)
    valid_ds = None
    if train_valid_split >= 0:
        assert type(train_ds) is tuple
        train_ds, valid_ds, train_idx, valid_idx = train_ds
    train_ds_len = len(train_ds)
    print('Length of training dataset: {}'.format(train_ds_len))
    if train_valid_split < 0:
        print('initializing validation dataset')
        valid_ds = dataset.get_dataset(data_file_path, valid_pats, num_classes=c['num-classes'],
                                       pad_img_dim=c['pad-img-size'], device=dev)
    print('Length of validation dataset: {}'.format(len(valid_ds)))

    print('creating network')
    net = dfl_amd.UNet(n_classes=c['num-classes'], depth=c['depth'], wf=c['init-feats-exp'], batch_norm=c['batch-norm'],
                       padding=c['padding'], max_pool=not c['no-max-pool'], num_lands=num_lands, do_res=c['unet-use-res'],
                       block_depth=c['unet-block-depth'])
    if load_from_checkpoint:
        net.load_state_dict(prev_state['model-state-dict'])
    print('moving network to device...')
    net.to(dev)
    print('creating loss function')
    if num_lands > 0:
        print('  Dice + Heatmap Loss...')
        criterion = dfl_amd.DiceAndHeatMapLoss2D(skip_bg=False, heatmap_wgt=c['heat-coeff'])
    else:
        print('  Dice only...')
        criterion = dfl_amd.DiceLoss2D(skip_bg=False)

    lr_sched = None
    if c['optim-type'] == 'sgd':
        print('creating SGD optimizer and LR scheduler')
        optimizer = dfl_amd.SGD(net.parameters(), lr=args.init_lr, momentum=c['opt-momentum'],
                                weight_decay=c['opt-wgt-decay'], nesterov=c['opt-nesterov'])
        if lrs_is_cos:
            lr_sched = dfl_amd.WarmRestartLR(optimizer, init_run_period_epochs=c['lrs-num-epochs'],
                                             growth_factor=c['lrs-growth-factor'])
        elif lrs_plateau:
            lr_sched = optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.1, patience=c['lrs-patience'],
                                                            cooldown=c['lrs-cooldown'])
        else:
            assert lrs_none
    elif c['optim-type'] == 'adam':
        print('creating ADAM optimizer')
        optimizer = optim.Adam(net.parameters(), lr=args.init_lr, weight_decay=c['opt-wgt-decay'])
        assert lrs_none
    elif c['optim-type'] == 'rmsprop':
        print('creating RMSProp optimizer')
        optimizer = optim.RMSprop(net.parameters(), lr=args.init_lr, weight_decay=c['opt-wgt-decay'],
                                  momentum=c['opt-momentum'])
        assert lrs_none
    else:
        raise ValueError('unknown --optim {!r}'.format(c['optim-type']))

    best_valid_loss = None
    epoch = 0
    if load_from_checkpoint:
        optimizer.load_state_dict(prev_state['optimizer-state-dict'])
        if lr_sched is not None:
            lr_sched.load_state_dict(prev_state['scheduler-state-dict'])
        best_valid_loss = prev_state['best-valid-loss']
        epoch = prev_state['epoch']
    del prev_state

    train_iter_loss_out = util.RunningFloatWriter(args.train_loss_txt, new_file=not load_from_checkpoint)
    valid_loss_out = util.RunningFloatWriter(args.valid_loss_txt, new_file=not load_from_checkpoint)
    tot_time_this_session_hours = 0.0
    num_epochs_completed_this_session = 0
    loss = None

    def save_net(net_path):
        state = dict(c)
        state.update({'epoch': epoch, 'model-state-dict': net.state_dict(), 'optimizer-state-dict': optimizer.state_dict(),
                      'scheduler-state-dict': lr_sched.state_dict() if lr_sched is not None else None,
                      'loss': loss.detach() if loss is not None else None, 'best-valid-loss': best_valid_loss,
                      'train-idx': train_idx, 'valid-idx': valid_idx})
        tmp_name = '{}.tmp'.format(net_path)
        torch.save({k: state[k] for k in CHECKPOINT_KEYS}, tmp_name)
        shutil.move(tmp_name, net_path)

    print('Start Training...')
    keep_training = True
    while keep_training:
        epoch_start_time = time.time()
        print('Epoch: {:03d}'.format(epoch))
        net.train()
        num_batches, avg_loss = 0, 0.0
        running_loss, running_loss_iter = 0.0, 0
        running_loss_num_iters = int(0.05 * train_ds_len)
        num_examples_run = 0
        for projs, masks, lands, heats in train_ds.batches(c['batch-size'], shuffle=True):
            if num_lands > 0 and heats.dim() > 4:
                assert heats.dim() == 5 and heats.shape[2] == 1
                heats = heats.view(heats.shape[0], heats.shape[1], heats.shape[3], heats.shape[4])
            optimizer.zero_grad()
            net_out = net(projs)
            if num_lands > 0:
                pred_masks = dfl_amd.center_crop(net_out[0], masks.shape)
                pred_heat_maps = dfl_amd.center_crop(net_out[1], heats.shape)
                loss = criterion((pred_masks, pred_heat_maps), (masks, heats))
            else:
                loss = criterion(dfl_amd.center_crop(net_out, masks.shape), masks)
            loss.backward()
            optimizer.step()
            num_examples_run += projs.shape[0]
            if lr_sched is not None and lrs_is_cos:
                lr_sched.intra_epoch_step(num_examples_run / train_ds_len)
            l = loss.item()
            train_iter_loss_out.write(l)
            avg_loss += l
            num_batches += 1
            running_loss += l
            running_loss_iter += 1
            if running_loss_iter == running_loss_num_iters:
                print('    Running Avg. Loss: {:.6f}'.format(running_loss / running_loss_num_iters))
                running_loss_iter, running_loss = 0, 0.0
        avg_loss /= num_batches

        print('  Running validation')
        avg_valid_loss, std_valid_loss = util.test_dataset(valid_ds, net, dev=dev,
                                                           num_lands=0 if c['use-dice-valid'] else num_lands)
        avg_valid_loss, std_valid_loss = float(avg_valid_loss), float(std_valid_loss)
        valid_loss_out.write(avg_valid_loss)
        print('  Avg. Training Loss: {:.6f}'.format(avg_loss))
        print('  Validation Loss: {:.6f} +/- {:.6f}'.format(avg_valid_loss, std_valid_loss))
        if lr_sched is not None:
            if lrs_plateau:
                lr_sched.step(avg_valid_loss)
            else:
                lr_sched.step()
            if lrs_is_cos and lr_sched.just_restarted:
                print('  Next epoch is warm restart...')
                c['lrs-num-restarts'] += 1
        epoch += 1
        new_best_valid = best_valid_loss is None or avg_valid_loss < best_valid_loss
        if new_best_valid:
            best_valid_loss = avg_valid_loss

        net_saved_this_epoch_path = None
        if epoch % c['checkpoint-freq'] == 0:
            print('  Saving checkpoint')
            save_net(args.checkpoint_net)
            net_saved_this_epoch_path = args.checkpoint_net
        if new_best_valid and c['save-best-valid']:
            print('  Saving best validation (loss: {:.6f})'.format(best_valid_loss))
            if net_saved_this_epoch_path is not None:
                shutil.copy(net_saved_this_epoch_path, args.best_net)
            else:
                save_net(args.best_net)
                net_saved_this_epoch_path = args.best_net
        prefix = c['lrs-save-restart-net-prefix']
        if lrs_is_cos and lr_sched.just_restarted and prefix is not None \
                and c['lrs-num-restarts'] >= c['lrs-save-after-n-restarts']:
            restart_net_path = '{}_{:02d}.pt'.format(prefix, c['lrs-num-restarts'] - 1)
            print('  Saving network before restart {} to {}'.format(c['lrs-num-restarts'], restart_net_path))
            if net_saved_this_epoch_path is not None:
                shutil.copy(net_saved_this_epoch_path, restart_net_path)
            else:
                save_net(restart_net_path)
                net_saved_this_epoch_path = restart_net_path

        this_epoch_hours = (time.time() - epoch_start_time) / 3600.0
        print('  This epoch took {:.4f} hours!'.format(this_epoch_hours))
        tot_time_this_session_hours += this_epoch_hours
        num_epochs_completed_this_session += 1
        avg_epoch_time_hours = tot_time_this_session_hours / num_epochs_completed_this_session
        print('  Current average epoch runtime: {:.4f} hours'.format(avg_epoch_time_hours))
        if enforce_max_hours and tot_time_this_session_hours + avg_epoch_time_hours > args.max_hours:
            print('  Exiting - did not expect to be able to complete next expoch within time limit!')
            keep_training = False
        if enforce_max_num_restarts:
            if c['lrs-num-restarts'] >= c['lrs-max-num-restarts']:
                keep_training = False
                print('  Exiting - maximum number of restarts performed!')
        elif epoch >= args.max_num_epochs:
            keep_training = False
            print('  Exiting - maximum number of epochs performed!')
        if not keep_training:
            print('    saving checkpoint before exit!')
            if net_saved_this_epoch_path is None:
                save_net(args.checkpoint_net)
            elif net_saved_this_epoch_path != args.checkpoint_net:
                shutil.copy(net_saved_this_epoch_path, args.checkpoint_net)
    train_iter_loss_out.close()
    valid_loss_out.close()
    print('Training Hours: {:.4f}'.format(tot_time_this_session_hours))


if __name__ == '__main__':
    main()
