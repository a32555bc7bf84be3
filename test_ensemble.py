#!/usr/bin/env python3
"""Ensemble inference entry point: the command line and output file of the reference's train_test_code/test_ensemble.py
(flags :25-37, checkpoint fields :61-77, output layout :123-135 + util.py:300-310), driving the MI355X path.

    python test_ensemble.py data.h5 out.h5 --pats 1 --nets net_a.pt net_b.pt ... [--times times.txt]

Loads the checkpoints train.py (this one or the reference's) wrote, runs every net on every projection of the chosen
patients -- each forward is one hipGraph replay -- and writes 'nn-segs' (uint8 labels of the averaged soft-max) and
'nn-heats' (average of the per-net min-max-normalised heat maps) plus the 'land-names' group.  Containers: the
reference's HDF5 (read and written by the dependency-free dfl_amd.h5lite -- gzip-9, one chunk per image / heat map, streamed
to disk), or .npz files with the same dataset names (dataset.open_output_container).  Ensemble members must agree on
class count, landmark count and padded image size (the reference silently takes the last net's).  --no-gpu is refused:
there is no CPU path.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dfl_amd  # noqa: E402
from dfl_amd import dataset, util  # noqa: E402
from dfl_amd._native import DflError  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description='Segment and locate landmarks with an ensemble of trained U-Nets.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('input_data_file_path', type=str, help='pre-processed data file with the projections')
    p.add_argument('output_data_file_path', type=str, help='file to write nn-segs / nn-heats to')
    p.add_argument('--nets', type=str, nargs='+', help='checkpoints of the ensemble members (after the positional arguments)')
    p.add_argument('--pats', type=str, help='specimen numbers to process, comma separated')
    p.add_argument('--no-gpu', action='store_true', help='(refused: this implementation has no CPU path)')
    p.add_argument('--times', type=str, default='', help='write the seconds spent per image to this file')
    return p


SHOWN = (('num-classes', 'classes'), ('depth', 'levels'), ('init-feats-exp', 'first-level channels (log2)'),
         ('batch-norm', 'BatchNorm'), ('padding', 'padded convolutions'), ('no-max-pool', 'strided-conv down-sampling'),
         ('pad-img-size', 'padded image side'), ('unet-use-res', 'residual blocks'), ('unet-block-depth', 'convs per block'),
         ('num-lands', 'landmarks'), ('epoch', 'epochs trained'), ('best-valid-loss', 'best validation loss'))


def load_member(path, dev):
    """One ensemble member from a checkpoint of train.py (this build's or the reference's: same dictionary)."""
    state = torch.load(path, map_location='cpu', weights_only=False)
    print('member {}'.format(path))
    for key, label in SHOWN:
        print('  {:<28} {}'.format(label + ':', state[key]))
    last = state['loss']
    print('  {:<28} {}'.format('last training loss:', None if last is None else float(last)))
    net = dfl_amd.UNet(n_classes=state['num-classes'], depth=state['depth'], wf=state['init-feats-exp'],
                       batch_norm=state['batch-norm'], padding=state['padding'], max_pool=not state['no-max-pool'],
                       num_lands=state['num-lands'], do_res=state['unet-use-res'], block_depth=state['unet-block-depth'])
    net.load_state_dict(state['model-state-dict'])
    return net.to(dev).eval(), (state['num-classes'], state['num-lands'], state['pad-img-size'])


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.no_gpu:
        raise DflError('--no-gpu: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')
    if not args.pats or not args.nets:
        raise SystemExit('--pats and --nets are required')
    test_pats = [int(i) for i in args.pats.split(',')]
    dev = dfl_amd.get_device()
    if dev.type != 'cuda':
        raise DflError('no GPU visible: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')
    members = [load_member(path, dev) for path in args.nets]
    nets = [m[0] for m in members]
    shapes = {m[1] for m in members}
    if len(shapes) != 1:
        raise ValueError('ensemble members disagree on (classes, landmarks, padded image side): {}'.format(sorted(shapes)))
    num_classes, num_lands, proj_unet_dim = shapes.pop()

    land_names = None
    if num_lands > 0:
        land_names = dataset.get_land_names_from_dataset(args.input_data_file_path)
        if len(land_names) != num_lands:
            raise ValueError('the data file names {} landmarks, the networks predict {}'.format(len(land_names), num_lands))
    test_ds = dataset.get_dataset(args.input_data_file_path, test_pats, num_classes=num_classes,
                                  pad_img_dim=proj_unet_dim, no_seg=True, device=dev)
    print('Length of testing dataset: {}'.format(len(test_ds)))
    f = dataset.open_output_container(args.output_data_file_path)
    if land_names:                                   # the group est_lands_csv.py reads the names from (test_ensemble.py:124-129)
        g = f.create_group('land-names')
        g['num-lands'] = num_lands
        for l, name in enumerate(land_names):
            g['land-{:02d}'.format(l)] = name
    times = []
    print('running {} network(s) on {} projections'.format(len(nets), len(test_ds)))
    util.seg_dataset_ensemble(test_ds, nets, f, dev=dev, num_lands=num_lands, times=times)
    f.flush()
    f.close()
    if args.times:
        with open(args.times, 'w') as out:
            for t in times:
                out.write('{:.6f}\n'.format(t))


if __name__ == '__main__':
    main()
