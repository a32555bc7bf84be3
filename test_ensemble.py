#!/usr/bin/env python3
"""Ensemble inference entry point: the command line and output file of the reference's train_test_code/test_ensemble.py
(flags :25-37, checkpoint fields :61-77, output layout :123-135 + util.py:300-310), driving the MI355X path.

    python test_ensemble.py data.h5 out.h5 --pats 1 --nets net_a.pt net_b.pt ... [--times times.txt]

Loads the checkpoints train.py (this one or the reference's) wrote, runs every net on every projection of the chosen
patients -- each forward is one hipGraph replay -- and writes 'nn-segs' (uint8 labels of the averaged soft-max) and
'nn-heats' (average of the per-net min-max-normalised heat maps) plus the 'land-names' group.  Containers: HDF5 through
h5py when it is installed, or .npz files with the same dataset names (dataset.open_output_container).  --no-gpu is
refused: there is no CPU path.
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
    p = argparse.ArgumentParser(description='Run ensemble segmentation and heatmap estimation.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('input_data_file_path', type=str, help='Path to the datafile containing projections')
    p.add_argument('output_data_file_path', type=str, help='Path to the output datafile containing segmentations')
    p.add_argument('--nets', type=str, nargs='+',
                   help='Paths to the networks used to perform segmentation - specify this after the positional arguments')
    p.add_argument('--pats', type=str, help='comma delimited list of patient IDs used for testing')
    p.add_argument('--no-gpu', action='store_true', help='Only use CPU - do not use GPU even if it is available')
    p.add_argument('--times', type=str, default='', help='Path to file storing runtimes for each image')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.pats is not None
    test_pats = [int(i) for i in args.pats.split(',')]
    assert len(test_pats) > 0
    assert args.nets, 'no networks given (--nets)'
    if args.no_gpu:
        raise DflError('--no-gpu: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')
    dev = dfl_amd.get_device()
    if dev.type != 'cuda':
        raise DflError('no GPU visible: this implementation runs on the MI355X only (HIP kernels, no CPU fallback)')

    nets = []
    num_classes = num_lands = proj_unet_dim = None
    for net_path in args.nets:
        print('  loading state from disk for: {}'.format(net_path))
        state = torch.load(net_path, map_location='cpu', weights_only=False)
        print('  loading unet params from checkpoint state dict...')
        num_classes, num_lands, proj_unet_dim = state['num-classes'], state['num-lands'], state['pad-img-size']
        for label, k in (('num. classes', 'num-classes'), ('depth', 'depth'), ('init. feats. exp.', 'init-feats-exp'),
                         ('batch norm.', 'batch-norm'), ('unet do pad img.', 'padding'), ('no max pool', 'no-max-pool'),
                         ('reflect pad img. dim.', 'pad-img-size'), ('unet use res.', 'unet-use-res'),
                         ('unet block depth', 'unet-block-depth'), ('batch size', 'batch-size'),
                         ('num. lands.', 'num-lands')):
            print('{:>25}: {}'.format(label, state[k]))
        print('          Last Epoch: {}'.format(state['epoch']))
        print('           Last Loss: {}'.format(state['loss'].item() if state['loss'] is not None else None))
        print('    Best Valid. Loss: {}'.format(state['best-valid-loss']))
        print('    creating network')
        net = dfl_amd.UNet(n_classes=num_classes, depth=state['depth'], wf=state['init-feats-exp'],
                           batch_norm=state['batch-norm'], padding=state['padding'], max_pool=not state['no-max-pool'],
                           num_lands=num_lands, do_res=state['unet-use-res'], block_depth=state['unet-block-depth'])
        net.load_state_dict(state['model-state-dict'])
        del state
        print('  moving network to device...')
        net.to(dev)
        nets.append(net)

    land_names = None
    if num_lands > 0:
        land_names = dataset.get_land_names_from_dataset(args.input_data_file_path)
        assert len(land_names) == num_lands
    print('initializing testing dataset')
    test_ds = dataset.get_dataset(args.input_data_file_path, test_pats, num_classes=num_classes,
                                  pad_img_dim=proj_unet_dim, no_seg=True, device=dev)
    print('Length of testing dataset: {}'.format(len(test_ds)))
    print('opening destination file for writing')
    f = dataset.open_output_container(args.output_data_file_path)
    if land_names:
        land_names_g = f.create_group('land-names')
        land_names_g['num-lands'] = num_lands
        for l in range(num_lands):
            land_names_g['land-{:02d}'.format(l)] = land_names[l]
    times = []
    print('running network on projections')
    util.seg_dataset_ensemble(test_ds, nets, f, dev=dev, num_lands=num_lands, times=times)
    print('closing file...')
    f.flush()
    f.close()
    if args.times:
        with open(args.times, 'w') as times_out:
            for t in times:
                times_out.write('{:.6f}\n'.format(t))


if __name__ == '__main__':
    main()
