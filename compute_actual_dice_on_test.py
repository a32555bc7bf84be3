#!/usr/bin/env python3
"""Hard Dice per projection and foreground label as CSV: the command line and output format of the reference's
train_test_code/compute_actual_dice_on_test.py (:20-31 arguments, :61 header, :93 rows ``pat,proj,label,dice`` with two
decimals), computed on the GPU by dfl_hard_dice (one launch for the whole patient instead of a Python loop per label).

    python compute_actual_dice_on_test.py data.h5 out.h5 nn-segs dice.csv 4 [--no-hdr] [--num-classes 7]

Files: the reference's HDF5 (dependency-free reader dfl_amd.h5lite) or .npz with the same dataset names.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dfl_amd  # noqa: E402,F401
from dfl_amd import dataset, util  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description='compute actual dice coefficients between estimated segmentations and ground '
                                            'truth. Scores are written out in CSV format.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('ds_path', type=str, help='Path to dataset containing projections')
    p.add_argument('seg_file', type=str, help='Path to H5 file with estimated segmentations')
    p.add_argument('seg_group', type=str, help='Path within H5 file of estimated segmentations')
    p.add_argument('csv_out', type=str, help='Path to output CSV file')
    p.add_argument('pat_ind', type=int, help='patient index')
    p.add_argument('--no-hdr', action='store_true', help='No CSV header')
    p.add_argument('--num-classes', type=int, default=7, help='number of classes in segmentation')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    get, close = dataset._open_container(args.ds_path)
    gt_segs = torch.from_numpy(np.asarray(get('{:02d}/segs'.format(args.pat_ind))))
    close()
    get, close = dataset._open_container(args.seg_file)
    est_segs = torch.from_numpy(np.asarray(get(args.seg_group)))
    close()
    num_projs = gt_segs.shape[0]
    assert num_projs == est_segs.shape[0]
    dev = dfl_amd.get_device()
    dice = util.hard_dice(est_segs.to(dev), gt_segs.to(dev), args.num_classes).cpu()      # [projs, classes - 1]
    assert bool(((dice > -1.0e-8) & (dice < 1 + 1.0e-8)).all())
    with open(args.csv_out, 'w') as csv_out:
        if not args.no_hdr:
            csv_out.write('pat,proj,label,dice\n')
        for proj in range(num_projs):
            for l in range(1, args.num_classes):            # background excluded
                csv_out.write('{},{},{},{:.2f}\n'.format(args.pat_ind, proj, l, float(dice[proj, l - 1])))


if __name__ == '__main__':
    main()
