#!/usr/bin/env python3
"""Landmark locations from heat maps as CSV: the command line and output format of the reference's
train_test_code/est_lands_csv.py (:24-38 arguments, :77 header, :127-128 rows ``pat,proj,land,row,col,time``; row = col =
-1 when the landmark is not detected), computed on the GPU by dfl_est_lands: seg-masked arg-max, 25x25 Gaussian template,
normalised cross-correlation >= 0.9 (:96-124) for every (projection, landmark) pair in one launch.

    python est_lands_csv.py out.h5 nn-heats --use-seg nn-segs --pat 4 --out lands.csv

The time column holds the batch's wall time divided by the number of landmarks (the reference times each one).
Files: the reference's HDF5 (dependency-free reader dfl_amd.h5lite) or .npz with the same dataset names.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dfl_amd  # noqa: E402
from dfl_amd import dataset, util  # noqa: E402

# segmentation label a landmark must lie on (est_lands_csv.py:57-74: 1/2 = left/right hemipelvis, 5/6 = left/right femur)
SEG_LABEL_FOR_LAND = {'FH-l': 5, 'FH-r': 6, 'GSN-l': 1, 'GSN-r': 2, 'IOF-l': 1, 'IOF-r': 2, 'MOF-l': 1, 'MOF-r': 2,
                      'SPS-l': 1, 'SPS-r': 2, 'IPS-l': 1, 'IPS-r': 2, 'ASIS-l': 1, 'ASIS-r': 2, 'PSIS-l': 1, 'PSIS-r': 2,
                      'PIIS-l': 1, 'PIIS-r': 2}


def build_parser():
    p = argparse.ArgumentParser(description='estimate landmark locations and write to CSV',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('heat_file_path', type=str, help='Path to dataset file containing labelings.')
    p.add_argument('heats_group_path', type=str, help='H5 group path to heat maps')
    p.add_argument('--out', type=str, default='yy_lands_est.csv', help='output image path')
    p.add_argument('--pat', type=int, help='patient index')
    p.add_argument('--use-seg', type=str, default='', help='Path to segmentation dataset used to assist in detection')
    p.add_argument('--no-hdr', action='store_true', help='No CSV header')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    land_names = dataset.get_land_names_from_dataset(args.heat_file_path)
    labels = [SEG_LABEL_FOR_LAND[n] for n in land_names]          # an unknown name is an error, as in the reference
    print('reading heatmaps...')
    get, close = dataset._open_container(args.heat_file_path)
    heats = torch.from_numpy(np.asarray(get(args.heats_group_path), dtype=np.float32))
    segs = torch.from_numpy(np.asarray(get(args.use_seg))) if args.use_seg else None
    close()
    dev = dfl_amd.get_device()
    print('detecting landmark locations...')
    torch.cuda.synchronize()
    t0 = time.time()
    rc = util.est_lands(heats.to(dev), None if segs is None else segs.to(dev), labels if segs is not None else None)
    rc = rc.cpu()
    each = (time.time() - t0) / max(rc.shape[0] * rc.shape[1], 1)
    with open(args.out, 'w') as csv_out:
        if not args.no_hdr:
            csv_out.write('pat,proj,land,row,col,time\n')
        for i in range(rc.shape[0]):
            for l in range(rc.shape[1]):
                csv_out.write('{},{},{},{},{},{:3f}\n'.format(args.pat, i, l, int(rc[i, l, 0]), int(rc[i, l, 1]), each))


if __name__ == '__main__':
    main()
