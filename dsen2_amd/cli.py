"""GDAL-free command line for the drop-in path: the super-resolution part of the reference's
testing/s2_tiles_supres.py (:332-342 the DSen2_60 / DSen2_20 calls, :383-420 band assembly and the npz writer)
with arrays read from a .npz (or MATLAB v7.3 .mat when h5py is importable) instead of a SAFE product.

    python -m dsen2_amd.cli INPUT OUTPUT.npz [--run_60] [--copy_original_bands] [--roi_x_y x1,y1,x2,y2]
                                            [--models DIR] [--save_prefix P] [--precision fp32|bf16] [--deep]

INPUT .npz keys: data10 [x,y,4] (B2,B3,B4,B8), data20 [x/2,y/2,6] (B5,B6,B7,B8A,B11,B12), data60 [x/6,y/6,2]
(B1,B9) — the arrays s2_tiles_supres.py:311-329 reads from GDAL; aliases d10/d20/d60 and im10/im20/im60 (the
keys of the reference's data/*.mat, CHW, transposed like testing/demoDSen2.py:14-28) are accepted.
OUTPUT: np.savez(output, bands={description: 2-D array}) exactly like the reference's npz fallback (:419-420),
band descriptions "SR" + name.  Users with GDAL keep using the reference's own script: it only needs
`from supres import DSen2_20, DSen2_60` to resolve to dsen2_amd.supres.
"""
from __future__ import division

import argparse
import os
import re
import sys

import numpy as np

BANDS10 = ['B4', 'B3', 'B2', 'B8']            # order of the 10 m sub-dataset in a SAFE product
BANDS20 = ['B5', 'B6', 'B7', 'B8A', 'B11', 'B12']
BANDS60 = ['B1', 'B9']


def _load(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npz':
        z = np.load(path)
        def pick(*names):
            for n in names:
                if n in z:
                    return z[n]
            return None
        return pick('data10', 'd10'), pick('data20', 'd20'), pick('data60', 'd60')
    if ext == '.mat':
        try:
            import h5py
        except ImportError as e:
            raise ImportError('reading .mat needs h5py; convert to .npz with keys data10/data20/data60') from e
        with h5py.File(path, 'r') as f:        # testing/demoDSen2.py:14-28 (readh5): CHW -> HWC
            get = lambda k: np.array(f[k]).transpose() if k in f else None
            return get('im10'), get('im20'), get('im60')
    raise ValueError('unsupported input %r (use .npz or .mat)' % path)


def main(argv=None):
    ap = argparse.ArgumentParser(description='Perform super-resolution on Sentinel-2 arrays with DSen2 on MI355X.')
    ap.add_argument('data_file')
    ap.add_argument('output_file', nargs='?')
    ap.add_argument('--roi_x_y', default='', help='x_1,y_1,x_2,y_2 on the 10m bands; extended to 60m pixel boundaries')
    ap.add_argument('--run_60', action='store_true', help='also super-resolve the 60m bands (B1,B9)')
    ap.add_argument('--copy_original_bands', action='store_true')
    ap.add_argument('--save_prefix', default='')
    ap.add_argument('--models', default=None, help='directory with the checkpoints (default: supres.MDL_PATH)')
    ap.add_argument('--deep', action='store_true', help='VDSen2 (d=32, F=256)')
    ap.add_argument('--precision', default=None, choices=['fp32', 'bf16'])
    args = ap.parse_args(argv)

    from . import supres
    if args.models:
        supres.MDL_PATH = os.path.join(args.models, '')
    if args.precision:
        supres.PRECISION = args.precision

    data10, data20, data60 = _load(args.data_file)
    if data10 is None or data20 is None:
        print('No super-resolution performed, exiting')          # s2_tiles_supres.py:346-348
        return 0
    if args.roi_x_y:
        x1, y1, x2, y2 = [int(float(v)) for v in re.split(',', args.roi_x_y)]
        xmin, xmax, ymin, ymax = min(x1, x2), max(x1, x2), min(y1, y2), max(y1, y2)
        # nearest 60 m pixel boundaries, as s2_tiles_supres.py:131-134
        xmin, ymin = int(xmin / 6) * 6, int(ymin / 6) * 6
        xmax, ymax = int((xmax + 1) / 6) * 6 - 1, int((ymax + 1) / 6) * 6 - 1
        data10 = data10[ymin:ymax + 1, xmin:xmax + 1]
        data20 = data20[ymin // 2:(ymax + 1) // 2, xmin // 2:(xmax + 1) // 2]
        if data60 is not None:
            data60 = data60[ymin // 6:(ymax + 1) // 6, xmin // 6:(xmax + 1) // 6]

    output_file = args.output_file or os.path.split(args.data_file)[1] + '.npz'
    output_file = args.save_prefix + output_file

    sr60 = None
    if args.run_60 and data60 is not None:
        print('Super-resolving the 60m data into 10m bands')
        sr60 = supres.DSen2_60(data10, data20, data60, deep=args.deep)
    print('Super-resolving the 20m data into 10m bands')
    sr20 = supres.DSen2_20(data10, data20, deep=args.deep)

    bands = dict()
    if sr60 is not None:
        sr = np.concatenate((sr20, sr60), axis=2)
        names = BANDS20 + BANDS60
    else:
        sr, names = sr20, BANDS20
    sys.stdout.write('Writing')
    if args.copy_original_bands:
        sys.stdout.write(' the original 10m bands and')
        for bi, bn in enumerate(BANDS10[:data10.shape[2]]):
            bands[bn] = data10[:, :, bi]
    print(' the super-resolved bands in %s' % output_file)
    for bi, bn in enumerate(names):
        bands['SR' + bn] = sr[:, :, bi]
    for desc in bands:
        print(desc)
    np.savez(output_file, bands=bands)
    return 0


if __name__ == '__main__':
    sys.exit(main())
