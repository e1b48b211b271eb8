#!/usr/bin/env python3
"""Counterpart of the reference's testing/demoDSen2.py (its `readh5` / `RMSE` harness, :14-35) for this repo.

The reference demo needs the trained checkpoints and ground-truth tiles, all stripped from its checkout
(.MISSING_LARGE_BLOBS); what can be demonstrated here is the same flow on the tiles the reference DOES ship
(Copernicus Sentinel data, CC BY 4.0, committed as uint16 under tests/golden/): --tile crop (default: a 264x264 crop of
T33UUB, seconds), --tile T33UUB or --tile T49JGM (the whole 600x600 tiles, as testing/demoDSen2.py:42-43,67-68 reads
them from data/*.mat; about a minute of float64 oracle), or --tile FILE (.npz / .mat, via dsen2_amd.cli._load), with either
  * --models DIR : real checkpoints (keras .hdf5, or converted .npy) -> super-resolved bands, or
  * default      : seeded random-init weights, compared against the float64 oracle pipeline so the printed
                   RMSE is a parity figure, printed in the reference's format ("RMSE: %.4f").
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def RMSE(x1, x2):
    # testing/demoDSen2.py:31-35
    diff = x1.astype(np.float64) - x2.astype(np.float64)
    rms = np.sqrt(np.mean(np.power(diff, 2)))
    print('RMSE: {:.4f}'.format(rms))
    return rms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--models', default=None, help='directory with s2_032/s2_030 checkpoints (.hdf5 or .npy)')
    ap.add_argument('--no-oracle', action='store_true')
    ap.add_argument('--tile', default='crop', help='crop | T33UUB | T49JGM | a .npz / .mat file')
    args = ap.parse_args()

    from dsen2_amd import cli, supres, weights
    committed = {'crop': 'tile_T33UUB_crop.npz', 'T33UUB': 'tile_T33UUB_600.npz', 'T49JGM': 'tile_T49JGM_600.npz'}
    if args.tile in committed:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', committed[args.tile]))
        d10, d20, d60 = (g[k].astype(np.float32) for k in ('d10', 'd20', 'd60'))
    else:
        d10, d20, d60 = (np.asarray(a, np.float32) for a in cli._load(args.tile))        # readh5, testing/demoDSen2.py:14-28
    print('tile %s: im10 %s im20 %s im60 %s' % (args.tile, d10.shape, d20.shape, d60.shape))

    tmp = None
    if args.models:
        supres.MDL_PATH = os.path.join(args.models, '')
    else:
        tmp = tempfile.mkdtemp()
        np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11, bias_scale=0.02))
        np.save(os.path.join(tmp, 's2_030_lr_1e-05.npy'), weights.random_he_uniform(12, 2, 6, 128, seed=12, bias_scale=0.02))
        supres.MDL_PATH = os.path.join(tmp, '')
        print('no checkpoints given: using seeded random-init weights (%s)' % tmp)

    print('Super-resolving the 60m data into 10m bands')              # s2_tiles_supres.py:333
    sr60 = supres.DSen2_60(d10, d20, d60, deep=False)
    print('Super-resolving the 20m data into 10m bands')              # s2_tiles_supres.py:339
    sr20 = supres.DSen2_20(d10, d20, deep=False)
    print('sr20', sr20.shape, sr20.dtype, 'sr60', sr60.shape, sr60.dtype)

    if not args.models and not args.no_oracle:
        from oracle import c_oracle, patches_oracle as po               # checker only
        with contextlib.redirect_stdout(io.StringIO()):
            p10, p20 = po.get_test_patches(d10, d20, patchSize=128, border=8, f32_coords=True)
            pred = c_oracle.forward([p10 / np.float32(2000), p20 / np.float32(2000)],
                                    np.load(os.path.join(tmp, 's2_032_lr_1e-04.npy')), 6, 128)
            ref20 = po.recompose_images(pred, border=8, size=d10.shape).astype(np.float64) * 2000
        print('DSen2_20 vs float64 oracle pipeline (raw reflectance units; /2000 for the normalised domain):')
        r = RMSE(sr20, ref20)
        print('normalised RMSE: %.3e (gate 1e-4)' % (r / 2000))


if __name__ == '__main__':
    main()
