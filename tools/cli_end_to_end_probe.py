#!/usr/bin/env python3
"""Where the time of a whole command-line run goes (python -m dsen2_amd.cli in.npz out.npz --run_60 on a synthetic raster):
reading the input, the two super-resolution calls, assembling and writing the output — cProfile of cli.main, top entries.

    python tools/cli_end_to_end_probe.py [--size 3000]
"""
import argparse
import contextlib
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsen2_amd import cli, weights        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=3000)
ap.add_argument('--precision', default='fp32')
args = ap.parse_args()
n = args.size - args.size % 6
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(0)
np.savez(os.path.join(tmp, 'in.npz'), data10=rng.integers(35, 9000, size=(n, n, 4), dtype=np.uint16),
         data20=rng.integers(35, 9000, size=(n // 2, n // 2, 6), dtype=np.uint16),
         data60=rng.integers(35, 9000, size=(n // 6, n // 6, 2), dtype=np.uint16))
np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11))
np.save(os.path.join(tmp, 's2_030_lr_1e-05.npy'), weights.random_he_uniform(12, 2, 6, 128, seed=12))
argv = [os.path.join(tmp, 'in.npz'), os.path.join(tmp, 'out.npz'), '--run_60', '--models', tmp + os.sep, '--precision', args.precision]
with contextlib.redirect_stdout(io.StringIO()):
    cli.main([os.path.join(tmp, 'in.npz'), os.path.join(tmp, 'warm.npz'), '--roi_x_y', '0,0,479,479', '--models', tmp + os.sep,
              '--precision', args.precision])                       # library load, model build
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    cli.main(argv)
    pr.disable()
total = time.perf_counter() - t0
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22)
print('cli.main on a %d^2 raster, %s: %.3f s' % (n, args.precision, total))
print('\n'.join(ln[:170] for ln in s.getvalue().splitlines()[4:34]))
