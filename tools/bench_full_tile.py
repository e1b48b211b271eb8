#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU: DSen2_20 + DSen2_60 over a synthetic full-size Sentinel-2 tile
(10980 x 10980 @10 m), end to end through the drop-in surface (host ndarray in, host ndarray out).

    python tools/bench_full_tile.py [--size 10980] [--skip60]
Prints one JSON line with wall times of the stages.  Random-init weights; the data is synthetic.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import supres, weights        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=10980)
ap.add_argument('--skip60', action='store_true')
args = ap.parse_args()

n = args.size - args.size % 6
rng = np.random.default_rng(0)
d10 = rng.integers(35, 13110, size=(n, n, 4), dtype=np.uint16)
d20 = rng.integers(35, 13110, size=(n // 2, n // 2, 6), dtype=np.uint16)
d60 = rng.integers(35, 13110, size=(n // 6, n // 6, 2), dtype=np.uint16)
tmp = tempfile.mkdtemp()
np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11))
np.save(os.path.join(tmp, 's2_030_lr_1e-05.npy'), weights.random_he_uniform(12, 2, 6, 128, seed=12))
supres.MDL_PATH = os.path.join(tmp, '')
out = {'tile': [n, n], 'data': 'synthetic', 'patches20': int(np.ceil(n / 112.0) ** 2), 'patches60': int(np.ceil(n / 168.0) ** 2)}


def timed(fn, *a):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        y = fn(*a)
    torch.cuda.synchronize()
    return y, time.perf_counter() - t0


with contextlib.redirect_stdout(io.StringIO()):
    supres.DSen2_20(d10[:240, :240], d20[:120, :120])       # warm-up: library load, model build, weight upload
y20, t20 = timed(supres.DSen2_20, d10, d20)
out['dsen2_20_s'] = round(t20, 3)
out['dsen2_20_patches_per_s_128'] = round(out['patches20'] / t20, 1)
out['dsen2_20_equiv_32x32_patches_per_s'] = round(out['patches20'] * 16 / t20, 1)
assert y20.shape == (n, n, 6) and np.isfinite(y20[::97, ::89]).all()
if not args.skip60:
    with contextlib.redirect_stdout(io.StringIO()):
        supres.DSen2_60(d10[:384, :384], d20[:192, :192], d60[:64, :64])
    y60, t60 = timed(supres.DSen2_60, d10, d20, d60)
    out['dsen2_60_s'] = round(t60, 3)
    out['dsen2_60_equiv_32x32_patches_per_s'] = round(out['patches60'] * 36 / t60, 1)
    assert y60.shape == (n, n, 2)
out['peak_gpu_mem_gib'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
print(json.dumps(out))
