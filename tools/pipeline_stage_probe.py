#!/usr/bin/env python3
"""Durations of the stages of supres._run INSIDE the running full-tile pipeline (no profiler): HIP events recorded around every
gather / up-sampling / forward / recomposition call of DSen2_20 on a 10980^2 raster, read after the call has returned.

    python tools/pipeline_stage_probe.py [--precision bf16] [--banded 0|1]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsen2_amd import patches, supres, weights          # noqa: E402
from dsen2_amd.DSen2Net import S2Model                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='bf16')
ap.add_argument('--banded', default='1')
ap.add_argument('--size', type=int, default=10980)
args = ap.parse_args()
os.environ['DSEN2_BANDED_OUTPUT'] = args.banded
supres.PRECISION = args.precision
tmp = tempfile.mkdtemp()
np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11))
supres.MDL_PATH = os.path.join(tmp, '')
n = args.size - args.size % 6
rng = np.random.default_rng(0)
d10 = rng.integers(35, 13110, size=(n, n, 4), dtype=np.uint16)
d20 = rng.integers(35, 13110, size=(n // 2, n // 2, 6), dtype=np.uint16)
with contextlib.redirect_stdout(io.StringIO()):
    supres.DSen2_20(d10[:240, :240], d20[:120, :120])
    supres.DSen2_20(d10, d20)                                   # buffers exist from here on
records = []


def wrap(obj, name, tag):
    fn = getattr(obj, name)

    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        records.append((tag(*a, **k) if callable(tag) else tag, e0, e1))
        return out
    setattr(obj, name, timed)


wrap(patches, 'gather_patches_device', lambda img, org, s, *a, **k: 'gather x%d' % s)
wrap(patches, 'interp_patches_device', 'upsample')
wrap(patches, 'recompose_rows_device', 'recompose rows')
wrap(S2Model, 'forward_device', 'forward')
with contextlib.redirect_stdout(io.StringIO()):
    supres.DSen2_20(d10, d20)
torch.cuda.synchronize()
by = {}
for tag, e0, e1 in records:
    by.setdefault(tag, []).append(e0.elapsed_time(e1))
for tag, ms in by.items():
    print(json.dumps({'precision': args.precision, 'banded': args.banded, 'stage': tag, 'calls': len(ms), 'median_ms': round(float(np.median(ms)), 4),
                      'max_ms': round(max(ms), 4), 'total_ms': round(sum(ms), 2)}))
