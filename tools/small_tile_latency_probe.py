#!/usr/bin/env python3
"""Latency of the drop-in call on a SMALL raster — the reference's own demo workload (testing/demoDSen2.py on a 600 x 600 tile:
36 patches of 128^2 for DSen2_20, 16 of 192^2 for DSen2_60) — against the pure kernel time of its patches, per precision: what
the host side (uploads, launches, synchronisations, the download) adds when the GPU work is only a few milliseconds.

    python tools/small_tile_latency_probe.py [--reps 20]
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsen2_amd import supres, weights          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--tile', default='T33UUB')
ap.add_argument('--pinned-min-bytes', type=int, default=-1, help='override supres.PINNED_OUTPUT_MIN_BYTES (A/B of the page-locked download threshold)')
args = ap.parse_args()
if args.pinned_min_bytes >= 0:
    supres.PINNED_OUTPUT_MIN_BYTES = args.pinned_min_bytes
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'tile_%s_600.npz' % args.tile))
d10, d20, d60 = g['d10'], g['d20'], g['d60']
tmp = tempfile.mkdtemp()
np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11))
np.save(os.path.join(tmp, 's2_030_lr_1e-05.npy'), weights.random_he_uniform(12, 2, 6, 128, seed=12))
supres.MDL_PATH = os.path.join(tmp, '')
for prec in ('fp32', 'bf16x3', 'bf16'):
    supres.PRECISION = prec
    supres.clear_model_cache()
    for name, fn, a, patch, n in (('DSen2_20', supres.DSen2_20, (d10, d20), 128, 36), ('DSen2_60', supres.DSen2_60, (d10, d20, d60), 192, 16)):
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(3):
                fn(*a)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                fn(*a)
                ts.append(time.perf_counter() - t0)
        # the network alone on as many patches of that size, resident, back to back
        from dsen2_amd.DSen2Net import s2model
        shape = ((4, None, None), (6, None, None)) + (((2, None, None),) if name == 'DSen2_60' else ())
        m = s2model(shape, num_layers=6, feature_size=128, precision=prec)
        m.set_weights_flat(weights.random_he_uniform(sum(s[0] for s in shape), shape[-1][0], 6, 128, seed=3))
        xs = [torch.rand((n, s[0], patch, patch), device='cuda') for s in shape]
        out = torch.empty((n, shape[-1][0], patch, patch), device='cuda')
        for _ in range(5):
            m.forward_device(xs, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m.forward_device(xs, out=out)
        torch.cuda.synchronize()
        net = (time.perf_counter() - t0) / 20
        print(json.dumps({'pinned_min_bytes': supres.PINNED_OUTPUT_MIN_BYTES, 'call': name, 'precision': prec, 'tile': [600, 600], 'patches': n, 'call_ms_median': round(float(np.median(ts)) * 1e3, 2),
                          'call_ms_min': round(min(ts) * 1e3, 2), 'network_alone_ms': round(net * 1e3, 2),
                          'host_side_adds_ms': round((float(np.median(ts)) - net) * 1e3, 2)}), flush=True)
