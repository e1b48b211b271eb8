#!/usr/bin/env python3
"""HBM roofline of the patch kernels (tiling / up-sampling / recomposition, utils/patches.py counterparts).

Algorithmic bytes per element (float32): tile_gather 4 in + 4 out; upsample 4*(1/s^2) in + 4 out; recompose
4*(inner/P)^2.. in (only the kept interior is read) + 4 out.  Timed with events on the launch stream.
    python tools/bench_patch_ops.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import patches as P        # noqa: E402

PEAK = 8000.0   # GB/s, HBM3E spec (MI355X_MICROARCH.md)


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


dev = P.default_device()
n = 5490                                         # half a full tile edge at 10 m -> 2401 patches of 128
d10 = torch.rand((n, n, 4), device=dev) * 10000
d20 = torch.rand((n // 2, n // 2, 6), device=dev) * 10000
org, n_alloc = P.tile_origins(d20.shape, 64, 4)
used = org.shape[0]
res = []

ms = timeit(lambda: P.gather_patches_device(d10, org, 2, 8, 128, n_alloc, divisor=2000.0, first=0, count=used))
byt = used * 4 * 128 * 128 * 4 * 2
res.append(('tile_gather 10m (4 bands, 128^2, /2000)', ms, byt))
lr = P.gather_patches_device(d20, org, 1, 4, 64, n_alloc, first=0, count=used)
ms = timeit(lambda: P.gather_patches_device(d20, org, 1, 4, 64, n_alloc, first=0, count=used))
res.append(('tile_gather 20m (6 bands, 64^2)', ms, used * 6 * 64 * 64 * 4 * 2))
ms = timeit(lambda: P.interp_patches_device(lr, (128, 128), post_divisor=2000.0))
res.append(('upsample x2 (6 bands, 64^2 -> 128^2, /2000)', ms, used * 6 * (64 * 64 + 128 * 128) * 4))
lr60 = torch.rand((1024, 2, 32, 32), device=dev) * 10000
ms = timeit(lambda: P.interp_patches_device(lr60, (192, 192), post_divisor=2000.0))
res.append(('upsample x6 (2 bands, 32^2 -> 192^2, /2000)', ms, 1024 * 2 * (32 * 32 + 192 * 192) * 4))
pred = torch.rand((used, 6, 128, 128), device=dev)
ms = timeit(lambda: P.recompose_device(pred, 8, (n, n), scale=2000.0))
res.append(('recompose (6 bands, 128^2/8 -> %dx%d, *2000)' % (n, n), ms, n * n * 6 * 4 * 2))
out = []
for name, ms, byt in res:
    gbs = byt / ms / 1e6
    out.append({'kernel': name, 'ms': round(ms, 4), 'algorithmic_MB': round(byt / 1e6, 1), 'GB_per_s': round(gbs, 1),
                'frac_of_8TBps': round(gbs / PEAK, 3)})
    print(json.dumps(out[-1]))
