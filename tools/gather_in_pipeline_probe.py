#!/usr/bin/env python3
"""tile_gather on a FULL-SIZE raster, one batch of the tile path at a time (344 patches of 128^2 out of 9801), standalone with
events — against its duration inside the pipeline in the rocprofv3 trace of the full tile (profiles/r04_w_*: 790 us per launch).

    python tools/gather_in_pipeline_probe.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import patches as P        # noqa: E402

dev = P.default_device()
n = 10980
d10 = torch.rand((n, n, 4), device=dev) * 10000
org, n_alloc = P.tile_origins((n // 2, n // 2), 64, 4)
org_dev = torch.from_numpy(np.ascontiguousarray((org * 2).astype(np.int32))).to(dev)
used = org.shape[0]


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for count in (344, 1024, used):
    for first in (0, 4000, used - count):
        ms = timeit(lambda: P.gather_patches_device(d10, org, 2, 8, 128, n_alloc, divisor=2000.0, first=first, count=count,
                                                    origins_dev=org_dev))
        byt = count * 4 * 128 * 128 * 4 * 2
        print(json.dumps({'patches': count, 'first': first, 'ms': round(ms, 4), 'GB_per_s': round(byt / ms / 1e6, 1)}), flush=True)
        if count == used:
            break

# every batch ONCE, in tile order, each timed alone: its source rows have not been touched since the image was written
print('--- one pass over the raster, 344 patches per launch, each launch once (cold source rows)')
d10b = torch.rand((n, n, 4), device=dev) * 10000
torch.cuda.synchronize()
evs = []
for first in range(0, used, 344):
    count = min(344, used - first)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    P.gather_patches_device(d10b, org, 2, 8, 128, n_alloc, divisor=2000.0, first=first, count=count, origins_dev=org_dev)
    e1.record()
    evs.append((first, count, e0, e1))
torch.cuda.synchronize()
ms = [e0.elapsed_time(e1) for _, _, e0, e1 in evs]
print(json.dumps({'launches': len(ms), 'ms_first_5': [round(x, 4) for x in ms[:5]], 'ms_median': round(float(np.median(ms)), 4),
                  'ms_max': round(max(ms), 4), 'total_ms': round(sum(ms), 3)}))
# the same with the image as supres builds it: uint16 uploaded from the host, widened on the GPU
h10 = np.random.default_rng(0).integers(35, 13110, size=(n, n, 4), dtype=np.uint16)
d10c = P._to_device_f32(h10, dev)
torch.cuda.synchronize()
evs = []
for first in range(0, used, 344):
    count = min(344, used - first)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    P.gather_patches_device(d10c, org, 2, 8, 128, n_alloc, divisor=2000.0, first=first, count=count, origins_dev=org_dev)
    e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
ms = [e0.elapsed_time(e1) for e0, e1 in evs]
print(json.dumps({'image': 'uploaded uint16, widened', 'ms_first_5': [round(x, 4) for x in ms[:5]], 'ms_median': round(float(np.median(ms)), 4),
                  'ms_max': round(max(ms), 4), 'total_ms': round(sum(ms), 3)}))
