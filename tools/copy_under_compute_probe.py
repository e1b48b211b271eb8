#!/usr/bin/env python3
"""Does a host<->device copy on a second stream slow the body convolutions down?  (Both persistent kernels fill every CU, so a
copy done by a blit KERNEL would have to wait for — and then delay — their workgroups; a copy done by an SDMA engine would
not.)  Times `steps` forwards of the bench batch alone, then with a D2H / H2D of `mb` MB per forward running on a copy stream,
then with a device-to-device copy KERNEL of `mb` MB per forward on that stream (`d2d`: a stand-in for what RCCL's receive side
does on rank 0 at N = 8 — 7 x 12.6 = 88 MB per step written by copy workgroups while both body kernels fill every CU; round 5).
    python tools/copy_under_compute_probe.py [--precision fp32|bf16|bf16x3] [--mb 100]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import weights                      # noqa: E402
from dsen2_amd.DSen2Net import s2model             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='fp32')
ap.add_argument('--mb', type=int, default=100)
ap.add_argument('--steps', type=int, default=40)
args = ap.parse_args()
dev = torch.device('cuda', 0)
m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision=args.precision)
m.set_weights_flat(weights.random_he_uniform(10, 6, 6, 128, seed=1))
xs = [torch.rand((512, c, 32, 32), device=dev) * 5 for c in (4, 6)]
out = torch.empty((512, 6, 32, 32), device=dev)
n = args.mb * (1 << 20) // 4
dbuf = torch.empty(n, dtype=torch.float32, device=dev)
dbuf2 = torch.empty(n, dtype=torch.float32, device=dev)
hbuf = torch.empty(n, dtype=torch.float32, pin_memory=True)
copy = torch.cuda.Stream(dev)


def run(mode):
    for _ in range(5):
        m.forward_device(xs, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.forward_device(xs, out=out)
        if mode:
            with torch.cuda.stream(copy):
                if mode == 'd2h':
                    hbuf.copy_(dbuf, non_blocking=True)
                elif mode == 'd2d':
                    torch.add(dbuf, 0.0, out=dbuf2)         # an elementwise KERNEL (a plain copy_ may go to an SDMA engine)
                else:
                    dbuf.copy_(hbuf, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


res = {'precision': args.precision, 'copy_mb_per_forward': args.mb}
for mode in (None, 'd2h', None, 'h2d', None, 'd2d', None, 'd2d', None):
    res.setdefault(mode or 'alone', []).append(round(run(mode), 4))
t0 = time.perf_counter(); hbuf.copy_(dbuf, non_blocking=True); torch.cuda.synchronize()
res['d2h_alone_gbps'] = round(n * 4 / (time.perf_counter() - t0) / 1e9, 1)
print(json.dumps(res))
