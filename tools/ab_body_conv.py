#!/usr/bin/env python3
"""A/B the fp32 body-convolution structures in ONE process, interleaved rounds (guide §5.4 rule 24).
Needs the DIAGNOSTIC library (the product library has no structure switch):

    python -m dsen2_amd.build --diag
    DSEN2_HIP_LIB=build/libdsen2_hip_diag.so python tools/ab_body_conv.py [--variants 0,11,12,13,14] [--rounds 5] [--batch 512]

14 = default (conv3x3_body32.hip: deferred epilogue + wave-group stagger), 11-13 its sub-variants, 0 = one tile per
workgroup (conv3x3_mfma.hip).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--variants', default='0,14')
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--batch', type=int, default=512)
ap.add_argument('--hw', type=int, default=32)
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(',')]

flat = W.random_he_uniform(10, 6, 6, 128, seed=1, bias_scale=0.05)
models = {}
for v in variants:
    _lib.diag_set(0, v)
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128)
    m.set_weights_flat(flat)
    models[v] = m
_lib.diag_set(0, 14)

B, H = args.batch, args.hw
a = torch.randn((B, H, H, 128), device='cuda')
r = torch.randn((B, H, H, 128), device='cuda')
outs = {}
for v, m in models.items():
    o = torch.empty_like(a)
    m.time_body_conv(2, a, r, o, iters=1)
    outs[v] = o
base = outs[variants[0]]
for v in variants[1:]:
    d = (outs[v] - base).abs().max().item()
    print('variant %d max|diff| vs variant %d: %.3e' % (v, variants[0], d))
    assert d < 1e-4

flops = B * H * H * 2 * 9 * 128 * 128
times = {v: {'relu': [], 'res': []} for v in variants}
o = torch.empty_like(a)
for _ in range(args.rounds):
    for v, m in models.items():
        times[v]['relu'].append(m.time_body_conv(1, a, None, o, iters=args.iters))
        times[v]['res'].append(m.time_body_conv(2, a, r, o, iters=args.iters))
xs = [torch.rand((B, 4, H, H), device='cuda') * 5, torch.rand((B, 6, H, H), device='cuda') * 5]
fwd = {v: [] for v in variants}
for _ in range(args.rounds):
    for v, m in models.items():
        m.forward_device(xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.forward_device(xs)
        e1.record()
        torch.cuda.synchronize()
        fwd[v].append(e0.elapsed_time(e1) / 5)
for v in variants:
    tr, ts = np.array(times[v]['relu']), np.array(times[v]['res'])
    print(json.dumps({'variant': v, 'relu_ms_med': round(float(np.median(tr)), 4), 'relu_ms_min': round(float(tr.min()), 4),
                      'res_ms_med': round(float(np.median(ts)), 4), 'res_ms_min': round(float(ts.min()), 4),
                      'tflops_med': round(flops / np.median(np.concatenate([tr, ts])) / 1e9, 2),
                      'fwd_ms_med': round(float(np.median(fwd[v])), 3),
                      'patches_per_s': round(B * (H * H / 1024.0) / float(np.median(fwd[v])) * 1e3, 1)}))
