#!/usr/bin/env python3
"""Race / hazard screen for the body convolution: many launches on random data, every output element compared
with the one-tile-per-workgroup reference structure (variant 0), several batch sizes (1..8 items per workgroup)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

flat = W.random_he_uniform(10, 6, 6, 128, seed=1, bias_scale=0.05)
models = {}
TEST_VARIANT = int(os.environ.get('DSEN2_STRESS_VARIANT', '14'))
for v in (0, TEST_VARIANT):
    _lib.call('dsen2_set_tuning', 0, v)
    models[v] = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128)
    models[v].set_weights_flat(flat)
_lib.call('dsen2_set_tuning', 0, 14)
bad_total = 0
MULT = int(os.environ.get('DSEN2_STRESS_REPS', '1'))
SHAPES = [(3, 32, 32, 6), (64, 32, 32, 6), (65, 32, 32, 6), (200, 32, 32, 6), (512, 32, 32, 6), (1, 1, 1, 2), (2, 5, 70, 2),
          (5, 128, 128, 3), (2, 192, 192, 2), (7, 21, 37, 3), (40, 50, 17, 3), (1, 16, 16, 3), (300, 16, 16, 3)]
for B, HH, WW, REPS in SHAPES:
    for rep in range(REPS * MULT):
        a = torch.randn((B, HH, WW, 128), device='cuda'); r = torch.randn((B, HH, WW, 128), device='cuda')
        for layer in (1, 2):
            o0 = torch.empty_like(a); o4 = torch.empty_like(a)
            models[0].time_body_conv(layer, a, r if layer == 2 else None, o0, iters=1)
            models[TEST_VARIANT].time_body_conv(layer, a, r if layer == 2 else None, o4, iters=1)
            nbad = int(((o4 - o0).abs() > 1e-4).sum())
            bad_total += nbad
            if nbad:
                print('MISMATCH B=%d %dx%d rep=%d layer=%d: %d elements' % (B, HH, WW, rep, layer, nbad))
print('stress: total mismatching elements = %d' % bad_total)
sys.exit(1 if bad_total else 0)
