#!/usr/bin/env python3
"""Race / hazard screen for the bf16 body convolution: many launches on random data, every output element of the
DMA-fed kernel (tuning key 4 = DSEN2_STRESS_VARIANT, default 4) compared bit for bit with the register-staged
structure 0, at both widths, full / ragged / tiny shapes and 1..many items per workgroup.
DSEN2_STRESS_REPS scales the number of repetitions (default 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

TEST_VARIANT = int(os.environ.get('DSEN2_STRESS_VARIANT', '4'))
MULT = int(os.environ.get('DSEN2_STRESS_REPS', '1'))
bad_total = 0
for F in (256, 128):
    flat = W.random_he_uniform(10, 6, 2, F, seed=1, bias_scale=0.05)
    models = {}
    for v in (0, TEST_VARIANT):
        _lib.call('dsen2_set_tuning', 4, v)
        models[v] = s2model(((4, None, None), (6, None, None)), num_layers=2, feature_size=F, precision='bf16')
        models[v].set_weights_flat(flat)
    _lib.call('dsen2_set_tuning', 4, 4)
    SHAPES = [(3, 32, 32, 6), (64, 32, 32, 6), (65, 32, 32, 4), (256, 32, 32, 6), (5, 128, 128, 3), (2, 192, 192, 2),
              (7, 21, 37, 3), (40, 50, 17, 3), (1, 16, 16, 3), (300, 16, 16, 3), (1, 1, 1, 2), (2, 5, 70, 2)]
    for B, HH, WW, REPS in SHAPES:
        for rep in range(REPS * MULT):
            a = torch.randn((B, HH, WW, F), device='cuda').to(torch.bfloat16)
            r = torch.randn((B, HH, WW, F), device='cuda')
            for layer in (1, 2):
                outs = []
                for v in (0, TEST_VARIANT):
                    o = torch.zeros((B * 3 // 2 + 1, HH, WW, F), device='cuda')
                    models[v].time_body_conv(layer, a, r if layer == 2 else None, o, iters=1)
                    outs.append(o)
                nbad = int((outs[0] != outs[1]).sum())
                bad_total += nbad
                if nbad:
                    print('MISMATCH F=%d B=%d %dx%d rep=%d layer=%d: %d elements' % (F, B, HH, WW, rep, layer, nbad))
    del models
print('stress (bf16): total mismatching elements = %d' % bad_total)
sys.exit(1 if bad_total else 0)
