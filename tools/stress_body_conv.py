#!/usr/bin/env python3
"""Race / hazard screen for the fp32 body convolution: many launches on random data, every output element of the
persistent DMA-fed kernel (conv3x3_body32.hip, through dsen2_conv3x3_nhwc) compared BIT FOR BIT with the
one-tile-per-workgroup structure (dsen2_conv3x3_nhwc_ref), both epilogues, 1..many items per workgroup, full /
ragged / tiny shapes.  DSEN2_STRESS_REPS scales the number of repetitions (default 1)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd.DSen2Net import conv3x3_nhwc            # noqa: E402

MULT = int(os.environ.get('DSEN2_STRESS_REPS', '1'))
rng = np.random.default_rng(1)
bad_total = 0
for F in (128, 256):
    k = (rng.standard_normal((3, 3, F, F)) * np.sqrt(2.0 / (9 * F))).astype(np.float32)
    b = (rng.standard_normal(F) * 0.05).astype(np.float32)
    SHAPES = [(3, 32, 32, 6), (64, 32, 32, 6), (65, 32, 32, 6), (200, 32, 32, 6), (512, 32, 32, 6), (1, 1, 1, 2), (2, 5, 70, 2),
              (5, 128, 128, 3), (2, 192, 192, 2), (7, 21, 37, 3), (40, 50, 17, 3), (1, 16, 16, 3), (300, 16, 16, 3)]
    if F == 256:
        SHAPES = [(3, 32, 32, 3), (130, 32, 32, 3), (7, 21, 37, 2), (2, 128, 128, 2)]
    for B, HH, WW, REPS in SHAPES:
        for rep in range(REPS * MULT):
            a = torch.randn((B, HH, WW, F), device='cuda')
            r = torch.randn((B, HH, WW, F), device='cuda')
            for epi in (0, 1):
                aux = r if epi == 1 else None
                o_ref = conv3x3_nhwc(a, k, b, epilogue=epi, aux=aux, ref=True)
                o_new = conv3x3_nhwc(a, k, b, epilogue=epi, aux=aux)
                nbad = int((o_new != o_ref).sum())
                bad_total += nbad
                if nbad:
                    print('MISMATCH F=%d B=%d %dx%d rep=%d epilogue=%d: %d elements' % (F, B, HH, WW, rep, epi, nbad))
print('stress: total mismatching elements = %d' % bad_total)
sys.exit(1 if bad_total else 0)
