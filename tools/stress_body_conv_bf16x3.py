#!/usr/bin/env python3
"""Race / hazard screen for the bf16x3 body convolution (conv3x3_body16w.hip, X3): many launches on random data at both
widths, full / ragged / tiny shapes and 1..many items per workgroup, all three epilogues.  Reference = the fp32
one-tile-per-workgroup kernel (dsen2_conv3x3_nhwc_ref) on the same fp32 operands: bf16x3 differs from it by its 2^-17 per
operand (~1e-5 of the output) and fp32 summation order — a corrupted lane, a stale LDS read, a wrong plane or a missed zero
pad is O(0.1).  Also checked bit for bit: the stream the in-place epilogue leaves (hi | xl, lo16) is exactly
split3(its fp32 form), and conv-A's two output planes are (RNE(v), RNE(v - hi)) of some v.  DSEN2_STRESS_REPS scales the
repetitions (default 1)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd.DSen2Net import conv3x3_body_bf16x3, conv3x3_nhwc, split3_f32      # noqa: E402

MULT = int(os.environ.get('DSEN2_STRESS_REPS', '1'))
TOL = 3e-4
rng = np.random.default_rng(3)


def planes_value(p):
    """int16 [n, 2, c/8, h, w, 8] -> fp32 NHWC hi + lo (exact in fp32: two bf16 numbers 8 binades apart at most... summed in f64)."""
    n, _, b, h, w, _ = p.shape
    f = (p.to(torch.int32) << 16).view(torch.float32)                   # [n, 2, b, h, w, 8]
    v = f[:, 0].double() + f[:, 1].double()
    return v.permute(0, 2, 3, 1, 4).reshape(n, h, w, b * 8)


bad_total = 0
for F in (128, 256):
    k = (rng.standard_normal((3, 3, F, F)) * np.sqrt(2.0 / (9 * F))).astype(np.float32)
    b = (rng.standard_normal(F) * 0.05).astype(np.float32)
    SHAPES = [(3, 32, 32, 4), (64, 32, 32, 4), (65, 32, 32, 3), (256, 32, 32, 4), (5, 128, 128, 2), (2, 192, 192, 1),
              (7, 21, 37, 3), (40, 50, 17, 2), (1, 16, 16, 3), (300, 16, 16, 3), (1, 1, 1, 2), (2, 5, 70, 2), (3, 16, 33, 2)]
    for B, HH, WW, REPS in SHAPES:
        for rep in range(REPS * MULT):
            a = torch.randn((B, HH, WW, F), device='cuda')
            r = torch.randn((B, HH, WW, F), device='cuda')
            ax, _ = split3_f32(a)
            # conv-A: two planes of relu(conv + b)
            ref = conv3x3_nhwc(a, k, b, epilogue=0, ref=True)
            t = conv3x3_body_bf16x3(ax, k, b, epilogue=0)
            got = planes_value(t)
            nbad = int(((got - ref.double()).abs() > TOL).sum())
            # conv-B: fp32 form, then in place on the stream
            ref = conv3x3_nhwc(a, k, b, epilogue=1, aux=r, res_scale=0.1, ref=True)
            hx, lo = split3_f32(r)
            got32 = conv3x3_body_bf16x3(ax, k, b, epilogue=3, res_hx=hx, res_lo=lo, res_scale=0.1)
            nbad += int(((got32 - ref).abs() > TOL).sum())
            conv3x3_body_bf16x3(ax, k, b, epilogue=1, res_hx=hx, res_lo=lo, res_scale=0.1)
            h2, l2 = split3_f32(got32)
            nbad += int((h2 != hx).sum()) + int((l2 != lo).sum())
            bad_total += nbad
            if nbad:
                print('MISMATCH F=%d B=%d %dx%d rep=%d: %d elements' % (F, B, HH, WW, rep, nbad))
print('stress (bf16x3): total mismatching elements = %d' % bad_total)
sys.exit(1 if bad_total else 0)
