#!/usr/bin/env python3
"""Race / hazard screen for the bf16 body convolution (conv3x3_body16w.hip): many launches on random data at both
widths, full / ragged / tiny shapes and 1..many items per workgroup, all three epilogues.  Reference = the fp32
one-tile-per-workgroup kernel (dsen2_conv3x3_nhwc_ref) on the same bf16-rounded operands: every product is exact in
fp32, so the two differ only in fp32 summation order — a corrupted lane, a stale LDS read or a missed zero pad is
O(0.1).  Also checked bit for bit: the (hi, lo) planes the in-place epilogue leaves are exactly split(join(...)),
and the fp32 epilogue equals join of the in-place one.  DSEN2_STRESS_REPS scales the repetitions (default 1)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd.DSen2Net import conv3x3_body_bf16, conv3x3_nhwc, join_f32, split_f32      # noqa: E402

MULT = int(os.environ.get('DSEN2_STRESS_REPS', '1'))
TOL = 2e-4            # fp32 summation-order noise of 2304 products of O(1) x O(0.03) is ~1e-6
rng = np.random.default_rng(2)
bad_total = 0
for F in (256, 128):
    k = (rng.standard_normal((3, 3, F, F)) * np.sqrt(2.0 / (9 * F))).astype(np.float32)
    k = torch.from_numpy(k).to(torch.bfloat16).to(torch.float32).numpy()             # weights already bf16-exact
    b = (rng.standard_normal(F) * 0.05).astype(np.float32)
    SHAPES = [(3, 32, 32, 6), (64, 32, 32, 6), (65, 32, 32, 4), (256, 32, 32, 6), (5, 128, 128, 3), (2, 192, 192, 2),
              (7, 21, 37, 3), (40, 50, 17, 3), (1, 16, 16, 3), (300, 16, 16, 3), (1, 1, 1, 2), (2, 5, 70, 2), (3, 16, 33, 2)]
    for B, HH, WW, REPS in SHAPES:
        for rep in range(REPS * MULT):
            a = torch.randn((B, HH, WW, F), device='cuda').to(torch.bfloat16)
            a32 = a.to(torch.float32)
            r = torch.randn((B, HH, WW, F), device='cuda')
            # conv-A: bf16 output = RNE(relu(conv + b))
            ref = conv3x3_nhwc(a32, k, b, epilogue=0, ref=True)
            got = conv3x3_body_bf16(a, k, b, epilogue=0).to(torch.float32)
            nbad = int(((got - ref).abs() > TOL + ref.abs() * 2.0 ** -7).sum())
            # conv-B: exact fp32 residual stream on planes, in place; and the fp32 form of the last block
            ref = conv3x3_nhwc(a32, k, b, epilogue=1, aux=r, res_scale=0.1, ref=True)
            hi, lo = split_f32(r)
            assert torch.equal(join_f32(hi, lo), r)
            got32 = conv3x3_body_bf16(a, k, b, epilogue=3, res_hi=hi, res_lo=lo, res_scale=0.1)
            conv3x3_body_bf16(a, k, b, epilogue=1, res_hi=hi, res_lo=lo, res_scale=0.1)
            joined = join_f32(hi, lo)
            nbad += int(((got32 - ref).abs() > TOL).sum())
            nbad += int((joined != got32).sum())
            h2, l2 = split_f32(joined)
            nbad += int((h2 != hi).sum()) + int((l2 != lo).sum())
            bad_total += nbad
            if nbad:
                print('MISMATCH F=%d B=%d %dx%d rep=%d: %d elements' % (F, B, HH, WW, rep, nbad))
print('stress (bf16): total mismatching elements = %d' % bad_total)
sys.exit(1 if bad_total else 0)
