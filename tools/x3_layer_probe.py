#!/usr/bin/env python3
"""One bf16x3 residual-block convolution of each kind on FIXED dense random operands, many launches (for rocprofv3 kernel traces
of two builds of the library: timing-only variants must be compared on the same operand data — in the whole network a variant
that computes garbage feeds it forward, the matrix pipe draws less power on degenerate data and the clock goes up).
    rocprofv3 --kernel-trace --stats ... -- python3 tools/x3_layer_probe.py [n] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd.DSen2Net import conv3x3_body_bf16x3, split3_f32      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
F = 128
rng = np.random.default_rng(0)
k = (rng.standard_normal((3, 3, F, F)) * np.sqrt(2.0 / (9 * F))).astype(np.float32)
b = (rng.standard_normal(F) * 0.05).astype(np.float32)
a = torch.randn((n, 32, 32, F), device='cuda')
r = torch.randn((n, 32, 32, F), device='cuda')
ax, _ = split3_f32(a)
hx, lo = split3_f32(r)
for _ in range(reps):
    conv3x3_body_bf16x3(ax, k, b, epilogue=0)
for _ in range(reps):
    conv3x3_body_bf16x3(ax, k, b, epilogue=3, res_hx=hx, res_lo=lo, res_scale=0.1)
torch.cuda.synchronize()
print('done')
