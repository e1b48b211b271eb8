#!/usr/bin/env python3
"""Body convolution: ms per launch and per base batch at 1x / 4x / 8x the bench batch — how much of a launch is launch
overhead, prologue and tail (it amortises with the batch), for whatever library DSEN2_HIP_LIB points at.

    [DSEN2_HIP_LIB=build/lib_x.so] python tools/batch_scaling_body_conv.py [bf16|fp32]
bf16: F=256, base batch 256 (BASELINE configs[4]); fp32: F=128, base batch 512 (configs[1]).
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
F, H, D, BASE = (256, 32, 3, 256) if mode == 'bf16' else (128, 32, 3, 512)
m = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision=mode)
m.set_weights_flat(W.random_he_uniform(10, 6, D, F, seed=1))
for B in (BASE, 4 * BASE, 8 * BASE):
    a = torch.randn((B, H, H, F), device='cuda')
    if mode == 'bf16':
        a = a.to(torch.bfloat16)
    r = torch.randn((B, H, H, F), device='cuda')
    o = torch.empty_like(r)
    res = {}
    for name, layer in (('convA', 1), ('convB', 2)):
        best = 1e9
        for _ in range(4):
            best = min(best, m.time_body_conv(layer, a, r if layer == 2 else None, o, iters=20))
        res[name + '_ms'] = round(best, 4)
        res[name + '_ms_per_base'] = round(best * BASE / B, 4)
    print(json.dumps({'lib': os.environ.get('DSEN2_HIP_LIB', 'product'), 'mode': mode, 'batch': B, **res}), flush=True)
    del a, r, o
