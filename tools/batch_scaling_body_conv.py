#!/usr/bin/env python3
"""bf16 body convolution, F=256: ms per launch and per 256 patches at batch 256 / 1024 / 2048 — how much of a launch
is launch overhead and tail (it amortises with the batch), for whatever library DSEN2_HIP_LIB points at.

    [DSEN2_HIP_LIB=build/lib_x.so] python tools/batch_scaling_body_conv.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

F, H, D = 256, 32, 3
m = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision='bf16')
m.set_weights_flat(W.random_he_uniform(10, 6, D, F, seed=1))
for B in (256, 1024, 2048):
    a = torch.randn((B, H, H, F), device='cuda').to(torch.bfloat16)
    r = torch.randn((B, H, H, F), device='cuda')
    o = torch.empty_like(r)
    res = {}
    for name, layer in (('convA', 1), ('convB', 2)):
        best = 1e9
        for _ in range(4):
            best = min(best, m.time_body_conv(layer, a, r if layer == 2 else None, o, iters=20))
        res[name + '_ms'] = round(best, 4)
        res[name + '_ms_per_256'] = round(best * 256 / B, 4)
    print(json.dumps({'lib': os.environ.get('DSEN2_HIP_LIB', 'product'), 'batch': B, **res}), flush=True)
    del a, r, o
