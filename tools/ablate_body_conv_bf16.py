#!/usr/bin/env python3
"""Timing-only ablations of the bf16 body kernel (F=256, B=256): 1 no stores, 2 no residual loads, 4 no weight
stream, 8 no input stream, 16 no barriers.
The masks act on the DMA-fed kernels (conv3x3_body32.hip / conv3x3_body16.hip: the defaults); the register-staged
kernels ignore them."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W
from dsen2_amd.DSen2Net import s2model
m = s2model(((4, None, None), (6, None, None)), num_layers=2, feature_size=256, precision='bf16')
m.set_weights_flat(W.random_he_uniform(10, 6, 2, 256, seed=1))
B, H = 256, 32
a = torch.randn((B, H, H, 256), device='cuda').to(torch.bfloat16); r = torch.randn((B, H, H, 256), device='cuda')
o = torch.empty((B * 3 // 2 + 1, H, H, 256), device='cuda')
masks = [0, 1, 3, 4, 8, 12, 15, 16, 31]
res = {k: {'relu': [], 'res': []} for k in masks}
for rnd in range(4):
    for k in masks:
        _lib.call('dsen2_set_tuning', 1, k)
        res[k]['relu'].append(m.time_body_conv(1, a, None, o, iters=10))
        res[k]['res'].append(m.time_body_conv(2, a, r, o, iters=10))
_lib.call('dsen2_set_tuning', 1, 0)
flops = B * H * H * 2 * 9 * 256 * 256
for k in masks:
    tr, ts = float(np.median(res[k]['relu'])), float(np.median(res[k]['res']))
    print(json.dumps({'ablate': k, 'relu_ms': round(tr, 4), 'res_ms': round(ts, 4), 'relu_tflops': round(flops / tr / 1e9, 1),
                      'res_tflops': round(flops / ts / 1e9, 1)}))
