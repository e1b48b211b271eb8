#!/usr/bin/env python3
"""Timing-only ablations of the persistent body kernel (guide §7 'The diagnostic loop', step 2).
Masks: 1 no stores, 2 no residual loads, 4 no weight stream, 8 no input stream, 16 no barriers.
The masks act on the DMA-fed kernels (conv3x3_body32.hip / conv3x3_body16.hip: the defaults); the register-staged
kernels ignore them."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

flat = W.random_he_uniform(10, 6, 6, 128, seed=1)
m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128)
m.set_weights_flat(flat)
B, H = 512, 32
a = torch.randn((B, H, H, 128), device='cuda'); r = torch.randn((B, H, H, 128), device='cuda'); o = torch.empty_like(a)
masks = [0, 1, 2, 3, 4, 8, 12, 15, 16, 31]
res = {k: {'relu': [], 'res': []} for k in masks}
for rnd in range(4):
    for k in masks:
        _lib.call('dsen2_set_tuning', 1, k)
        res[k]['relu'].append(m.time_body_conv(1, a, None, o, iters=10))
        res[k]['res'].append(m.time_body_conv(2, a, r, o, iters=10))
_lib.call('dsen2_set_tuning', 1, 0)
flops = B * H * H * 2 * 9 * 128 * 128
for k in masks:
    tr, ts = float(np.median(res[k]['relu'])), float(np.median(res[k]['res']))
    print(json.dumps({'ablate': k, 'relu_ms': round(tr, 4), 'res_ms': round(ts, 4), 'relu_tflops': round(flops / tr / 1e9, 1),
                      'res_tflops': round(flops / ts / 1e9, 1)}))
