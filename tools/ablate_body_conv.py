#!/usr/bin/env python3
"""Timing-only ablations of the persistent body kernels (guide §7 'The diagnostic loop', step 2).
Masks: 1 no stores, 2 no residual loads, 4 no weight stream, 8 no input stream, 16 no barriers (outputs are wrong
by construction).  Needs the DIAGNOSTIC library:

    python -m dsen2_amd.build --diag
    DSEN2_HIP_LIB=build/libdsen2_hip_diag.so python tools/ablate_body_conv.py [fp32|bf16]

fp32: conv3x3_body32.hip at F=128, batch 512; bf16: conv3x3_body16w.hip at F=256, batch 256 (32x32 patches).
A model copies the mask when it is created, so there is one model per mask."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

bf = len(sys.argv) > 1 and sys.argv[1] == 'bf16'
F, B, H, D = (256, 256, 32, 3) if bf else (128, 512, 32, 2)
flat = W.random_he_uniform(10, 6, D, F, seed=1)
masks = ([int(m) for m in os.environ['ABLATE_MASKS'].split(',')] if os.environ.get('ABLATE_MASKS') else [0, 1, 3, 4, 8, 12, 15, 16, 31]) if bf else [0, 1, 2, 3, 4, 8, 12, 15, 16, 31]
models = {}
for k in masks:
    _lib.diag_set(1, k)
    models[k] = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision='bf16' if bf else 'fp32')
    models[k].set_weights_flat(flat)
_lib.diag_set(1, 0)
a = torch.randn((B, H, H, F), device='cuda'); r = torch.randn((B, H, H, F), device='cuda'); o = torch.empty_like(a)
if bf:
    a = a.to(torch.bfloat16)
res = {k: {'relu': [], 'res': []} for k in masks}
for rnd in range(4):
    for k in masks:
        res[k]['relu'].append(models[k].time_body_conv(1, a, None, o, iters=10))
        res[k]['res'].append(models[k].time_body_conv(2, a, r, o, iters=10))
flops = B * H * H * 2 * 9 * F * F
for k in masks:
    tr, ts = float(np.median(res[k]['relu'])), float(np.median(res[k]['res']))
    print(json.dumps({'ablate': k, 'relu_ms': round(tr, 4), 'res_ms': round(ts, 4), 'relu_tflops': round(flops / tr / 1e9, 1),
                      'res_tflops': round(flops / ts / 1e9, 1)}))
