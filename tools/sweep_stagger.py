#!/usr/bin/env python3
"""Sweep the start-stagger quantum of the persistent body kernel (fp32 F=128 B=512 and bf16 F=256 B=256)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W
from dsen2_amd.DSen2Net import s2model
H = 32
cfgs = [('fp32', 128, 512), ('bf16', 256, 256)]
for prec, F, B in cfgs:
    m = s2model(((4, None, None), (6, None, None)), num_layers=2, feature_size=F, precision=prec)
    m.set_weights_flat(W.random_he_uniform(10, 6, 2, F, seed=1))
    a = torch.randn((B, H, H, F), device='cuda'); r = torch.randn((B, H, H, F), device='cuda')
    if prec == 'bf16':
        a = a.to(torch.bfloat16)
    o = torch.empty((B * 3 // 2 + 1, H, H, F), device='cuda')
    qs = [0, 1, 2, 3, 4, 6, 8]
    res = {q: {'relu': [], 'res': []} for q in qs}
    for rnd in range(4):
        for q in qs:
            _lib.call('dsen2_set_tuning', 3, q)
            res[q]['relu'].append(m.time_body_conv(1, a, None, o, iters=10))
            res[q]['res'].append(m.time_body_conv(2, a, r, o, iters=10))
    _lib.call('dsen2_set_tuning', 3, 0)
    for q in qs:
        print(json.dumps({'prec': prec, 'stagger': q, 'relu_ms': round(float(np.median(res[q]['relu'])), 4),
                          'res_ms': round(float(np.median(res[q]['res'])), 4)}))
