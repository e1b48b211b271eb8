#!/usr/bin/env python3
"""A/B of the bf16 body-kernel structures (tuning key 4) at VDSen2 width: F=256, B=256, 32x32."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W
from dsen2_amd.DSen2Net import s2model
H, F, B = 32, 256, int(os.environ.get('AB_BATCH', 256))
flat = W.random_he_uniform(10, 6, 2, F, seed=1, bias_scale=0.05)
VARIANTS = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (2, 3, 4, 5, 6, 7)
ms = {}
for v in VARIANTS:
    _lib.call('dsen2_set_tuning', 4, v)
    ms[v] = s2model(((4, None, None), (6, None, None)), num_layers=2, feature_size=F, precision='bf16'); ms[v].set_weights_flat(flat)
_lib.call('dsen2_set_tuning', 4, 4)
a = torch.randn((B, H, H, F), device='cuda').to(torch.bfloat16); r = torch.randn((B, H, H, F), device='cuda')
outs = {}
for v in VARIANTS:
    o = torch.zeros((B * 3 // 2 + 1, H, H, F), device='cuda'); ms[v].time_body_conv(2, a, r, o, iters=1)
    outs[v] = [o[:B].clone(), o[B:B + B // 2].clone().view(torch.bfloat16).view(B, H, H, F).clone()]
    o.zero_(); ms[v].time_body_conv(1, a, None, o, iters=1)
    outs[v].append(o[:B // 2].clone().view(torch.bfloat16).view(B, H, H, F).clone())
v0 = VARIANTS[0]
for v in VARIANTS[1:]:
    print('variant %d vs %d: residual fp32 max diff %.3e, bf16 copy max diff %.3e, relu bf16 max diff %.3e' % (
        v, v0, (outs[v][0] - outs[v0][0]).abs().max().item(),
        (outs[v][1].float() - outs[v0][1].float()).abs().max().item(),
        (outs[v][2].float() - outs[v0][2].float()).abs().max().item()))
res = {v: {'relu': [], 'res': []} for v in VARIANTS}
o = torch.empty((B * 3 // 2 + 1, H, H, F), device='cuda')
for rnd in range(5):
    for v in VARIANTS:
        res[v]['relu'].append(ms[v].time_body_conv(1, a, None, o, iters=10))
        res[v]['res'].append(ms[v].time_body_conv(2, a, r, o, iters=10))
flops = B * H * H * 2 * 9 * F * F
for v in VARIANTS:
    tr, ts = float(np.median(res[v]['relu'])), float(np.median(res[v]['res']))
    print(json.dumps({'variant': v, 'relu_ms': round(tr, 4), 'res_ms': round(ts, 4), 'relu_tflops': round(flops / tr / 1e9, 1), 'res_tflops': round(flops / ts / 1e9, 1)}))
