#!/usr/bin/env python3
"""Per-CU versus chip-wide limits of the bf16 body convolution (diagnostic build): the same work per workgroup
(4 items) on G = 256, 128, 64, 32, 8 workgroups (batch = G patches of 32x32), full kernel and without the epilogue's
memory operations.

    DSEN2_HIP_LIB=build/libdsen2_hip_diag.so python tools/gridcap_body_conv.py
"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W
from dsen2_amd.DSen2Net import s2model
F, H, D = 256, 32, 3
flat = W.random_he_uniform(10, 6, D, F, seed=1)
for G in (256, 128, 64, 32, 8):
    ms = {}
    for mask in (0, 1, 3):
        _lib.diag_set(3, G); _lib.diag_set(1, mask)
        ms[mask] = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision='bf16')
        ms[mask].set_weights_flat(flat)
    _lib.diag_set(3, 0); _lib.diag_set(1, 0)
    B = G
    a = torch.randn((B, H, H, F), device='cuda').to(torch.bfloat16); r = torch.randn((B, H, H, F), device='cuda'); o = torch.empty_like(r)
    res = {}
    for mask in (0, 1, 3):
        tr = [ms[mask].time_body_conv(1, a, None, o, iters=10) for _ in range(3)]
        ts = [ms[mask].time_body_conv(2, a, r, o, iters=10) for _ in range(3)]
        res[mask] = (float(np.median(tr)), float(np.median(ts)))
    print(json.dumps({'G': G, 'convA_full': round(res[0][0], 4), 'convA_nostore': round(res[1][0], 4),
                      'convB_full': round(res[0][1], 4), 'convB_nostore': round(res[1][1], 4), 'convB_noepimem': round(res[3][1], 4)}))
