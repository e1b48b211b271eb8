#!/usr/bin/env python3
"""Shape fuzz of the bf16 path: random (n, h, w) including single pixels, widths around the 32-pixel item width and
heights around the 16-pixel item height; the bf16 network must stay within the bf16 gate of the fp32 network on the
same weights (both through the C ABI), and be bitwise reproducible.  `python tools/fuzz_bf16_shapes.py [cases] [seed]`"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import weights as W
from dsen2_amd.DSen2Net import s2model
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for F, D in ((128, 2), (256, 2)):
    flat = W.random_he_uniform(10, 6, D, F, seed=7, bias_scale=0.05)
    m16 = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision='bf16'); m16.set_weights_flat(flat)
    m32 = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F); m32.set_weights_flat(flat)
    for c in range(cases):
        n = int(rng.integers(1, 6))
        h = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, int(rng.integers(1, 80))]))
        w = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 96, int(rng.integers(1, 100))]))
        xs = [torch.rand((n, 4, h, w), device='cuda') * 5, torch.rand((n, 6, h, w), device='cuda') * 5]
        y16 = m16.forward_device(xs).clone()
        y32 = m32.forward_device(xs)
        again = m16.forward_device(xs)
        rel = float((y16 - y32).pow(2).mean().sqrt() / y32.pow(2).mean().sqrt())
        ok = bool(torch.isfinite(y16).all()) and rel < 5e-3 and bool(torch.equal(again, y16))
        if not ok:
            bad += 1
            print('FAIL F=%d n=%d h=%d w=%d rel=%.3e' % (F, n, h, w, rel))
print('fuzz: %d failures in %d cases' % (bad, 2 * cases))
sys.exit(1 if bad else 0)
