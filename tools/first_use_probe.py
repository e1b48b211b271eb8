#!/usr/bin/env python3
"""First, second and third forward of a fresh model per precision (batch 500 / 512 of 32 x 32 patches): what a process pays once
for a kernel's first launch (code load, launch attributes, workspace) — ~1 ms (profiles/r04_k_first_use.txt).

    python tools/first_use_probe.py
"""
import sys, time, torch
sys.path.insert(0, '.')
from dsen2_amd import weights as W
from dsen2_amd.DSen2Net import s2model
for prec, n in (('bf16', 500), ('bf16', 512), ('bf16x3', 512), ('fp32', 512)):
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision=prec)
    m.set_weights_flat(W.random_he_uniform(10, 6, 6, 128, seed=2))
    x = [torch.rand((n, c, 32, 32), device='cuda') * 5 for c in (4, 6)]
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); m.forward_device(x); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(prec, n, 'launches', m.body_launches(n, 32, 32), 'forward ms (1st, 2nd, 3rd):', ts, flush=True)
