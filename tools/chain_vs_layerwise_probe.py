#!/usr/bin/env python3
"""Where does the one-launch chain (conv3x3_body16w_chain_kernel: a workgroup owns whole patches through every layer) beat the
per-layer launches of the same item code?  It was adopted on the bench's 32x32 patches (+5 % at F = 128); this probe times both
forms of the SAME network per patch size: a batch that chains (a multiple of the CU count) against one a few patches smaller
that does not (body_launches tells which form ran), in us per patch.

    python tools/chain_vs_layerwise_probe.py [--precision bf16|bf16x3] [--feat 128|256]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import weights as W                 # noqa: E402
from dsen2_amd.DSen2Net import s2model             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='bf16')
ap.add_argument('--feat', type=int, default=128)
args = ap.parse_args()
d = 6 if args.feat == 128 else 4
m = s2model(((4, None, None), (6, None, None)), num_layers=d, feature_size=args.feat, precision=args.precision)
m.set_weights_flat(W.random_he_uniform(10, 6, d, args.feat, seed=2))
m.max_workspace_bytes = 64 << 30
cus = int(torch.cuda.get_device_properties(0).multi_processor_count)


def timed(n, p):
    x = [torch.rand((n, c, p, p), device='cuda') * 5 for c in (4, 6)]
    out = torch.empty((n, 6, p, p), device='cuda')
    t_end = time.perf_counter() + 0.06
    while time.perf_counter() < t_end:              # warm: clocks (profiles/r04_ablation.md §2) and the workspace
        m.forward_device(x, out=out)
        torch.cuda.synchronize()
    reps = max(3, int(0.25 / max(1e-4, 4e-6 * n * (p / 32.0) ** 2)))
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.forward_device(x, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


for p, ks in ((32, (2, 4)), (64, (1, 2)), (96, (1, 2)), (128, (1, 2, 3)), (192, (1, 2))):
    for k in ks:
        n_chain = k * cus
        n_layer = n_chain - 12
        row = {'precision': args.precision, 'feat': args.feat, 'patch': p, 'patches_per_workgroup': k}
        for name, n in (('chain', n_chain), ('layerwise', n_layer)):
            launches = m.body_launches(n, p, p)
            t = timed(n, p)
            row[name] = {'n': n, 'body_launches': launches, 'us_per_patch': round(t / n * 1e6, 2)}
        row['chain_over_layerwise'] = round(row['chain']['us_per_patch'] / row['layerwise']['us_per_patch'], 3)
        print(json.dumps(row), flush=True)
