#!/usr/bin/env python3
"""In-kernel timeline of the bf16 chain kernel (diagnostic build, mask 4096): s_memtime stamps of waves 0 and 7 of the
first four workgroups at every item-loop top and after every epilogue, for every layer of a VDSen2 forward
(256 x 32x32 patches, d = 32, F = 256: 64 layers x 4 items per workgroup).

    python -m dsen2_amd.build --diag
    DSEN2_HIP_LIB=build/libdsen2_hip_diag.so python tools/stamp_chain.py [mask=4096]     (5120 = drained boundaries)
s_memtime ticks are 10 ns (100 MHz) on gfx950.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

mask = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F, B, H, D = 256, 256, 32, 32
lib = _lib.load()
buf = torch.zeros(4 * 2 * 64 * 8 * 2, dtype=torch.int64, device='cuda')
lib.dsen2_diag_set_stamps.argtypes = [ctypes.c_void_p]
lib.dsen2_diag_set_stamps(ctypes.c_void_p(buf.data_ptr()))
_lib.diag_set(1, mask)
m = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision='bf16')
m.set_weights_flat(W.random_he_uniform(10, 6, D, F, seed=1))
_lib.diag_set(1, 0)
rng = np.random.Generator(np.random.PCG64(0))
xs = [torch.from_numpy(rng.random((B, c, H, H), dtype=np.float32) * np.float32(5.0)).cuda() for c in (4, 6)]
for _ in range(20):
    y = m.forward_device(xs)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(4, 2, 64, 8, 2).astype(np.int64)
TICK = 0.01    # us
for wg in (0, 1, 3):
    for wv in (0, 1):
        s = st[wg, wv]
        if s[0, 0, 0] == 0:
            continue
        total = (s[63, 4, 1] - s[0, 0, 0]) * TICK
        # per layer: items = from 'after epilogue' of iteration i to loop top of iteration i+1; epilogues = top -> after
        item = (s[:, 1:5, 0] - s[:, 0:4, 1]) * TICK                 # [64, 4] steps of item i
        epi = (s[:, 1:5, 1] - s[:, 1:5, 0]) * TICK                  # [64, 4] epilogue of item i (runs in iteration i+1)
        gap = (s[1:, 0, 1] - s[:-1, 4, 1]) * TICK                   # [63] last epilogue of layer l -> first item of l+1 starts
        a, b = slice(0, 64, 2), slice(1, 64, 2)
        print('workgroup %d wave %d: chain %.1f us = %.2f us per layer' % (wg, 7 * wv, total, total / 64))
        print('  conv-A: item steps %s us, epilogue %s us' % (np.round(item[a].mean(axis=0), 2), np.round(epi[a].mean(axis=0), 2)))
        print('  conv-B: item steps %s us, epilogue %s us' % (np.round(item[b].mean(axis=0), 2), np.round(epi[b].mean(axis=0), 2)))
        print("  boundary (last epilogue done -> the next layer's first item past its loop top): A->B %.2f us, B->A %.2f us"
              % (gap[0::2].mean(), gap[1::2].mean()))
        print('  layer sums: conv-A %.2f us, conv-B %.2f us' % ((item[a].sum(axis=1) + epi[a].sum(axis=1)).mean(), (item[b].sum(axis=1) + epi[b].sum(axis=1)).mean()))
