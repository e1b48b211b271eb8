#!/usr/bin/env python3
"""In-kernel timeline of the bf16 body convolution (diagnostic build, ablation mask 32): s_memtime stamps of waves 0
and 7 of the first four workgroups around the epilogue and the first steps of every item.

    python -m dsen2_amd.build --diag
    DSEN2_HIP_LIB=build/libdsen2_hip_diag.so python tools/stamp_body_conv.py [layer [mask]]     (layer 1 = conv-A, 2 = conv-B)

mask 32 (default): all stamps (the stamped workgroups run ~30 % slower); mask 96: only the two loop-top stamps per item —
the in-kernel clock (shader cycles per 100 MHz tick) and cycles per item of the kernel at its normal speed.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 32
F, B, H, D = 256, 256, 32, 3
lib = _lib.load()
buf = torch.zeros(4 * 2 * 16 * 32, dtype=torch.int64, device='cuda')
lib.dsen2_diag_set_stamps.argtypes = [ctypes.c_void_p]
lib.dsen2_diag_set_stamps(ctypes.c_void_p(buf.data_ptr()))
_lib.diag_set(1, mask)
m = s2model(((4, None, None), (6, None, None)), num_layers=D, feature_size=F, precision='bf16')
m.set_weights_flat(W.random_he_uniform(10, 6, D, F, seed=1))
_lib.diag_set(1, 0)
a = torch.randn((B, H, H, F), device='cuda').to(torch.bfloat16); r = torch.randn((B, H, H, F), device='cuda'); o = torch.empty_like(r)
for _ in range(3 if mask == 32 else 40):      # mask 96: ~0.1 s of back-to-back launches, the clock has settled
    ms = m.time_body_conv(layer, a, r if layer == 2 else None, o, iters=5 if mask == 32 else 20)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(4, 2, 16, 32)
print('layer %d: %.4f ms per launch (stamping build)' % (layer, ms))
names = {0: 'loop top', 14: 'residual loads issued', 15: 'pass 0 done', 16: 'pass 1 done', 17: 'pass 2 done', 18: 'pass 3 done',
         1: 'epilogue done', 2: 'first fragments', 3: 'step 0', 4: 'step 1', 5: 'step 2', 6: 'step 3', 7: 'step 4', 8: 'step 5',
         9: 'step 6', 10: 'step 7', 11: 'step 8', 12: 'end of chunk 1', 13: 'end of item',
         31: 'chunk 2 tap 0 end', 19: 'tap1 DMAs issued', 20: 'tap1 MFMAs issued', 21: 'tap1 own DMA wait', 22: 'tap1 barrier',
         30: 'tap 2 end', 23: 'tap3 DMAs issued', 24: 'tap3 MFMAs issued', 25: 'tap3 own DMA wait', 26: 'tap3 barrier'}
order = [0, 14, 15, 16, 17, 18, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]
# in-kernel clock: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime) between the loop tops of items 1 and 3
for wg in (0, 1):
    c = st[wg, 0]
    if c[3, 19] > c[1, 19]:
        print('workgroup %d: in-kernel clock %.3f GHz (items 1..3: %d cycles in %d ticks of 10 ns)' % (
            wg, (c[3, 0] - c[1, 0]) / (c[3, 19] - c[1, 19]) * 0.1, c[3, 0] - c[1, 0], c[3, 19] - c[1, 19]))
for wg in (0, 1):
    for wv in (0, 1):
        if mask != 32:
            c = st[wg, wv]
            print('workgroup %d wave %d: cycles per item %s' % (wg, 7 * wv, [int(c[i + 1, 0] - c[i, 0]) for i in range(4) if c[i + 1, 0] > 0]))
            continue
        print('workgroup %d wave %d: cycles since the item loop top of item 0 (100 MHz s_memtime ticks x clock ratio)' % (wg, 7 * wv))
        t0 = st[wg, wv, 0, 0]
        for it in range(5):
            row = st[wg, wv, it]
            if row[0] == 0:
                continue
            prev = row[0]
            parts = []
            for k in order:
                if row[k] == 0:
                    continue
                parts.append('%s +%d' % (names[k], row[k] - prev))
                prev = row[k]
            print('  item %d @%d: %s' % (it, row[0] - t0, '; '.join(parts)))
