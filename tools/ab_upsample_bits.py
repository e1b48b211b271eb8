#!/usr/bin/env python3
"""Up-sampler outputs of one library as an .npz over many shapes (x2, x3, x6, ragged, mirror edges, one-pixel dims, sizes
that are not multiples of the 4 x 8 thread tile), for bit-for-bit comparison of two builds in one gpurun call:
    DSEN2_HIP_LIB=a.so python tools/ab_upsample_bits.py a.npz ; DSEN2_HIP_LIB=b.so python tools/ab_upsample_bits.py b.npz
    python tools/ab_upsample_bits.py --compare a.npz b.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == '--compare':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))]
    print('%d cases, %d differ %s' % (len(a.files), len(bad), bad[:5]))
    sys.exit(1 if bad else 0)

import torch                                            # noqa: E402
from dsen2_amd import patches as P                       # noqa: E402

rng = np.random.default_rng(5)
cases = [(7, 6, 64, 64, 128, 128), (3, 2, 32, 32, 192, 192), (2, 6, 96, 96, 192, 192), (1, 2, 37, 53, 74, 106),
         (2, 3, 1, 1, 2, 2), (1, 1, 2, 3, 12, 18), (1, 2, 10, 10, 27, 27), (2, 1, 5, 7, 15, 21), (1, 1, 33, 1, 66, 5),
         (1, 2, 9, 11, 18, 23), (1, 1, 20, 30, 41, 61), (3, 1, 16, 16, 97, 101), (1, 1, 50, 40, 100, 80), (1, 1, 13, 13, 26, 91),
         (1, 2, 64, 64, 100, 128), (1, 1, 8, 8, 12, 12)]          # the last two: less than x2 in one / both directions
res = {}
for n, c, h, w, oh, ow in cases:
    x = torch.from_numpy((rng.random((n, c, h, w), dtype=np.float32) * 12000).astype(np.float32)).cuda()
    for pd in (1.0, 2000.0):
        res['%dx%dx%dx%d_to_%dx%d_div%g' % (n, c, h, w, oh, ow, pd)] = P.interp_patches_device(x, (oh, ow), post_divisor=pd).cpu().numpy()
np.savez(sys.argv[1], **res)
print('wrote %d cases' % len(res))
