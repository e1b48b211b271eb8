#!/opt/conda/bin/python3.9
"""tests/golden/interp_shapes.npz: the reference's own interp_patches (utils/patches.py:11-16 over scikit-image 0.18.3) on
shapes and factors beyond the x2 / x6 of the tile path — non-integer factors, odd and tiny planes, a factor barely above 1,
planes with a constant plateau at their maximum and minimum (warp()'s clip), values up to 65535 — so that the bit-exact
restatement of skimage's float32 arithmetic (oracle/patches_oracle.py, dsen2_amd/csrc/patch_ops.hip) is pinned on more than
the cases the path happens to use.  DATA only: seeded inputs and what the reference returned.

    /opt/conda/bin/python3.9 tests/golden/make_golden_interp_shapes.py      (build container; numpy 1.26.4, scikit-image 0.18.3)
"""
import os
import sys

import numpy as np

sys.path.insert(0, '/root/reference')
from utils.patches import interp_patches  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(10, 10, 25, 25), (7, 9, 23, 31), (13, 5, 40, 17), (3, 3, 18, 18), (2, 2, 12, 12), (5, 7, 10, 14), (11, 13, 33, 39),
         (8, 8, 50, 50), (20, 30, 47, 61), (30, 30, 31, 31), (8, 8, 8, 8), (12, 12, 72, 72), (24, 24, 48, 48), (6, 40, 36, 80)]


def main():
    rng = np.random.default_rng(20261005)
    out = {}
    for k, (h, w, oh, ow) in enumerate(CASES):
        x = rng.integers(0, 65536 if k % 3 == 0 else 13110, size=(2, 3, h, w)).astype(np.float32)
        if h >= 5 and w >= 5:
            x[0, 0, :3, :3] = x.max() + 7          # a plateau at the plane's maximum, in a corner (mirrored taps)
            x[1, 2, -3:, 1:4] = 0                  # ... and at its minimum
        x[1, 1] = 4321                             # a constant plane
        y = interp_patches(x.copy(), (2, 3, oh, ow))
        assert y.dtype == np.float32 and y.shape == (2, 3, oh, ow)
        out['in_%02d' % k] = x
        out['out_%02d' % k] = y
    np.savez_compressed(os.path.join(HERE, 'interp_shapes.npz'), **out)
    print('%d cases, %d bytes' % (len(CASES), os.path.getsize(os.path.join(HERE, 'interp_shapes.npz'))))


if __name__ == '__main__':
    main()
