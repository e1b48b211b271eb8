#!/usr/bin/env python3
"""The C oracle (oracle/dsen2_oracle.c, the checker of the GPU parity tests) under AddressSanitizer + UBSan on the CPU — GPU
sanitizers are not available on this pool.  Build and run:

    mkdir -p build/asan && gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -fopenmp \
        -fPIC -std=c11 -shared -o build/asan/libdsen2_oracle.so oracle/dsen2_oracle.c -lm
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/oracle_sanitizer_run.py
"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, dsen2_oracle as do
c_oracle._SO = os.path.join(ROOT, 'build', 'asan', 'libdsen2_oracle.so')
c_oracle._lib = None
rng = np.random.default_rng(0)
for bands, d, f, h, w in (((4, 6), 2, 128, 9, 7), ((4, 6, 2), 1, 128, 5, 11), ((4, 6), 1, 256, 3, 3), ((4, 6), 1, 128, 1, 1)):
    xs = [rng.random((2, c, h, w)).astype(np.float32) * 5 for c in bands]
    flat = do.he_uniform_weights(sum(bands), bands[-1], d, f, seed=3, bias_scale=0.1)
    y = c_oracle.forward(xs, flat, d, f)
    assert np.abs(y - do.forward(xs, flat, d, f)).max() < 1e-10
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'interp_shapes.npz'))
for k in range(14):
    x, want = g['in_%02d' % k], g['out_%02d' % k]
    assert c_oracle.upsample(x, want.shape[2], want.shape[3], skimage=True).tobytes() == want.tobytes()
    c_oracle.upsample(x, want.shape[2], want.shape[3])
print('C oracle under ASan + UBSan: forward x 4 shapes, both up-samplers x 14 shapes: clean')
