#!/usr/bin/env python3
"""In-kernel timeline of the matrix-core output convolution (diagnostic build, key 6 mask 32): s_memtime stamps of every wave
of workgroups 0-3 for the bench config (512 patches of 32x32: two jobs of one phase per workgroup).

    python -m dsen2_amd.build --diag
    DSEN2_HIP_LIB=build/libdsen2_hip_diag.so python tools/stamp_out_conv.py [extra mask bits]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib, weights as W          # noqa: E402
from dsen2_amd.DSen2Net import s2model            # noqa: E402

extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib = _lib.load()
buf = torch.zeros(4 * 8 * 32, dtype=torch.int64, device='cuda')
lib.dsen2_diag_set_stamps.argtypes = [ctypes.c_void_p]
lib.dsen2_diag_set_stamps(ctypes.c_void_p(buf.data_ptr()))
_lib.diag_set(6, 32 | extra)
m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128)
m.set_weights_flat(W.random_he_uniform(10, 6, 6, 128, seed=1))
rng = np.random.Generator(np.random.PCG64(0))
xs = [torch.from_numpy(rng.random((512, c, 32, 32), dtype=np.float32) * np.float32(5.0)).cuda() for c in (4, 6)]
for _ in range(10):
    y = m.forward_device(xs)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(4, 8, 32).astype(np.int64)
names = ['start', 'weights in LDS', 'first fetch issued']
for j in range(2):
    names += ['job %d: phase top' % j, 'skip loads issued', 'rows done', 'past barrier 1', 'second stage done', 'past barrier 2']
for wg in range(4):
    t0 = st[wg, :, 0].min()
    print('workgroup %d (ticks since its first wave started; waves 0..7)' % wg)
    for i, nm in enumerate(names):
        print('  %-22s %s' % (nm, ' '.join('%7d' % (v - t0) for v in st[wg, :, i])))
    print('  %-22s %s' % ('end', ' '.join('%7d' % (v - t0) for v in st[wg, :, 31])))
