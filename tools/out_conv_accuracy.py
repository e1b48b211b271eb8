#!/usr/bin/env python3
"""The two output-convolution kernels against the float64 oracle on the same operands (diagnostic library: key 2 = 2 the
matrix-core kernel, 3 the vector-unit kernel): they sum in different orders, so their bits differ; this prints how far each
is from the exact result and from the other.
    DSEN2_HIP_LIB=build/libdsen2_hip_diag.so python tools/out_conv_accuracy.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import _lib                              # noqa: E402
from dsen2_amd.DSen2Net import conv3x3_nhwc            # noqa: E402
from oracle import c_oracle                            # noqa: E402   (checker only)

for feat, cout, n, h, w in [(128, 6, 8, 32, 32), (128, 2, 4, 48, 192), (256, 6, 4, 32, 32)]:
    rng = np.random.default_rng(feat + cout)
    x = (rng.standard_normal((n, feat, h, w)) * 1.0).astype(np.float32)
    skip = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    k = (rng.standard_normal((3, 3, feat, cout)) * np.sqrt(2.0 / (9 * feat))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    ref = c_oracle.conv3x3(x, k, b) + skip
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda()
    sd = torch.from_numpy(skip).cuda()
    out = {}
    for v in (2, 3):
        _lib.diag_set(2, v)
        out[v] = conv3x3_nhwc(xd, k, b, epilogue=2, aux=sd).cpu().numpy().astype(np.float64)
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    print('F=%d Cout=%d %dx%dx%d: rms error vs float64  matrix cores %.3e   vector units %.3e   (output rms %.2f);  '
          'max |difference| between the two %.3e, identical values %.1f %%'
          % (feat, cout, n, h, w, rms(out[2] - ref), rms(out[3] - ref), rms(ref), np.abs(out[2] - out[3]).max(),
             100.0 * np.mean(out[2] == out[3])))
