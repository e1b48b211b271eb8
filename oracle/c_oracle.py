"""TEST INFRASTRUCTURE ONLY — ctypes loader for oracle/libdsen2_oracle.so (built from dsen2_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libdsen2_oracle.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'dsen2_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libdsen2_oracle.so'], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        dp, fp, i = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.c_int
        _lib.dsen2_oracle_conv3x3_f64.argtypes = [dp, fp, fp, dp, i, i, i, i, i, i]
        _lib.dsen2_oracle_conv3x3_f64.restype = None
        _lib.dsen2_oracle_forward_f64.argtypes = [dp, dp, fp, dp, i, i, i, i, i, i, i]
        _lib.dsen2_oracle_forward_f64.restype = ctypes.c_int
        _lib.dsen2_oracle_upsample_f64.argtypes = [fp, fp, i, i, i, i, i]
        _lib.dsen2_oracle_upsample_f64.restype = None
        _lib.dsen2_oracle_upsample_skimage.argtypes = [fp, fp, i, i, i, i, i]
        _lib.dsen2_oracle_upsample_skimage.restype = None
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def conv3x3(x, kernel, bias, relu=False):
    x = np.ascontiguousarray(x, np.float64)
    kernel = np.ascontiguousarray(kernel, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    n, cin, h, w = x.shape
    cout = kernel.shape[3]
    out = np.empty((n, cout, h, w), np.float64)
    lib().dsen2_oracle_conv3x3_f64(_dp(x), _fp(kernel), _fp(bias), _dp(out), n, cin, cout, h, w, int(relu))
    return out


def forward(inputs, flat_weights, num_layers, feature_size):
    xs = [np.ascontiguousarray(a, np.float64) for a in inputs]
    xcat = np.ascontiguousarray(np.concatenate(xs, axis=1))
    skip = xs[-1]
    flat = np.ascontiguousarray(flat_weights, np.float32)
    n, cin, h, w = xcat.shape
    cout = skip.shape[1]
    out = np.empty((n, cout, h, w), np.float64)
    rc = lib().dsen2_oracle_forward_f64(_dp(xcat), _dp(skip), _fp(flat), _dp(out), n, cin, cout, h, w,
                                        num_layers, feature_size)
    if rc != 0:
        raise MemoryError('dsen2_oracle_forward_f64 failed')
    return out


def upsample(image_lr, oh, ow, skimage=False):
    """interp_patches on the C side.  skimage=False: exact (double) coordinates and blend — the mathematical definition;
    skimage=True: scikit-image 0.18.3's float32 arithmetic operation by operation (the reference's bits)."""
    x = np.ascontiguousarray(image_lr, np.float32)
    lead = x.shape[:-2]
    h, w = x.shape[-2:]
    planes = int(np.prod(lead)) if lead else 1
    out = np.empty(lead + (oh, ow), np.float32)
    fn = lib().dsen2_oracle_upsample_skimage if skimage else lib().dsen2_oracle_upsample_f64
    fn(_fp(x), _fp(out), planes, h, w, oh, ow)
    return out
