"""Drop-in for the reference's testing/supres.py: same names, signatures, constants, prints and returns.

    from supres import DSen2_20, DSen2_60          (testing/s2_tiles_supres.py:9, testing/demoDSen2.py:5)

DSen2_20(d10, d20, deep=False)      -> [x, y, 6] float32     testing/supres.py:15-30
DSen2_60(d10, d20, d60, deep=False) -> [x, y, 2] float32     testing/supres.py:33-50
_predict(test, input_shape, deep=False, run_60=False)        testing/supres.py:53-66

The whole chain — symmetric pad + overlapped tiling, mirror-bilinear up-sampling, /2000, the CNN,
recomposition, *2000 — runs on one MI355X without touching the host between stages; with
torch.distributed initialised (one process per GPU) the patches are sharded across ranks
(dsen2_amd/dist.py).  Models are cached per (device, architecture, weight file) instead of being rebuilt
and re-read on every call as the reference does (supres.py:59,63) — observable behaviour is unchanged.
"""
from __future__ import division

import os

import numpy as np
import torch

from . import dist as _dist
from . import patches as _patches
from .DSen2Net import s2model

SCALE = 2000
MDL_PATH = '../models/'
# Not in the reference: arithmetic of the residual-block convolutions.  'fp32' (default, what keras computes) or
# 'bf16' (bf16 operands, fp32 accumulate and residual stream; ~7x faster for deep=True, ~1e-3 relative error).
PRECISION = os.environ.get('DSEN2_PRECISION', 'fp32')

_MODEL_CACHE = {}


def _weight_file(deep, run_60):
    # testing/supres.py:55-60
    if deep:
        return MDL_PATH + 's2_034_lr_1e-04.hdf5' if run_60 else MDL_PATH + 's2_033_lr_1e-04.hdf5'
    return MDL_PATH + 's2_030_lr_1e-05.hdf5' if run_60 else MDL_PATH + 's2_032_lr_1e-04.hdf5'


def _get_model(input_shape, deep, run_60):
    if deep:
        num_layers, feature_size = 32, 256      # supres.py:56
    else:
        num_layers, feature_size = 6, 128       # supres.py:59
    predict_file = _weight_file(deep, run_60)
    dev = _patches.default_device()
    key = (str(dev), tuple(s[0] for s in input_shape), num_layers, feature_size, os.path.abspath(predict_file), PRECISION)
    model = _MODEL_CACHE.get(key)
    if model is None:
        model = s2model(input_shape, num_layers=num_layers, feature_size=feature_size, device=dev, precision=PRECISION)
        print('Symbolic Model Created.')
        model.load_weights(predict_file)
        _MODEL_CACHE[key] = model
    else:
        print('Symbolic Model Created.')
    print("Predicting using file: {}".format(predict_file))
    return model


def clear_model_cache():
    _MODEL_CACHE.clear()


def _predict(test, input_shape, deep=False, run_60=False):
    """testing/supres.py:53-66 — list of NCHW float32 arrays in, [N, Cout, H, W] float32 array out."""
    model = _get_model(input_shape, deep, run_60)
    return model.predict(test, verbose=1)


def _run(dsets, scales, patch, border, deep, run_60):
    dev = _patches.default_device()
    imgs = [_patches._to_device_f32(d, dev) for d in dsets]
    patch_sizes = [patch // (scales[0] // s) for s in scales]          # P, P//2(, P//6)
    borders = [border // (scales[0] // s) for s in scales]             # b, b//2(, b//6)
    org, n_alloc = _patches.tile_origins(imgs[-1].shape, patch_sizes[-1], borders[-1])
    used = org.shape[0]
    input_shape = tuple((int(d.shape[2]), None, None) for d in imgs)
    model = _get_model(input_shape, deep, run_60)
    cout = model.cout
    # this rank's contiguous share of the USED patches (the reference's trailing all-zero patches are
    # never read by recompose_images, so they are not computed)
    first, count = _dist.shard_range(used)
    bs = model.batch_limit(patch, patch)
    pred_local = torch.empty((count, cout, patch, patch), dtype=torch.float32, device=dev)
    org_dev = [_patches.device_origins(org, s, dev) for s in scales]   # uploaded once: the loop below only enqueues
    for i0 in range(0, count, bs):
        n = min(bs, count - i0)
        xs = []
        for k, (img, s, ps, b) in enumerate(zip(imgs, scales, patch_sizes, borders)):
            if k == 0:
                # `p10 /= SCALE` (supres.py:23) folded into the gather (IEEE divide, bit-identical)
                xs.append(_patches.gather_patches_device(img, org, s, b, ps, n_alloc, divisor=SCALE,
                                                         first=first + i0, count=n, origins_dev=org_dev[k]))
            else:
                lr = _patches.gather_patches_device(img, org, s, b, ps, n_alloc, first=first + i0, count=n,
                                                    origins_dev=org_dev[k])
                # up-sample raw values, then `/= SCALE` (supres.py:24,43-44)
                xs.append(_patches.interp_patches_device(lr, (patch, patch), post_divisor=SCALE))
        model.forward_device(xs, out=pred_local[i0:i0 + n])
    size = dsets[0].shape
    # Everything above is only ENQUEUED (seconds of GPU work for a full tile): the page-locked buffer the result is
    # downloaded into is allocated now, under that work (0.18 s for a 10980^2 x 6 image; the download itself then
    # runs at 57 GB/s instead of 11 GB/s from pageable memory: 0.05 s instead of 0.26 s).
    host = _host_output((int(size[0]), int(size[1]), cout)) if n_alloc > 1 else None
    pred = _dist.gather_patches(pred_local, used)                      # every rank gets all `used` patches
    if n_alloc == 1:
        # recompose_images' single-patch shortcut (patches.py:375-376): a[0] uncropped
        images = pred[0].permute(1, 2, 0).contiguous() * SCALE
        return images.cpu().numpy()
    print((cout, size[0], size[1]))                                    # patches.py:392
    # `images *= SCALE` (supres.py:29) folded into the recomposition
    images = _patches.recompose_device(pred, border, size, scale=SCALE)
    if host is None:
        return images.cpu().numpy()
    host.copy_(images, non_blocking=True)
    torch.cuda.synchronize(dev)
    return host.numpy()            # ndarray view of the page-locked tensor (kept alive by the array)


PINNED_OUTPUT_MIN_BYTES = 64 << 20     # DSEN2_PINNED_OUTPUT=0 disables; torch caches page-locked blocks for reuse


def _host_output(shape):
    nbytes = 4 * shape[0] * shape[1] * shape[2]
    if os.environ.get('DSEN2_PINNED_OUTPUT', '1') == '0' or nbytes < PINNED_OUTPUT_MIN_BYTES:
        return None
    try:
        return torch.empty(shape, dtype=torch.float32, pin_memory=True)
    except RuntimeError:           # page-locked memory exhausted: the pageable path still works
        return None


def DSen2_20(d10, d20, deep=False):
    """20 m -> 10 m.  d10 [x, y, 4] (B2 B3 B4 B8), d20 [x/2, y/2, 6] (B5 B6 B7 B8A B11 B12), any real dtype, HWC.
    deep=True selects VDSen2 (d=32, F=256).  Returns [x, y, 6] float32.  Geometry of testing/supres.py:21-22:
    patches of 128 with an 8-pixel border."""
    return _run([d10, d20], [2, 1], patch=128, border=8, deep=deep, run_60=False)


def DSen2_60(d10, d20, d60, deep=False):
    """60 m -> 10 m.  As DSen2_20 plus d60 [x/6, y/6, 2] (B1 B9; B10 is not super-resolved).  Returns [x, y, 2]
    float32.  Geometry of testing/supres.py:40-41: patches of 192 with a 12-pixel border."""
    return _run([d10, d20, d60], [6, 3, 1], patch=192, border=12, deep=deep, run_60=True)
