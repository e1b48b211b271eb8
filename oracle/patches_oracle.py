"""TEST INFRASTRUCTURE ONLY — numpy restatement of the inference half of utils/patches.py.

Restates (paths relative to /root/reference):
  * utils/patches.py:11-16    interp_patches      -> interp_patches()
  * utils/patches.py:19-80    get_test_patches    -> get_test_patches()
  * utils/patches.py:83-156   get_test_patches60  -> get_test_patches60()
  * utils/patches.py:374-405  recompose_images    -> recompose_images()

PARITY STATUS: pinned.  tests/golden/patches_*.npz hold outputs of the reference's own functions
(imported from /root/reference under /opt/conda/bin/python3.9, scikit-image 0.18.3) produced by
tests/golden/make_golden_patches.py; tests/test_oracle_patches.py checks this file against them
(tiling / recompose bit-exact; up-sampling BIT-EXACT too in the f32_coords=True mode, which follows scikit-image
0.18.3's float32 arithmetic operation by operation — see interp_patches).

Third-party arithmetic: scikit-image ``transform.resize`` (unpinned in the reference's README;
0.18.3 is what the capture used): order-1 interpolation at half-pixel centres
``src = (dst + 0.5) * in/out - 0.5`` between floor(src) and ceil(src), out-of-range neighbours
mirrored WITHOUT repeating the edge sample (skimage mode 'reflect' == numpy.pad 'reflect').
"""
import math
import numpy as np


def mirror_index(idx, dim):
    """skimage _warps_cy.coord_map(mode='R') for an integer array ``idx``."""
    idx = np.asarray(idx, np.int64)
    if dim == 1:
        return np.zeros_like(idx)
    cmax = dim - 1
    k = np.abs(idx)
    odd = (k // cmax) % 2 != 0
    folded = np.where(odd, cmax - (k % cmax), k % cmax)
    inside = (idx >= 0) & (idx <= cmax)
    return np.where(inside, idx, folded)


def _axis_taps(n_in, n_out, f32_coords=False):
    if f32_coords:
        # what skimage 0.18.3 does: warp() casts the 3x3 matrix to the image dtype (float32), so the
        # source coordinate is  float32(scale) * float32(dst) + float32(offset)  rounded to float32.
        scale = np.float32(n_in / n_out)
        offset = np.float32(0.5 * n_in / n_out - 0.5)
        src = (scale * np.arange(n_out, dtype=np.float32) + offset).astype(np.float32)
        lo = np.floor(src)
        hi = np.ceil(src)
        frac = (src - lo).astype(np.float64)
        return mirror_index(lo.astype(np.int64), n_in), mirror_index(hi.astype(np.int64), n_in), frac
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    lo = np.floor(src)
    hi = np.ceil(src)
    frac = src - lo
    return mirror_index(lo.astype(np.int64), n_in), mirror_index(hi.astype(np.int64), n_in), frac


def interp_patches(image_lr, hr_shape, f32_coords=False):
    """[N,C,h,w] float32 -> [N,C,H,W] float32 (patches.py:11-16). hr_shape = 4-tuple, H,W read from [2:4].

    f32_coords=False: exact (float64) sample positions and blend — the mathematical definition.
    f32_coords=True : scikit-image 0.18.3's arithmetic for a float32 image, operation by operation — BIT-IDENTICAL to the
                      captured reference outputs (tests/test_oracle_patches.py).  What `resize` -> `warp` -> `_warp_fast[float32]`
                      -> `bilinear_interpolation` computes there (established from the compiled extension's instruction
                      sequence and confirmed bit for bit on random images, tests/golden/make_golden_patches.py's captures):
                        c, r   = float32(scale) * float32(index) + float32(offset), each operation rounded to float32
                        dc, dr = c - floor(c), r - floor(r) in float32
                        top    = (1.0 - double(dc)) * double(top_left) + double(float32(dc * top_right))
                        bottom = (1.0 - double(dc)) * double(bottom_left) + double(float32(dc * bottom_right))
                        out    = float32((1.0 - double(dr)) * top + double(dr) * bottom)
                      — the `dc * neighbour` product alone is a float32 multiply (both operands are float32 in the Cython
                      source), everything else is promoted to double by the literal 1.  `warp` then clips to the input's
                      [min, max]: a no-op for this arithmetic (a plateau of equal taps returns its value exactly — the blend's
                      error stays below half a float32 ulp — and every operation is monotone in the taps), kept for fidelity.
    """
    image_lr = np.asarray(image_lr, np.float32)
    oh, ow = int(hr_shape[2]), int(hr_shape[3])
    h, w = image_lr.shape[2:4]
    r0, r1, fr = _axis_taps(h, oh, f32_coords)
    c0, c1, fc = _axis_taps(w, ow, f32_coords)
    xq = image_lr / np.float32(30000)                              # float32 divide, as patches.py:15
    x = xq.astype(np.float64)
    fcb = fc[None, None, None, :]
    frb = fr[None, None, :, None]
    if f32_coords:
        dcf = fc.astype(np.float32)[None, None, None, :]
        right_top = (dcf * xq[:, :, r0][:, :, :, c1]).astype(np.float32).astype(np.float64)
        right_bot = (dcf * xq[:, :, r1][:, :, :, c1]).astype(np.float32).astype(np.float64)
        top = (1.0 - fcb) * x[:, :, r0][:, :, :, c0] + right_top
        bot = (1.0 - fcb) * x[:, :, r1][:, :, :, c0] + right_bot
        out = ((1.0 - frb) * top + frb * bot).astype(np.float32)
        lo = xq.min(axis=(2, 3), keepdims=True)
        hi = xq.max(axis=(2, 3), keepdims=True)
        out = np.clip(out, lo, hi)                                 # warp(clip=True): skimage/transform/_warps.py
    else:
        top = (1 - fcb) * x[:, :, r0][:, :, :, c0] + fcb * x[:, :, r0][:, :, :, c1]
        bot = (1 - fcb) * x[:, :, r1][:, :, :, c0] + fcb * x[:, :, r1][:, :, :, c1]
        out = ((1 - frb) * top + frb * bot).astype(np.float32)
    return out * np.float32(30000)                                # float32 multiply, as patches.py:15


def _starts(extent, patch, stride):
    """Patch origins along one axis of the UNPADDED low-res extent (patches.py:45-53 / :114-122).

    ``extent // stride`` regular origins, plus one clamped origin when stride does not divide.
    In padded coordinates the clamped origin is  padded_extent - patch = extent + 2*border - patch.
    """
    k = extent // stride
    s = [i * stride for i in range(k)]
    return s, k


def _tile(dsets, scales, patch_sizes, borders):
    """Common body of get_test_patches / get_test_patches60.

    dsets[-1] is the lowest-resolution image; ``scales[i]`` = resolution ratio of dsets[i] to it;
    ``patch_sizes[i]`` / ``borders[i]`` = the per-resolution patch edge and pad (each obtained by the
    reference with its own floor division, patches.py:21-24 / :85-90).
    Returns the list of cropped (not yet up-sampled) NCHW float32 patch arrays.
    """
    padded = [np.pad(d, ((b, b), (b, b), (0, 0)), mode='symmetric') for d, b in zip(dsets, borders)]
    low = padded[-1]
    patch_lr, border_lr = patch_sizes[-1], borders[-1]
    stride = patch_lr - 2 * border_lr
    ext_i, ext_j = low.shape[0] - 2 * border_lr, low.shape[1] - 2 * border_lr
    si, ki = _starts(ext_i, patch_lr, stride)
    sj, kj = _starts(ext_j, patch_lr, stride)
    if ext_i % stride != 0:
        si.append(low.shape[0] - patch_lr)
    if ext_j % stride != 0:
        sj.append(low.shape[1] - patch_lr)
    n_alloc = (ki + 1) * (kj + 1)                      # patches.py:35 / :103 — always (k+1)^2
    outs = [np.zeros((n_alloc, d.shape[2], p, p), np.float32) for d, p in zip(padded, patch_sizes)]
    count = 0
    for i0 in si:
        for j0 in sj:
            for o, d, s in zip(outs, padded, scales):
                crop = d[i0 * s:(i0 + patch_lr) * s, j0 * s:(j0 + patch_lr) * s]
                o[count] = np.moveaxis(crop, 2, 0)
            count += 1
    return outs


def get_test_patches(dset_10, dset_20, patchSize=128, border=4, interp=True, f32_coords=False):
    """patches.py:19-80 -> (image_10 [N,4,P,P], data20 [N,6,P,P] (up-sampled if interp))."""
    p10, p20 = _tile([dset_10, dset_20], [2, 1], [patchSize, patchSize // 2], [border, border // 2])
    if interp:
        p20 = interp_patches(p20, p10.shape, f32_coords)
    return p10, p20


def get_test_patches60(dset_10, dset_20, dset_60, patchSize=128, border=8, interp=True, f32_coords=False):
    """patches.py:83-156 -> (image_10, data20, data60), all [N,*,P,P] when interp."""
    p10, p20, p60 = _tile([dset_10, dset_20, dset_60], [6, 3, 1],
                          [patchSize, patchSize // 2, patchSize // 6], [border, border // 2, border // 6])
    if interp:
        p20 = interp_patches(p20, p10.shape, f32_coords)
        p60 = interp_patches(p60, p10.shape, f32_coords)
    return p10, p20, p60


def recompose_images(a, border, size=None):
    """patches.py:374-405.  [N,C,P,P] -> [size0,size1,C] float32; N==1 returns a[0] uncropped (HWC)."""
    a = np.asarray(a)
    if a.shape[0] == 1:
        img = a[0]
    else:
        inner = a.shape[2] - 2 * border
        x_tiles = int(math.ceil(size[1] / float(inner)))
        y_tiles = int(math.ceil(size[0] / float(inner)))
        img = np.zeros((a.shape[1], size[0], size[1]), np.float32)
        k = 0
        for y in range(y_tiles):
            y0 = min(y * inner, size[0] - inner)
            for x in range(x_tiles):
                x0 = min(x * inner, size[1] - inner)
                img[:, y0:y0 + inner, x0:x0 + inner] = a[k, :, border:a.shape[2] - border, border:a.shape[3] - border]
                k += 1
    return img.transpose((1, 2, 0))
