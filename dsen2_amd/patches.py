"""Host-side mirror of the inference half of the reference's utils/patches.py, running on the GPU.

Same function names, arguments, defaults, return types and quirks as the reference:
  interp_patches(image_20, image_10_shape)                              utils/patches.py:11-16
  get_test_patches(dset_10, dset_20, patchSize=128, border=4, interp)   utils/patches.py:19-80
  get_test_patches60(dset_10, dset_20, dset_60, patchSize=128, border=8, interp)   :83-156
  recompose_images(a, border, size=None)                                utils/patches.py:374-405
numpy in, numpy out; the ``*_device`` variants keep everything in HBM (torch CUDA tensors) so that
supres.DSen2_20/60 never bounce patches through the host.  The gathers, the mirror-bilinear
up-sampling and the recomposition are HIP kernels behind the C ABI (include/dsen2_hip.h); only the
O(#patches) origin arithmetic runs on the host.
"""
import ctypes
from math import ceil

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('dsen2_amd needs a ROCm GPU (gfx950); there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _storage_plan(a):
    """How a non-C-contiguous ndarray can be uploaded WITHOUT a strided pass over it on the host.

    The reference's own callers hand over views: `np.rollaxis(ds.ReadAsArray(...), 0, 3)` (testing/s2_tiles_supres.py: an HWC
    view of a CHW array), `f['im10'][()].transpose()` (testing/demoDSen2.py:16: Fortran order), and under torch.distributed a
    row slab `d[r0:r1]` of either.  numpy needs seconds to make a 10980^2 x 4 view of that kind C-contiguous — longer than the
    GPU needs for the whole tile in the bf16 modes — while the storage underneath is contiguous as it is.  Returns
    (order, mode): `a.transpose(order)` lists the axes by decreasing stride; mode 'whole' = that array is C-contiguous (one
    copy), 'planes' = each `a.transpose(order)[k]` is (a row slab of a plane-major array: one copy per plane), None = neither
    (the caller falls back to np.ascontiguousarray)."""
    if a.ndim < 2 or a.size == 0 or any(st <= 0 for st in a.strides):
        return None, None
    order = tuple(int(i) for i in np.argsort([-st for st in a.strides], kind='stable'))
    base = a.transpose(order)
    if base.flags.c_contiguous:
        return order, 'whole'
    if base.shape[0] <= 64 and base[0].flags.c_contiguous:
        return order, 'planes'
    return None, None


_TORCH_BITS = {np.uint16: torch.int16, np.int16: torch.int16, np.uint8: torch.uint8, np.int8: torch.int8, np.int32: torch.int32,
               np.int64: torch.int64, np.float32: torch.float32, np.float64: torch.float64}


def _widen(t, np_dtype):
    """The uploaded integer tensor (uint16 travels as int16 bits) -> float32, on the GPU."""
    if np_dtype == np.uint16:
        return (t.to(torch.int32) & 0xFFFF).to(torch.float32)
    return t.to(torch.float32)


def _to_device_f32(a, device):
    """Any real array -> contiguous float32 CUDA tensor (the reference's float32 patch arrays take any dtype).
    Integer rasters (Sentinel-2 L1C is uint16) are uploaded as they are and widened on the GPU: half the PCIe
    bytes and no host-side conversion pass (0.15 s of a 10980^2 tile).  Views whose storage is contiguous in another axis
    order (_storage_plan) are uploaded in THAT order and permuted on the GPU."""
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    a = np.asarray(a)
    direct = a.dtype.type in _TORCH_BITS and a.dtype.isnative
    if direct and not a.flags.c_contiguous:
        order, mode = _storage_plan(a)
        if mode is not None:
            base = a.transpose(order)
            if not base.flags.writeable:
                base = None if mode == 'whole' else base       # (read-only maps: the plane copies below make their own arrays)
            if base is not None:
                as_bits = (lambda x: x.view(np.int16)) if a.dtype == np.uint16 else (lambda x: x)
                if mode == 'whole':
                    t = torch.from_numpy(as_bits(base)).to(device)
                else:
                    t = torch.empty(base.shape, dtype=_TORCH_BITS[a.dtype.type], device=device)
                    for k in range(base.shape[0]):
                        plane = base[k] if base[k].flags.writeable else np.array(base[k])
                        t[k].copy_(torch.from_numpy(as_bits(plane)))
                inverse = [order.index(i) for i in range(a.ndim)]
                # permuted at the storage width (2 B for a Sentinel-2 raster), widened after
                return _widen(t.permute(*inverse).contiguous(), a.dtype)
    a = np.ascontiguousarray(a)
    if not a.flags.writeable:
        a = np.array(a)           # a read-only memory map (cli._load under torch.distributed): copy the slab, not a view torch would warn about
    if a.dtype == np.uint16:
        return _widen(torch.from_numpy(a.view(np.int16)).to(device), np.uint16)
    if direct:
        # converted on the GPU with the same round-to-nearest as numpy's cast into the reference's float32 patch arrays (a
        # float64 raster: twice the PCIe bytes, but no pass over it on the host)
        return torch.from_numpy(a).to(device).to(torch.float32)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


# ---- interp_patches ---------------------------------------------------------------------------
def interp_patches_device(image_lr, hr_hw, post_divisor=1.0, ref=False):
    """[N,C,h,w] float32 CUDA tensor -> [N,C,H,W]; optionally folds the later ``/= SCALE``.  ref=True: the general
    kernel whatever the scale (dsen2_upsample_mirror_bilinear_ref, kernel-level cross-check)."""
    n, c, h, w = image_lr.shape
    oh, ow = int(hr_hw[0]), int(hr_hw[1])
    out = torch.empty((n, c, oh, ow), dtype=torch.float32, device=image_lr.device)
    with torch.cuda.device(image_lr.device):
        _lib.call('dsen2_upsample_mirror_bilinear_ref' if ref else 'dsen2_upsample_mirror_bilinear', _ptr(image_lr), _ptr(out), n * c, h, w, oh, ow,
                  float(post_divisor), _stream(image_lr.device))
    return out


def interp_patches(image_20, image_10_shape):
    """utils/patches.py:11-16 — bilinear (order 1) resize with mirror boundary, per patch and band."""
    dev = default_device()
    x = _to_device_f32(image_20, dev)
    return interp_patches_device(x, image_10_shape[2:4]).cpu().numpy()


# ---- tiling -----------------------------------------------------------------------------------
def _axis_origins(extent, patch, border):
    """Patch origins (padded low-res coordinates) along one axis: patches.py:45-53 / :114-122."""
    stride = patch - 2 * border
    if stride <= 0:
        raise ValueError('border %d leaves no interior in a patch of %d' % (border, patch))
    if extent + 2 * border < patch:
        # the reference would index with a negative origin here and fail on a shape mismatch
        raise ValueError('image extent %d (+2*%d border) is smaller than one patch of %d' % (extent, border, patch))
    k = extent // stride
    starts = [i * stride for i in range(k)]
    if extent % stride != 0:
        starts.append(extent + 2 * border - patch)      # = padded_extent - patch
    return starts, k


def tile_origins(lr_shape, patch_lr, border_lr):
    """(origins [used,2] int32 in padded LR coordinates (row-major: i outer, j inner), n_alloc).

    n_alloc = (k_i+1)*(k_j+1) is what the reference allocates (patches.py:35/:103); when the stride
    divides an extent only ``used`` < n_alloc patches are filled and the rest stay zero.
    """
    si, ki = _axis_origins(int(lr_shape[0]), patch_lr, border_lr)
    sj, kj = _axis_origins(int(lr_shape[1]), patch_lr, border_lr)
    org = np.array([(i, j) for i in si for j in sj], dtype=np.int32).reshape(-1, 2)
    return org, (ki + 1) * (kj + 1)


def gather_patches_device(img_dev, origins_lr, scale, border, patch, n_alloc, divisor=1.0, first=0, count=None,
                          origins_dev=None):
    """Crop patches [first, first+count) of one resolution from the (virtually symmetric-padded) image.

    img_dev: [H,W,C] float32 CUDA tensor; origins_lr: int32 [used,2] in padded low-res coordinates;
    ``scale`` = resolution ratio to the low-res grid (crop origin and size multiply by it,
    patches.py:67 / :136-137).  Returns [count_or_alloc, C, patch, patch]; rows beyond the used
    patches are zero as in the reference.  ``origins_dev``: the same origins already multiplied by ``scale`` as an
    int32 [used,2] tensor on the device (``device_origins``) — callers that loop over batches pass it so that no
    call uploads anything (a pageable host-to-device copy blocks the host until the stream has drained).
    """
    H, W, C = img_dev.shape
    used = origins_lr.shape[0]
    if count is None:
        first, count, n_out = 0, used, n_alloc
    else:
        n_out = count
    out = torch.empty((n_out, C, patch, patch), dtype=torch.float32, device=img_dev.device)
    if n_out > count:
        out[count:].zero_()          # the reference's trailing never-filled patches (patches.py:35)
    if count > 0:
        if origins_dev is not None:
            org = origins_dev[first:first + count]
        else:
            org = device_origins(origins_lr[first:first + count], scale, img_dev.device)
        with torch.cuda.device(img_dev.device):
            _lib.call('dsen2_tile_gather', _ptr(img_dev), H, W, C, border, _ptr(org), count, patch, float(divisor),
                      _ptr(out), _stream(img_dev.device))
    return out


def device_origins(origins_lr, scale, device):
    """int32 [n,2] crop origins at one resolution (low-res origins x scale) as a device tensor."""
    return torch.from_numpy(np.ascontiguousarray(origins_lr * scale, dtype=np.int32)).to(device)


def _check_ratio(hi, lo, scale):
    if hi.shape[0] < scale * lo.shape[0] or hi.shape[1] < scale * lo.shape[1]:
        # the reference's crop loop fails with a broadcasting ValueError here (patches.py:67, :136-137)
        raise ValueError('image of shape %r does not cover %d x the lower-resolution image %r'
                         % (tuple(hi.shape), scale, tuple(lo.shape)))


def get_test_patches(dset_10, dset_20, patchSize=128, border=4, interp=True):
    """utils/patches.py:19-80.  Returns (image_10 [N,B10,P,P], data20 [N,B20,P,P]) float32 ndarrays."""
    _check_ratio(dset_10, dset_20, 2)
    dev = default_device()
    p_lr, b_lr = patchSize // 2, border // 2
    d10, d20 = _to_device_f32(dset_10, dev), _to_device_f32(dset_20, dev)
    org, n_alloc = tile_origins(d20.shape, p_lr, b_lr)
    image_10 = gather_patches_device(d10, org, 2, border, patchSize, n_alloc)
    image_20 = gather_patches_device(d20, org, 1, b_lr, p_lr, n_alloc)
    data20 = interp_patches_device(image_20, image_10.shape[2:4]) if interp else image_20
    return image_10.cpu().numpy(), data20.cpu().numpy()


def get_test_patches60(dset_10, dset_20, dset_60, patchSize=128, border=8, interp=True):
    """utils/patches.py:83-156.  Returns (image_10, data20, data60) float32 ndarrays."""
    _check_ratio(dset_10, dset_60, 6)
    _check_ratio(dset_20, dset_60, 3)
    dev = default_device()
    p20, p60 = patchSize // 2, patchSize // 6
    b20, b60 = border // 2, border // 6
    d10, d20, d60 = (_to_device_f32(a, dev) for a in (dset_10, dset_20, dset_60))
    org, n_alloc = tile_origins(d60.shape, p60, b60)
    image_10 = gather_patches_device(d10, org, 6, border, patchSize, n_alloc)
    image_20 = gather_patches_device(d20, org, 3, b20, p20, n_alloc)
    image_60 = gather_patches_device(d60, org, 1, b60, p60, n_alloc)
    if interp:
        image_20 = interp_patches_device(image_20, image_10.shape[2:4])
        image_60 = interp_patches_device(image_60, image_10.shape[2:4])
    return image_10.cpu().numpy(), image_20.cpu().numpy(), image_60.cpu().numpy()


# ---- recomposition ----------------------------------------------------------------------------
def recompose_grid(size, patch, border):
    inner = patch - 2 * border
    return int(ceil(size[1] / float(inner))), int(ceil(size[0] / float(inner)))    # x_tiles, y_tiles


def final_row_runs(have, done_rows, size, inner):
    """Which image rows can be recomposed now.  `have`: bool per patch (row-major patch index, x_tiles * y_tiles of them) —
    the patches that are final; `done_rows`: bool per TILE ROW, rows already recomposed (updated in place).  A tile row is
    ready when all its patches are; tile row ty decides image rows [ty * inner, min((ty + 1) * inner, H - inner)), the last
    one [H - inner, H) (its start is clamped, patches.py:396-401, and it overwrites what the row before it put there).
    Returns merged [(row0, row1)] runs in ascending order."""
    H = int(size[0])
    x_tiles = int(ceil(size[1] / float(inner)))
    y_tiles = int(ceil(H / float(inner)))
    ready = np.asarray(have[:x_tiles * y_tiles], bool).reshape(y_tiles, x_tiles).all(axis=1) & ~done_rows
    runs = []
    for ty in np.nonzero(ready)[0]:
        r0, r1 = (H - inner, H) if ty == y_tiles - 1 else (int(ty) * inner, min((int(ty) + 1) * inner, H - inner))
        if runs and runs[-1][1] == r0:
            runs[-1] = (runs[-1][0], r1)
        else:
            runs.append((r0, r1))
        done_rows[ty] = True
    return runs


def recompose_device(a_dev, border, size, scale=1.0):
    """[N,C,P,P] CUDA tensor -> [size0,size1,C] CUDA tensor (N > 1)."""
    n, c, p, _ = a_dev.shape
    H, W = int(size[0]), int(size[1])
    img = torch.empty((H, W, c), dtype=torch.float32, device=a_dev.device)
    with torch.cuda.device(a_dev.device):
        _lib.call('dsen2_recompose', _ptr(a_dev), n, c, p, border, _ptr(img), H, W, float(scale),
                  _stream(a_dev.device))
    return img


def recompose_rows_device(a_dev, border, img, row0, row1, scale=1.0):
    """Rows [row0, row1) of `img` ([H,W,C] CUDA tensor) from the patch buffer a_dev [N,C,P,P] (dsen2_recompose_rows): only the
    patches those rows read need to be final."""
    n, c, p, _ = a_dev.shape
    H, W = int(img.shape[0]), int(img.shape[1])
    with torch.cuda.device(a_dev.device):
        _lib.call('dsen2_recompose_rows', _ptr(a_dev), n, c, p, border, _ptr(img), H, W, float(scale), int(row0), int(row1),
                  _stream(a_dev.device))


def recompose_images(a, border, size=None):
    """utils/patches.py:374-405 — including the single-patch shortcut (:375-376) and the shape print (:392)."""
    if a.shape[0] == 1:
        images = np.asarray(a[0]) if not isinstance(a, torch.Tensor) else a[0].cpu().numpy()
        return images.transpose((1, 2, 0))
    dev = default_device()
    x = _to_device_f32(a, dev)
    print((a.shape[1], size[0], size[1]))
    return recompose_device(x, border, size).cpu().numpy()
