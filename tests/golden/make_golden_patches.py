#!/opt/conda/bin/python3.9
"""Generate tests/golden/patches_*.npz by RUNNING THE REFERENCE'S OWN utils/patches.py.

Run in the build container only (the reference never travels to the GPU box):
    /opt/conda/bin/python3.9 tests/golden/make_golden_patches.py
Environment at capture time: numpy 1.26.4, scikit-image 0.18.3, h5py 3.3.0.

The .npz files hold DATA only: seeded integer-valued inputs and the arrays the reference's
get_test_patches / get_test_patches60 / interp_patches / recompose_images returned for them.
The real-tile cases use /root/reference/data/S2A_MSIL1C_20170527_T33UUB.mat (a 264x264 crop, and the whole
600x600 tile) and S2B_MSIL1C_20171022_T49JGM.mat (whole) — Copernicus Sentinel data, CC BY 4.0, see the
reference's data/LICENSE.md — stored as uint16.  `make_golden_patches.py bundled` regenerates only the whole tiles.
"""
import contextlib
import io
import os
import sys

import numpy as np

sys.path.insert(0, '/root/reference')
from utils.patches import (get_test_patches, get_test_patches60, interp_patches,  # noqa: E402
                           recompose_images)

HERE = os.path.dirname(os.path.abspath(__file__))


def synth(rng, h, w, bands):
    # integer-valued reflectances in the range seen in the bundled tiles (35..13109)
    return rng.integers(35, 13110, size=(h, w, bands)).astype(np.uint16)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):     # recompose_images prints the shape
        return fn(*a, **k)


def case20(name, rng, h, w, patch, border):
    d10 = synth(rng, h, w, 4)
    d20 = synth(rng, h // 2, w // 2, 6)
    p10, p20 = get_test_patches(d10.astype(np.float32), d20.astype(np.float32), patchSize=patch, border=border)
    _, p20_raw = get_test_patches(d10.astype(np.float32), d20.astype(np.float32), patchSize=patch,
                                  border=border, interp=False)
    pred = rng.standard_normal((p10.shape[0], 6, patch, patch)).astype(np.float32)
    rec = quiet(recompose_images, pred, border=border, size=d10.shape)
    rec_id = quiet(recompose_images, p10, border=border, size=d10.shape)
    np.savez_compressed(os.path.join(HERE, name), kind='20', d10=d10, d20=d20, patch=patch, border=border,
                        p10=p10, p20=p20, p20_raw=p20_raw, pred=pred, rec=rec, rec_identity=rec_id)
    print(name, p10.shape, p20.shape, rec.shape)


def case60(name, rng, h, w, patch, border):
    d10 = synth(rng, h, w, 4)
    d20 = synth(rng, h // 2, w // 2, 6)
    d60 = synth(rng, h // 6, w // 6, 2)
    f = [a.astype(np.float32) for a in (d10, d20, d60)]
    p10, p20, p60 = get_test_patches60(*f, patchSize=patch, border=border)
    _, p20_raw, p60_raw = get_test_patches60(*f, patchSize=patch, border=border, interp=False)
    pred = rng.standard_normal((p10.shape[0], 2, patch, patch)).astype(np.float32)
    rec = quiet(recompose_images, pred, border=border, size=d10.shape)
    np.savez_compressed(os.path.join(HERE, name), kind='60', d10=d10, d20=d20, d60=d60, patch=patch,
                        border=border, p10=p10, p20=p20, p60=p60, p20_raw=p20_raw, p60_raw=p60_raw,
                        pred=pred, rec=rec)
    print(name, p10.shape, p20.shape, p60.shape, rec.shape)


def case_interp(name, rng):
    out = {}
    ramp = np.array([[0, 10, 20, 30]], np.float32).repeat(4, axis=0)[None, None]
    out['ramp'] = ramp
    out['ramp_x2'] = interp_patches(ramp, (1, 1, 8, 8))
    out['ramp_x6'] = interp_patches(ramp, (1, 1, 24, 24))
    a = rng.integers(35, 13110, size=(3, 2, 16, 16)).astype(np.float32)
    out['a'] = a
    out['a_x2'] = interp_patches(a, (3, 2, 32, 32))
    out['a_x6'] = interp_patches(a, (3, 2, 96, 96))
    b = rng.integers(35, 13110, size=(2, 3, 5, 7)).astype(np.float32)      # non-square, odd
    out['b'] = b
    out['b_x2'] = interp_patches(b, (2, 3, 10, 14))
    out['b_x6'] = interp_patches(b, (2, 3, 30, 42))
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


def case_real_tile(name):
    import h5py
    with h5py.File('/root/reference/data/S2A_MSIL1C_20170527_T33UUB.mat', 'r') as f:
        im10 = np.array(f['im10']).transpose()      # testing/demoDSen2.py:14-28 readh5: CHW -> HWC
        im20 = np.array(f['im20']).transpose()
        im60 = np.array(f['im60']).transpose()
    assert im10.shape == (600, 600, 4) and im20.shape == (300, 300, 6) and im60.shape == (100, 100, 2)
    # 264x264 @10 m window aligned to the 60 m grid (multiples of 6), origin (120, 60)
    y0, x0, n = 120, 60, 264
    d10 = im10[y0:y0 + n, x0:x0 + n]
    d20 = im20[y0 // 2:(y0 + n) // 2, x0 // 2:(x0 + n) // 2]
    d60 = im60[y0 // 6:(y0 + n) // 6, x0 // 6:(x0 + n) // 6]
    for a in (d10, d20, d60):
        assert np.array_equal(a, np.round(a)) and a.min() >= 0 and a.max() < 65536
    # default geometry of testing/supres.py:21-22 and :40-41
    p10, p20 = get_test_patches(d10.astype(np.float32), d20.astype(np.float32), patchSize=128, border=8)
    q10, q20, q60 = get_test_patches60(d10.astype(np.float32), d20.astype(np.float32), d60.astype(np.float32),
                                       patchSize=192, border=12)
    sub = (slice(None), slice(None), slice(3, None, 7), slice(2, None, 5))   # strided subsample
    np.savez_compressed(
        os.path.join(HERE, name), d10=d10.astype(np.uint16), d20=d20.astype(np.uint16), d60=d60.astype(np.uint16),
        n20=p10.shape[0], n60=q10.shape[0],
        p10_sum=p10.astype(np.float64).sum(axis=(2, 3)), p20_sum=p20.astype(np.float64).sum(axis=(2, 3)),
        p20_sub=p20[sub], p20_patch4=p20[4, :2],
        q10_sum=q10.astype(np.float64).sum(axis=(2, 3)), q20_sum=q20.astype(np.float64).sum(axis=(2, 3)),
        q60_sum=q60.astype(np.float64).sum(axis=(2, 3)), q20_sub=q20[sub], q60_sub=q60[sub],
        q60_patch0=q60[0, :1])
    print(name, p10.shape, q10.shape)


BUNDLED = {'tile_T33UUB_600.npz': 'S2A_MSIL1C_20170527_T33UUB.mat',
           'tile_T49JGM_600.npz': 'S2B_MSIL1C_20171022_T49JGM.mat'}
SUB = (slice(None), slice(None), slice(3, None, 7), slice(2, None, 5))   # strided subsample of [N, C, H, W]


def case_bundled_tile(name, mat):
    """One of the two tiles the reference ships (data/*.mat, 600x600 @10 m), whole: the arrays themselves (uint16) and
    what the reference's own tiling / up-sampling / recomposition return for them at the geometry of
    testing/supres.py:21-22,40-41 — patch counts, per-(patch, band) sums, strided subsamples, the clamped last patch in
    full, and recompose_images of the up-sampled 20 m patches (row / column sums + a strided subsample)."""
    import h5py
    with h5py.File('/root/reference/data/' + mat, 'r') as f:
        im10 = np.array(f['im10']).transpose()      # testing/demoDSen2.py:14-28 readh5: CHW -> HWC
        im20 = np.array(f['im20']).transpose()
        im60 = np.array(f['im60']).transpose()
    assert im10.shape == (600, 600, 4) and im20.shape == (300, 300, 6) and im60.shape == (100, 100, 2)
    for a in (im10, im20, im60):
        assert np.array_equal(a, np.round(a)) and a.min() >= 0 and a.max() < 65536
    f10, f20, f60 = (a.astype(np.float32) for a in (im10, im20, im60))
    p10, p20 = get_test_patches(f10, f20, patchSize=128, border=8)
    q10, q20, q60 = get_test_patches60(f10, f20, f60, patchSize=192, border=12)
    rec10 = quiet(recompose_images, p10, border=8, size=im10.shape)          # == d10 exactly
    rec10_60 = quiet(recompose_images, q10, border=12, size=im10.shape)
    assert np.array_equal(rec10, f10) and np.array_equal(rec10_60, f10)
    rec20 = quiet(recompose_images, p20, border=8, size=im10.shape)          # [600, 600, 6] float32
    rec60 = quiet(recompose_images, q60, border=12, size=im10.shape)         # [600, 600, 2]
    sums = lambda a: a.astype(np.float64).sum(axis=(2, 3))
    np.savez_compressed(
        os.path.join(HERE, name), source=mat, d10=im10.astype(np.uint16), d20=im20.astype(np.uint16),
        d60=im60.astype(np.uint16), n20=p10.shape[0], n60=q10.shape[0],
        p10_sum=sums(p10), p20_sum=sums(p20), p20_sub=p20[SUB], p10_last=p10[-1], p20_last=p20[-1],
        q10_sum=sums(q10), q20_sum=sums(q20), q60_sum=sums(q60), q20_sub=q20[SUB], q60_sub=q60[SUB],
        q20_last=q20[-1, :2], q60_last=q60[-1],
        rec20_rows=rec20.astype(np.float64).sum(axis=1), rec20_cols=rec20.astype(np.float64).sum(axis=0),
        rec20_sub=rec20[1::5, 2::7], rec60_rows=rec60.astype(np.float64).sum(axis=1),
        rec60_cols=rec60.astype(np.float64).sum(axis=0), rec60_sub=rec60[1::5, 2::7])
    print(name, p10.shape, q10.shape, rec20.shape)


def case_single_patch(name, rng):
    pred = rng.standard_normal((1, 6, 32, 32)).astype(np.float32)
    rec = quiet(recompose_images, pred, border=4, size=(24, 24, 4))
    np.savez_compressed(os.path.join(HERE, name), pred=pred, rec=rec)
    print(name, rec.shape)


if __name__ == '__main__':
    if sys.argv[1:] == ['bundled']:          # only the two whole tiles (the other fixtures stay as committed)
        for name, mat in BUNDLED.items():
            case_bundled_tile(name, mat)
        sys.exit(0)
    rng = np.random.default_rng(20170527)
    case20('patches_20_div.npz', rng, 72, 72, 32, 4)        # stride divides: (k+1)^2 alloc, trailing zeros
    case20('patches_20_nondiv.npz', rng, 80, 92, 32, 4)     # clamped last row / column, non-square
    case20('patches_20_b8.npz', rng, 100, 76, 48, 8)        # other patch/border
    case60('patches_60_div.npz', rng, 108, 108, 48, 6)
    case60('patches_60_nondiv.npz', rng, 96, 132, 48, 6)
    case60('patches_60_b12.npz', rng, 144, 114, 96, 12)
    case_interp('interp.npz', rng)
    case_single_patch('recompose_single.npz', rng)
    case_real_tile('tile_T33UUB_crop.npz')
    for name, mat in BUNDLED.items():
        case_bundled_tile(name, mat)
