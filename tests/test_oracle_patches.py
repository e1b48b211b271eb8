"""Pin oracle/patches_oracle.py against outputs of the reference's own utils/patches.py
(tests/golden/*.npz, produced by tests/golden/make_golden_patches.py)."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import patches_oracle as po

CASES20 = ['patches_20_div.npz', 'patches_20_nondiv.npz', 'patches_20_b8.npz']
CASES60 = ['patches_60_div.npz', 'patches_60_nondiv.npz', 'patches_60_b12.npz']

# Two checks of the up-sampler against the captured outputs of the reference's interp_patches (scikit-image 0.18.3):
#  * the oracle's f32_coords mode restates skimage's float32 arithmetic operation by operation: BIT-IDENTICAL;
#  * the exact-coordinate mode (the mathematical definition): coordinate error <= ~2e-6 of a pixel times the local
#    gradient (<= 13110 / pixel) -> 0.03 raw reflectance units = 1.5e-5 after /2000 (gate: 1e-4 RMSE).
LOOSE = dict(rtol=0, atol=3e-2)


from bits import assert_same_bits      # noqa: E402


def check_interp(fn, expect, err_msg=''):
    assert_same_bits(fn(True), expect, err_msg + ' (skimage arithmetic)')
    np.testing.assert_allclose(fn(False), expect, err_msg=err_msg + ' (exact coords)', **LOOSE)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize('name', CASES20)
def test_get_test_patches_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    d10, d20 = g['d10'].astype(np.float32), g['d20'].astype(np.float32)
    patch, border = int(g['patch']), int(g['border'])
    p10, p20 = po.get_test_patches(d10, d20, patchSize=patch, border=border)
    assert p10.dtype == np.float32 and p20.dtype == np.float32
    assert np.array_equal(p10, g['p10'])                       # tiling is a pure copy: bit-exact
    _, raw = po.get_test_patches(d10, d20, patchSize=patch, border=border, interp=False)
    assert np.array_equal(raw, g['p20_raw'])
    check_interp(lambda f: po.get_test_patches(d10, d20, patchSize=patch, border=border, f32_coords=f)[1], g['p20'])


@pytest.mark.parametrize('name', CASES60)
def test_get_test_patches60_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    d = [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]
    patch, border = int(g['patch']), int(g['border'])
    p10, p20, p60 = po.get_test_patches60(*d, patchSize=patch, border=border)
    assert np.array_equal(p10, g['p10'])
    _, r20, r60 = po.get_test_patches60(*d, patchSize=patch, border=border, interp=False)
    assert np.array_equal(r20, g['p20_raw']) and np.array_equal(r60, g['p60_raw'])
    check_interp(lambda f: po.get_test_patches60(*d, patchSize=patch, border=border, f32_coords=f)[1], g['p20'])
    check_interp(lambda f: po.get_test_patches60(*d, patchSize=patch, border=border, f32_coords=f)[2], g['p60'])


@pytest.mark.parametrize('name', CASES20 + CASES60)
def test_recompose_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    rec = quiet(po.recompose_images, g['pred'], border=int(g['border']), size=g['d10'].shape)
    assert rec.dtype == np.float32 and rec.shape == g['rec'].shape
    assert np.array_equal(rec, g['rec'])


@pytest.mark.parametrize('name', CASES20)
def test_recompose_of_tiling_is_identity(golden_dir, name):
    """recompose(get_test_patches(d10,...)[0]) == d10 exactly."""
    g = load(golden_dir, name)
    d10 = g['d10'].astype(np.float32)
    p10, _ = po.get_test_patches(d10, g['d20'].astype(np.float32), patchSize=int(g['patch']), border=int(g['border']))
    rec = quiet(po.recompose_images, p10, border=int(g['border']), size=d10.shape)
    assert np.array_equal(rec, d10)
    assert np.array_equal(rec, g['rec_identity'])


def test_trailing_patches_are_zero_when_stride_divides(golden_dir):
    """patches.py:35 allocates (k+1)^2 patches; with a dividing stride the trailing ones stay zero."""
    g = load(golden_dir, 'patches_20_div.npz')
    p10, p20 = po.get_test_patches(g['d10'].astype(np.float32), g['d20'].astype(np.float32),
                                   patchSize=int(g['patch']), border=int(g['border']))
    assert p10.shape[0] == 16 and not p10[9:].any() and not p20[9:].any()
    assert p10[:9].all()


def test_single_patch_is_returned_uncropped(golden_dir):
    g = load(golden_dir, 'recompose_single.npz')
    rec = po.recompose_images(g['pred'], border=4, size=(24, 24, 4))
    assert rec.shape == (32, 32, 6) and np.array_equal(rec, g['rec'])


def test_interp_patches_matches_reference(golden_dir):
    g = load(golden_dir, 'interp.npz')
    for src, key, shape in [('ramp', 'ramp_x2', (1, 1, 8, 8)), ('ramp', 'ramp_x6', (1, 1, 24, 24)),
                            ('a', 'a_x2', (3, 2, 32, 32)), ('a', 'a_x6', (3, 2, 96, 96)),
                            ('b', 'b_x2', (2, 3, 10, 14)), ('b', 'b_x6', (2, 3, 30, 42))]:
        out = po.interp_patches(g[src], shape)
        assert out.dtype == np.float32 and out.shape == g[key].shape
        check_interp(lambda f: po.interp_patches(g[src], shape, f32_coords=f), g[key], key)


def test_mirror_bilinear_ramp_known_answer():
    """ramp 0 10 20 30, x2 -> 2.5 2.5 7.5 ... 27.5 27.5 (mirror, not clamp)."""
    ramp = np.array([[0, 10, 20, 30]], np.float32).repeat(4, axis=0)[None, None]
    x2 = po.interp_patches(ramp, (1, 1, 8, 8))[0, 0, 0]
    np.testing.assert_allclose(x2, [2.5, 2.5, 7.5, 12.5, 17.5, 22.5, 27.5, 27.5], rtol=1e-6)
    x6 = po.interp_patches(ramp, (1, 1, 24, 24))[0, 0, 0]
    np.testing.assert_allclose(x6[:5], [25 / 6, 2.5, 5 / 6, 5 / 6, 2.5], rtol=1e-5)


def test_real_tile_crop_default_geometry(golden_dir):
    """Default geometry of testing/supres.py (128/8 and 192/12) on a crop of the bundled T33UUB tile."""
    g = load(golden_dir, 'tile_T33UUB_crop.npz')
    d = [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]
    p10, p20 = po.get_test_patches(d[0], d[1], patchSize=128, border=8, f32_coords=True)
    assert p10.shape[0] == int(g['n20']) == 9
    np.testing.assert_array_equal(p10.astype(np.float64).sum(axis=(2, 3)), g['p10_sum'])
    sub = (slice(None), slice(None), slice(3, None, 7), slice(2, None, 5))
    assert_same_bits(p20[sub], g['p20_sub'])
    assert_same_bits(p20[4, :2], g['p20_patch4'])
    np.testing.assert_allclose(p20.astype(np.float64).sum(axis=(2, 3)), g['p20_sum'], rtol=1e-6)
    q10, q20, q60 = po.get_test_patches60(*d, patchSize=192, border=12, f32_coords=True)
    assert q10.shape[0] == int(g['n60']) == 4
    np.testing.assert_array_equal(q10.astype(np.float64).sum(axis=(2, 3)), g['q10_sum'])
    assert_same_bits(q20[sub], g['q20_sub'])
    assert_same_bits(q60[sub], g['q60_sub'])
    assert_same_bits(q60[0, :1], g['q60_patch0'])
    # round trip on the real data
    rec = quiet(po.recompose_images, p10, border=8, size=d[0].shape)
    assert np.array_equal(rec, d[0])


@pytest.mark.parametrize('name', ['tile_T33UUB_600.npz', 'tile_T49JGM_600.npz'])
def test_whole_bundled_tiles(golden_dir, name):
    """The two tiles the reference ships, whole (600x600 @10 m): 36 / 16 patches, sums, subsamples, the
    clamped last patch, and recompose_images of the up-sampled patches — all as captured from utils/patches.py."""
    g = load(golden_dir, name)
    d = [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]
    assert d[0].shape == (600, 600, 4) and d[1].shape == (300, 300, 6) and d[2].shape == (100, 100, 2)
    sub = (slice(None), slice(None), slice(3, None, 7), slice(2, None, 5))
    sums = lambda a: a.astype(np.float64).sum(axis=(2, 3))
    p10, p20 = po.get_test_patches(d[0], d[1], patchSize=128, border=8, f32_coords=True)
    assert p10.shape[0] == int(g['n20']) == 36
    np.testing.assert_array_equal(sums(p10), g['p10_sum'])
    assert np.array_equal(p10[-1], g['p10_last'])
    assert_same_bits(p20[sub], g['p20_sub'])
    assert_same_bits(p20[-1], g['p20_last'])
    np.testing.assert_allclose(sums(p20), g['p20_sum'], rtol=1e-6)
    q10, q20, q60 = po.get_test_patches60(*d, patchSize=192, border=12, f32_coords=True)
    assert q10.shape[0] == int(g['n60']) == 16
    np.testing.assert_array_equal(sums(q10), g['q10_sum'])
    assert_same_bits(q20[sub], g['q20_sub'])
    assert_same_bits(q60[sub], g['q60_sub'])
    assert_same_bits(q20[-1, :2], g['q20_last'])
    assert_same_bits(q60[-1], g['q60_last'])
    assert np.array_equal(quiet(po.recompose_images, p10, border=8, size=d[0].shape), d[0])
    assert np.array_equal(quiet(po.recompose_images, q10, border=12, size=d[0].shape), d[0])
    rec20 = quiet(po.recompose_images, p20, border=8, size=d[0].shape)
    assert_same_bits(rec20[1::5, 2::7], g['rec20_sub'])
    np.testing.assert_allclose(rec20.astype(np.float64).sum(axis=1), g['rec20_rows'], rtol=1e-6)
    rec60 = quiet(po.recompose_images, q60, border=12, size=d[0].shape)
    assert_same_bits(rec60[1::5, 2::7], g['rec60_sub'])
    np.testing.assert_allclose(rec60.astype(np.float64).sum(axis=0), g['rec60_cols'], rtol=1e-6)


def test_a_patch_of_a_large_tile_is_a_patch_of_an_aligned_crop():
    """The method tests/test_gpu_full_tile.py uses to check windows of a 10980^2 run without tiling 9801 patches on the CPU:
    an aligned 336 x 336 crop of the tile has, in ITS patch grid, exactly the full grid's first / interior / clamped last
    patch.  Shown here with the oracle's own tiling of a whole (smaller) tile whose size, like 10980 = 98 * 112 + 4, is not a
    multiple of the stride."""
    n = 112 * 9 + 100
    rng = np.random.default_rng(0)
    d10 = rng.integers(35, 13110, size=(n, n, 4)).astype(np.float32)
    d20 = rng.integers(35, 13110, size=(n // 2, n // 2, 6)).astype(np.float32)
    full10, full20 = po.get_test_patches(d10, d20, patchSize=128, border=8, interp=False)
    per = n // 2 // 56 + 1                                   # used patches per axis (the clamped one included)
    assert full10.shape[0] == per * per == 100

    def crop(r0):
        return po.get_test_patches(d10[r0:r0 + 336, r0:r0 + 336], d20[r0 // 2:(r0 + 336) // 2, r0 // 2:(r0 + 336) // 2],
                                   patchSize=128, border=8, interp=False)
    for r0, pick, idx in [(0, 0, 0), (3 * 112, 4, 4 * per + 4), (n - 336, 8, per * per - 1)]:
        c10, c20 = crop(r0)
        assert c10.shape[0] == 16
        assert np.array_equal(c10[pick], full10[idx]) and np.array_equal(c20[pick], full20[idx]), (r0, pick, idx)


def test_interp_patches_bit_exact_beyond_the_tile_path_factors(golden_dir):
    """tests/golden/interp_shapes.npz (make_golden_interp_shapes.py): the reference's interp_patches on non-integer factors, odd
    / tiny planes, plateaus at a plane's extremes, a constant plane, values up to 65535 — the oracle's restatement of
    scikit-image 0.18.3's float32 arithmetic gives the same bits, with and without warp()'s clip (a no-op for it)."""
    g = load(golden_dir, 'interp_shapes.npz')
    n = len([k for k in g.files if k.startswith('in_')])
    assert n >= 12
    for k in range(n):
        x, want = g['in_%02d' % k], g['out_%02d' % k]
        got = po.interp_patches(x, want.shape, f32_coords=True)
        assert_same_bits(got, want, 'case %d %r -> %r' % (k, x.shape[2:], want.shape[2:]))
        assert_same_bits(want[1, 1], np.full(want.shape[2:], 4321, np.float32), 'constant plane, case %d' % k)
