"""The two tiles the reference ships (data/S2A_MSIL1C_20170527_T33UUB.mat, data/S2B_MSIL1C_20171022_T49JGM.mat), WHOLE
(600x600 @10 m), through the drop-in surface on the GPU.

tests/golden/tile_*_600.npz hold the tiles (uint16; Copernicus Sentinel data, CC BY 4.0) and what the reference's own
utils/patches.py returned for them (tests/golden/make_golden_patches.py): 36 / 16 patches, per-patch sums,
strided subsamples, the clamped last patch, recompose_images of the up-sampled patches.  The network half
(testing/supres.py:15-50 on testing/demoDSen2.py:42-43,67-68's inputs) is compared with the oracle pipeline: oracle
tiling + float64 C oracle CNN + oracle recomposition, seeded he_uniform weights (the checkpoints are stripped from the
reference checkout)."""
import contextlib
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dsen2_oracle as do

TILES = ['tile_T33UUB_600.npz', 'tile_T49JGM_600.npz']
from bits import assert_same_bits          # noqa: E402  (the up-sampler reproduces the reference's captured bits)
SUB = (slice(None), slice(None), slice(3, None, 7), slice(2, None, 5))
RMSE_GATE_NORMALISED = 1e-4                 # BASELINE.md §2 (fp32, normalised domain)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def bands(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    return g, [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]


@pytest.fixture()
def model_dir(tmp_path, monkeypatch):
    from dsen2_amd import supres
    files = {}
    for stem, (cin, cout, seed) in {'s2_032_lr_1e-04': (10, 6, 31), 's2_030_lr_1e-05': (12, 2, 32)}.items():
        files[stem] = do.he_uniform_weights(cin, cout, 6, 128, seed=seed, bias_scale=0.02)
        np.save(str(tmp_path / (stem + '.npy')), files[stem])
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    supres.clear_model_cache()
    yield files
    supres.clear_model_cache()


@pytest.mark.parametrize('name', TILES)
def test_tiling_of_the_whole_bundled_tile_equals_the_reference_capture(golden_dir, name):
    """get_test_patches(128, 8) / get_test_patches60(192, 12) / recompose_images (utils/patches.py:19-156,374-405)."""
    from dsen2_amd import patches as gp
    g, d = bands(golden_dir, name)
    sums = lambda a: a.astype(np.float64).sum(axis=(2, 3))
    p10, p20 = gp.get_test_patches(d[0], d[1], patchSize=128, border=8)
    assert p10.shape == (36, 4, 128, 128) and p20.shape == (36, 6, 128, 128) and int(g['n20']) == 36
    np.testing.assert_array_equal(sums(p10), g['p10_sum'])
    assert np.array_equal(p10[-1], g['p10_last'])                        # the clamped last patch, bit for bit
    assert_same_bits(p20[SUB], g['p20_sub'])
    assert_same_bits(p20[-1], g['p20_last'])
    np.testing.assert_allclose(sums(p20), g['p20_sum'], rtol=1e-6)
    q10, q20, q60 = gp.get_test_patches60(*d, patchSize=192, border=12)
    assert q10.shape == (16, 4, 192, 192) and q60.shape == (16, 2, 192, 192) and int(g['n60']) == 16
    np.testing.assert_array_equal(sums(q10), g['q10_sum'])
    assert_same_bits(q20[SUB], g['q20_sub'])
    assert_same_bits(q60[SUB], g['q60_sub'])
    assert_same_bits(q20[-1, :2], g['q20_last'])
    assert_same_bits(q60[-1], g['q60_last'])
    np.testing.assert_allclose(sums(q60), g['q60_sum'], rtol=1e-6)
    # recompose_images: the 10 m patches give the tile back exactly (both geometries); the up-sampled patches give
    # the image the reference's recompose_images gave
    assert np.array_equal(quiet(gp.recompose_images, p10, border=8, size=d[0].shape), d[0])
    assert np.array_equal(quiet(gp.recompose_images, q10, border=12, size=d[0].shape), d[0])
    rec20 = quiet(gp.recompose_images, p20, border=8, size=d[0].shape)
    assert rec20.shape == (600, 600, 6) and rec20.dtype == np.float32
    assert_same_bits(rec20[1::5, 2::7], g['rec20_sub'])
    np.testing.assert_allclose(rec20.astype(np.float64).sum(axis=1), g['rec20_rows'], rtol=1e-6)
    np.testing.assert_allclose(rec20.astype(np.float64).sum(axis=0), g['rec20_cols'], rtol=1e-6)
    rec60 = quiet(gp.recompose_images, q60, border=12, size=d[0].shape)
    assert_same_bits(rec60[1::5, 2::7], g['rec60_sub'])
    np.testing.assert_allclose(rec60.astype(np.float64).sum(axis=1), g['rec60_rows'], rtol=1e-6)
    np.testing.assert_allclose(rec60.astype(np.float64).sum(axis=0), g['rec60_cols'], rtol=1e-6)


@pytest.mark.parametrize('name', TILES)
def test_dsen2_20_on_the_whole_bundled_tile(golden_dir, model_dir, oracle_dsen2_tile, name):
    """DSen2_20(im10, im20) as testing/demoDSen2.py:42-43 calls it: all 36 patches against the float64 oracle."""
    from dsen2_amd.supres import DSen2_20
    _, d = bands(golden_dir, name)
    out = quiet(DSen2_20, d[0], d[1], deep=False)
    assert out.shape == (600, 600, 6) and out.dtype == np.float32 and np.isfinite(out).all()
    ref = oracle_dsen2_tile(name, model_dir['s2_032_lr_1e-04'])
    err = do.rmse(out, ref) / 2000
    print('%s DSen2_20: normalised rmse %.3e' % (name, err))
    assert err < RMSE_GATE_NORMALISED


@pytest.mark.parametrize('name', TILES)
def test_dsen2_60_on_the_whole_bundled_tile(golden_dir, model_dir, oracle_dsen2_tile, name):
    """DSen2_60(im10, im20, im60) as testing/demoDSen2.py:67-68 calls it: all 16 patches against the float64 oracle."""
    from dsen2_amd.supres import DSen2_60
    _, d = bands(golden_dir, name)
    out = quiet(DSen2_60, d[0], d[1], d[2], deep=False)
    assert out.shape == (600, 600, 2) and out.dtype == np.float32 and np.isfinite(out).all()
    ref = oracle_dsen2_tile(name, model_dir['s2_030_lr_1e-05'], run_60=True)
    err = do.rmse(out, ref) / 2000
    print('%s DSen2_60: normalised rmse %.3e' % (name, err))
    assert err < RMSE_GATE_NORMALISED
