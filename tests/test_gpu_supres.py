"""End-to-end drop-in surface: DSen2_20 / DSen2_60 / _predict (testing/supres.py) on the GPU vs the
oracle pipeline (oracle tiling + float64 oracle CNN + oracle recomposition)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import dsen2_oracle as do
from oracle import patches_oracle as po

RMSE_GATE_NORMALISED = 1e-4     # fp32 gate, in the network's normalised (/2000) domain


@pytest.fixture()
def model_dir(tmp_path, monkeypatch):
    """A MDL_PATH holding synthetic checkpoints under the reference's file names (as .npy)."""
    from dsen2_amd import supres
    files = {}
    for stem, (cin, cout, d, f, seed) in {
        's2_032_lr_1e-04': (10, 6, 6, 128, 11), 's2_030_lr_1e-05': (12, 2, 6, 128, 12),
        's2_033_lr_1e-04': (10, 6, 32, 256, 13), 's2_034_lr_1e-04': (12, 2, 32, 256, 14),
    }.items():
        if d == 32:
            continue                        # VDSen2 oracle at float64 is too slow for a unit test
        flat = do.he_uniform_weights(cin, cout, d, f, seed=seed, bias_scale=0.02)
        np.save(str(tmp_path / (stem + '.npy')), flat)
        files[stem] = flat
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    supres.clear_model_cache()
    yield files
    supres.clear_model_cache()


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def oracle_dsen2_20(d10, d20, flat):
    p10, p20 = po.get_test_patches(d10, d20, patchSize=128, border=8, f32_coords=True)
    p10 = p10 / np.float32(2000); p20 = p20 / np.float32(2000)
    used = int(np.ceil(d10.shape[0] / 112.0) * np.ceil(d10.shape[1] / 112.0))
    pred = np.zeros((p10.shape[0], 6, 128, 128))
    pred[:used] = c_oracle.forward([p10[:used], p20[:used]], flat, 6, 128)
    with contextlib.redirect_stdout(io.StringIO()):
        img = po.recompose_images(pred, border=8, size=d10.shape)
    return img.astype(np.float64) * 2000


def test_dsen2_20_matches_oracle_pipeline(model_dir):
    from dsen2_amd.supres import DSen2_20
    rng = np.random.default_rng(1)
    d10 = rng.integers(35, 6000, size=(240, 150, 4)).astype(np.float32)     # non-dividing: clamped last tiles
    d20 = rng.integers(35, 6000, size=(120, 75, 6)).astype(np.float32)
    keep10, keep20 = d10.copy(), d20.copy()
    out, printed = quiet(DSen2_20, d10, d20, deep=False)
    assert out.shape == (240, 150, 6) and out.dtype == np.float32
    assert np.array_equal(d10, keep10) and np.array_equal(d20, keep20)      # caller's arrays untouched
    assert 'Symbolic Model Created.' in printed and 's2_032_lr_1e-04.hdf5' in printed
    ref = oracle_dsen2_20(d10, d20, model_dir['s2_032_lr_1e-04'])
    err = do.rmse(out, ref) / 2000
    print('DSen2_20 normalised rmse', err)
    assert err < RMSE_GATE_NORMALISED


def test_dsen2_60_matches_oracle_pipeline(model_dir):
    from dsen2_amd.supres import DSen2_60
    rng = np.random.default_rng(2)
    d10 = rng.integers(35, 6000, size=(216, 180, 4)).astype(np.float32)
    d20 = rng.integers(35, 6000, size=(108, 90, 6)).astype(np.float32)
    d60 = rng.integers(35, 6000, size=(36, 30, 2)).astype(np.float32)
    out, _ = quiet(DSen2_60, d10, d20, d60, deep=False)
    assert out.shape == (216, 180, 2) and out.dtype == np.float32
    flat = model_dir['s2_030_lr_1e-05']
    p = po.get_test_patches60(d10, d20, d60, patchSize=192, border=12, f32_coords=True)
    p = [a / np.float32(2000) for a in p]
    used = int(np.ceil(216 / 168.0) * np.ceil(180 / 168.0))
    pred = np.zeros((p[0].shape[0], 2, 192, 192))
    pred[:used] = c_oracle.forward([a[:used] for a in p], flat, 6, 128)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = po.recompose_images(pred, border=12, size=d10.shape).astype(np.float64) * 2000
    assert do.rmse(out, ref) / 2000 < RMSE_GATE_NORMALISED


def test_predict_surface(model_dir):
    """_predict(test, input_shape, deep, run_60): list of NCHW arrays in, NCHW float32 out, every patch
    computed (trailing zero patches included, as keras would)."""
    from dsen2_amd.supres import _predict
    xs = do.synthetic_inputs(3, 32, 32, (4, 6), seed=8)
    xs[0][2] = 0; xs[1][2] = 0
    out, _ = quiet(_predict, xs, ((4, None, None), (6, None, None)))
    ref = c_oracle.forward(xs, model_dir['s2_032_lr_1e-04'], 6, 128)
    assert out.shape == (3, 6, 32, 32) and out.dtype == np.float32
    assert do.rmse(out, ref) < 5e-6


def test_predict_deep_selects_vdsen2(tmp_path, monkeypatch):
    """deep=True -> (d, F) = (32, 256) and the s2_033 checkpoint (testing/supres.py:55-57); fp32 on the F=256
    kernels.  One 16x16 patch keeps the float64 oracle (66 convolutions of 256 channels) to a few seconds."""
    from dsen2_amd import supres
    flat = do.he_uniform_weights(10, 6, 32, 256, seed=13, bias_scale=0.02)
    np.save(str(tmp_path / 's2_033_lr_1e-04.npy'), flat)
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    supres.clear_model_cache()
    xs = do.synthetic_inputs(1, 16, 16, (4, 6), seed=21)
    out, printed = quiet(supres._predict, xs, ((4, None, None), (6, None, None)), True)
    assert 's2_033_lr_1e-04.hdf5' in printed
    ref = c_oracle.forward(xs, flat, 32, 256)
    err = do.rmse(out, ref)
    print('VDSen2 fp32 rmse', err)
    assert err < 2e-5
    supres.clear_model_cache()


def test_image_smaller_than_a_patch_is_rejected(model_dir):
    """The reference indexes with a negative origin and dies on a shape mismatch (patches.py:52,62-72);
    here the same input raises ValueError before any GPU work."""
    from dsen2_amd.supres import DSen2_20
    with pytest.raises(ValueError):
        quiet(DSen2_20, np.zeros((100, 100, 4), np.float32), np.zeros((50, 50, 6), np.float32))


def test_missing_checkpoint_raises_oserror(tmp_path, monkeypatch):
    from dsen2_amd import supres
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path / 'nowhere') + os.sep)
    supres.clear_model_cache()
    with pytest.raises(OSError):
        quiet(supres.DSen2_20, np.zeros((240, 240, 4), np.float32), np.zeros((120, 120, 6), np.float32))


def test_cli_npz_round_trip(model_dir, tmp_path, golden_dir):
    """python -m dsen2_amd.cli: the super-resolution + npz-writer part of s2_tiles_supres.py (:332-342,:383-420)."""
    from dsen2_amd import cli, supres
    g = np.load(os.path.join(golden_dir, 'tile_T33UUB_crop.npz'))
    inp = str(tmp_path / 'tile.npz')
    np.savez(inp, data10=g['d10'], data20=g['d20'], data60=g['d60'])
    out = str(tmp_path / 'sr.npz')
    rc, printed = quiet(cli.main, [inp, out, '--run_60', '--copy_original_bands', '--models', supres.MDL_PATH])
    assert rc == 0 and 'Super-resolving the 60m data into 10m bands' in printed
    bands = np.load(out, allow_pickle=True)['bands'].item()
    assert list(bands) == ['B4', 'B3', 'B2', 'B8', 'SRB5', 'SRB6', 'SRB7', 'SRB8A', 'SRB11', 'SRB12', 'SRB1', 'SRB9']
    assert all(b.shape == (264, 264) for b in bands.values())
    ref20, _ = quiet(supres.DSen2_20, g['d10'], g['d20'])
    assert np.array_equal(bands['SRB5'], ref20[:, :, 0])
    # ROI rounded to 60 m pixel boundaries (s2_tiles_supres.py:131-134)
    out2 = str(tmp_path / 'roi.npz')
    quiet(cli.main, [inp, out2, '--roi_x_y', '5,7,250,245', '--models', supres.MDL_PATH])
    b2 = np.load(out2, allow_pickle=True)['bands'].item()
    assert b2['SRB5'].shape == (240, 246)


def test_demo_script_prints_a_parity_rmse():
    """demo.py (the counterpart of testing/demoDSen2.py:14-35): runs DSen2_60 + DSen2_20 on the committed T33UUB crop
    with seeded random-init weights and prints the reference's `RMSE: %.4f` line against the float64 oracle pipeline."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'demo.py')], capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    m = re.search(r'^RMSE: (\d+\.\d{4})$', p.stdout, re.M)
    assert m, p.stdout[-1500:]
    assert float(m.group(1)) / 2000 < 1e-4                      # raw reflectance units -> normalised domain
    assert 'Super-resolving the 60m data into 10m bands' in p.stdout and 'sr20 (264, 264, 6) float32' in p.stdout


BF16_GATE_REL = 5e-3        # rmse / signal rms in the normalised domain: 1.8x the 2.8e-3 measured at d=32 (profiles)


def test_dsen2_20_and_60_in_bf16_through_the_drop_in_surface(model_dir, monkeypatch):
    """supres.PRECISION = 'bf16' (DSEN2_PRECISION / --precision): the same DSen2_20 / DSen2_60 calls with bf16
    operands on the residual-block convolutions, against the float64 oracle pipeline on real tile geometry
    (128 / 192-pixel patches, non-dividing sizes, clamped last tiles)."""
    from dsen2_amd import supres
    monkeypatch.setattr(supres, 'PRECISION', 'bf16')
    supres.clear_model_cache()
    rng = np.random.default_rng(4)
    d10 = rng.integers(35, 6000, size=(240, 150, 4)).astype(np.float32)
    d20 = rng.integers(35, 6000, size=(120, 75, 6)).astype(np.float32)
    out, _ = quiet(supres.DSen2_20, d10, d20, deep=False)
    ref = oracle_dsen2_20(d10, d20, model_dir['s2_032_lr_1e-04'])
    rel = do.rmse(out, ref) / float(np.sqrt(np.mean(ref ** 2)))
    print('DSen2_20 bf16: rmse / signal rms = %.3e' % rel)
    assert out.shape == (240, 150, 6) and out.dtype == np.float32 and rel < BF16_GATE_REL
    monkeypatch.setattr(supres, 'PRECISION', 'fp32')
    supres.clear_model_cache()
    out32, _ = quiet(supres.DSen2_20, d10, d20, deep=False)
    assert do.rmse(out32, ref) / 2000 < RMSE_GATE_NORMALISED                  # and the switch really switches
    assert not np.array_equal(out, out32)
    monkeypatch.setattr(supres, 'PRECISION', 'bf16')
    supres.clear_model_cache()
    d10 = rng.integers(35, 6000, size=(216, 180, 4)).astype(np.float32)
    d20 = rng.integers(35, 6000, size=(108, 90, 6)).astype(np.float32)
    d60 = rng.integers(35, 6000, size=(36, 30, 2)).astype(np.float32)
    out, _ = quiet(supres.DSen2_60, d10, d20, d60, deep=False)
    p = po.get_test_patches60(d10, d20, d60, patchSize=192, border=12, f32_coords=True)
    p = [a / np.float32(2000) for a in p]
    used = int(np.ceil(216 / 168.0) * np.ceil(180 / 168.0))
    pred = np.zeros((p[0].shape[0], 2, 192, 192))
    pred[:used] = c_oracle.forward([a[:used] for a in p], model_dir['s2_030_lr_1e-05'], 6, 128)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = po.recompose_images(pred, border=12, size=d10.shape).astype(np.float64) * 2000
    rel = do.rmse(out, ref) / float(np.sqrt(np.mean(ref ** 2)))
    print('DSen2_60 bf16: rmse / signal rms = %.3e' % rel)
    assert out.shape == (216, 180, 2) and rel < BF16_GATE_REL


def test_predict_deep_in_bf16(tmp_path, monkeypatch):
    """deep=True with PRECISION = 'bf16': BASELINE configs[4]'s network through `_predict` (testing/supres.py:53-66)."""
    from dsen2_amd import supres
    flat = do.he_uniform_weights(10, 6, 32, 256, seed=13, bias_scale=0.02)
    np.save(str(tmp_path / 's2_033_lr_1e-04.npy'), flat)
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    monkeypatch.setattr(supres, 'PRECISION', 'bf16')
    supres.clear_model_cache()
    xs = do.synthetic_inputs(2, 16, 16, (4, 6), seed=22)
    out, printed = quiet(supres._predict, xs, ((4, None, None), (6, None, None)), True)
    assert 's2_033_lr_1e-04.hdf5' in printed and out.shape == (2, 6, 16, 16) and out.dtype == np.float32
    ref = c_oracle.forward(xs, flat, 32, 256)
    rel = do.rmse(out, ref) / float(np.sqrt(np.mean(ref ** 2)))
    print('VDSen2 bf16 through _predict: rmse / signal rms = %.3e' % rel)
    assert rel < BF16_GATE_REL
    supres.clear_model_cache()


@pytest.mark.parametrize('shape,limit', [((600, 600), 5), ((570, 342), 7), ((240, 198), 1), ((1200, 348), 11)])
def test_banded_output_equals_the_one_shot_output(model_dir, monkeypatch, shape, limit):
    """supres._run on one rank recomposes the rows that are final after every batch (dsen2_recompose_rows) and downloads them
    band by band on a copy stream under the batches still computing.  Forced here on small images (page-locked threshold 0,
    a handful of patches per batch, so every band boundary and the clamped last tile row are exercised): the same bits and
    prints as the one-shot path (DSEN2_BANDED_OUTPUT=0), for DSen2_20 and DSen2_60."""
    from dsen2_amd import supres
    from dsen2_amd.DSen2Net import S2Model
    rng = np.random.default_rng(shape[0] + shape[1])
    h, w = shape
    d10 = rng.integers(35, 9000, size=(h, w, 4), dtype=np.uint16)
    d20 = rng.integers(35, 9000, size=(h // 2, w // 2, 6), dtype=np.uint16)
    d60 = rng.integers(35, 9000, size=(h // 6, w // 6, 2), dtype=np.uint16)
    monkeypatch.setattr(supres, 'PINNED_OUTPUT_MIN_BYTES', 0)
    monkeypatch.setattr(S2Model, 'batch_limit', lambda self, hh, ww: limit)
    for fn, args in ((supres.DSen2_20, (d10, d20)), (supres.DSen2_60, (d10, d20, d60))):
        monkeypatch.setenv('DSEN2_BANDED_OUTPUT', '1')
        a, pa = quiet(fn, *args, deep=False)
        monkeypatch.setenv('DSEN2_BANDED_OUTPUT', '0')
        b, pb = quiet(fn, *args, deep=False)
        assert a.shape == b.shape == (h, w, a.shape[2]) and a.dtype == np.float32
        assert np.array_equal(a, b), (fn.__name__, shape, limit)
        assert pa == pb


def test_recompose_rows_is_the_full_recomposition_cut_into_bands():
    """dsen2_recompose_rows over any partition of the rows = dsen2_recompose, bit for bit; rows outside the band are not
    touched; a bad range is refused."""
    import torch
    from dsen2_amd import _lib, patches as gp
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.standard_normal((12, 6, 32, 32)).astype(np.float32)).cuda()     # 4 x 3 tiles of inner 24
    size = (80, 60)
    full = gp.recompose_device(a, 4, size, scale=2000.0)
    img = torch.full((80, 60, 6), -7.0, device='cuda')
    for r0, r1 in ((0, 1), (1, 24), (24, 55), (55, 56), (56, 80)):
        gp.recompose_rows_device(a, 4, img, r0, r1, scale=2000.0)
        assert torch.equal(img[:r1], full[:r1]) and bool((img[r1:] == -7.0).all())
    with pytest.raises(_lib.DSen2Error):
        gp.recompose_rows_device(a, 4, img, 10, 81)
    with pytest.raises(_lib.DSen2Error):
        gp.recompose_rows_device(a, 4, img, 30, 20)


def test_smallest_images_the_tiling_accepts(model_dir):
    """The smallest images whose symmetric padding still covers one patch: 112 x 112 for DSen2_20 (ONE used patch of the four
    the reference allocates, patches.py:35) and 168 x 168 for DSen2_60, plus one pixel row / 60 m cell more (a clamped second
    patch that overlaps the first almost completely) — each against the oracle pipeline."""
    from dsen2_amd.supres import DSen2_20, DSen2_60
    rng = np.random.default_rng(8)
    for h, w in ((112, 112), (114, 112), (112, 124)):
        d10 = rng.integers(35, 6000, size=(h, w, 4)).astype(np.float32)
        d20 = rng.integers(35, 6000, size=(h // 2, w // 2, 6)).astype(np.float32)
        out, _ = quiet(DSen2_20, d10, d20, deep=False)
        assert out.shape == (h, w, 6)
        p10, p20 = po.get_test_patches(d10, d20, patchSize=128, border=8, f32_coords=True)
        used = int(np.ceil(h / 112.0) * np.ceil(w / 112.0))
        pred = np.zeros((p10.shape[0], 6, 128, 128))
        pred[:used] = c_oracle.forward([p10[:used] / np.float32(2000), p20[:used] / np.float32(2000)], model_dir['s2_032_lr_1e-04'], 6, 128)
        with contextlib.redirect_stdout(io.StringIO()):
            ref = po.recompose_images(pred, border=8, size=d10.shape).astype(np.float64) * 2000
        assert do.rmse(out, ref) / 2000 < RMSE_GATE_NORMALISED, (h, w)
    for h, w in ((168, 168), (174, 168)):
        d10 = rng.integers(35, 6000, size=(h, w, 4)).astype(np.float32)
        d20 = rng.integers(35, 6000, size=(h // 2, w // 2, 6)).astype(np.float32)
        d60 = rng.integers(35, 6000, size=(h // 6, w // 6, 2)).astype(np.float32)
        out, _ = quiet(DSen2_60, d10, d20, d60, deep=False)
        assert out.shape == (h, w, 2)
        p = po.get_test_patches60(d10, d20, d60, patchSize=192, border=12, f32_coords=True)
        used = int(np.ceil(h / 168.0) * np.ceil(w / 168.0))
        pred = np.zeros((p[0].shape[0], 2, 192, 192))
        pred[:used] = c_oracle.forward([a[:used] / np.float32(2000) for a in p], model_dir['s2_030_lr_1e-05'], 6, 128)
        with contextlib.redirect_stdout(io.StringIO()):
            ref = po.recompose_images(pred, border=12, size=d10.shape).astype(np.float64) * 2000
        assert do.rmse(out, ref) / 2000 < RMSE_GATE_NORMALISED, (h, w)


def test_any_real_dtype_and_memory_layout_gives_the_float32_result(model_dir):
    """testing/supres.py takes whatever GDAL's ReadAsArray returns ("any real dtype": np.pad + a float32 patch array,
    patches.py:27-28,37-39): uint16 (Sentinel-2 L1C), int16, int32, uint8-range, float64, a Fortran-ordered array, a
    non-contiguous view and a read-only array all give the bits of the float32 C-contiguous call."""
    from dsen2_amd.supres import DSen2_20
    rng = np.random.default_rng(9)
    base10 = rng.integers(35, 9000, size=(226, 150, 4))
    base20 = rng.integers(35, 9000, size=(113, 75, 6))
    want, _ = quiet(DSen2_20, base10.astype(np.float32), base20.astype(np.float32), deep=False)
    for cast in (np.uint16, np.int16, np.int32, np.float64):
        got, _ = quiet(DSen2_20, base10.astype(cast), base20.astype(cast), deep=False)
        assert np.array_equal(got, want), cast
    f10, f20 = np.asfortranarray(base10.astype(np.float32)), np.asfortranarray(base20.astype(np.uint16))
    assert np.array_equal(quiet(DSen2_20, f10, f20, deep=False)[0], want)
    wide10 = np.zeros((226, 150, 9), np.uint16); wide10[:, :, ::2][:, :, :4] = base10
    view10 = wide10[:, :, 0:7:2]                                   # strided channel view
    assert not view10.flags.c_contiguous
    ro20 = base20.astype(np.uint16); ro20.setflags(write=False)
    assert np.array_equal(quiet(DSen2_20, view10, ro20, deep=False)[0], want)
    small = (base10 % 256).astype(np.uint8), (base20 % 256).astype(np.uint8)
    want8, _ = quiet(DSen2_20, small[0].astype(np.float32), small[1].astype(np.float32), deep=False)
    assert np.array_equal(quiet(DSen2_20, small[0], small[1], deep=False)[0], want8)
    # float64 values that are NOT float32 numbers (converted on the GPU: the same round-to-nearest as numpy's cast), int64, and a
    # big-endian array (converted on the host)
    frac10, frac20 = base10 + rng.random(base10.shape) * 0.999, base20 + rng.random(base20.shape) * 0.999
    assert frac10.dtype == np.float64 and not np.array_equal(frac10, frac10.astype(np.float32))
    wantf, _ = quiet(DSen2_20, frac10.astype(np.float32), frac20.astype(np.float32), deep=False)
    assert np.array_equal(quiet(DSen2_20, frac10, frac20, deep=False)[0], wantf)
    assert np.array_equal(quiet(DSen2_20, np.asfortranarray(frac10), frac20, deep=False)[0], wantf)
    assert np.array_equal(quiet(DSen2_20, base10.astype(np.int64), base20.astype('>u2'), deep=False)[0], want)


@pytest.mark.parametrize('dtype', [np.uint16, np.float32, np.int16])
def test_views_as_the_reference_scripts_build_them_give_the_same_image(dtype, model_dir):
    """The reference's callers pass VIEWS: `np.rollaxis(ds.ReadAsArray(...), 0, 3)` (an HWC view of a CHW array,
    testing/s2_tiles_supres.py) and `f['im10'][()].transpose()` (Fortran order, testing/demoDSen2.py:16).  They are uploaded in
    their storage order and permuted on the GPU (patches._to_device_f32); the image is the one the C-contiguous copy gives, bit
    for bit, and the caller's arrays are left as they were."""
    from dsen2_amd import patches as gp, supres
    rng = np.random.default_rng(8)
    hi = 9000 if dtype != np.int16 else 3000
    c10 = rng.integers(35, hi, size=(4, 288, 252)).astype(dtype)
    c20 = rng.integers(35, hi, size=(6, 144, 126)).astype(dtype)
    c60 = rng.integers(35, hi, size=(2, 48, 42)).astype(dtype)
    roll = [np.rollaxis(a, 0, 3) for a in (c10, c20, c60)]                                  # HWC views of CHW
    flat = [np.ascontiguousarray(a) for a in roll]                                            # what they mean
    fort = [np.asfortranarray(a) for a in flat]                                               # readh5-style
    assert not roll[0].flags.c_contiguous and not fort[0].flags.c_contiguous
    for views in (roll, fort):
        for a, b in zip(views, flat):
            t = gp._to_device_f32(a, gp.default_device())
            assert t.is_contiguous() and t.dtype == torch.float32 and np.array_equal(t.cpu().numpy(), b.astype(np.float32))
    # a row slab of a rolled view (what a rank uploads under torch.distributed): one copy per plane
    t = gp._to_device_f32(roll[0][40:200], gp.default_device())
    assert np.array_equal(t.cpu().numpy(), flat[0][40:200].astype(np.float32))
    want20, _ = quiet(supres.DSen2_20, flat[0], flat[1])
    want60, _ = quiet(supres.DSen2_60, *flat)
    keep = [a.copy() for a in (c10, c20, c60)]
    for views in (roll, fort):
        got20, _ = quiet(supres.DSen2_20, views[0], views[1])
        got60, _ = quiet(supres.DSen2_60, *views)
        assert np.array_equal(got20, want20) and np.array_equal(got60, want60)
    assert all(np.array_equal(a, b) for a, b in zip((c10, c20, c60), keep))
