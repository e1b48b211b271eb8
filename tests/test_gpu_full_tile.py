"""BASELINE configs[3] at its real size under -m gpu: DSen2_20 over a seeded synthetic 10980 x 10980 Sentinel-2 tile — 9801
patches of 128 x 128 (testing/s2_tiles_supres.py:311-342 -> testing/supres.py:15-30), the whole pipeline on the GPU.

What a full-size run can be checked against in seconds:
  * run-to-run bit identity of the whole 10980 x 10980 x 6 image;
  * three 112 x 112 output windows — the first patch (symmetric padding at the top-left corner), an interior patch, the
    CLAMPED last row / column patch (patches.py:45-53) — against the oracle pipeline: oracle tiling + up-sampling of a crop
    of the tile whose patch grid contains exactly that patch, float64 C oracle CNN, x 2000; gate 1e-4 in the normalised domain;
  * the patch-sharded run (2 ranks over gloo sharing the one GPU: row-slab uploads, weights broadcast from rank 0, inner
    crops gathered to rank 0) returns the single-rank image bit for bit at that size.
"""
import contextlib
import io
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import dsen2_oracle as do
from oracle import patches_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 10980
PATCH, BORDER = 128, 8              # testing/supres.py:21-22
INNER = PATCH - 2 * BORDER          # 112
STRIDE_LR = PATCH // 2 - BORDER     # 56: low-res stride of the patch grid (patches.py:32)
RMSE_GATE_NORMALISED = 1e-4


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def oracle_window(d10, d20, flat, which):
    """The 112 x 112 x 6 block of the output image that patch `which` owns, from the oracle pipeline, and where it lies.
    The oracle tiles a CROP of the tile chosen so that one patch of the crop's grid IS that patch of the full grid:
      'first'    the top-left 336 x 336: patch (0, 0) of both grids (same symmetric padding at the corner);
      'interior' 10 m rows / columns [(m-1)*112, (m+2)*112): aligned with the grid, so the crop's patch (1, 1) — whose
                 window lies a full stride inside the crop, beyond the reach of its padding — is the tile's patch (m, m);
      'last'     the bottom-right 336 x 336: both grids end with the clamped patch (origin = padded extent - patch)."""
    m = 47
    if which == 'first':
        r0, pick, y0 = 0, 0, 0
    elif which == 'interior':
        r0, pick, y0 = (m - 1) * INNER, 4, m * INNER          # crop grid is 3 x 3: patch (1, 1) = index 4
    else:
        r0, pick, y0 = N - 3 * INNER, 8, N - INNER
    c10 = d10[r0:r0 + 3 * INNER, r0:r0 + 3 * INNER].astype(np.float32)
    c20 = d20[r0 // 2:(r0 + 3 * INNER) // 2, r0 // 2:(r0 + 3 * INNER) // 2].astype(np.float32)
    p10, p20 = po.get_test_patches(c10, c20, patchSize=PATCH, border=BORDER, f32_coords=True)
    assert p10.shape[0] == 16          # 336 = 3 strides exactly: 3 x 3 used patches, (3 + 1)^2 allocated (patches.py:35)
    # the oracle's patch order is row-major over the USED grid (3 per row)
    xs = [p10[pick:pick + 1] / np.float32(2000), p20[pick:pick + 1] / np.float32(2000)]
    pred = c_oracle.forward(xs, flat, 6, 128)[0]                              # [6, 128, 128], normalised domain
    return pred[:, BORDER:PATCH - BORDER, BORDER:PATCH - BORDER].transpose(1, 2, 0).astype(np.float64), y0


@pytest.fixture()
def model_dir(tmp_path, monkeypatch):
    from dsen2_amd import supres
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=41, bias_scale=0.02)
    np.save(str(tmp_path / 's2_032_lr_1e-04.npy'), flat)
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    supres.clear_model_cache()
    yield flat
    supres.clear_model_cache()


def test_dsen2_20_full_tile_identity_and_windows_vs_oracle(model_dir):
    from dsen2_amd.supres import DSen2_20
    rng = np.random.default_rng(2026)
    d10 = rng.integers(35, 13110, size=(N, N, 4), dtype=np.uint16)           # the bundled tiles' value range
    d20 = rng.integers(35, 13110, size=(N // 2, N // 2, 6), dtype=np.uint16)
    out = quiet(DSen2_20, d10, d20, deep=False)
    assert out.shape == (N, N, 6) and out.dtype == np.float32
    assert np.isfinite(out[::61, ::67]).all()
    # the second run takes the rasters the way testing/s2_tiles_supres.py:313-322 passes them — band-major storage behind an HWC
    # view (np.rollaxis of a CHW read) — which are uploaded in storage order and permuted on the GPU
    v10 = np.rollaxis(np.ascontiguousarray(np.rollaxis(d10, 2, 0)), 0, 3)
    v20 = np.rollaxis(np.ascontiguousarray(np.rollaxis(d20, 2, 0)), 0, 3)
    assert not v10.flags.c_contiguous and np.array_equal(v10[5000:5003], d10[5000:5003])
    again = quiet(DSen2_20, v10, v20, deep=False)
    assert np.array_equal(out, again)                                         # 9801 patches, bit for bit
    del again, v10, v20
    for which in ('first', 'interior', 'last'):
        ref, y0 = oracle_window(d10, d20, model_dir, which)
        got = out[y0:y0 + INNER, y0:y0 + INNER].astype(np.float64) / 2000
        err = do.rmse(got, ref)
        print('10980^2 DSen2_20, %s patch window at (%d, %d): normalised rmse %.3e, signal rms %.3f'
              % (which, y0, y0, err, float(np.sqrt(np.mean(ref * ref)))))
        assert err < RMSE_GATE_NORMALISED, which


def test_dsen2_20_full_tile_in_bf16x3_stays_inside_the_fp32_gate(model_dir, monkeypatch):
    """The opt-in bf16x3 arithmetic over all 9801 patches of the 10980^2 tile against the fp32 run of the same call (which the
    test above ties to the oracle at 2e-6): the WHOLE image within the 1e-4 normalised gate, every band finite."""
    from dsen2_amd import supres
    rng = np.random.default_rng(2026)
    d10 = rng.integers(35, 13110, size=(N, N, 4), dtype=np.uint16)
    d20 = rng.integers(35, 13110, size=(N // 2, N // 2, 6), dtype=np.uint16)
    ref = quiet(supres.DSen2_20, d10, d20, deep=False)
    monkeypatch.setattr(supres, 'PRECISION', 'bf16x3')
    out = quiet(supres.DSen2_20, d10, d20, deep=False)
    assert out.shape == ref.shape == (N, N, 6) and not np.array_equal(out, ref)
    err2 = 0.0
    for r0 in range(0, N, 1098):                     # float64 accumulation in slabs: no 5.8 GB temporary
        dlt = out[r0:r0 + 1098].astype(np.float64) - ref[r0:r0 + 1098]
        assert np.isfinite(dlt).all()
        err2 += float((dlt * dlt).sum())
    err = np.sqrt(err2 / out.size) / 2000
    print('10980^2 DSen2_20 bf16x3 vs fp32: normalised rmse %.3e over the whole image' % err)
    assert err < 3e-5 < RMSE_GATE_NORMALISED


def test_dsen2_60_full_tile_identity_and_a_window_vs_oracle(tmp_path, monkeypatch):
    """The other half of testing/s2_tiles_supres.py:332-342 at the real size: DSen2_60 over 10980^2 — 4356 patches of 192^2
    (borders 12 / 6 / 2, 60 m stride 28, crops x 6 / x 3 / x 1: patches.py:83-156) — run-to-run identity and the interior
    patch's 168 x 168 window against the oracle pipeline (same aligned-crop method: 504 x 504 = 3 x 3 patches)."""
    from dsen2_amd import supres
    flat = do.he_uniform_weights(12, 2, 6, 128, seed=42, bias_scale=0.02)
    np.save(str(tmp_path / 's2_030_lr_1e-05.npy'), flat)
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    supres.clear_model_cache()
    rng = np.random.default_rng(2027)
    d10 = rng.integers(35, 13110, size=(N, N, 4), dtype=np.uint16)
    d20 = rng.integers(35, 13110, size=(N // 2, N // 2, 6), dtype=np.uint16)
    d60 = rng.integers(35, 13110, size=(N // 6, N // 6, 2), dtype=np.uint16)
    out = quiet(supres.DSen2_60, d10, d20, d60, deep=False)
    assert out.shape == (N, N, 2) and out.dtype == np.float32 and np.isfinite(out[::61, ::67]).all()
    assert np.array_equal(out, quiet(supres.DSen2_60, d10, d20, d60, deep=False))
    supres.clear_model_cache()
    inner, m = 192 - 24, 31
    r0 = (m - 1) * inner
    c = [d10[r0:r0 + 3 * inner, r0:r0 + 3 * inner].astype(np.float32),
         d20[r0 // 2:(r0 + 3 * inner) // 2, r0 // 2:(r0 + 3 * inner) // 2].astype(np.float32),
         d60[r0 // 6:(r0 + 3 * inner) // 6, r0 // 6:(r0 + 3 * inner) // 6].astype(np.float32)]
    p = po.get_test_patches60(c[0], c[1], c[2], patchSize=192, border=12, f32_coords=True)
    assert p[0].shape[0] == 16
    pred = c_oracle.forward([a[4:5] / np.float32(2000) for a in p], flat, 6, 128)[0]
    ref = pred[:, 12:180, 12:180].transpose(1, 2, 0).astype(np.float64)
    y0 = m * inner
    err = do.rmse(out[y0:y0 + inner, y0:y0 + inner].astype(np.float64) / 2000, ref)
    print('10980^2 DSen2_60, interior patch window at (%d, %d): normalised rmse %.3e' % (y0, y0, err))
    assert err < RMSE_GATE_NORMALISED


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def test_full_tile_two_ranks_equal_the_single_rank_image_at_10980():
    """supres._run sharded over 2 ranks (gloo, both on the one GPU) at the full size: 4901 + 4900 patches, each rank
    uploads the row slab its patches read (of band-major rasters as the reference's tile script builds them: one copy per band), only rank 0 holds the weight file (C1 broadcast), the inner crops (2.95 GB)
    are gathered to rank 0; its image equals the single-rank image bit for bit, rank 1 returns None."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', _free_port(), os.path.join(ROOT, 'tools', 'bench_full_tile.py'), '--size', str(N),
           '--skip60', '--backend', 'gloo', '--check', '--layout', 'rollaxis']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert r['tile'] == [N, N] and r['n_gpus'] == 2 and r['patches20'] == 9801 and r['matches_single_rank'] is True


def test_full_tile_chunked_gather_equals_the_single_rank_image_at_10980():
    """The same run with DSEN2_CHUNKED_GATHER=1: 8 gathers of ~613 crops per rank issued under the shards' work, rank 0
    recomposes and downloads band by band on its own stream into the page-locked buffer (the full-size image takes that
    path; the small rehearsal tiles do not)."""
    env = dict(os.environ, DSEN2_CHUNKED_GATHER='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', _free_port(), os.path.join(ROOT, 'tools', 'bench_full_tile.py'), '--size', str(N),
           '--skip60', '--backend', 'gloo', '--check']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert r['n_gpus'] == 2 and r['chunked_gather'] is True and r['matches_single_rank'] is True
