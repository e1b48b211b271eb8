"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the header declares
(no compute without a GPU), and the host logic above it (origin arithmetic, weight containers,
error behaviour) agrees with the oracle."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import dsen2_oracle as do
from oracle import patches_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from dsen2_amd import _lib, build
    build.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'dsen2_hip.h')).read()
    declared = set(re.findall(r'\b(dsen2_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert b'gfx950' in lib.dsen2_version()


def test_product_abi_has_no_tuning_or_diagnostic_switches():
    """no global mutable state.  Kernel-structure switches and timing-only ablations exist only in the
    diagnostic build (python -m dsen2_amd.build --diag), never in the product library or its header."""
    from dsen2_amd import _lib
    lib = _lib.load()
    for name in ('dsen2_set_tuning', 'dsen2_diag_set'):
        assert not hasattr(lib, name), name
    header = open(os.path.join(ROOT, 'include', 'dsen2_hip.h')).read()
    assert 'set_tuning' not in header and 'diag_set' not in header
    with pytest.raises(_lib.DSen2Error):
        _lib.diag_set(0, 0)
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'DIAGNOSTIC build' not in blob


def test_code_object_targets_gfx950_only():
    from dsen2_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob
    for other in (b'gfx942', b'gfx90a', b'sm_90', b'sm_80'):
        assert other not in blob


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-box behaviour')
def test_no_gpu_fails_loudly_not_silently():
    from dsen2_amd import _lib
    from dsen2_amd import supres
    assert _lib.load().dsen2_device_count() < 0
    with pytest.raises(RuntimeError):
        supres.DSen2_20(np.zeros((240, 240, 4), np.float32), np.zeros((120, 120, 6), np.float32))
    with pytest.raises(RuntimeError):
        from dsen2_amd.DSen2Net import s2model
        s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'dsen2_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src, os.path.join(dirpath, f)


@pytest.mark.parametrize('shape,patch,border', [((36, 36), 16, 2), ((40, 46), 16, 2), ((300, 300), 64, 4),
                                                ((5490, 5490), 64, 4), ((18, 18), 8, 1), ((16, 22), 8, 1),
                                                ((1830, 1830), 32, 2), ((56, 57), 64, 4)])
def test_tile_origins_match_oracle_geometry(shape, patch, border):
    """tile_origins() (host arithmetic feeding dsen2_tile_gather) vs the oracle's tiling: compare by
    tiling an index image so every patch reveals its origin."""
    from dsen2_amd.patches import tile_origins
    org, n_alloc = tile_origins(shape, patch, border)
    idx = np.arange(shape[0] * shape[1], dtype=np.float64).reshape(shape + (1,))
    padded = np.pad(idx, ((border, border), (border, border), (0, 0)), mode='symmetric')
    stride = patch - 2 * border
    ki, kj = shape[0] // stride, shape[1] // stride
    assert n_alloc == (ki + 1) * (kj + 1)
    used = (ki + (shape[0] % stride != 0)) * (kj + (shape[1] % stride != 0))
    assert org.shape == (used, 2) and org.dtype == np.int32
    if shape[0] * shape[1] <= 100000:
        pats = po._tile([idx.astype(np.float32)], [1], [patch], [border])[0]
        assert pats.shape[0] == n_alloc
        for k, (i0, j0) in enumerate(org):
            assert np.array_equal(pats[k, 0], padded[i0:i0 + patch, j0:j0 + patch, 0].astype(np.float32))
        assert not pats[used:].any()
    # full 10980^2 tile: 9801 patches
    if shape == (5490, 5490):
        assert used == 9801


def test_tile_origins_reject_images_smaller_than_a_patch():
    from dsen2_amd.patches import tile_origins
    with pytest.raises(ValueError):
        tile_origins((50, 50), 64, 4)


def test_inconsistent_image_sizes_raise_value_error_before_the_gpu():
    """The reference fails with a ValueError when the 10 m / 20 m / 60 m images do not have the 2x / 6x size ratio
    its crop loops assume (patches.py:67,136-137); the drop-in says so up front (crop origins are multiples of the
    low-resolution ones, so a too-small image would otherwise be read out of bounds)."""
    from dsen2_amd import patches, supres
    d20 = np.zeros((120, 120, 6), np.float32)
    with pytest.raises(ValueError):
        supres.DSen2_20(np.zeros((200, 240, 4), np.float32), d20)
    with pytest.raises(ValueError):
        supres.DSen2_60(np.zeros((240, 240, 4), np.float32), d20, np.zeros((64, 40, 2), np.float32))
    with pytest.raises(ValueError):
        patches.get_test_patches(np.zeros((100, 240, 4), np.float32), d20)
    with pytest.raises(ValueError):
        patches.get_test_patches60(np.zeros((240, 240, 4), np.float32), np.zeros((100, 120, 6), np.float32),
                                   np.zeros((40, 40, 2), np.float32))
    with pytest.raises(ValueError):
        supres.DSen2_20(np.zeros((240, 240), np.float32), d20)


def test_row_slabs_cover_what_a_shard_reads():
    """Multi-GPU: a rank uploads only rows [r0, r1) of each image; every row its patches read (after the symmetric
    reflection at the true image edges) must lie inside, and the slab touches an image edge whenever a patch
    reflects there."""
    from dsen2_amd import supres
    from dsen2_amd.patches import tile_origins
    for (h, w), patch, border in [((300, 300), 64, 4), ((285, 300), 64, 4), ((190, 190), 32, 2)]:
        org, _ = tile_origins((h, w), patch, border)
        for world in (1, 2, 3, 8):
            per = (len(org) + world - 1) // world
            for r in range(world):
                mine = org[r * per:(r + 1) * per]
                if len(mine) == 0:
                    continue
                for scale in (1, 2, 6):
                    H, P, b = h * scale, patch * scale, border * scale
                    r0, r1 = supres._row_slab(mine, scale, P, b, H)
                    rows = (mine[:, 0:1].astype(np.int64) * scale - b + np.arange(P)[None, :]).ravel()
                    refl = np.where(rows < 0, -1 - rows, np.where(rows >= H, 2 * H - 1 - rows, rows))
                    assert refl.min() >= r0 and refl.max() < r1, (h, patch, world, r, scale)
                    assert (rows.min() >= 0) or r0 == 0
                    assert (rows.max() < H) or r1 == H


def test_weight_container_matches_oracle_layout(tmp_path):
    from dsen2_amd import weights as w
    for cin, cout, d, f in [(10, 6, 6, 128), (12, 2, 6, 128), (10, 6, 32, 256)]:
        assert w.num_params(cin, cout, d, f) == do.num_params(cin, cout, d, f)
        assert w.layer_shapes(cin, cout, d, f) == do.layer_shapes(cin, cout, d, f)
    a = w.random_he_uniform(10, 6, 6, 128, seed=1)
    b = do.he_uniform_weights(10, 6, 6, 128, seed=1)
    assert np.array_equal(a, b)
    np.save(str(tmp_path / 's2_032_lr_1e-04.npy'), a)
    # the reference asks for '<stem>.hdf5'; a converted '<stem>.npy' next to it is picked up
    got = w.load_flat(str(tmp_path / 's2_032_lr_1e-04.hdf5'), 10, 6, 6, 128)
    assert np.array_equal(got, a)
    with pytest.raises(OSError):
        w.load_flat(str(tmp_path / 'missing.hdf5'), 10, 6, 6, 128)
    with pytest.raises(ValueError):
        w.load_flat(str(tmp_path / 's2_032_lr_1e-04.npy'), 12, 2, 6, 128)


def test_supres_constants_and_weight_file_selection():
    from dsen2_amd import supres
    assert supres.SCALE == 2000 and supres.MDL_PATH == '../models/'
    assert supres._weight_file(False, False).endswith('s2_032_lr_1e-04.hdf5')     # testing/supres.py:60
    assert supres._weight_file(False, True).endswith('s2_030_lr_1e-05.hdf5')
    assert supres._weight_file(True, False).endswith('s2_033_lr_1e-04.hdf5')      # testing/supres.py:57
    assert supres._weight_file(True, True).endswith('s2_034_lr_1e-04.hdf5')
    import inspect
    assert list(inspect.signature(supres.DSen2_20).parameters) == ['d10', 'd20', 'deep']
    assert list(inspect.signature(supres.DSen2_60).parameters) == ['d10', 'd20', 'd60', 'deep']
    assert list(inspect.signature(supres._predict).parameters) == ['test', 'input_shape', 'deep', 'run_60']
    from dsen2_amd import patches
    assert list(inspect.signature(patches.get_test_patches).parameters) == ['dset_10', 'dset_20', 'patchSize', 'border', 'interp']
    assert inspect.signature(patches.get_test_patches).parameters['border'].default == 4
    assert inspect.signature(patches.get_test_patches60).parameters['border'].default == 8
    assert list(inspect.signature(patches.recompose_images).parameters) == ['a', 'border', 'size']


def test_overriding_the_product_library_is_never_silent(tmp_path):
    """DSEN2_HIP_LIB exists for tools/ (A/B of experimental and diagnostic builds whose outputs can be wrong under ablation
    masks): when it is set, _lib.load() says on stderr which file it loaded; without it, nothing is printed."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, 'dsen2_amd', 'libdsen2_hip.so')
    code = 'from dsen2_amd import _lib; _lib.load(); print(_lib.LIB_PATH)'
    env = dict(os.environ, DSEN2_HIP_LIB=lib)
    p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1000:]
    assert 'DSEN2_HIP_LIB overrides the product library' in p.stderr and lib in p.stderr
    env.pop('DSEN2_HIP_LIB')
    p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'overrides' not in p.stderr


def test_storage_plan_of_the_views_the_reference_scripts_hand_over():
    """patches._storage_plan (host logic of the upload): the HWC view of a CHW array (`np.rollaxis(ReadAsArray(...), 0, 3)`,
    testing/s2_tiles_supres.py) and the transposed array of `readh5` (testing/demoDSen2.py:16) are contiguous in another axis
    order — uploaded as they lie, permuted on the GPU; a row slab of the former is one contiguous piece per plane; anything
    else (a column stride, a reversed axis) is not claimed."""
    from dsen2_amd.patches import _storage_plan
    chw = np.arange(4 * 6 * 8, dtype=np.uint16).reshape(4, 6, 8)
    hwc = np.rollaxis(chw, 0, 3)
    assert not hwc.flags.c_contiguous and _storage_plan(hwc) == ((2, 0, 1), 'whole')
    assert hwc.transpose(2, 0, 1).flags.c_contiguous
    assert _storage_plan(chw.transpose()) == ((2, 1, 0), 'whole')
    # exactly what testing/s2_tiles_supres.py:313-315 passes: the rolled view indexed by the validated band list — numpy lays
    # the result of that advanced index out band-major (also when the source is C-contiguous)
    picked = np.rollaxis(chw, 0, 3)[:, :, [2, 1, 0, 3]]
    assert not picked.flags.c_contiguous and _storage_plan(picked) == ((2, 0, 1), 'whole')
    assert _storage_plan(np.ascontiguousarray(hwc)[:, :, [2, 1, 0, 3]]) == ((2, 0, 1), 'whole')
    order, mode = _storage_plan(hwc[2:5])
    assert (order, mode) == ((2, 0, 1), 'planes') and all(hwc[2:5].transpose(order)[k].flags.c_contiguous for k in range(4))
    for other in (hwc[:, ::2], np.ascontiguousarray(hwc)[::-1], chw.transpose()[2:5], np.zeros((0, 3, 4))[:, ::2]):
        assert _storage_plan(other) == (None, None)
