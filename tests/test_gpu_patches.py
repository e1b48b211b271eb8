"""GPU tiling / up-sampling / recomposition (through the C ABI) vs the reference's own outputs
(tests/golden/*.npz captured from /root/reference/utils/patches.py) and vs the oracle."""
import contextlib
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import patches_oracle as po

CASES20 = ['patches_20_div.npz', 'patches_20_nondiv.npz', 'patches_20_b8.npz']
CASES60 = ['patches_60_div.npz', 'patches_60_nondiv.npz', 'patches_60_b12.npz']
# The up-sampler follows scikit-image 0.18.3's float32 arithmetic operation by operation (patch_ops.hip): the captured
# outputs of the reference's interp_patches, and the oracle's f32_coords mode, are reproduced BIT FOR BIT.
from bits import assert_same_bits      # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('name', CASES20)
def test_get_test_patches(golden_dir, name):
    from dsen2_amd import patches as gp
    g = load(golden_dir, name)
    d10, d20 = g['d10'].astype(np.float32), g['d20'].astype(np.float32)
    patch, border = int(g['patch']), int(g['border'])
    p10, p20 = gp.get_test_patches(d10, d20, patchSize=patch, border=border)
    assert p10.dtype == np.float32 and p10.shape == g['p10'].shape
    assert np.array_equal(p10, g['p10'])                         # bit-exact copy, incl. trailing zero patches
    assert_same_bits(p20, g['p20'])
    _, raw = gp.get_test_patches(d10, d20, patchSize=patch, border=border, interp=False)
    assert np.array_equal(raw, g['p20_raw'])
    # accepts the integer dtype the tiles are stored in, like the reference (np.pad + float32 assignment)
    q10, _ = gp.get_test_patches(g['d10'], g['d20'], patchSize=patch, border=border, interp=False)
    assert np.array_equal(q10, g['p10'])


@pytest.mark.parametrize('name', CASES60)
def test_get_test_patches60(golden_dir, name):
    from dsen2_amd import patches as gp
    g = load(golden_dir, name)
    d = [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]
    patch, border = int(g['patch']), int(g['border'])
    p10, p20, p60 = gp.get_test_patches60(*d, patchSize=patch, border=border)
    assert np.array_equal(p10, g['p10'])
    assert_same_bits(p20, g['p20'])
    assert_same_bits(p60, g['p60'])
    _, r20, r60 = gp.get_test_patches60(*d, patchSize=patch, border=border, interp=False)
    assert np.array_equal(r20, g['p20_raw']) and np.array_equal(r60, g['p60_raw'])


@pytest.mark.parametrize('name', CASES20 + CASES60)
def test_recompose(golden_dir, name):
    from dsen2_amd import patches as gp
    g = load(golden_dir, name)
    rec = quiet(gp.recompose_images, g['pred'], border=int(g['border']), size=g['d10'].shape)
    assert rec.dtype == np.float32 and np.array_equal(rec, g['rec'])


def test_recompose_single_patch_quirk(golden_dir):
    from dsen2_amd import patches as gp
    g = load(golden_dir, 'recompose_single.npz')
    rec = gp.recompose_images(g['pred'], border=4, size=(24, 24, 4))
    assert np.array_equal(rec, g['rec'])


def test_interp_patches(golden_dir):
    from dsen2_amd import patches as gp
    g = load(golden_dir, 'interp.npz')
    for src, key in [('ramp', 'ramp_x2'), ('ramp', 'ramp_x6'), ('a', 'a_x2'), ('a', 'a_x6'), ('b', 'b_x2'),
                     ('b', 'b_x6')]:
        out = gp.interp_patches(g[src], g[key].shape)
        assert out.dtype == np.float32
        assert_same_bits(out, g[key], key)
    ramp = gp.interp_patches(g['ramp'], (1, 1, 8, 8))[0, 0, 0]
    np.testing.assert_allclose(ramp, [2.5, 2.5, 7.5, 12.5, 17.5, 22.5, 27.5, 27.5], rtol=1e-6)


def test_real_tile_crop_default_geometry(golden_dir):
    """128/8 and 192/12 (testing/supres.py:21-22,40-41) on the crop of the bundled T33UUB tile."""
    from dsen2_amd import patches as gp
    g = load(golden_dir, 'tile_T33UUB_crop.npz')
    d = [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]
    sub = (slice(None), slice(None), slice(3, None, 7), slice(2, None, 5))
    p10, p20 = gp.get_test_patches(d[0], d[1], patchSize=128, border=8)
    assert p10.shape == (9, 4, 128, 128)
    np.testing.assert_array_equal(p10.astype(np.float64).sum(axis=(2, 3)), g['p10_sum'])
    assert_same_bits(p20[sub], g['p20_sub'])
    assert_same_bits(p20[4, :2], g['p20_patch4'])
    q10, q20, q60 = gp.get_test_patches60(*d, patchSize=192, border=12)
    assert q10.shape == (4, 4, 192, 192)
    np.testing.assert_array_equal(q10.astype(np.float64).sum(axis=(2, 3)), g['q10_sum'])
    assert_same_bits(q20[sub], g['q20_sub'])
    assert_same_bits(q60[sub], g['q60_sub'])
    rec = quiet(gp.recompose_images, p10, border=8, size=d[0].shape)
    assert np.array_equal(rec, d[0])                             # tiling -> recompose round trip


def test_large_image_round_trip_and_oracle():
    """Size-independent property at a production-like size: recompose(tile(x)) == x, and the GPU patches
    equal the oracle's on a 1098x1098 image (a 10 % edge of a real 10980^2 tile)."""
    from dsen2_amd import patches as gp
    rng = np.random.default_rng(0)
    d10 = rng.integers(35, 13110, size=(1098, 1098, 4)).astype(np.float32)
    d20 = rng.integers(35, 13110, size=(549, 549, 6)).astype(np.float32)
    p10, p20 = gp.get_test_patches(d10, d20, patchSize=128, border=8)
    rec = quiet(gp.recompose_images, p10, border=8, size=d10.shape)
    assert np.array_equal(rec, d10)
    o10, o20 = po.get_test_patches(d10, d20, patchSize=128, border=8, f32_coords=True)
    assert np.array_equal(p10, o10)
    assert_same_bits(p20, o20)


def test_integer_rasters_are_widened_on_device(golden_dir):
    """uint16 / int16 / int32 inputs give exactly the patches of their float32 conversion (values up to 65535)."""
    from dsen2_amd import patches as gp
    rng = np.random.default_rng(5)
    d10 = rng.integers(0, 65536, size=(72, 72, 4)).astype(np.uint16)
    d20 = rng.integers(0, 65536, size=(36, 36, 6)).astype(np.uint16)
    ref = gp.get_test_patches(d10.astype(np.float32), d20.astype(np.float32), patchSize=32, border=4)
    for cast in (lambda a: a, lambda a: a.astype(np.int32), lambda a: (a // 2).astype(np.int16)):
        a10, a20 = cast(d10), cast(d20)
        got = gp.get_test_patches(a10, a20, patchSize=32, border=4)
        want = ref if cast(d10).dtype != np.int16 else gp.get_test_patches(a10.astype(np.float32), a20.astype(np.float32),
                                                                            patchSize=32, border=4)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize('size,patch,border', [(10980, 128, 8), (10980, 192, 12), (1000, 128, 8)])
def test_full_tile_gather_then_recompose_is_the_identity(size, patch, border):
    """BASELINE configs[3] at its real size (a 10980^2 Sentinel-2 tile: 9801 patches of 128, 4356 of 192): tiling the
    10 m image with get_test_patches' geometry and recomposing the patches' inner crops must return the image bit
    for bit — every pixel comes back from exactly the patch recompose_images reads it from (patches.py:374-405),
    whatever the clamped last row / column of tiles overlaps."""
    import torch
    from dsen2_amd import patches as P
    dev = P.default_device()
    g = torch.Generator(device='cpu').manual_seed(size + patch)
    # two bands keep the round trip within a few GB at full size; values as Sentinel-2 digital numbers
    img = torch.randint(35, 13110, (size, size, 2), generator=g, dtype=torch.int32).to(torch.float32).to(dev)
    lr = patch // 2                      # tile on the 20 m grid like DSen2_20 (crop origins x 2)
    org, n_alloc = P.tile_origins((size // 2, size // 2), lr, border // 2)
    assert org.shape[0] == int(np.ceil(size / float(patch - 2 * border))) ** 2
    pats = P.gather_patches_device(img, org, 2, border, patch, n_alloc, first=0, count=org.shape[0])
    back = P.recompose_device(pats, border, (size, size))
    assert torch.equal(back, img)
    # and the gather-to-root form: inner crops recomposed with border 0 (dsen2_amd/supres.py, multi-GPU path)
    crops = pats[:, :, border:patch - border, border:patch - border].contiguous()
    del pats
    assert torch.equal(P.recompose_device(crops, 0, (size, size)), img)


def _fuzz_geometries(seed, n, sixty):
    """Random (low-res extent, patch, border) triples the reference accepts: per-resolution patch and border come from
    its floor divisions (patches.py:21-24, :85-90), so patch and border are multiples of the scale here; images from
    smaller than one stride (a single clamped patch per axis) to ragged multi-tile extents."""
    rng = np.random.default_rng(seed)
    scale = 6 if sixty else 2
    out = []
    while len(out) < n:
        p_lr = int(rng.integers(3, 17))
        b_lr = int(rng.integers(0, (p_lr - 1) // 2 + 1))
        stride = p_lr - 2 * b_lr
        if stride < 1:
            continue
        lo_h = int(rng.integers(max(p_lr - 2 * b_lr, b_lr + 1), 5 * p_lr))
        lo_w = int(rng.integers(max(p_lr - 2 * b_lr, b_lr + 1), 5 * p_lr))
        if lo_h + 2 * b_lr < p_lr or lo_w + 2 * b_lr < p_lr or b_lr > min(lo_h, lo_w):   # np.pad('symmetric') handles b <= extent
            continue
        out.append((lo_h, lo_w, p_lr * scale, b_lr * scale))
    return out


@pytest.mark.parametrize('sixty', [False, True])
def test_tiling_geometry_fuzz_matches_oracle(sixty):
    """Seeded sweep over 40 geometries (ragged extents, one-patch images, zero borders, borders up to half a patch):
    every crop is bit-exact against the oracle's restatement of patches.py:19-156, the up-sampled bands within the
    float32-coordinate tolerance, and recompose_images inverts the tiling of the 10 m image exactly."""
    from dsen2_amd import patches as gp
    rng = np.random.default_rng(77 + sixty)
    for lo_h, lo_w, patch, border in _fuzz_geometries(1234 + sixty, 40, sixty):
        s = 6 if sixty else 2
        d10 = rng.integers(0, 13110, size=(lo_h * s, lo_w * s, 4)).astype(np.float32)
        if sixty:
            d20 = rng.integers(0, 13110, size=(lo_h * 3, lo_w * 3, 6)).astype(np.float32)
            d60 = rng.integers(0, 13110, size=(lo_h, lo_w, 2)).astype(np.float32)
            got = gp.get_test_patches60(d10, d20, d60, patchSize=patch, border=border)
            want = po.get_test_patches60(d10, d20, d60, patchSize=patch, border=border, f32_coords=True)
            raw = gp.get_test_patches60(d10, d20, d60, patchSize=patch, border=border, interp=False)
            raw_want = po.get_test_patches60(d10, d20, d60, patchSize=patch, border=border, interp=False)
        else:
            d20 = rng.integers(0, 13110, size=(lo_h, lo_w, 6)).astype(np.float32)
            got = gp.get_test_patches(d10, d20, patchSize=patch, border=border)
            want = po.get_test_patches(d10, d20, patchSize=patch, border=border, f32_coords=True)
            raw = gp.get_test_patches(d10, d20, patchSize=patch, border=border, interp=False)
            raw_want = po.get_test_patches(d10, d20, patchSize=patch, border=border, interp=False)
        tag = 'lo=%dx%d patch=%d border=%d' % (lo_h, lo_w, patch, border)
        assert got[0].shape == want[0].shape, tag
        assert np.array_equal(got[0], want[0]), tag
        for a, b in zip(raw[1:], raw_want[1:]):
            assert np.array_equal(a, b), tag
        for a, b in zip(got[1:], want[1:]):
            assert_same_bits(a, b, tag)
        rec = quiet(gp.recompose_images, got[0], border=border, size=d10.shape)
        rec_want = quiet(po.recompose_images, want[0], border, d10.shape)
        assert np.array_equal(rec, rec_want), tag
        if got[0].shape[0] > 1 and min(d10.shape[:2]) >= patch - 2 * border:
            assert np.array_equal(rec, d10), tag


def test_windowed_upsampler_gives_the_general_kernels_bits():
    """Up-sampling by 2 or more takes the branch-free windowed kernel (all taps loaded at once); the general kernel
    (dsen2_upsample_mirror_bilinear_ref: taps fetched on demand, any scale) must give the same bits — x2, x3, x6, ragged
    sizes, mirror edges, one-pixel dims, sizes that are not multiples of a thread's 4 x 8 outputs, both divisors."""
    import torch
    from dsen2_amd import patches as P
    rng = np.random.default_rng(5)
    for n, c, h, w, oh, ow in [(7, 6, 64, 64, 128, 128), (3, 2, 32, 32, 192, 192), (2, 6, 96, 96, 192, 192), (1, 2, 37, 53, 74, 106),
                               (2, 3, 1, 1, 2, 2), (1, 1, 2, 3, 12, 18), (1, 2, 10, 10, 27, 27), (2, 1, 5, 7, 15, 21), (1, 1, 33, 1, 66, 5),
                               (1, 2, 9, 11, 18, 23), (1, 1, 20, 30, 41, 61), (3, 1, 16, 16, 97, 101), (1, 1, 13, 13, 26, 91),
                               (1, 1, 7, 9, 25, 19), (1, 2, 64, 64, 100, 128)]:
        x = torch.from_numpy((rng.random((n, c, h, w), dtype=np.float32) * 12000).astype(np.float32)).cuda()
        x[0, 0, 0, 0] = 0.0
        for pd in (1.0, 2000.0):
            a = P.interp_patches_device(x, (oh, ow), post_divisor=pd)
            b = P.interp_patches_device(x, (oh, ow), post_divisor=pd, ref=True)
            assert torch.equal(a, b), (n, c, h, w, oh, ow, pd)


def test_interp_patches_bit_exact_beyond_the_tile_path_factors(golden_dir):
    """The reference's interp_patches on non-integer factors, odd / tiny planes, plateaus at a plane's extremes, a constant plane
    and values up to 65535 (tests/golden/interp_shapes.npz): the HIP up-sampler — the windowed kernel where the factor allows
    it, the general one elsewhere, and the general one forced — gives the reference's bits."""
    from dsen2_amd import patches as gp
    g = load(golden_dir, 'interp_shapes.npz')
    n = len([k for k in g.files if k.startswith('in_')])
    assert n >= 12
    for k in range(n):
        x, want = g['in_%02d' % k], g['out_%02d' % k]
        tag = 'case %d %r -> %r' % (k, x.shape[2:], want.shape[2:])
        assert_same_bits(gp.interp_patches(x, want.shape), want, tag)
        import torch
        ref = gp.interp_patches_device(torch.from_numpy(x).cuda(), want.shape[2:], ref=True).cpu().numpy()
        assert_same_bits(ref, want, tag + ' (general kernel)')
