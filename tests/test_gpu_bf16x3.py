"""precision 2, "bf16x3" (conv3x3_body16w.hip, X3): fp32-grade results from the bf16 matrix cores — every fp32 operand is
two bf16 numbers (hi + lo, 16 significant bits), a product is hi*hi + hi*lo + lo*hi in fp32.

Adoption gate: the SAME 1e-4 normalised RMSE gate as fp32 on the golden fixtures and on both bundled tiles
 — not a bf16-appropriate tolerance.  Kernel level: the operand planes are checked bit for bit against their
definition; a convolution against the float64 oracle on the fp32 operands (what is lost is 2^-17 per operand)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import dsen2_oracle as do

RMSE_GATE = 1e-4                    # BASELINE.md §2, normalised domain — the fp32 gate
X3_EXPECTED = 3e-5                  # what the arithmetic should achieve at d = 6 (emulation: 1e-5); a regression alarm


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _planes_to_f32(planes):
    """int16 [n, 2, c/8, h, w, 8] (bf16 bit patterns) -> two fp32 NHWC arrays (hi, lo)."""
    n, _, b, h, w, _ = planes.shape
    u = planes.cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    f = u.view(np.float32)                                            # [n, 2, b, h, w, 8]
    f = f.transpose(0, 1, 3, 4, 2, 5).reshape(n, 2, h, w, b * 8)
    return f[:, 0], f[:, 1]


def _bf16_rne(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _hi_ties_away(x):
    u = x.view(np.uint32).astype(np.uint64)
    return ((((u + 0x8000) >> 16) << 16) & 0xffffffff).astype(np.uint32).view(np.float32)


def test_split3_planes_are_what_the_header_says():
    """dsen2_split3_f32: plane 0 = the bf16 rounding (ties away) of the bit pattern = dsen2_split_f32's hi; lo16 = its lo
    (the pair restores x bit for bit); plane 1 = bf16_rne(x - hi), so hi + xl carries 16 significant bits of x."""
    from dsen2_amd.DSen2Net import join_f32, split3_f32, split_f32
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 5, 13, 128)) * np.exp(rng.uniform(-6, 6, (2, 5, 13, 128)))).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    hx, lo = split3_f32(xd)
    hi_ref, lo_ref = split_f32(xd)
    assert torch.equal(hx[:, 0], hi_ref) and torch.equal(lo, lo_ref)
    assert np.array_equal(join_f32(hx[:, 0].contiguous(), lo).cpu().numpy().view(np.uint32), x.view(np.uint32))
    hi, xl = _planes_to_f32(hx)
    assert np.array_equal(hi, _hi_ties_away(x))
    assert np.array_equal(xl.view(np.uint32), _bf16_rne(x - hi).view(np.uint32))
    rel = np.abs((hi.astype(np.float64) + xl) - x) / np.abs(x)
    assert rel.max() <= 2.0 ** -16 and np.sqrt(np.mean(rel ** 2)) < 2.0 ** -17


@pytest.mark.parametrize('feat,n,h,w', [(128, 2, 32, 32), (128, 1, 21, 37), (256, 1, 16, 32), (128, 3, 16, 16)])
def test_bf16x3_convolutions_match_the_oracle_on_the_fp32_operands(feat, n, h, w):
    """One residual block at kernel level: conv-A (relu, two output planes), conv-B in place on the stream (hi | xl, lo16) and
    its fp32-output form — each against the float64 oracle run on the SAME fp32 operands (no pre-rounding: the 2^-17 per
    operand is the error being measured)."""
    from dsen2_amd.DSen2Net import conv3x3_body_bf16x3, join_f32, split3_f32
    rng = np.random.default_rng(feat + h)
    x = rng.standard_normal((n, feat, h, w)).astype(np.float32)
    ka = (rng.standard_normal((3, 3, feat, feat)) * np.sqrt(2.0 / (9 * feat))).astype(np.float32)
    kb = (rng.standard_normal((3, 3, feat, feat)) * np.sqrt(2.0 / (9 * feat))).astype(np.float32)
    ba = (rng.standard_normal(feat) * 0.1).astype(np.float32)
    bb = (rng.standard_normal(feat) * 0.1).astype(np.float32)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda()
    hx, lo = split3_f32(xd)
    # conv-A
    t = conv3x3_body_bf16x3(hx, ka, ba, epilogue=0)
    th, tl = _planes_to_f32(t)
    t_gpu = (th.astype(np.float64) + tl).transpose(0, 3, 1, 2)
    t_ref = c_oracle.conv3x3(x, ka, ba, relu=True)
    ea = do.rmse(t_gpu, t_ref)
    assert (th >= 0).all()
    # conv-B, fp32 output: x + 0.1 * (conv(t) + b) with t = what conv-A wrote (so only conv-B's own error is measured)
    t_in = (th.astype(np.float64) + tl).astype(np.float32).transpose(0, 3, 1, 2)
    yb = conv3x3_body_bf16x3(t, kb, bb, epilogue=3, res_hx=hx, res_lo=lo).cpu().numpy().transpose(0, 3, 1, 2)
    ref_b = x.astype(np.float64) + 0.1 * c_oracle.conv3x3(t_in, kb, bb, relu=False)
    eb = do.rmse(yb, ref_b)
    # conv-B in place: the stream's (hi, lo16) now hold exactly the fp32 values of the fp32-output form, xl is their second plane
    hx2, lo2 = conv3x3_body_bf16x3(t, kb, bb, epilogue=1, res_hx=hx.clone(), res_lo=lo.clone())
    back = join_f32(hx2[:, 0].contiguous(), lo2).cpu().numpy()
    assert np.array_equal(back.view(np.uint32), np.ascontiguousarray(yb.transpose(0, 2, 3, 1)).view(np.uint32))
    hi2, xl2 = _planes_to_f32(hx2)
    assert np.array_equal(hi2, _hi_ties_away(back))
    assert np.array_equal(xl2.view(np.uint32), _bf16_rne(back - hi2).view(np.uint32))
    print('bf16x3 F=%d %dx%dx%d: conv-A rmse %.3e (output rms %.2f), conv-B rmse %.3e' % (feat, n, h, w, ea, np.sqrt(np.mean(t_ref ** 2)), eb))
    assert ea < 2e-5 and eb < 3e-6          # measured 4-6e-6 / 5e-7 (profiles/r04_bf16x3.md): operands lose 2^-17, conv-B scales by 0.1


def _model(bands, d, f, flat, precision='bf16x3'):
    from dsen2_amd.DSen2Net import s2model
    m = s2model(tuple((b, None, None) for b in bands), num_layers=d, feature_size=f, precision=precision)
    m.set_weights_flat(flat)
    return m


@pytest.mark.parametrize('name', ['cnn_20_d6_f128', 'cnn_60_d6_f128', 'cnn_20_d2_f256', 'cnn_20_d6_f128_ragged'])
def test_bf16x3_forward_meets_the_fp32_gate_on_the_golden_fixtures(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    bands = tuple(int(b) for b in g['bands'])
    d, f = int(g['num_layers']), int(g['feature_size'])
    flat = do.he_uniform_weights(sum(bands), bands[-1], d, f, seed=int(g['weight_seed']), bias_scale=float(g['bias_scale']))
    xs = [g['x%d' % i] for i in range(len(bands))]
    y = _model(bands, d, f, flat).predict(xs)
    err = do.rmse(y, g['out'])
    print(name, 'bf16x3 rmse %.3e (fp32 gate %.0e)' % (err, RMSE_GATE))
    assert err < RMSE_GATE and err < X3_EXPECTED


def test_bf16x3_batch_512_properties_and_vdsen2_depth():
    """BASELINE configs[1]'s shape in bf16x3: deterministic, permutation-equivariant, batch-invariant, oracle on a sample;
    and the deep network (d = 32, F = 256) still inside the fp32 gate."""
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=1)
    xs = do.synthetic_inputs(512, 32, 32, (4, 6), seed=0)
    m = _model((4, 6), 6, 128, flat)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    y1 = m.forward_device(dev).clone()
    assert torch.equal(y1, m.forward_device(dev))
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(0)).cuda()
    assert torch.equal(m.forward_device([d[perm].contiguous() for d in dev]), y1[perm])
    assert torch.equal(m.forward_device([d[100:107].contiguous() for d in dev]), y1[100:107])
    idx = [0, 255, 511]
    ref = c_oracle.forward([a[idx] for a in xs], flat, 6, 128)
    e = do.rmse(y1.cpu().numpy()[idx], ref)
    print('bf16x3 batch 512: sampled rmse %.3e' % e)
    assert e < RMSE_GATE and e < X3_EXPECTED
    flat = do.he_uniform_weights(10, 6, 32, 256, seed=5, bias_scale=0.02)
    xs = do.synthetic_inputs(2, 16, 16, (4, 6), seed=4)
    y = _model((4, 6), 32, 256, flat).predict(xs)
    ref = c_oracle.forward(xs, flat, 32, 256)
    e = do.rmse(y, ref)
    print('bf16x3 VDSen2 d=32 F=256: rmse %.3e, signal rms %.2f' % (e, np.sqrt(np.mean(ref ** 2))))
    assert e < RMSE_GATE


@pytest.mark.parametrize('name', ['tile_T33UUB_600.npz', 'tile_T49JGM_600.npz'])
def test_bf16x3_dsen2_20_on_the_whole_bundled_tiles(golden_dir, tmp_path, monkeypatch, oracle_dsen2_tile, name):
    """DSen2_20 through the drop-in surface with supres.PRECISION = 'bf16x3' on the two tiles the reference ships, all 36
    patches against the float64 oracle pipeline: the fp32 gate.  (Same weights as tests/test_gpu_bundled_tiles.py: the oracle
    image is computed once per session.)"""
    from dsen2_amd import supres
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=31, bias_scale=0.02)
    np.save(str(tmp_path / 's2_032_lr_1e-04.npy'), flat)
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    monkeypatch.setattr(supres, 'PRECISION', 'bf16x3')
    supres.clear_model_cache()
    g = np.load(os.path.join(golden_dir, name))
    d10, d20 = g['d10'].astype(np.float32), g['d20'].astype(np.float32)
    out = quiet(supres.DSen2_20, d10, d20, deep=False)
    supres.clear_model_cache()
    ref = oracle_dsen2_tile(name, flat)
    err = do.rmse(out.astype(np.float64), ref) / 2000
    print('%s DSen2_20 bf16x3: normalised rmse %.3e' % (name, err))
    assert out.shape == (600, 600, 6) and err < RMSE_GATE and err < X3_EXPECTED


def test_bf16x3_dsen2_60_on_a_whole_bundled_tile(golden_dir, tmp_path, monkeypatch, oracle_dsen2_tile):
    """DSen2_60 (12-band input, two up-sampling passes, 16 patches of 192^2) in bf16x3 on a tile the reference ships."""
    from dsen2_amd import supres
    flat = do.he_uniform_weights(12, 2, 6, 128, seed=32, bias_scale=0.02)
    np.save(str(tmp_path / 's2_030_lr_1e-05.npy'), flat)
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    monkeypatch.setattr(supres, 'PRECISION', 'bf16x3')
    supres.clear_model_cache()
    g = np.load(os.path.join(golden_dir, 'tile_T33UUB_600.npz'))
    d = [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]
    out = quiet(supres.DSen2_60, d[0], d[1], d[2], deep=False)
    supres.clear_model_cache()
    ref = oracle_dsen2_tile('tile_T33UUB_600.npz', flat, run_60=True)
    err = do.rmse(out.astype(np.float64), ref) / 2000
    print('T33UUB DSen2_60 bf16x3: normalised rmse %.3e' % err)
    assert out.shape == (600, 600, 2) and err < RMSE_GATE and err < X3_EXPECTED


@pytest.mark.parametrize('d,feat,n,h,w', [(2, 128, 2, 32, 32), (1, 256, 1, 21, 37), (3, 128, 1, 16, 33)])
def test_precision2_forward_is_the_chain_of_its_kernel_level_operations(d, feat, n, h, w):
    """dsen2_model_forward with precision 2 against the same network assembled from the kernel-level entry points (first
    convolution in bf16x3 on the matrix cores writing the stream's three planes, dsen2_conv3x3_first_planes -> bf16x3 conv-A /
    conv-B on the planes -> fp32 output convolution): bit for bit.  Pins what the forward strings together — the in-place
    updates, the last block's fp32 form — to operations tested on their own (the first one: test_gpu_first16.py)."""
    from dsen2_amd.DSen2Net import conv3x3_body_bf16x3, conv3x3_first_planes, conv3x3_nhwc, s2model
    flat = do.he_uniform_weights(10, 6, d, feat, seed=d + feat, bias_scale=0.05)
    xs = do.synthetic_inputs(n, h, w, (4, 6), seed=3)
    m = s2model(((4, None, None), (6, None, None)), num_layers=d, feature_size=feat, precision='bf16x3')
    m.set_weights_flat(flat)
    xd = [torch.from_numpy(a).cuda() for a in xs]
    y = m.forward_device(xd)
    layers = do.split_weights(np.asarray(flat), 10, 6, d, feat)
    hx, lo = conv3x3_first_planes(xd, layers[0][0], layers[0][1], precision=2)
    for i in range(d):
        (ka, ba), (kb, bb) = layers[1 + 2 * i], layers[2 + 2 * i]
        t = conv3x3_body_bf16x3(hx, ka, ba, epilogue=0)
        if i + 1 < d:
            conv3x3_body_bf16x3(t, kb, bb, epilogue=1, res_hx=hx, res_lo=lo, res_scale=0.1)
        else:
            a = conv3x3_body_bf16x3(t, kb, bb, epilogue=3, res_hx=hx, res_lo=lo, res_scale=0.1)
    want = conv3x3_nhwc(a, layers[-1][0], layers[-1][1], epilogue=2, aux=xd[1].contiguous())
    assert torch.equal(y, want)


@pytest.mark.parametrize('feat,d,n,h,w', [
    (128, 2, 512, 32, 32),      # BASELINE configs[1]'s batch: two patches per workgroup, seamless boundaries
    (128, 3, 256, 32, 32),      # one patch per workgroup at F = 128: drained boundaries
    (128, 2, 401, 32, 32),      # odd batch: the tail workgroup owns one patch and drains, the others run seamless
    (128, 2, 301, 16, 32),      # ... with a single item per layer in the tail workgroup
    (256, 1, 256, 32, 32),      # F = 256, one patch per workgroup: seamless (slab 1 = virtual input chunks 12-23)
    (256, 1, 256, 48, 40),      # 3 x 2 tiles per patch, ragged last column
    (256, 2, 511, 32, 32),      # F = 256, two patches per workgroup except the TAIL one (one patch): seamless everywhere
    (256, 2, 511, 16, 32),      # ... with ONE tile (two items: slab 0, slab 1) per layer in the tail workgroup
])
def test_bf16x3_chain_kernel_equals_the_per_layer_kernels_bit_for_bit(feat, d, n, h, w):
    """precision 2: a batch that gives every CU whole patches runs its 2d body convolutions as ONE chain launch
    (conv3x3_body16w_x3_chain_kernel); a 5-patch sub-batch of the same inputs runs layer by layer (or, for single-tile
    patches, as a drained chain).  Same arithmetic per item: the same bits, for every form of the layer boundary."""
    flat = do.he_uniform_weights(10, 6, d, feat, seed=d + feat + n, bias_scale=0.05)
    rng = np.random.Generator(np.random.PCG64(n + h))
    xs = [rng.random((n, c, h, w), dtype=np.float32) * np.float32(5.0) for c in (4, 6)]
    m = _model((4, 6), d, feat, flat)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    assert m.body_launches(n, h, w) == 1
    if h * w > 16 * 32:
        assert m.body_launches(5, h, w) == 2 * d
    y = m.forward_device(dev)
    for first in (0, n // 2 - 2, n - 5):
        sub = m.forward_device([t[first:first + 5].contiguous() for t in dev])
        assert torch.equal(sub, y[first:first + 5]), (feat, d, n, h, w, first)
    ref = c_oracle.forward([a[:2] for a in xs], flat, d, feat)
    assert do.rmse(y[:2].cpu().numpy(), ref) < X3_EXPECTED
