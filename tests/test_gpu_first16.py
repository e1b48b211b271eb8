"""The first convolution of the bf16-operand modes on the bf16 matrix cores (conv3x3_first16.hip), at kernel
level through dsen2_conv3x3_first_planes.

precision 1: the kernel multiplies bf16(x) by bf16(w) — every product exact in fp32 — so against the float64 oracle run on
the SAME rounded operands only the fp32 summation differs: the check is tight.  precision 2 (bf16x3): three bf16 products
per fp32 product, error ~2^-17 of each: checked against the float64 oracle on the fp32 operands with the body kernels' gate.
Both: the planes written are exactly the split of one fp32 result (what the residual blocks read), zero padding at the image
edges, ragged sizes, DSen2_60's 12 channels, batches larger than the grid."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import dsen2_oracle as do

SHAPES = [((4, 6), 128, 2, 32, 32), ((4, 6, 2), 128, 1, 21, 37), ((4, 6), 256, 1, 16, 33), ((4, 6, 2), 256, 2, 5, 70),
          ((4, 6), 128, 1, 1, 1), ((4, 6), 128, 300, 16, 16), ((4, 6), 256, 3, 48, 40)]


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def np_split(u32):
    u = u32.astype(np.uint64)
    return (((u + 0x8000) >> 16) & 0xffff).astype(np.uint16), (u & 0xffff).astype(np.uint16)


def _case(bands, feat, n, h, w, seed):
    rng = np.random.default_rng(seed)
    xs = [(rng.random((n, c, h, w), dtype=np.float32) * np.float32(5.0)) for c in bands]      # U[0, 1) * 5
    cin = sum(bands)
    k = (rng.uniform(-1, 1, (3, 3, cin, feat)) * np.sqrt(6.0 / (9 * cin))).astype(np.float32)   # he_uniform
    b = (rng.standard_normal(feat) * 0.1).astype(np.float32)
    return xs, k, b


@pytest.mark.parametrize('bands,feat,n,h,w', SHAPES)
def test_first_convolution_bf16_exact_products(bands, feat, n, h, w):
    from dsen2_amd.DSen2Net import conv3x3_first_planes, from_blocked, join_f32
    xs, k, b = _case(bands, feat, n, h, w, feat + h + n)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    hi, lo = conv3x3_first_planes(dev, k, b, precision=1)
    assert hi.shape == (n, feat // 8, h, w, 8) and lo.shape == hi.shape
    y32 = join_f32(hi, lo)                                             # the exact fp32 value the planes hold, NHWC
    y = y32.cpu().numpy().transpose(0, 3, 1, 2)
    xr = bf16_round(np.concatenate(xs, axis=1))
    ref = c_oracle.conv3x3(xr, bf16_round(k), b, relu=True)
    scale = float(np.sqrt(np.mean(ref ** 2)))
    assert do.rmse(y, ref) < 1e-6 * max(scale, 1.0), (do.rmse(y, ref), scale)          # measured ~1e-7: fp32 summation only
    assert np.abs(y - ref).max() < 2e-5 * max(scale, 1.0)
    assert (y >= 0).all()                                                                # ReLU
    # the planes are exactly the split of that fp32 value: hi = the next convolution's bf16 operand (ties away)
    eh, el = np_split(y32.cpu().numpy().view(np.uint32))
    assert np.array_equal(from_blocked(hi).cpu().numpy().view(np.uint16), eh)
    assert np.array_equal(from_blocked(lo).cpu().numpy().view(np.uint16), el)
    # against the fp32 network's first layer (operands NOT rounded): the bf16-operand error, ~2^-9 per operand
    full = c_oracle.conv3x3(np.concatenate(xs, axis=1), k, b, relu=True)
    assert do.rmse(y, full) < 6e-3 * max(scale, 1.0)


@pytest.mark.parametrize('bands,feat,n,h,w', SHAPES)
def test_first_convolution_bf16x3_meets_the_fp32_grade(bands, feat, n, h, w):
    from dsen2_amd.DSen2Net import conv3x3_first_planes, from_blocked, join_f32
    xs, k, b = _case(bands, feat, n, h, w, 3 * feat + w)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    hx, lo = conv3x3_first_planes(dev, k, b, precision=2)
    assert hx.shape == (n, 2, feat // 8, h, w, 8) and lo.shape == (n, feat // 8, h, w, 8)
    hi = hx[:, 0].contiguous()
    y32 = join_f32(hi, lo)
    y = y32.cpu().numpy().transpose(0, 3, 1, 2)
    ref = c_oracle.conv3x3(np.concatenate(xs, axis=1), k, b, relu=True)
    scale = float(np.sqrt(np.mean(ref ** 2)))
    err = do.rmse(y, ref)
    assert err < 5e-6 * max(scale, 1.0), (err, scale)            # the gate of the bf16x3 conv-A kernels (measured there: 5e-7 ... 5e-6)
    # plane 1 of hx = xl = bf16(x - hi), round to nearest even: the operand pair (hi | xl) the bf16x3 conv-A multiplies
    x = y32.cpu().numpy()
    hi_f = (from_blocked(hi).cpu().numpy().view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    want_xl = torch.from_numpy(x - hi_f).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    got_xl = from_blocked(hx[:, 1].contiguous()).cpu().numpy().view(np.uint16)
    assert np.array_equal(got_xl, want_xl)
    eh, el = np_split(x.view(np.uint32))
    assert np.array_equal(from_blocked(hi).cpu().numpy().view(np.uint16), eh)
    assert np.array_equal(from_blocked(lo).cpu().numpy().view(np.uint16), el)


def test_first_convolution_planes_refuse_what_they_do_not_support():
    from dsen2_amd.DSen2Net import conv3x3_first_planes
    xs, k, b = _case((4, 6), 128, 1, 8, 8, 1)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    with pytest.raises(RuntimeError):
        conv3x3_first_planes(dev, k, b, precision=0)                  # fp32 has no plane form: dsen2_conv3x3_nhwc
    with pytest.raises(RuntimeError):
        conv3x3_first_planes([dev[0][:, :3].contiguous(), dev[1]], k[:, :, :9], b, precision=1)      # 3 + 6 bands


def test_models_with_other_band_groups_still_run_through_the_generic_first_layer():
    """A precision-1 model whose inputs are not the Sentinel-2 groups (4 + 6 (+ 2)) has no bf16 first convolution: the packed
    fp32 path + plane-writing epilogue of conv3x3_mfma.hip takes over, same gates."""
    from dsen2_amd.DSen2Net import s2model
    flat = do.he_uniform_weights(8, 5, 1, 128, seed=9, bias_scale=0.05)
    xs = do.synthetic_inputs(2, 16, 16, (3, 5), seed=2)
    for prec, gate in (('bf16', 2e-2), ('bf16x3', 1e-4)):
        m = s2model(((3, None, None), (5, None, None)), num_layers=1, feature_size=128, precision=prec)
        m.set_weights_flat(flat)
        y = m.predict(xs)
        ref = c_oracle.forward(xs, flat, 1, 128)
        assert do.rmse(y, ref) < gate


@pytest.mark.parametrize('precision,feat,bands', [(1, 128, (4, 6)), (2, 128, (4, 6)), (2, 256, (4, 6, 2))])
def test_first_convolution_planes_over_many_launches_on_fresh_data(precision, feat, bands):
    """Hazard screen (the kernel's 128-bit buffer stores leave under the next tile's MFMAs and v_permlane32_swap feeds them;
    gfx950 reads store data late): 40 launches on fresh random inputs at several items per workgroup, every plane compared with a
    SECOND launch on the same inputs bit for bit, and the fp32 value the planes hold with the float64 restatement."""
    from dsen2_amd.DSen2Net import conv3x3_first_planes, join_f32
    reps = 40 * int(os.environ.get('DSEN2_STRESS_REPS', '1'))
    rng = np.random.default_rng(precision * 1000 + feat)
    cin = sum(bands)
    k = (rng.uniform(-1, 1, (3, 3, cin, feat)) * np.sqrt(6.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(feat) * 0.1).astype(np.float32)
    bad = 0
    for i in range(reps):
        n, h, w = [(64, 32, 32), (300, 16, 16), (7, 48, 40), (513, 32, 32)][i % 4]
        dev = [torch.rand((n, c, h, w), device='cuda') * 5 for c in bands]
        o1, l1 = conv3x3_first_planes(dev, k, b, precision=precision)
        o2, l2 = conv3x3_first_planes(dev, k, b, precision=precision)
        bad += int(not (torch.equal(o1, o2) and torch.equal(l1, l2)))
        if i % 10 == 0:
            xs = np.concatenate([d[:2].cpu().numpy() for d in dev], axis=1)
            kk = bf16_round(k) if precision == 1 else k
            ref = c_oracle.conv3x3(bf16_round(xs) if precision == 1 else xs, kk, b, relu=True)
            hi = o1[:2] if precision == 1 else o1[:2, 0].contiguous()
            y = join_f32(hi.contiguous(), l1[:2].contiguous()).cpu().numpy().transpose(0, 3, 1, 2)
            assert do.rmse(y, ref) < 5e-6 * max(float(np.sqrt(np.mean(ref ** 2))), 1.0)
    assert bad == 0
