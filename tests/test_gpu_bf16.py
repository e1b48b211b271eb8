"""bf16-operand path (v_mfma_f32_16x16x32_bf16, fp32 accumulate; conv3x3_body16w.hip): kernel-level and
whole-network parity.

Kernel level the check is TIGHT: inputs and weights are rounded to bf16 first, so every product is exact in
fp32 and only the summation order differs from the float64 oracle run on the same rounded values.
Network level the tolerance is a bf16 one (the reference is fp32; BASELINE.md §2: "bf16 reported, gated
against a bf16-appropriate tolerance"); the full-depth VDSen2 cases are in test_gpu_vdsen2_bf16.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import dsen2_oracle as do


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def nhwc_bf16(x_nchw):
    return torch.from_numpy(np.ascontiguousarray(x_nchw.transpose(0, 2, 3, 1))).cuda().to(torch.bfloat16)


def np_split(u32):
    """The residual stream's plane format restated in numpy (include/dsen2_hip.h, dsen2_split_f32)."""
    u = u32.astype(np.uint64)
    return (((u + 0x8000) >> 16) & 0xffff).astype(np.uint16), (u & 0xffff).astype(np.uint16)


def test_split_join_round_trips_every_kind_of_value():
    """hi = (u + 0x8000) >> 16, lo = u & 0xffff; join restores u bit for bit — including NaNs, infinities,
    denormals, values that round up to the next exponent and the 2^16 ties."""
    from dsen2_amd.DSen2Net import from_blocked, join_f32, split_f32
    rng = np.random.default_rng(0)
    special = np.array([0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0xffffffff, 0x7f7fffff, 0x00000001,
                        0x00008000, 0x00018000, 0x3f808000, 0x3f818000, 0x3f80ffff, 0xbf808000, 0x7f7f8000, 0xffff8000],
                       np.uint32)
    u = np.concatenate([special, rng.integers(0, 2 ** 32, size=4096 - special.size, dtype=np.uint64).astype(np.uint32)])
    x = torch.from_numpy(u.view(np.float32).reshape(2, 4, 8, 64)).cuda()        # NHWC, 8 blocks of 8 channels
    hi, lo = split_f32(x)
    assert hi.shape == (2, 8, 4, 8, 8)                                          # blocked: [n][c/8][h][w][8]
    eh, el = np_split(u)
    assert np.array_equal(from_blocked(hi).cpu().numpy().view(np.uint16).ravel(), eh)
    assert np.array_equal(from_blocked(lo).cpu().numpy().view(np.uint16).ravel(), el)
    back = join_f32(hi, lo).cpu().numpy().view(np.uint32).ravel()
    assert np.array_equal(back, u)
    # hi is the bf16 rounding of the value, ties away from zero: never more than half a bf16 ulp off
    finite = np.isfinite(u.view(np.float32)) & (np.abs(u.view(np.float32)) < 1e38) & (np.abs(u.view(np.float32)) > 1e-30)
    as_f = (eh.astype(np.uint32) << 16).view(np.float32)
    rel = np.abs(as_f[finite].astype(np.float64) - u.view(np.float32)[finite]) / np.abs(u.view(np.float32)[finite])
    assert rel.max() <= 2.0 ** -8


@pytest.mark.parametrize('c,n,h,w', [(512, 2, 5, 13), (8, 3, 7, 9), (504, 1, 1, 33)])
def test_split_join_at_the_documented_channel_limits(c, n, h, w):
    """include/dsen2_hip.h: c % 8 == 0, c <= 512.  c = 512 needs 66,048 B of dynamic LDS — above the 64 KiB a kernel
    gets without the MaxDynamicSharedMemorySize attribute; c = 520 is refused, not launched."""
    from dsen2_amd import _lib
    from dsen2_amd.DSen2Net import from_blocked, join_f32, split_f32
    u = np.random.default_rng(c).integers(0, 2 ** 32, size=n * h * w * c, dtype=np.uint64).astype(np.uint32)
    x = torch.from_numpy(u.view(np.float32).reshape(n, h, w, c)).cuda()
    hi, lo = split_f32(x)
    eh, el = np_split(u)
    assert np.array_equal(from_blocked(hi).cpu().numpy().view(np.uint16).ravel(), eh)
    assert np.array_equal(from_blocked(lo).cpu().numpy().view(np.uint16).ravel(), el)
    assert np.array_equal(join_f32(hi, lo).cpu().numpy().view(np.uint32).ravel(), u)
    with pytest.raises(RuntimeError):
        split_f32(torch.zeros((1, 2, 2, 520), device='cuda'))


@pytest.mark.parametrize('feat,n,h,w', [(256, 2, 32, 32), (256, 1, 21, 37), (128, 2, 32, 32), (128, 1, 16, 48),
                                        (256, 1, 16, 33), (128, 3, 5, 70), (256, 1, 1, 1)])
def test_bf16_conv_relu_exact_products(feat, n, h, w):
    from dsen2_amd.DSen2Net import conv3x3_body_bf16
    rng = np.random.default_rng(feat + h)
    x = bf16_round(rng.standard_normal((n, feat, h, w)))
    k = bf16_round(rng.standard_normal((3, 3, feat, feat)) * np.sqrt(2.0 / (9 * feat)))
    b = (rng.standard_normal(feat) * 0.1).astype(np.float32)
    y = conv3x3_body_bf16(nhwc_bf16(x), k, b, epilogue=0).to(torch.float32).cpu().numpy().transpose(0, 3, 1, 2)
    ref = c_oracle.conv3x3(x, k, b, relu=True)
    # output is rounded to bf16 (8 significant bits): half an ulp = 2^-9 relative
    np.testing.assert_allclose(y, ref, rtol=2.0 ** -8, atol=1e-3)
    assert do.rmse(y, ref) < 4e-3


@pytest.mark.parametrize('feat,n,h,w', [(256, 1, 32, 32), (128, 2, 19, 32), (256, 2, 17, 40), (128, 1, 3, 5)])
def test_bf16_conv_residual_exact_fp32_stream(feat, n, h, w):
    """conv-B: the fp32 residual stream lives on two 16-bit planes and is updated in place; its fp32 form (the last
    block's epilogue) must be the same bits."""
    from dsen2_amd.DSen2Net import conv3x3_body_bf16, from_blocked, join_f32, split_f32
    rng = np.random.default_rng(7 + feat)
    x = bf16_round(rng.standard_normal((n, feat, h, w)))
    k = bf16_round(rng.standard_normal((3, 3, feat, feat)) * np.sqrt(2.0 / (9 * feat)))
    b = (rng.standard_normal(feat) * 0.1).astype(np.float32)
    res = rng.standard_normal((n, feat, h, w)).astype(np.float32)
    aux = torch.from_numpy(np.ascontiguousarray(res.transpose(0, 2, 3, 1))).cuda()
    hi, lo = split_f32(aux)
    out32 = conv3x3_body_bf16(nhwc_bf16(x), k, b, epilogue=3, res_hi=hi, res_lo=lo, res_scale=0.1)
    assert torch.equal(join_f32(hi, lo), aux)                     # epilogue 3 leaves the planes alone
    conv3x3_body_bf16(nhwc_bf16(x), k, b, epilogue=1, res_hi=hi, res_lo=lo, res_scale=0.1)
    joined = join_f32(hi, lo)
    assert torch.equal(joined, out32)
    ref = res.astype(np.float64) + 0.1 * c_oracle.conv3x3(x, k, b)
    y = out32.cpu().numpy().transpose(0, 3, 1, 2)
    assert do.rmse(y, ref) < 2e-6                                  # fp32 stream: exact products, fp32 accumulate
    # the planes are exactly the split of the fp32 result (hi = the next convolution's operand)
    eh, el = np_split(out32.cpu().numpy().view(np.uint32))
    assert np.array_equal(from_blocked(hi).cpu().numpy().view(np.uint16), eh)
    assert np.array_equal(from_blocked(lo).cpu().numpy().view(np.uint16), el)


def test_bf16_network_vs_fp32_oracle():
    """DSen2-width network (d=6, F=128) and a shallow VDSen2-width one (d=4, F=256) in bf16 vs the float64
    oracle with fp32 weights: error budget ~ 2^-9 per operand rounding, damped by the 0.1 residual scale."""
    from dsen2_amd.DSen2Net import s2model
    for d, f, seed in [(6, 128, 1), (4, 256, 2), (1, 256, 3), (0, 128, 4)]:
        flat = do.he_uniform_weights(10, 6, d, f, seed=seed, bias_scale=0.05)
        xs = do.synthetic_inputs(2, 32, 32, (4, 6), seed=seed)
        m = s2model(((4, None, None), (6, None, None)), num_layers=d, feature_size=f, precision='bf16')
        m.set_weights_flat(flat)
        y = m.predict(xs)
        ref = c_oracle.forward(xs, flat, d, f)
        m32 = s2model(((4, None, None), (6, None, None)), num_layers=d, feature_size=f)
        m32.set_weights_flat(flat)
        y32 = m32.predict(xs)
        e16, e32 = do.rmse(y, ref), do.rmse(y32, ref)
        scale = float(np.sqrt(np.mean(ref ** 2)))
        print('d=%d F=%d: bf16 rmse %.3e (%.2e of signal rms %.2f), fp32 rmse %.3e' % (d, f, e16, e16 / scale, scale, e32))
        assert e32 < 5e-6
        assert e16 / scale < 5e-3                                   # bf16-appropriate gate: 0.5 % of signal rms
        if d == 0:
            assert np.array_equal(y, y32)                           # no residual block: nothing runs in bf16


def test_bf16_network_ragged():
    """Ragged image sizes through the bf16 network (tiles of 16 x 32 pixels: both edges partial)."""
    from dsen2_amd.DSen2Net import s2model
    flat = do.he_uniform_weights(10, 6, 3, 256, seed=9, bias_scale=0.05)
    m = s2model(((4, None, None), (6, None, None)), num_layers=3, feature_size=256, precision='bf16')
    m.set_weights_flat(flat)
    for n, h, w in [(1, 21, 37), (2, 16, 33), (1, 3, 70), (3, 1, 1)]:
        xs = do.synthetic_inputs(n, h, w, (4, 6), seed=9 + h)
        ref = c_oracle.forward(xs, flat, 3, 256)
        scale = float(np.sqrt(np.mean(ref ** 2)))
        assert do.rmse(m.predict(xs), ref) / scale < 5e-3, (n, h, w)


def test_bf16_zero_weights_return_skip_input_exactly_and_full_batch_f128():
    """Size-independent properties of the bf16 path at DSen2 width: all-zero parameters make the network the identity
    on its low-resolution input (the residual planes carry the first convolution's zeros exactly), and a full batch
    of 512 patches is deterministic and permutation-equivariant."""
    from dsen2_amd.DSen2Net import s2model
    xs = do.synthetic_inputs(3, 32, 32, (4, 6, 2), seed=4)
    m = s2model(((4, None, None), (6, None, None), (2, None, None)), num_layers=6, feature_size=128, precision='bf16')
    m.set_weights_flat(np.zeros(do.num_params(12, 2, 6, 128), np.float32))
    assert np.array_equal(m.predict(xs), xs[2])
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=1, bias_scale=0.05)
    xs = do.synthetic_inputs(512, 32, 32, (4, 6), seed=0)
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision='bf16')
    m.set_weights_flat(flat)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    y1 = m.forward_device(dev).clone()
    assert torch.equal(m.forward_device(dev), y1)
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(3)).cuda()
    assert torch.equal(m.forward_device([d[perm].contiguous() for d in dev]), y1[perm])
    idx = [0, 255, 511]
    ref = c_oracle.forward([a[idx] for a in xs], flat, 6, 128)
    assert do.rmse(y1.cpu().numpy()[idx], ref) / float(np.sqrt(np.mean(ref ** 2))) < 5e-3


@pytest.mark.parametrize('d,feat,n,h,w', [(2, 128, 2, 32, 32), (1, 256, 1, 21, 37), (3, 128, 1, 16, 33)])
def test_precision1_forward_is_the_chain_of_its_kernel_level_operations(d, feat, n, h, w):
    """dsen2_model_forward with precision 1 against the same network assembled from the kernel-level entry points
    (first convolution on the bf16 matrix cores writing the (hi, lo) planes, dsen2_conv3x3_first_planes -> bf16 conv-A /
    conv-B on the planes -> fp32 output convolution): bit for bit.  Pins what the forward strings together — the in-place
    plane updates, the last block's fp32 form — to operations tested on their own (the first one: test_gpu_first16.py)."""
    from dsen2_amd.DSen2Net import (conv3x3_body_bf16, conv3x3_first_planes, conv3x3_nhwc, from_blocked, s2model)
    flat = do.he_uniform_weights(10, 6, d, feat, seed=d + feat, bias_scale=0.05)
    xs = do.synthetic_inputs(n, h, w, (4, 6), seed=3)
    m = s2model(((4, None, None), (6, None, None)), num_layers=d, feature_size=feat, precision='bf16')
    m.set_weights_flat(flat)
    xd = [torch.from_numpy(a).cuda() for a in xs]
    y = m.forward_device(xd)
    layers = do.split_weights(np.asarray(flat), 10, 6, d, feat)
    hi, lo = conv3x3_first_planes(xd, layers[0][0], layers[0][1], precision=1)
    for i in range(d):
        (ka, ba), (kb, bb) = layers[1 + 2 * i], layers[2 + 2 * i]
        t = conv3x3_body_bf16(from_blocked(hi.view(torch.bfloat16)), ka, ba, epilogue=0)
        if i + 1 < d:
            conv3x3_body_bf16(t, kb, bb, epilogue=1, res_hi=hi, res_lo=lo, res_scale=0.1)
        else:
            a = conv3x3_body_bf16(t, kb, bb, epilogue=3, res_hi=hi, res_lo=lo, res_scale=0.1)
    want = conv3x3_nhwc(a, layers[-1][0], layers[-1][1], epilogue=2, aux=xd[1].contiguous())
    assert torch.equal(y, want)
