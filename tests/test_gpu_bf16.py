"""bf16-operand path (v_mfma_f32_32x32x16_bf16, fp32 accumulate): kernel-level and whole-network parity.

Kernel level the check is TIGHT: inputs and weights are rounded to bf16 first, so every product is exact in
fp32 and only the summation order differs from the float64 oracle run on the same rounded values.
Network level the tolerance is a bf16 one (the reference is fp32; BASELINE.md §2: "bf16 reported, gated
against a bf16-appropriate tolerance")."""
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


@pytest.mark.parametrize('feat,n,h,w', [(256, 2, 32, 32), (256, 1, 21, 37), (128, 2, 32, 32), (128, 1, 16, 48)])
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


@pytest.mark.parametrize('feat,n,h,w', [(256, 1, 32, 32), (128, 2, 19, 32)])
def test_bf16_conv_residual_fp32_stream(feat, n, h, w):
    from dsen2_amd.DSen2Net import conv3x3_body_bf16
    rng = np.random.default_rng(7 + feat)
    x = bf16_round(rng.standard_normal((n, feat, h, w)))
    k = bf16_round(rng.standard_normal((3, 3, feat, feat)) * np.sqrt(2.0 / (9 * feat)))
    b = (rng.standard_normal(feat) * 0.1).astype(np.float32)
    res = rng.standard_normal((n, feat, h, w)).astype(np.float32)
    aux = torch.from_numpy(np.ascontiguousarray(res.transpose(0, 2, 3, 1))).cuda()
    out, out_bf = conv3x3_body_bf16(nhwc_bf16(x), k, b, epilogue=1, aux=aux, res_scale=0.1)
    ref = res.astype(np.float64) + 0.1 * c_oracle.conv3x3(x, k, b)
    y = out.cpu().numpy().transpose(0, 3, 1, 2)
    assert do.rmse(y, ref) < 2e-6                                  # fp32 stream: exact products, fp32 accumulate
    # the bf16 copy is the RNE rounding of the fp32 output
    assert torch.equal(out_bf, out.to(torch.bfloat16))


def test_bf16_network_vs_fp32_oracle():
    """DSen2-width network (d=6, F=128) and a shallow VDSen2-width one (d=4, F=256) in bf16 vs the float64
    oracle with fp32 weights: error budget ~ 2^-9 per operand rounding, damped by the 0.1 residual scale."""
    from dsen2_amd.DSen2Net import s2model
    for d, f, seed in [(6, 128, 1), (4, 256, 2)]:
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


def test_bf16_network_ragged_and_variant():
    """Ragged image size through the bf16 network, every kernel structure (tuning key 4)."""
    from dsen2_amd import _lib
    from dsen2_amd.DSen2Net import s2model
    flat = do.he_uniform_weights(10, 6, 3, 256, seed=9, bias_scale=0.05)
    xs = do.synthetic_inputs(1, 21, 37, (4, 6), seed=9)
    ref = c_oracle.forward(xs, flat, 3, 256)
    scale = float(np.sqrt(np.mean(ref ** 2)))
    outs = []
    try:
        for v in (0, 2, 3, 4, 5, 6, 7):       # every structure of the 256->256 bf16 body convolution (tuning key 4)
            _lib.call('dsen2_set_tuning', 4, v)
            m = s2model(((4, None, None), (6, None, None)), num_layers=3, feature_size=256, precision='bf16')
            m.set_weights_flat(flat)
            outs.append(m.predict(xs))
            assert do.rmse(outs[-1], ref) / scale < 5e-3
    finally:
        _lib.call('dsen2_set_tuning', 4, 4)      # the library default
    # same products, same fp32 accumulation order per output element? (64- vs 32-channel steps differ only in
    # where the k loop is cut, not in its order) -> the two structures agree to fp32 rounding of the bf16 copies
    for o in outs[1:]:
        assert np.abs(outs[0] - o).max() < 1e-2 * scale
