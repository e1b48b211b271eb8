"""Kernel-level parity: one HIP convolution (through the C ABI) vs the float64 C oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle                      # checker only
from oracle import dsen2_oracle as do


def _nhwc(x_nchw):
    return torch.from_numpy(np.ascontiguousarray(x_nchw.transpose(0, 2, 3, 1))).cuda()


def _rand(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


# Gates = 10 x what the kernels measure against the float64 oracle on these operands (inputs N(0,1), he-normal kernels:
# outputs of rms ~1.4; `python -m pytest tests/test_gpu_conv.py -m gpu -s` prints every case's figures, and
# profiles/r04_conv_gates.txt holds the run the numbers below were read from).  An fp32 MFMA chain over K = 9*Cin products
# has rms error ~ 6e-8 * sqrt(K) * |a*b|, so the gate scales with sqrt(Cin); the residual epilogue scales the
# convolution's error by res_scale = 0.1 and adds one rounding of the sum.  (Round 3 gated at 2e-6 * sqrt(9*Cin) = 6.8e-5 at
# Cin 128: a systematic 1e-5 error would have passed.)
# measured (profiles/r04_conv_gates.txt): relu 2.1e-7 / 5.9e-7 / 8.3e-7 at Cin 16 / 128 / 256; residual 8.7e-8 / 1.2e-7; output
# convolution on the matrix cores 3.0e-7 / 4.2e-7 (nine partial sums of F products), on the vector units 8.3e-7 (one chain of 9F)
RMS_GATE = {'relu': {16: 2e-6, 128: 5e-6, 256: 8e-6}, 'residual': {128: 9e-7, 256: 1.2e-6}, 'out': {128: 3e-6, 256: 4e-6},
            'out_valu': {128: 8e-6}}
MAX_GATE_FACTOR = 12          # max |error| <= 12 x the rms gate (a 5-sigma tail over <= 1e6 outputs is ~ 5 x the rms)


def _check(y, ref, kind, cin, what=''):
    e, mx = do.rmse(y, ref), float(np.abs(y - ref).max())
    gate = RMS_GATE[kind][cin]
    print('%-9s cin %3d %s: rmse %.3e (gate %.1e)  max %.3e (gate %.1e)  output rms %.3f'
          % (kind, cin, what, e, gate, mx, MAX_GATE_FACTOR * gate, float(np.sqrt(np.mean(ref * ref)))))
    assert e < gate, (kind, cin, what, e)
    assert mx < MAX_GATE_FACTOR * gate, (kind, cin, what, mx)


@pytest.mark.parametrize('cin,cout,n,h,w', [
    (16, 128, 2, 32, 32),       # conv_in geometry (10 or 12 real channels zero-padded to 16)
    (128, 128, 2, 32, 32),      # body conv, BASELINE config patch size
    (128, 128, 1, 16, 16),      # single tile
    (128, 128, 1, 21, 37),      # ragged: H, W not multiples of the 16x16 tile
    (128, 128, 3, 48, 16),      # non-square, several tiles
    (256, 256, 1, 32, 32),      # VDSen2 width: two output slabs, 8 channel chunks
    (16, 256, 1, 19, 16),
])
def test_conv_relu_matches_oracle(cin, cout, n, h, w):
    rng = np.random.default_rng(cin * 1000 + cout + h)
    x = _rand(rng, (n, cin, h, w))
    k = _rand(rng, (3, 3, cin, cout), np.sqrt(2.0 / (9 * cin)))
    b = _rand(rng, (cout,), 0.1)
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    y = conv3x3_nhwc(_nhwc(x), k, b, epilogue=0).cpu().numpy().transpose(0, 3, 1, 2)
    ref = c_oracle.conv3x3(x, k, b, relu=True)
    _check(y, ref, 'relu', cin, '%dx%dx%d->%d' % (n, h, w, cout))
    assert (y >= 0).all()


@pytest.mark.parametrize('feat,n,h,w', [(128, 2, 32, 32), (128, 1, 23, 18), (256, 1, 16, 32)])
def test_conv_residual_matches_oracle(feat, n, h, w):
    rng = np.random.default_rng(feat + h)
    x = _rand(rng, (n, feat, h, w))
    res = _rand(rng, (n, feat, h, w))
    k = _rand(rng, (3, 3, feat, feat), np.sqrt(2.0 / (9 * feat)))
    b = _rand(rng, (feat,), 0.1)
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    y = conv3x3_nhwc(_nhwc(x), k, b, epilogue=1, aux=_nhwc(res), res_scale=0.1).cpu().numpy().transpose(0, 3, 1, 2)
    ref = res.astype(np.float64) + 0.1 * c_oracle.conv3x3(x, k, b, relu=False)     # DSen2Net.py:12-15
    _check(y, ref, 'residual', feat, '%dx%dx%d' % (n, h, w))


def test_conv_residual_in_place():
    """forward() runs conv-B in place on the residual stream (aux == out)."""
    rng = np.random.default_rng(5)
    x = _rand(rng, (1, 128, 32, 32)); res = _rand(rng, (1, 128, 32, 32))
    k = _rand(rng, (3, 3, 128, 128), 0.05); b = _rand(rng, (128,), 0.1)
    import ctypes
    from dsen2_amd import _lib
    xin, r = _nhwc(x), _nhwc(res)
    _lib.call('dsen2_conv3x3_nhwc', ctypes.c_void_p(xin.data_ptr()), k.ctypes.data_as(_lib.c_float_p),
              b.ctypes.data_as(_lib.c_float_p), ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(r.data_ptr()),
              1, 32, 32, 128, 128, 1, 0.1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    ref = res.astype(np.float64) + 0.1 * c_oracle.conv3x3(x, k, b)
    _check(r.cpu().numpy().transpose(0, 3, 1, 2), ref, 'residual', 128, 'in place')


@pytest.mark.parametrize('feat,cout,n,h,w', [(128, 6, 2, 32, 32), (128, 2, 1, 20, 35), (256, 6, 1, 16, 16)])
def test_conv_out_skip_nchw_matches_oracle(feat, cout, n, h, w):
    rng = np.random.default_rng(feat + cout)
    x = _rand(rng, (n, feat, h, w))
    skip = _rand(rng, (n, cout, h, w))
    k = _rand(rng, (3, 3, feat, cout), np.sqrt(2.0 / (9 * feat)))
    b = _rand(rng, (cout,), 0.1)
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    y = conv3x3_nhwc(_nhwc(x), k, b, epilogue=2, aux=torch.from_numpy(skip).cuda()).cpu().numpy()
    ref = c_oracle.conv3x3(x, k, b) + skip                                         # DSen2Net.py:35,38,41
    assert y.shape == (n, cout, h, w)
    _check(y, ref, 'out', feat, '%dx%dx%d->%d' % (n, h, w, cout))


# shapes of the tap-expanded matrix-core output kernel (conv3x3_out_mfma.hip): rows cut into 32-pixel blocks (edges at
# x = 31 | 32), several rows per wave and phase (W <= 64), strips of rows when there are few images, Cout 6 and 2, F = 256;
# and shapes it hands to the vector-unit kernel (Cout 7, a row of Q too wide for LDS)
OUT_SHAPES = [(128, 6, 1, 128, 128), (128, 2, 1, 50, 192), (128, 6, 3, 40, 70), (256, 6, 1, 33, 64), (128, 5, 2, 7, 31),
              (128, 3, 1, 1, 1), (256, 2, 2, 67, 33), (128, 6, 1, 70, 97), (128, 7, 1, 20, 40), (128, 6, 1, 9, 230)]


@pytest.mark.parametrize('feat,cout,n,h,w', OUT_SHAPES)
def test_conv_out_shapes_match_oracle(feat, cout, n, h, w):
    rng = np.random.default_rng(feat + cout + h)
    x = _rand(rng, (n, feat, h, w))
    skip = _rand(rng, (n, cout, h, w))
    k = _rand(rng, (3, 3, feat, cout), np.sqrt(2.0 / (9 * feat)))
    b = _rand(rng, (cout,), 0.1)
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    y = conv3x3_nhwc(_nhwc(x), k, b, epilogue=2, aux=torch.from_numpy(skip).cuda()).cpu().numpy()
    ref = c_oracle.conv3x3(x, k, b) + skip
    assert y.shape == (n, cout, h, w)
    # (Cout 7-8 and rows of Q too wide for LDS go to the vector-unit kernel: conv3x3_out.hip)
    kind = 'out_valu' if cout > 6 or w > 224 else 'out'
    _check(y, ref, kind, feat, '%dx%dx%d->%d' % (n, h, w, cout))      # incl. max: no single pixel off (a block edge, a strip's first row)


@pytest.mark.parametrize('feat,cout,h,w', [(128, 6, 32, 32), (128, 6, 45, 100), (128, 2, 40, 192), (256, 6, 20, 64)])
def test_conv_out_delta_kernels_shift_exactly(feat, cout, h, w):
    """Known-answer for the output convolution: a one-hot kernel copies a shifted input channel bit-exactly into ONE
    output channel (zero outside the image, across every block edge and strip boundary), bias and skip added after."""
    rng = np.random.default_rng(h + w)
    x = _rand(rng, (1, feat, h, w))
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    zero = torch.zeros(1, cout, h, w, device='cuda')
    for dy, dx, ci, co in [(0, 0, 3, cout - 1), (2, 1, feat - 1, 0), (1, 1, feat // 2, 1), (0, 2, 31, 0), (2, 2, 64, cout - 1),
                           (1, 0, 95, 1), (2, 0, 7, 0)]:
        k = np.zeros((3, 3, feat, cout), np.float32); k[dy, dx, ci, co] = 1
        y = conv3x3_nhwc(_nhwc(x), k, np.zeros(cout, np.float32), epilogue=2, aux=zero).cpu().numpy()
        exp = np.zeros((h, w), np.float32)
        ys = slice(max(0, 1 - dy), min(h, h + 1 - dy)); xs = slice(max(0, 1 - dx), min(w, w + 1 - dx))
        exp[ys, xs] = x[0, ci, ys.start + dy - 1:ys.stop + dy - 1, xs.start + dx - 1:xs.stop + dx - 1]
        assert np.array_equal(y[0, co], exp), (dy, dx, ci, co)
        assert not np.delete(y[0], co, axis=0).any()


def test_conv_out_does_not_depend_on_how_the_image_is_cut():
    """The same image alone (cut into strips of rows, one per workgroup) and inside a batch of 600 (a workgroup takes the
    whole image): the same bits — every sum's order is fixed by (y, x)."""
    rng = np.random.default_rng(77)
    x = _rand(rng, (1, 128, 64, 64)); skip = _rand(rng, (1, 6, 64, 64))
    k = _rand(rng, (3, 3, 128, 6), np.sqrt(2.0 / (9 * 128))); b = _rand(rng, (6,), 0.1)
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    alone = conv3x3_nhwc(_nhwc(x), k, b, epilogue=2, aux=torch.from_numpy(skip).cuda())
    xb = torch.zeros(600, 64, 64, 128, device='cuda'); xb[417] = _nhwc(x)[0]
    sb = torch.zeros(600, 6, 64, 64, device='cuda'); sb[417] = torch.from_numpy(skip[0]).cuda()
    batch = conv3x3_nhwc(xb, k, b, epilogue=2, aux=sb)
    assert torch.equal(batch[417], alone[0])
    assert torch.equal(batch[0], torch.from_numpy(np.broadcast_to(b[:, None, None], (6, 64, 64)).copy()).cuda())


def test_delta_kernel_shifts_exactly():
    """Known-answer: a one-hot kernel copies a shifted input channel bit-exactly (zero outside)."""
    rng = np.random.default_rng(9)
    x = _rand(rng, (1, 128, 32, 32))
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    for dy, dx, ci, co in [(0, 0, 3, 5), (2, 1, 127, 0), (1, 1, 64, 127), (0, 2, 31, 32)]:
        k = np.zeros((3, 3, 128, 128), np.float32); k[dy, dx, ci, co] = 1
        # use the residual epilogue with zero aux and scale 1 so negatives survive
        y = conv3x3_nhwc(_nhwc(x), k, np.zeros(128, np.float32), epilogue=1,
                         aux=torch.zeros(1, 32, 32, 128, device='cuda'), res_scale=1.0).cpu().numpy()
        exp = np.zeros((32, 32), np.float32)
        ys = slice(max(0, 1 - dy), min(32, 33 - dy)); xs = slice(max(0, 1 - dx), min(32, 33 - dx))
        exp[ys, xs] = x[0, ci, ys.start + dy - 1:ys.stop + dy - 1, xs.start + dx - 1:xs.stop + dx - 1]
        assert np.array_equal(y[0, :, :, co], exp), (dy, dx, ci, co)
        other = np.delete(y[0], co, axis=2)
        assert not other.any()


@pytest.mark.parametrize('feat,n,h,w', [(128, 3, 21, 37), (128, 40, 32, 32), (256, 2, 32, 32), (256, 1, 19, 16)])
def test_persistent_kernel_is_bit_identical_to_the_reference_structure(feat, n, h, w):
    """The persistent DMA-fed body kernel (conv3x3_body32.hip) and the one-tile-per-workgroup kernel
    (dsen2_conv3x3_nhwc_ref) compute the same sums in the same order: bit-identical, both epilogues."""
    rng = np.random.default_rng(feat + n)
    x = _nhwc(_rand(rng, (n, feat, h, w)))
    res = _nhwc(_rand(rng, (n, feat, h, w)))
    k = _rand(rng, (3, 3, feat, feat), np.sqrt(2.0 / (9 * feat)))
    b = _rand(rng, (feat,), 0.1)
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    for epi, aux in ((0, None), (1, res)):
        a = conv3x3_nhwc(x, k, b, epilogue=epi, aux=aux)
        r = conv3x3_nhwc(x, k, b, epilogue=epi, aux=aux, ref=True)
        assert torch.equal(a, r), (feat, epi)


def test_bad_arguments_raise():
    from dsen2_amd import _lib
    from dsen2_amd.DSen2Net import conv3x3_nhwc
    x = torch.zeros(1, 8, 8, 24, device='cuda')
    with pytest.raises(_lib.DSen2Error):
        conv3x3_nhwc(x, np.zeros((3, 3, 24, 128), np.float32), np.zeros(128, np.float32))
