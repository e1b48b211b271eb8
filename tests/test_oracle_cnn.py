"""The CNN oracle has no reference output to be pinned against (keras absent: "parity unpinned").
What can be checked on CPU: three independent implementations of the same graph agree
(numpy float64, plain C float64, torch conv2d float64) and reproduce the frozen fixtures."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_oracle
from oracle import dsen2_oracle as do

CNN_CASES = ['cnn_20_d6_f128', 'cnn_60_d6_f128', 'cnn_20_d2_f256', 'cnn_20_d6_f128_ragged']


def load_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    bands = tuple(int(b) for b in g['bands'])
    d, f = int(g['num_layers']), int(g['feature_size'])
    flat = do.he_uniform_weights(sum(bands), bands[-1], d, f, seed=int(g['weight_seed']),
                                 bias_scale=float(g['bias_scale']))
    assert abs(float(flat.astype(np.float64).sum()) - float(g['weights_checksum'])) < 1e-9
    inputs = [g['x%d' % i] for i in range(len(bands))]
    return inputs, flat, d, f, g['out']


def torch_forward(inputs, flat, d, f, dtype=torch.float64):
    xs = [torch.from_numpy(np.asarray(a)).to(dtype) for a in inputs]
    x = torch.cat(xs, dim=1)
    layers = do.split_weights(flat, x.shape[1], xs[-1].shape[1], d, f)

    def conv(t, k, b):
        w = torch.from_numpy(np.ascontiguousarray(k.transpose(3, 2, 0, 1))).to(dtype)   # HWIO -> OIHW
        return F.conv2d(t, w, torch.from_numpy(b).to(dtype), padding=1)

    x = F.relu(conv(x, *layers[0]))
    for i in range(d):
        t = F.relu(conv(x, *layers[1 + 2 * i]))
        x = x + conv(t, *layers[2 + 2 * i]) * 0.1
    return (conv(x, *layers[-1]) + xs[-1]).numpy()


def test_param_counts_match_survey():
    assert do.num_params(10, 6, 6, 128) == 1789574        # DSen2_20
    assert do.num_params(12, 2, 6, 128) == 1787266        # DSen2_60
    assert do.num_params(10, 6, 32, 256) == 37802246      # VDSen2_20


def test_conv3x3_is_cross_correlation_with_zero_pad():
    """Known-answer: a delta kernel at tap (dy,dx) shifts the image by (dy-1, dx-1), zeros outside."""
    x = np.arange(2 * 1 * 4 * 5, dtype=np.float64).reshape(2, 1, 4, 5) + 1
    for dy in range(3):
        for dx in range(3):
            k = np.zeros((3, 3, 1, 1), np.float32); k[dy, dx] = 1
            exp = np.zeros_like(x)
            ys = slice(max(0, 1 - dy), min(4, 5 - dy)); xs = slice(max(0, 1 - dx), min(5, 6 - dx))
            exp[:, :, ys, xs] = x[:, :, ys.start + dy - 1:ys.stop + dy - 1, xs.start + dx - 1:xs.stop + dx - 1]
            for impl in (do.conv3x3, c_oracle.conv3x3):
                got = impl(x, k, np.zeros(1, np.float32))
                assert np.array_equal(got, exp), (dy, dx, impl)


@pytest.mark.parametrize('name', CNN_CASES)
def test_numpy_c_torch_agree_and_match_fixture(golden_dir, name):
    inputs, flat, d, f, frozen = load_case(golden_dir, name)
    y_np = do.forward(inputs, flat, d, f)
    y_c = c_oracle.forward(inputs, flat, d, f)
    y_t = torch_forward(inputs, flat, d, f)
    for y in (y_np, y_c, y_t):
        assert y.shape == frozen.shape
        assert do.rmse(y, frozen) < 1e-12, name          # float64: summation-order noise only
        assert np.abs(y - frozen).max() < 1e-11


def test_float32_cpu_graph_is_within_gate_of_oracle(golden_dir):
    """The cpu_baseline graph (torch CPU fp32) stays within the 1e-4 RMSE gate of the float64 oracle."""
    inputs, flat, d, f, frozen = load_case(golden_dir, 'cnn_20_d6_f128')
    y32 = torch_forward(inputs, flat, d, f, dtype=torch.float32)
    assert do.rmse(y32, frozen) < 1e-5


def test_c_upsample_matches_numpy_and_reference(golden_dir):
    from oracle import patches_oracle as po
    g = np.load(os.path.join(golden_dir, 'interp.npz'))
    for src, key in [('a', 'a_x2'), ('a', 'a_x6'), ('b', 'b_x2'), ('b', 'b_x6'), ('ramp', 'ramp_x6')]:
        oh, ow = g[key].shape[2:]
        c = c_oracle.upsample(g[src], oh, ow)
        n = po.interp_patches(g[src], g[key].shape)
        np.testing.assert_allclose(c, n, rtol=2e-7, atol=1e-3)           # same maths, both exact coords
        np.testing.assert_allclose(c, g[key], rtol=0, atol=3e-2)         # vs skimage (f32 coords)
        # the C restatement of scikit-image's own float32 arithmetic: the reference's bits, like the numpy one
        cs = c_oracle.upsample(g[src], oh, ow, skimage=True)
        assert cs.tobytes() == g[key].tobytes() == po.interp_patches(g[src], g[key].shape, f32_coords=True).tobytes(), key
    gs = np.load(os.path.join(golden_dir, 'interp_shapes.npz'))
    for k in range(len([f for f in gs.files if f.startswith('in_')])):
        x, want = gs['in_%02d' % k], gs['out_%02d' % k]
        assert c_oracle.upsample(x, want.shape[2], want.shape[3], skimage=True).tobytes() == want.tobytes(), k
