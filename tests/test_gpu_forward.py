"""Whole-network parity through the C ABI: dsen2_model_forward vs the float64 oracle.

Gate: RMSE <= 1e-4 in the network's normalised domain for fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle
from oracle import dsen2_oracle as do

RMSE_GATE = 1e-4
CNN_CASES = ['cnn_20_d6_f128', 'cnn_60_d6_f128', 'cnn_20_d2_f256', 'cnn_20_d6_f128_ragged']


def _model(bands, d, f, flat):
    from dsen2_amd.DSen2Net import s2model
    m = s2model(tuple((b, None, None) for b in bands), num_layers=d, feature_size=f)
    assert m.count_params() == flat.size
    m.set_weights_flat(flat)
    return m


@pytest.mark.parametrize('name', CNN_CASES)
def test_forward_matches_golden_fixture(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    bands = tuple(int(b) for b in g['bands'])
    d, f = int(g['num_layers']), int(g['feature_size'])
    flat = do.he_uniform_weights(sum(bands), bands[-1], d, f, seed=int(g['weight_seed']),
                                 bias_scale=float(g['bias_scale']))
    xs = [g['x%d' % i] for i in range(len(bands))]
    y = _model(bands, d, f, flat).predict(xs)
    assert y.dtype == np.float32 and y.shape == g['out'].shape
    err = do.rmse(y, g['out'])
    print(name, 'rmse', err, 'max', np.abs(y - g['out']).max())
    assert err < RMSE_GATE
    assert err < 5e-6              # what exact-f32 MFMA should actually achieve


def test_forward_config_patch_vs_c_oracle():
    """BASELINE configs[1] geometry (32x32x(4+6), d=6, F=128) at a batch the oracle finishes in seconds."""
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=1)
    xs = do.synthetic_inputs(4, 32, 32, (4, 6), seed=0)
    y = _model((4, 6), 6, 128, flat).predict(xs)
    ref = c_oracle.forward(xs, flat, 6, 128)
    assert do.rmse(y, ref) < 5e-6


def test_forward_batching_is_invisible():
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=3)
    xs = do.synthetic_inputs(5, 32, 32, (4, 6), seed=2)
    m = _model((4, 6), 6, 128, flat)
    a = m.predict(xs)
    b = m.predict(xs, batch_size=2)
    assert np.array_equal(a, b)


def test_zero_weights_return_skip_input_exactly():
    """Size-independent property: with all-zero parameters the network is the identity on its
    low-resolution input (DSen2Net.py:38,41)."""
    xs = do.synthetic_inputs(3, 32, 32, (4, 6, 2), seed=4)
    m = _model((4, 6, 2), 6, 128, np.zeros(do.num_params(12, 2, 6, 128), np.float32))
    assert np.array_equal(m.predict(xs), xs[2])


def test_full_batch_512_properties():
    """BASELINE configs[1] at full size (512 x 32x32): properties that need no oracle at that size,
    plus an oracle check of a sampled subset."""
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=1)
    xs = do.synthetic_inputs(512, 32, 32, (4, 6), seed=0)
    m = _model((4, 6), 6, 128, flat)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    y1 = m.forward_device(dev).clone()
    y2 = m.forward_device(dev)
    assert torch.equal(y1, y2)                                   # deterministic, bit for bit
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(0)).cuda()
    yp = m.forward_device([d[perm].contiguous() for d in dev])
    assert torch.equal(yp, y1[perm])                             # patches are independent units
    y = y1.cpu().numpy()
    assert np.isfinite(y).all()
    idx = [0, 17, 255, 511]
    ref = c_oracle.forward([a[idx] for a in xs], flat, 6, 128)
    assert do.rmse(y[idx], ref) < 5e-6


def test_forward_errors():
    from dsen2_amd import _lib
    from dsen2_amd.DSen2Net import s2model
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128)
    x = [np.zeros((1, 4, 16, 16), np.float32), np.zeros((1, 6, 16, 16), np.float32)]
    with pytest.raises(_lib.DSen2Error) as e:
        m.predict(x)                                             # no weights loaded
    assert e.value.code == _lib.ERR_NO_WEIGHTS
    with pytest.raises(_lib.DSen2Error):
        m.set_weights_flat(np.zeros(10, np.float32))             # wrong parameter count
    with pytest.raises(_lib.DSen2Error):
        s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=100)


def test_empty_batch_returns_empty():
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=1)
    m = _model((4, 6), 6, 128, flat)
    y = m.predict([np.zeros((0, 4, 32, 32), np.float32), np.zeros((0, 6, 32, 32), np.float32)])
    assert y.shape == (0, 6, 32, 32) and y.dtype == np.float32


def test_single_pixel_and_tiny_images():
    """Smallest legal shapes: 1x1 and 3x5 images exercise the zero-padding select and the ragged-tile path alone."""
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=2, bias_scale=0.05)
    m = _model((4, 6), 6, 128, flat)
    for h, w in [(1, 1), (3, 5), (17, 16)]:
        xs = do.synthetic_inputs(2, h, w, (4, 6), seed=h * 10 + w)
        y = m.predict(xs)
        ref = c_oracle.forward(xs, flat, 6, 128)
        assert do.rmse(y, ref) < 5e-6, (h, w)


def test_forward_timed_runs_the_same_forward():
    """dsen2_model_forward_timed (bench.py's roofline hook): same output as dsen2_model_forward, a plausible time."""
    flat = do.he_uniform_weights(10, 6, 2, 128, seed=6, bias_scale=0.05)
    xs = do.synthetic_inputs(4, 32, 32, (4, 6), seed=6)
    m = _model((4, 6), 2, 128, flat)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    y0 = m.forward_device(dev).clone()
    y1 = torch.empty_like(y0)
    ms = m.time_body_in_forward(dev, out=y1, iters=3)
    assert torch.equal(y0, y1)
    assert 0.0 < ms < 50.0


def test_forward_profile_closes_on_itself():
    """dsen2_model_forward_profile (what bench.py's roofline object is built from): the same output as
    dsen2_model_forward; first + body + out = forward (consecutive intervals between the same four events); the host
    clock per instrumented pass is not shorter than the events' forward."""
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=6, bias_scale=0.05)
    xs = do.synthetic_inputs(64, 32, 32, (4, 6), seed=6)
    m = _model((4, 6), 6, 128, flat)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    y0 = m.forward_device(dev).clone()
    y1 = torch.empty_like(y0)
    p = m.profile_forward(dev, out=y1, iters=5)
    assert torch.equal(y0, y1)
    assert all(p[k] > 0 for k in ('forward_ms', 'first_ms', 'body_ms', 'out_ms', 'wall_ms')), p
    assert abs(p['first_ms'] + p['body_ms'] + p['out_ms'] - p['forward_ms']) < 1e-3 * p['forward_ms'] + 2e-3, p
    assert p['body_ms'] > p['first_ms'] and p['body_ms'] > p['out_ms'], p          # 12 of the 14 convolutions
    assert p['wall_ms'] > 0.9 * p['forward_ms'], p


def test_a_handle_belongs_to_the_device_it_was_created_on():
    """include/dsen2_hip.h: one handle per device — a call with another current device returns DSEN2_ERR_INVALID instead
    of handing device A's weights to kernels on device B.  (Needs two GPUs: skipped on a one-GPU box.)"""
    from dsen2_amd import _lib
    if torch.cuda.device_count() < 2:
        pytest.skip('one GPU visible: the mismatch cannot be produced')
    flat = do.he_uniform_weights(10, 6, 1, 128, seed=6)
    with torch.cuda.device(0):
        m = _model((4, 6), 1, 128, flat)
    xs = [torch.zeros((1, c, 16, 16), device='cuda:1') for c in (4, 6)]
    out = torch.empty((1, 6, 16, 16), device='cuda:1')
    ws = torch.empty(m.workspace_bytes(1, 16, 16), dtype=torch.uint8, device='cuda:1')
    with torch.cuda.device(1):
        rc = _lib.load().dsen2_model_forward(m._handle, xs[0].data_ptr(), xs[1].data_ptr(), None, out.data_ptr(), 1, 16, 16,
                                             ws.data_ptr(), ws.numel(), None)
        assert rc == _lib.ERR_INVALID and b'device' in _lib.load().dsen2_last_error()
        assert _lib.load().dsen2_model_body_launches(m._handle, 1, 16, 16) == _lib.ERR_INVALID
        f = np.zeros(m.count_params(), np.float32)
        assert _lib.load().dsen2_model_load_weights(m._handle, f.ctypes.data_as(_lib.c_float_p), f.size) == _lib.ERR_INVALID
    with torch.cuda.device(0):
        assert m.body_launches(1, 16, 16) == 2


@pytest.mark.parametrize('bands,feat', [((4, 6), 128), ((4, 6, 2), 128), ((4, 6), 256)])
def test_first_layer_without_padding_mfmas_gives_the_same_bits(bands, feat):
    """The model's first convolution issues MFMAs for its 10 / 12 real input channels only; the single-layer entry
    point on the SAME data zero-padded to 16 channels runs the generic kernel with all 16.  Skipping zero terms must
    not change one bit (ragged image; a d=0 network = first convolution + output convolution)."""
    from dsen2_amd.DSen2Net import conv3x3_nhwc, s2model
    cin, cout = sum(bands), bands[-1]
    flat = do.he_uniform_weights(cin, cout, 0, feat, seed=8, bias_scale=0.05)
    xs = do.synthetic_inputs(3, 21, 37, bands, seed=8)
    m = s2model(tuple((b, None, None) for b in bands), num_layers=0, feature_size=feat)
    m.set_weights_flat(flat)
    dev = [torch.from_numpy(a).cuda() for a in xs]
    y = m.forward_device(dev)
    (k0, b0), (k1, b1) = do.split_weights(flat, cin, cout, 0, feat)
    x16 = torch.zeros((3, 21, 37, 16), device='cuda')
    x16[..., :cin] = torch.cat(dev, dim=1).permute(0, 2, 3, 1)
    k16 = np.zeros((3, 3, 16, feat), np.float32)
    k16[:, :, :cin] = k0
    a = conv3x3_nhwc(x16, k16, b0, epilogue=0)
    y_ref = conv3x3_nhwc(a, k1, b1, epilogue=2, aux=dev[-1])
    assert torch.equal(y, y_ref)


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16x3'])
def test_one_model_on_two_streams_and_two_threads(precision):
    """calls on a handle are serialised by the stream they are given, so one model used from two streams
    (forwards enqueued alternately, nothing synchronised in between: their kernels interleave layer by layer) and from
    two host threads must give each caller the result of its own inputs — every stream has its own activation
    workspace (S2Model._get_workspace); with a shared one the second stream overwrites the first one's layers."""
    import threading
    from dsen2_amd.DSen2Net import s2model
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=5, bias_scale=0.02)
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision=precision)
    m.set_weights_flat(flat)
    dev = m.device
    ins = [[torch.from_numpy(a).to(dev) for a in do.synthetic_inputs(64, 32, 32, (4, 6), seed=s)] for s in (11, 12)]
    want = [m.forward_device(x).clone() for x in ins]          # one at a time, default stream
    torch.cuda.synchronize()
    assert not torch.equal(want[0], want[1])
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    got = [None, None]
    for rep in range(3):
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                got[k] = m.forward_device(ins[k])
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert len(m._workspaces) == 3                               # default stream + the two side streams

    def worker(k):
        with torch.cuda.device(dev), torch.cuda.stream(streams[k]):
            for _ in range(3):
                got[k] = m.forward_device(ins[k])
            streams[k].synchronize()
    got = [None, None]
    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    m.release_workspaces()
    assert torch.equal(m.forward_device(ins[0]), want[0])


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16x3'])
def test_a_nan_pixel_stays_inside_its_receptive_field(precision):
    """A NaN input pixel (Sentinel-2 rasters are integers, so this is outside the parity claim — and the reference itself is
    not well defined here: Eigen's vectorised `cwiseMax` of TF 1.x returns 0 for relu(NaN), its scalar tail NaN) changes
    nothing outside the receptive field of that pixel (14 pixels: 1 + 2 * 6 + 1 convolutions of 3 x 3) and nothing of any other
    patch, bit for bit.  Here relu is `fmaxf(v, 0)`: the first ReLU turns the poisoned activations into zeros, so a NaN in a
    10 m band leaves every output finite; a NaN in a 20 m band also reaches the output through the skip connection
    (DSen2Net.py:41) — at exactly that pixel and band.  Screens the zero padding (materialised by out-of-range loads, not by
    multiplying by zero), the tile halos and the plane formats of the bf16 modes for stray contamination."""
    from dsen2_amd.DSen2Net import s2model
    flat = do.he_uniform_weights(10, 6, 6, 128, seed=3, bias_scale=0.05)
    m = s2model(((4, None, None), (6, None, None)), num_layers=6, feature_size=128, precision=precision)
    m.set_weights_flat(flat)
    xs = [torch.from_numpy(a).cuda() for a in do.synthetic_inputs(5, 48, 40, (4, 6), seed=2)]
    clean = m.forward_device(xs).clone()
    assert bool(torch.isfinite(clean).all())
    py, px = 30, 7                                                    # near the left edge and a 16 / 32-pixel tile boundary
    yy, xx = torch.meshgrid(torch.arange(48, device='cuda'), torch.arange(40, device='cuda'), indexing='ij')
    inside = ((yy - py).abs() <= 14) & ((xx - px).abs() <= 14)
    xs[0][2, 1, py, px] = float('nan')                                # a 10 m band
    dirty = m.forward_device(xs).clone()
    assert torch.equal(dirty[[0, 1, 3, 4]], clean[[0, 1, 3, 4]])      # other patches: bit-identical
    assert torch.equal(dirty[2][:, ~inside], clean[2][:, ~inside])    # outside the receptive field: bit-identical
    assert bool(torch.isfinite(dirty).all())                          # swallowed by the first ReLU
    assert not torch.equal(dirty[2][:, inside], clean[2][:, inside])  # ... but the neighbourhood did change
    xs[0][2, 1, py, px] = 1.0
    xs[1][2, 4, py, px] = float('nan')                                # a 20 m band: also the skip input of output band 4
    dirty = m.forward_device(xs)
    bad = ~torch.isfinite(dirty)
    assert int(bad.sum()) == 1 and bool(bad[2, 4, py, px])
    assert torch.equal(dirty[[0, 1, 3, 4]], clean[[0, 1, 3, 4]]) and torch.equal(dirty[2][:, ~inside], clean[2][:, ~inside])
