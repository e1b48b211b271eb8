"""TEST INFRASTRUCTURE ONLY — float64 numpy restatement of the DSen2 / VDSen2 forward pass.

Follows (reference paths relative to /root/reference):
  * utils/DSen2Net.py:9-15   resBlock:  x + 0.1 * conv3x3(relu(conv3x3(x)))
  * utils/DSen2Net.py:18-43  s2model:   concat(axis=1) -> conv+relu -> d x resBlock -> conv -> + low-res input
  * testing/supres.py:53-66  _predict:  (d,F) = (6,128) or (32,256); Cout = input_shape[-1][0]

Keras semantics that are not spelled out in the reference but that its calls rely on
(keras Conv2D with data_format='channels_first', padding='same', kernel 3x3, stride 1):
cross-correlation (no kernel flip), one pixel of zero padding, kernel stored HWIO
``(3, 3, Cin, Cout)``, bias added before the activation, ``he_uniform`` = U(+-sqrt(6 / fan_in))
with fan_in = 9 * Cin, bias initialised to zero.

PARITY STATUS: **parity unpinned for the CNN's arithmetic** — keras/tensorflow are not installed in the build
container and the trained checkpoints are stripped from the reference checkout, so no output of
the reference network itself could be captured.  What IS pinned is the WIRING: tests/golden/graph_trace.json
records what the reference's own utils/DSen2Net.py builds when it is executed (under a recording stand-in for
the keras names it imports, tests/golden/make_golden_graph.py), and tests/test_oracle_graph_trace.py requires
forward() below to equal that graph, evaluated with conv3x3 / ReLU / x 0.1 / add / concatenate, bit for bit —
input order, fused ReLU of the first convolution, residual structure, block count, which input is added back,
weight order.  What a keras Conv2D computes (next paragraph) remains a reading.  The restatement is also
cross-checked against an independent C implementation (oracle/dsen2_oracle.c) and against torch's conv2d in tests/.

Weight container ("keras flat" order) used by every implementation in this repo:
  [conv_in.kernel(3,3,Cin,F), conv_in.bias(F),
   res0.convA.kernel(3,3,F,F), res0.convA.bias(F), res0.convB.kernel, res0.convB.bias, ... (d blocks),
   conv_out.kernel(3,3,F,Cout), conv_out.bias(Cout)]   all float32, C-order, concatenated.
"""
import numpy as np

RES_SCALE = 0.1  # utils/DSen2Net.py:9 (scale=0.1)


def layer_shapes(cin, cout, num_layers, feature_size):
    """[(Cin, Cout)] for the 2*d+2 convolutions, in graph order (DSen2Net.py:29-35)."""
    shapes = [(cin, feature_size)]
    for _ in range(num_layers):
        shapes += [(feature_size, feature_size), (feature_size, feature_size)]
    shapes.append((feature_size, cout))
    return shapes


def num_params(cin, cout, num_layers, feature_size):
    return sum(9 * a * b + b for a, b in layer_shapes(cin, cout, num_layers, feature_size))


def he_uniform_weights(cin, cout, num_layers, feature_size, seed=1, bias_scale=0.0):
    """Seeded synthetic weights in keras-flat order.

    he_uniform limit = sqrt(6 / (9*Cin)) (keras default named at DSen2Net.py:10,12,29,35).
    ``bias_scale`` > 0 draws non-zero biases (U(+-bias_scale)) so tests exercise the bias path;
    keras' own initial bias is zero.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    for a, b in layer_shapes(cin, cout, num_layers, feature_size):
        limit = np.sqrt(6.0 / (9 * a))
        parts.append(rng.uniform(-limit, limit, size=(3, 3, a, b)).astype(np.float32).ravel())
        if bias_scale > 0:
            parts.append(rng.uniform(-bias_scale, bias_scale, size=(b,)).astype(np.float32))
        else:
            parts.append(np.zeros((b,), np.float32))
    return np.concatenate(parts)


def split_weights(flat, cin, cout, num_layers, feature_size):
    """keras-flat -> [(kernel HWIO, bias)]"""
    out, off = [], 0
    for a, b in layer_shapes(cin, cout, num_layers, feature_size):
        k = flat[off:off + 9 * a * b].reshape(3, 3, a, b); off += 9 * a * b
        bias = flat[off:off + b]; off += b
        out.append((k, bias))
    assert off == flat.size, (off, flat.size)
    return out


def conv3x3(x, kernel, bias, dtype=np.float64):
    """'same' 3x3 cross-correlation, NCHW, zero pad 1 (keras Conv2D as used at DSen2Net.py:10,12,29,35)."""
    x = np.asarray(x, dtype)
    n, c, h, w = x.shape
    kernel = np.asarray(kernel, dtype)
    xp = np.zeros((n, c, h + 2, w + 2), dtype)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((n, kernel.shape[3], h, w), dtype)
    for dy in range(3):
        for dx in range(3):
            # out[n,o,y,x] += sum_c xp[n,c,y+dy,x+dx] * K[dy,dx,c,o]
            out += np.einsum('nchw,co->nohw', xp[:, :, dy:dy + h, dx:dx + w], kernel[dy, dx], optimize=True)
    out += np.asarray(bias, dtype)[None, :, None, None]
    return out


def res_block(x, ka, ba, kb, bb, dtype=np.float64):
    """utils/DSen2Net.py:9-15"""
    t = np.maximum(conv3x3(x, ka, ba, dtype), 0)
    t = conv3x3(t, kb, bb, dtype) * dtype(RES_SCALE)
    return x + t


def forward(inputs, flat_weights, num_layers, feature_size, dtype=np.float64, return_features=False):
    """s2model forward (utils/DSen2Net.py:18-43).  ``inputs`` = [x10, x20] or [x10, x20, x60], NCHW.

    Output has the channel count of the LAST input and that input is added back (DSen2Net.py:35-41).
    """
    xs = [np.asarray(a, dtype) for a in inputs]
    x = np.concatenate(xs, axis=1)
    cin, cout = x.shape[1], xs[-1].shape[1]
    layers = split_weights(np.asarray(flat_weights), cin, cout, num_layers, feature_size)
    feats = []
    x = np.maximum(conv3x3(x, *layers[0], dtype=dtype), 0)
    feats.append(x)
    for i in range(num_layers):
        (ka, ba), (kb, bb) = layers[1 + 2 * i], layers[2 + 2 * i]
        x = res_block(x, ka, ba, kb, bb, dtype)
        feats.append(x)
    x = conv3x3(x, *layers[-1], dtype=dtype)
    y = x + xs[-1]
    return (y, feats) if return_features else y


def rmse(a, b):
    """testing/demoDSen2.py:31-35 — float64 RMSE over every element."""
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.sqrt(np.mean(d * d)))


def synthetic_inputs(n, h, w, bands=(4, 6), seed=0):
    """U[0,1)*5 float32, PCG64(seed) — post-/2000 reflectance range."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return [(rng.random((n, c, h, w), dtype=np.float32) * np.float32(5.0)) for c in bands]
