"""The oracle's network WIRING against the reference's own code: tests/golden/graph_trace.json is what /root/reference/
utils/DSen2Net.py builds when it is EXECUTED (under a recording stand-in for the keras names it imports;
tests/golden/make_golden_graph.py).  The recorded graphs are evaluated here node by node with the oracle's primitives, the
weights being consumed in the order the reference created its Conv2D layers, and the oracle's forward() — written from reading
the same file — must give the same numbers BIT FOR BIT.  This pins: the input order of the concatenation and its axis, the
first convolution's fused ReLU, the residual block (conv -> ReLU -> conv -> x 0.1 -> add to the block's input), the number of
blocks and features per configuration, the output channel count, WHICH input is added back at the end, the 'keras flat'
weight order, and that no convolution was given a stride / dilation / use_bias / data_format argument.  It does not pin what
keras' Conv2D computes (cross-correlation, zero 'same' padding, HWIO): that part of the CNN oracle stays a reading."""
import json
import os

import numpy as np
import pytest

from oracle import dsen2_oracle as do

TRACE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'graph_trace.json')))


def evaluate(rec, inputs, flat, dtype=np.float64):
    """The recorded graph with the oracle's primitives; returns (output, [(cin, cout)] of the convolutions in creation order)."""
    convs = [n for n in rec['nodes'] if n['op'] == 'Conv2D']
    # weights in CREATION order: split by the channel chain the trace itself implies
    val, shapes, off = {}, [], 0
    flat = np.asarray(flat)
    for n in rec['nodes']:
        a, ins = n['attrs'], n['inputs']
        if n['op'] == 'Input':
            k = rec['inputs'].index(n['id'])
            assert a['shape'] == [inputs[k].shape[1], None, None] and a['extra'] == []
            val[n['id']] = np.asarray(inputs[k], dtype)
        elif n['op'] == 'Concatenate':
            assert a['args'] == [] and a['kwargs'] == {'axis': 1}
            val[n['id']] = np.concatenate([val[i] for i in ins], axis=1)
        elif n['op'] == 'Conv2D':
            filters, ksize = a['args']
            kw = dict(a['kwargs'])
            assert list(ksize) == [3, 3] and kw.pop('padding') == 'same' and kw.pop('kernel_initializer') == 'he_uniform'
            act = kw.pop('activation', None)
            assert kw == {} and act in (None, 'relu'), kw            # no strides, dilation_rate, use_bias, data_format ...
            x = val[ins[0]]
            ci = x.shape[1]
            kern = flat[off:off + 9 * ci * filters].reshape(3, 3, ci, filters); off += 9 * ci * filters
            bias = flat[off:off + filters]; off += filters
            shapes.append((ci, filters))
            y = do.conv3x3(x, kern, bias, dtype)
            val[n['id']] = np.maximum(y, 0) if act == 'relu' else y
        elif n['op'] == 'Activation':
            assert a['args'] == ['relu'] and a['kwargs'] == {}
            val[n['id']] = np.maximum(val[ins[0]], 0)
        elif n['op'] == 'Lambda':
            # the function, executed on probe numbers when the trace was made, is a multiplication by scale = 0.1
            assert a['values'] == [p * 0.1 for p in a['probes']] and a['extra'] == [] and a['kwargs'] == {}
            val[n['id']] = val[ins[0]] * dtype(0.1)
        elif n['op'] == 'Add':
            assert a['args'] == [] and a['kwargs'] == {} and len(ins) == 2
            val[n['id']] = val[ins[0]] + val[ins[1]]
        else:
            raise AssertionError('unexpected layer %r' % n['op'])
    assert off == flat.size and len(shapes) == len(convs)
    return val[rec['output']], shapes


@pytest.mark.parametrize('name,h,w', [('DSen2_20', 9, 7), ('DSen2_60', 8, 8), ('VDSen2_20', 4, 5), ('VDSen2_60', 4, 4)])
def test_oracle_forward_is_the_reference_graph_bit_for_bit(name, h, w):
    rec = TRACE['models'][name]
    bands = tuple(s[0] for s in rec['input_shape'])
    d, f = rec['num_layers'], rec['feature_size']
    rng = np.random.default_rng(len(name) + h)
    xs = [rng.random((2, c, h, w)).astype(np.float32) * np.float32(5) for c in bands]
    flat = do.he_uniform_weights(sum(bands), bands[-1], d, f, seed=h + w, bias_scale=0.1)
    got, shapes = evaluate(rec, xs, flat)
    want = do.forward(xs, flat, d, f)
    assert got.dtype == want.dtype == np.float64 and np.array_equal(got, want)
    # the weight containers' layer list (host side and oracle) = the reference's Conv2D creation order
    from dsen2_amd import weights as W
    assert shapes == do.layer_shapes(sum(bands), bands[-1], d, f) == W.layer_shapes(sum(bands), bands[-1], d, f)
    # ... and float32 evaluation of the trace = the oracle's float32 mode
    got32, _ = evaluate(rec, xs, flat, np.float32)
    assert np.array_equal(got32, do.forward(xs, flat, d, f, dtype=np.float32))


def test_what_the_reference_code_wires():
    assert TRACE['module_level_calls'] == [['keras.backend.set_image_data_format', 'channels_first']]       # DSen2Net.py:6
    assert TRACE['defaults'] == {'convs': 66, 'first_conv_filters': 256}                                      # s2model(num_layers=32, feature_size=256)
    for name, rec in TRACE['models'].items():
        nodes = rec['nodes']
        ins = [n['id'] for n in nodes if n['op'] == 'Input']
        assert rec['inputs'] == ins and rec['model_extra_kwargs'] == []               # Model(inputs=[input10, input20(, input60)])
        last = nodes[rec['output']]
        assert last['op'] == 'Add' and last['inputs'][1] == ins[-1]                   # + input20 / + input60 (DSen2Net.py:38,41)
        out_conv = nodes[last['inputs'][0]]
        assert out_conv['op'] == 'Conv2D' and out_conv['attrs']['args'][0] == rec['input_shape'][-1][0]
        assert 'activation' not in out_conv['attrs']['kwargs']
        ops = [n['op'] for n in nodes]
        d = rec['num_layers']
        assert ops.count('Conv2D') == 2 * d + 2 and ops.count('Lambda') == d and ops.count('Activation') == d
        assert ops.count('Add') == d + 1 and ops.count('Concatenate') == 1
        # every residual Add takes (block input, scaled branch): the skip is the block's INPUT, not the first convolution
        adds = [n for n in nodes if n['op'] == 'Add'][:-1]
        prev = next(n['id'] for n in nodes if n['op'] == 'Conv2D')
        for a in adds:
            assert a['inputs'][0] == prev and nodes[a['inputs'][1]]['op'] == 'Lambda'
            prev = a['id']
