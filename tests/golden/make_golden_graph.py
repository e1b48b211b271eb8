#!/usr/bin/env python3
"""Writes tests/golden/graph_trace.json: what the REFERENCE's own utils/DSen2Net.py builds when it is executed.

keras / tensorflow are installed nowhere in the build image, so no output of the reference network exists (the CNN's
ARITHMETIC stays unpinned: oracle/dsen2_oracle.py).  What CAN be pinned by running the reference's code is the WIRING: this
script imports /root/reference/utils/DSen2Net.py unmodified with a recording stand-in for the `keras` names it imports
(Model, Input, Conv2D, Concatenate, Activation, Lambda, Add, backend.set_image_data_format) — every layer construction and
every layer call becomes a node with the arguments the reference passed; a Lambda's function is EXECUTED on probe numbers,
so `lambda x: x * scale` is recorded by what it does — and calls s2model() with the four configurations
testing/supres.py:55-60 asks for.  tests/test_oracle_graph_trace.py then evaluates the recorded graphs with the oracle's own
primitives (convolution, ReLU, scaling, addition, concatenation) and requires the oracle's forward() — which was written from
READING the same file — to give the same numbers bit for bit, the weights being consumed in the order the reference created
its Conv2D layers (= keras' load_weights order = the "keras flat" order of every weight container in this repo).

    python tests/golden/make_golden_graph.py        (in the build container; any python)
"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
NODES = []
CALLS = []


def node(op, inputs=(), **attrs):
    NODES.append({'id': len(NODES), 'op': op, 'inputs': [int(i) for i in inputs], 'attrs': attrs})
    return Sym(len(NODES) - 1)


class Sym(object):
    """A symbolic tensor: only its node id; arithmetic on it is NOT defined (the reference does none outside Lambda)."""
    def __init__(self, nid):
        self.nid = nid


def _plain(v):
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if v is None or isinstance(v, (int, float, str, bool)):
        return v
    return repr(v)


def Input(shape=None, **kw):
    return node('Input', shape=_plain(shape), extra=_plain(sorted(kw)))


class _Layer(object):
    op = None

    def __init__(self, *args, **kw):
        self.args, self.kw = args, kw

    def __call__(self, x):
        ins = [t.nid for t in x] if isinstance(x, (list, tuple)) else [x.nid]
        return node(self.op, ins, args=_plain(self.args), kwargs={k: _plain(v) for k, v in sorted(self.kw.items())})


class Conv2D(_Layer):
    op = 'Conv2D'


class Concatenate(_Layer):
    op = 'Concatenate'


class Activation(_Layer):
    op = 'Activation'


class Add(_Layer):
    op = 'Add'


class Lambda(_Layer):
    op = 'Lambda'
    PROBES = (1.0, -2.5, 8.0, 0.0)

    def __call__(self, x):
        fn = self.args[0]
        return node('Lambda', [x.nid], probes=list(self.PROBES), values=[float(fn(p)) for p in self.PROBES],
                    extra=_plain(self.args[1:]), kwargs={k: _plain(v) for k, v in sorted(self.kw.items())})


class Model(object):
    def __init__(self, inputs=None, outputs=None, **kw):
        self.inputs = [t.nid for t in inputs]
        self.outputs = outputs.nid
        self.extra = _plain(sorted(kw))


def install():
    keras = types.ModuleType('keras')
    models = types.ModuleType('keras.models')
    layers = types.ModuleType('keras.layers')
    backend = types.ModuleType('keras.backend')
    models.Model, models.Input = Model, Input
    for cls in (Conv2D, Concatenate, Activation, Lambda, Add):
        setattr(layers, cls.__name__, cls)
    backend.set_image_data_format = lambda fmt: CALLS.append(['keras.backend.set_image_data_format', fmt])
    keras.models, keras.layers, keras.backend = models, layers, backend
    sys.modules.update({'keras': keras, 'keras.models': models, 'keras.layers': layers, 'keras.backend': backend})


def main():
    install()
    spec = importlib.util.spec_from_file_location('reference_DSen2Net', '/root/reference/utils/DSen2Net.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)                     # the reference's file, unmodified
    out = {'module_level_calls': list(CALLS), 'models': {}}
    # testing/supres.py:55-60: (deep, run_60) -> num_layers / feature_size; :19-21 / :38-40: the input shapes
    configs = {'DSen2_20': (((4, None, None), (6, None, None)), 6, 128),
               'DSen2_60': (((4, None, None), (6, None, None), (2, None, None)), 6, 128),
               'VDSen2_20': (((4, None, None), (6, None, None)), 32, 256),
               'VDSen2_60': (((4, None, None), (6, None, None), (2, None, None)), 32, 256)}
    for name, (shape, d, f) in configs.items():
        del NODES[:]
        m = ref.s2model(shape, num_layers=d, feature_size=f)
        out['models'][name] = {'input_shape': _plain(shape), 'num_layers': d, 'feature_size': f, 'inputs': m.inputs,
                               'output': m.outputs, 'model_extra_kwargs': m.extra, 'nodes': [dict(n) for n in NODES]}
    # the defaults of s2model itself (DSen2Net.py:18)
    del NODES[:]
    m = ref.s2model(((4, None, None), (6, None, None)))
    out['defaults'] = {'convs': sum(1 for n in NODES if n['op'] == 'Conv2D'),
                       'first_conv_filters': next(n for n in NODES if n['op'] == 'Conv2D')['attrs']['args'][0]}
    json.dump(out, open(os.path.join(HERE, 'graph_trace.json'), 'w'), indent=0, sort_keys=True)
    for name, rec in out['models'].items():
        ops = [n['op'] for n in rec['nodes']]
        print(name, {k: ops.count(k) for k in sorted(set(ops))})
    print('module-level calls:', out['module_level_calls'], ' defaults:', out['defaults'])


if __name__ == '__main__':
    main()
