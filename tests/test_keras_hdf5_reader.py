"""weights._from_keras_hdf5 against a synthetic file laid out like a keras 2.x full-model checkpoint
(training/supres_train.py:195-201: ModelCheckpoint(save_weights_only=False) -> /model_weights/<layer>/<layer>/...).
Needs h5py, which the system python of the build image lacks (skipped there; run under /opt/conda/bin/python3.9)."""
import os
import sys

import numpy as np
import pytest

h5py = pytest.importorskip('h5py')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_keras_like(path, cin, cout, d, f, flat):
    from dsen2_amd import weights as W
    shapes = W.layer_shapes(cin, cout, d, f)
    names, off = [], 0
    with h5py.File(path, 'w') as fh:
        root = fh.create_group('model_weights')
        def add(name, weights=None):
            g = root.create_group(name)
            wn = []
            if weights is not None:
                k, b = weights
                sub = g.create_group(name)
                sub.create_dataset('kernel:0', data=k)
                sub.create_dataset('bias:0', data=b)
                wn = [('%s/kernel:0' % name).encode(), ('%s/bias:0' % name).encode()]
            g.attrs['weight_names'] = wn
            names.append(name.encode())
        add('input_1'); add('input_2'); add('concatenate_1')
        ci = 0
        for li, (a, o) in enumerate(shapes):
            k = flat[off:off + 9 * a * o].reshape(3, 3, a, o); off += 9 * a * o
            b = flat[off:off + o]; off += o
            ci += 1
            add('conv2d_%d' % ci, (k, b))
            if 0 < li < len(shapes) - 1:
                add('activation_%d' % ci if li % 2 == 1 else 'lambda_%d' % ci)
                if li % 2 == 0:
                    add('add_%d' % ci)
        add('add_final')
        root.attrs['layer_names'] = names


def test_reader_round_trip(tmp_path):
    from dsen2_amd import weights as W
    flat = W.random_he_uniform(10, 6, 6, 128, seed=3, bias_scale=0.1)
    p = str(tmp_path / 's2_032_lr_1e-04.hdf5')
    write_keras_like(p, 10, 6, 6, 128, flat)
    got = W.load_flat(p, 10, 6, 6, 128)
    assert np.array_equal(got, flat)
    with pytest.raises(ValueError):
        W.load_flat(p, 12, 2, 6, 128)
