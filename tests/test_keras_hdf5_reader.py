"""weights._from_keras_hdf5 and cli._load (.mat) against synthetic files laid out like a keras 2.x full-model
checkpoint (training/supres_train.py:195-201: ModelCheckpoint(save_weights_only=False) ->
/model_weights/<layer>/<layer>/...) and like the reference's MATLAB v7.3 tiles (testing/demoDSen2.py:14-28).
Needs h5py, which the system python of the build image lacks: there tests/test_h5py_suite.py runs this file under
/opt/conda/bin/python3.9 (which has h5py but no torch — nothing here may import torch)."""
import os
import sys

import numpy as np
import pytest

h5py = pytest.importorskip('h5py')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_keras_like(path, cin, cout, d, f, flat, first_index=1, shuffle_names=False, bias_first=False, scope='',
                     conv_name=None, swap_body=False):
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
                sub = g.create_group(name + scope)      # scope='_1': a name scope TensorFlow made unique
                sub.create_dataset('kernel:0', data=k)
                sub.create_dataset('bias:0', data=b)
                wn = [('%s%s/kernel:0' % (name, scope)).encode(), ('%s%s/bias:0' % (name, scope)).encode()]
                if bias_first:
                    wn.reverse()
            g.attrs['weight_names'] = wn
            names.append(name.encode())
        add('input_1'); add('input_2'); add('concatenate_1')
        ci = first_index - 1
        for li, (a, o) in enumerate(shapes):
            k = flat[off:off + 9 * a * o].reshape(3, 3, a, o); off += 9 * a * o
            b = flat[off:off + o]; off += o
            ci += 1
            add(conv_name(li) if conv_name else 'conv2d_%d' % ci, (k, b))
            if 0 < li < len(shapes) - 1:
                add('activation_%d' % ci if li % 2 == 1 else 'lambda_%d' % ci)
                if li % 2 == 0:
                    add('add_%d' % ci)
        add('add_final')
        if swap_body:                                # two body convolutions trade places in layer_names: both orders chain
            i, j = names.index(b'conv2d_%d' % (first_index + 1)), names.index(b'conv2d_%d' % (first_index + 2))
            names[i], names[j] = names[j], names[i]
        if shuffle_names:
            names = [names[i] for i in np.random.default_rng(0).permutation(len(names))]
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


def test_reader_does_not_trust_attribute_or_weight_order(tmp_path):
    """conv layers are ordered by the numeric suffix of their names (conv2d_37.. when other models were built in the
    same keras session), kernel / bias are picked by name: a shuffled layer_names attribute, an offset numbering and
    bias-before-kernel weight_names all read back the same flat vector."""
    from dsen2_amd import weights as W
    flat = W.random_he_uniform(12, 2, 6, 128, seed=4, bias_scale=0.1)
    for kw in (dict(shuffle_names=True), dict(first_index=37), dict(bias_first=True),
               dict(first_index=9, shuffle_names=True, bias_first=True)):
        p = str(tmp_path / ('s2_030_%s.hdf5' % '_'.join(sorted(kw))))
        write_keras_like(p, 12, 2, 6, 128, flat, **kw)
        assert np.array_equal(W.load_flat(p, 12, 2, 6, 128), flat), kw


def test_reader_accepts_tf_keras_names_unique_scopes_and_unnumbered_layers(tmp_path):
    """(a) tf.keras numbers 'conv2d', 'conv2d_1', ... (the first layer has no suffix); (b) a weight path whose
    scope TensorFlow made unique ('conv2d_1_1/kernel:0' under layer 'conv2d_1') is legitimate; (c) layer names without
    a usable numbering fall back to the file's layer_names order — validated by the shape chain, not by names."""
    from dsen2_amd import weights as W
    flat = W.random_he_uniform(10, 6, 2, 128, seed=6, bias_scale=0.1)
    cases = dict(tf=dict(conv_name=lambda li: 'conv2d' if li == 0 else 'conv2d_%d' % li),
                 scope=dict(scope='_1'),
                 unnumbered=dict(conv_name=lambda li: 'layer_' + 'abcdefgh'[li]),
                 repeated=dict(conv_name=lambda li: 'block%d_conv_1' % li))
    import warnings
    for tag, kw in cases.items():
        p = str(tmp_path / ('%s.hdf5' % tag))
        write_keras_like(p, 10, 6, 2, 128, flat, **kw)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            assert np.array_equal(W.load_flat(p, 10, 6, 2, 128), flat), tag
        # the fallback to the file's layer_names order (which the shape chain cannot validate among the F -> F body layers)
        # is never silent; a usable numbering needs no warning
        fell_back = [w for w in caught if 'layer_names order' in str(w.message)]
        assert bool(fell_back) == (tag in ('unnumbered', 'repeated')), (tag, [str(w.message) for w in caught])
    # unnumbered AND shuffled: no order can be trusted, and the shape chain says so
    p = str(tmp_path / 'hopeless.hdf5')
    write_keras_like(p, 10, 6, 2, 128, flat, conv_name=lambda li: 'layer_' + 'abcdefgh'[li], shuffle_names=True)
    with pytest.raises(ValueError):
        W.load_flat(p, 10, 6, 2, 128)


def test_reader_rejects_a_layer_that_is_not_a_conv(tmp_path):
    from dsen2_amd import weights as W
    flat = W.random_he_uniform(10, 6, 1, 128, seed=5)
    p = str(tmp_path / 'odd.hdf5')
    write_keras_like(p, 10, 6, 1, 128, flat)
    with h5py.File(p, 'a') as fh:
        g = fh['model_weights'].create_group('batch_normalization_1')
        g.attrs['weight_names'] = [b'batch_normalization_1/gamma:0', b'batch_normalization_1/beta:0']
        names = [n if isinstance(n, bytes) else str(n).encode() for n in fh['model_weights'].attrs['layer_names']] + [b'batch_normalization_1']
        fh['model_weights'].attrs['layer_names'] = names
    with pytest.raises(ValueError):
        W.load_flat(p, 10, 6, 1, 128)


def test_cli_reads_matlab_v73_tiles(tmp_path):
    """cli._load on a .mat laid out like data/S2A_MSIL1C_20170527_T33UUB.mat: MATLAB stores [x, y, c] column-major,
    so h5py sees datasets [c, y, x]; readh5 (testing/demoDSen2.py:14-28) transposes them back to HWC."""
    from dsen2_amd import cli
    rng = np.random.default_rng(1)
    im10 = rng.integers(0, 9000, size=(24, 18, 4)).astype(np.uint16)
    im20 = rng.integers(0, 9000, size=(12, 9, 6)).astype(np.uint16)
    im60 = rng.integers(0, 9000, size=(4, 3, 2)).astype(np.uint16)
    p = str(tmp_path / 'tile.mat')
    with h5py.File(p, 'w') as fh:
        for k, a in (('im10', im10), ('im20', im20), ('im60', im60)):
            fh.create_dataset(k, data=np.ascontiguousarray(a.transpose()))
    d10, d20, d60 = cli._load(p)
    assert np.array_equal(d10, im10) and np.array_equal(d20, im20) and np.array_equal(d60, im60)
    with h5py.File(p, 'a') as fh:
        del fh['im60']
    assert cli._load(p)[2] is None


REFERENCE_DATA = '/root/reference/data'


@pytest.mark.skipif(not os.path.isdir(REFERENCE_DATA), reason='the reference checkout is only present in the build container')
@pytest.mark.parametrize('mat, golden', [('S2A_MSIL1C_20170527_T33UUB.mat', 'tile_T33UUB_600.npz'),
                                         ('S2B_MSIL1C_20171022_T49JGM.mat', 'tile_T49JGM_600.npz')])
def test_cli_reads_the_tiles_the_reference_ships(mat, golden):
    """cli._load on the REAL data/*.mat files (what testing/demoDSen2.py:14-28,42,67 reads) gives the arrays committed
    in tests/golden/ — which is what every GPU test on the bundled tiles runs on."""
    from dsen2_amd import cli
    d10, d20, d60 = cli._load(os.path.join(REFERENCE_DATA, mat))
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', golden))
    assert d10.shape == (600, 600, 4) and d20.shape == (300, 300, 6) and d60.shape == (100, 100, 2)
    assert str(g['source']) == mat
    for got, key in ((d10, 'd10'), (d20, 'd20'), (d60, 'd60')):
        assert np.array_equal(got, g[key].astype(got.dtype)), key


def _same_tree(mine, theirs, where=''):
    assert mine.keys() == list(theirs.keys()), where
    assert set(mine.attrs) == set(theirs.attrs.keys()), where
    for k in mine.attrs:
        a, b = np.asarray(mine.attrs[k]), np.asarray(theirs.attrs[k])
        assert a.shape == b.shape and [str(x) for x in a.ravel()] == [str(x) for x in b.ravel()], (where, k)
    for k in mine.keys():
        x, y = mine[k], theirs[k]
        if isinstance(y, h5py.Group):
            _same_tree(x, y, where + '/' + k)
        else:
            a, b = np.asarray(x), np.asarray(y)
            assert a.dtype == b.dtype.newbyteorder('=') and a.shape == b.shape and a.tobytes() == b.astype(a.dtype).tobytes(), (where, k)


def test_own_reader_and_h5py_agree_and_the_own_reader_is_the_one_used(tmp_path, monkeypatch):
    """The product reads HDF5 through dsen2_amd/hdf5_min.py (h5py is only the fall-back for format features that reader
    names as unsupported): on a full-size checkpoint written by h5py the two read the same tree, and load_flat does not
    touch h5py."""
    from dsen2_amd import hdf5_min, weights as W
    flat = W.random_he_uniform(10, 6, 6, 128, seed=11, bias_scale=0.1)
    p = str(tmp_path / 's2_032_lr_1e-04.hdf5')
    write_keras_like(p, 10, 6, 6, 128, flat, first_index=7)
    with hdf5_min.File(p) as mine, h5py.File(p, 'r') as theirs:
        _same_tree(mine, theirs)
    monkeypatch.setattr(h5py, 'File', lambda *a, **k: (_ for _ in ()).throw(AssertionError('h5py used')))
    assert np.array_equal(W.load_flat(p, 10, 6, 6, 128), flat)


def test_when_numbering_and_layer_names_order_both_fit_but_differ_keras_order_wins_loudly(tmp_path):
    """keras' load_weights (testing/supres.py:63) pairs weighted layers in the FILE's layer_names order; a file whose body
    layers are listed in another order than their numbering fits the architecture either way — the loader does what keras
    would and names both orders in a warning instead of choosing silently."""
    import warnings
    from dsen2_amd import weights as W
    flat = W.random_he_uniform(10, 6, 2, 128, seed=9, bias_scale=0.1)
    p = str(tmp_path / 'swapped.hdf5')
    write_keras_like(p, 10, 6, 2, 128, flat, swap_body=True)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        got = W.load_flat(p, 10, 6, 2, 128)
    assert any('differs from the layers\' numbering' in str(w.message) for w in caught), [str(w.message) for w in caught]
    n0, nb = 9 * 10 * 128 + 128, 9 * 128 * 128 + 128
    a, b = flat[n0:n0 + nb], flat[n0 + nb:n0 + 2 * nb]
    assert np.array_equal(got[:n0], flat[:n0]) and np.array_equal(got[n0:n0 + nb], b) and np.array_equal(got[n0 + nb:n0 + 2 * nb], a)
    assert np.array_equal(got[n0 + 2 * nb:], flat[n0 + 2 * nb:])


@pytest.mark.skipif(not os.path.isdir(REFERENCE_DATA), reason='the reference checkout is only present in the build container')
def test_the_reference_demo_readh5_runs_on_the_own_reader(monkeypatch):
    """testing/demoDSen2.py:14-28 `readh5` — the reference's own function, imported from its unmodified file (its imports of
    `supres` and matplotlib stubbed: neither is used by readh5) — gives the same arrays with this package's hdf5_min standing in
    for h5py as with h5py itself, and they are what cli._load returns: a user without h5py can run the demo's reader as it is
    (`sys.modules['h5py'] = dsen2_amd.hdf5_min`)."""
    import importlib.util
    import types
    from dsen2_amd import cli, hdf5_min

    def load_demo(h5):
        stubs = {'supres': types.ModuleType('supres'), 'matplotlib': types.ModuleType('matplotlib'),
                 'matplotlib.pyplot': types.ModuleType('matplotlib.pyplot'), 'h5py': h5}
        stubs['supres'].DSen2_20 = stubs['supres'].DSen2_60 = None
        stubs['matplotlib'].pyplot = stubs['matplotlib.pyplot']
        saved = {k: sys.modules.get(k) for k in list(stubs) + ['utils', 'utils.imresize']}
        sys.modules.update(stubs)
        sys.path.insert(0, '/root/reference/testing')
        sys.path.insert(0, '/root/reference')
        try:
            spec = importlib.util.spec_from_file_location('reference_demo_%s' % h5.__name__.replace('.', '_'),
                                                          '/root/reference/testing/demoDSen2.py')
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)         # `if __name__ == '__main__'` keeps the demo itself from running
        finally:
            sys.path.remove('/root/reference/testing')
            sys.path.remove('/root/reference')
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
        mod.DATA_PATH = REFERENCE_DATA + '/'
        return mod
    with_h5py, with_own = load_demo(h5py), load_demo(hdf5_min)
    for mat in ('S2A_MSIL1C_20170527_T33UUB.mat', 'S2B_MSIL1C_20171022_T49JGM.mat'):
        a = with_h5py.readh5(mat, im60=True)
        b = with_own.readh5(mat, im60=True)
        c = cli._load(os.path.join(REFERENCE_DATA, mat))
        assert len(a) == len(b) == 3
        for x, y, z in zip(a, b, c):
            assert x.dtype == y.dtype == z.dtype and x.shape == y.shape and np.array_equal(x, y) and np.array_equal(x, z)
        assert len(with_own.readh5(mat)) == 2
