#!/opt/conda/bin/python3.9
"""Writes tests/golden/hdf5/*: small HDF5 files made by the real library (h5py 3.3.0 / HDF5 1.12 under
/opt/conda/bin/python3.9 in the build image) and, beside them, what h5py itself reads back from each
(expected.npz / expected.json).  tests/test_hdf5_min.py holds dsen2_amd/hdf5_min.py — the dependency-free reader the
product uses for keras checkpoints and MATLAB v7.3 tiles — to those read-backs in the interpreter that has no h5py.

The files are laid out like what this path meets:
  keras_full_model.h5     ModelCheckpoint(save_weights_only=False) (training/supres_train.py:195-201): /model_weights/<layer>/
                          <layer>/{kernel:0,bias:0}, layer_names / weight_names attributes, a long model_config string, an
                          optimizer_weights group; the Conv2D numbering starts at 7 (other models built first in the session)
  keras_weights_only.h5   model.save_weights(): the same without the model_weights level; vlen-string version attributes
  keras_split_attrs.h5    layer_names saved in pieces layer_names0, layer_names1 (keras does that past 64 KB)
  matlab_v73.mat          512-byte user block with MATLAB's text header, im10 / im20 / im60 as CHW float32, chunked + gzip
                          (the layout of data/*.mat: chunks (C, H, few columns), gzip 3, MATLAB_class attribute)
  features.h5             the rest of what hdf5_min implements: big-endian and 64-bit types, shuffle + fletcher32, a chunked
                          dataset with unallocated chunks and a fill value, chunk and group B-trees with two levels, compact
                          storage, fixed and variable-length strings, a scalar
  latest.h5               libver='latest': v2 object headers, link messages, layout v4 (contiguous, single-chunk)
  unsupported_*.h5        valid files using what hdf5_min does NOT implement (it must say so by name)
"""
import json
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hdf5')
CIN, COUT, D, F = 10, 6, 2, 8                       # a DSen2_20-shaped network small enough to commit


def shapes():
    s = [(CIN, F)]
    for _ in range(D):
        s += [(F, F)] * 2
    return s + [(F, COUT)]


def keras_layers(root, rng, first):
    names, flat = [], []

    def add(name, weights=None):
        g = root.create_group(name)
        wn = []
        if weights is not None:
            sub = g.create_group(name)
            sub.create_dataset('kernel:0', data=weights[0])
            sub.create_dataset('bias:0', data=weights[1])
            wn = [('%s/kernel:0' % name).encode(), ('%s/bias:0' % name).encode()]
        g.attrs['weight_names'] = np.array(wn, dtype='S%d' % max(1, max([len(w) for w in wn] or [1])))
        names.append(name.encode())
    add('input_1'); add('input_2'); add('concatenate_1')
    ci = first - 1
    for li, (a, o) in enumerate(shapes()):
        k = rng.standard_normal((3, 3, a, o)).astype(np.float32)
        b = rng.standard_normal(o).astype(np.float32)
        flat += [k.ravel(), b]
        ci += 1
        add('conv2d_%d' % ci, (k, b))
        if 0 < li < len(shapes()) - 1:
            add('activation_%d' % ci if li % 2 == 1 else 'lambda_%d' % ci)
            if li % 2 == 0:
                add('add_%d' % ci)
    add('add_%d' % (ci + 1))
    return names, np.concatenate(flat)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260404)
    flats = {}
    with h5py.File(os.path.join(OUT, 'keras_full_model.h5'), 'w') as f:
        root = f.create_group('model_weights')
        names, flats['keras_full_model.h5'] = keras_layers(root, rng, first=7)
        root.attrs['layer_names'] = np.array(names)
        root.attrs['backend'] = np.bytes_(b'tensorflow')
        root.attrs['keras_version'] = np.bytes_(b'2.2.4')
        f.attrs['keras_version'] = np.bytes_(b'2.2.4')
        f.attrs['backend'] = np.bytes_(b'tensorflow')
        f.attrs['model_config'] = np.bytes_(json.dumps({'class_name': 'Model', 'config': {'layers': [
            {'name': n.decode(), 'class_name': 'Layer', 'config': {'filler': 'x' * 150}} for n in names]}}).encode())
        opt = f.create_group('optimizer_weights')
        opt.attrs['weight_names'] = np.array([b'Nadam/iterations:0'])
        opt.create_group('Nadam')['iterations:0'] = np.int64(12345)
    with h5py.File(os.path.join(OUT, 'keras_weights_only.h5'), 'w') as f:
        names, flats['keras_weights_only.h5'] = keras_layers(f, rng, first=1)
        f.attrs['layer_names'] = np.array(names)
        f.attrs['backend'] = 'tensorflow'                               # str -> variable-length string (global heap)
        f.attrs['keras_version'] = '2.3.1'
    with h5py.File(os.path.join(OUT, 'keras_split_attrs.h5'), 'w') as f:
        names, flats['keras_split_attrs.h5'] = keras_layers(f, rng, first=1)
        f.attrs['layer_names0'] = np.array(names[:7])
        f.attrs['layer_names1'] = np.array(names[7:])
    with h5py.File(os.path.join(OUT, 'matlab_v73.mat'), 'w', userblock_size=512) as f:
        for key, shape, cols in (('im10', (4, 36, 36), 6), ('im20', (6, 18, 18), 9), ('im60', (2, 6, 6), 6)):
            ds = f.create_dataset(key, data=(rng.random(shape) * 9000).astype(np.float32), chunks=shape[:2] + (cols,),
                                  compression='gzip', compression_opts=3)
            ds.attrs['MATLAB_class'] = np.bytes_(b'single')
    with open(os.path.join(OUT, 'matlab_v73.mat'), 'r+b') as fh:
        head = b'MATLAB 7.3 MAT-file, Platform: PCWIN64, Created on: (synthetic fixture) HDF5 schema 1.00 .'
        fh.write(head.ljust(124, b' ') + b'\x00\x02IM')
    with h5py.File(os.path.join(OUT, 'features.h5'), 'w') as f:
        f.create_dataset('be_f8', data=rng.standard_normal(5).astype('>f8'))
        f.create_dataset('be_i2', data=np.arange(-3, 4).astype('>i2'))
        f.create_dataset('i64', data=np.arange(-5, 5, dtype='i8') * (1 << 40))
        f.create_dataset('u8', data=np.arange(250, 256, dtype='u1'))
        f.create_dataset('f2', data=np.linspace(-2, 2, 9).astype('f2'))
        f.create_dataset('scalar', data=np.float32(3.5))
        f.create_dataset('shuffled', data=rng.integers(0, 60000, (50, 70)).astype('u2'), chunks=(7, 9), compression='gzip',
                         shuffle=True, fletcher32=True)
        sp = f.create_dataset('sparse', shape=(10, 12), dtype='f4', chunks=(5, 5), fillvalue=2.5)
        sp[0:5, 0:5] = 1.0
        sp[5:10, 10:12] = -1.0
        f.create_dataset('never_written', shape=(3, 3), dtype='i4', fillvalue=-7)
        f.create_dataset('many_chunks', data=rng.standard_normal((150, 4)).astype('f4'), chunks=(1, 4))
        g = f.create_group('many_links')
        for i in range(200):
            g['d%03d' % i] = np.int16(i)
        f.create_dataset('fixed_strings', data=np.array([b'ab', b'cde', b'']))
        f.create_dataset('vlen_strings', data=np.array(['ab', 'cdé', ''], dtype=object), dtype=h5py.string_dtype())
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        data = np.arange(12, dtype='f4').reshape(3, 4)
        space = h5py.h5s.create_simple(data.shape)
        dsid = h5py.h5d.create(f.id, b'compact', h5py.h5t.NATIVE_FLOAT, space, dcpl)
        dsid.write(h5py.h5s.ALL, h5py.h5s.ALL, data)
        f['many_links'].attrs['ints'] = np.arange(4, dtype='i4')
        f['many_links'].attrs['float'] = np.float64(0.1)
        f['many_links'].attrs['text'] = 'variable length'
        n = f.create_group('nested')
        n.create_group('deeper')['leaf'] = np.arange(3, dtype='f4')
    with h5py.File(os.path.join(OUT, 'latest.h5'), 'w', libver='latest') as f:
        f.attrs['layer_names'] = np.array([b'x', b'y'])
        g = f.create_group('x')
        g['w'] = rng.standard_normal((3, 3)).astype('f4')
        g.attrs['weight_names'] = np.array([b'w'])
        f['c'] = np.arange(6, dtype='f4').reshape(2, 3)
        f.create_dataset('one_chunk', data=rng.standard_normal((6, 6)).astype('f4'), chunks=(6, 6), compression='gzip')
    with h5py.File(os.path.join(OUT, 'unsupported_compound.h5'), 'w') as f:
        f['ok'] = np.arange(3, dtype='f4')
        f['table'] = np.zeros(3, dtype=[('a', 'i4'), ('b', 'f4')])
    with h5py.File(os.path.join(OUT, 'unsupported_dense_attrs.h5'), 'w', libver='latest') as f:
        f['ok'] = np.arange(3, dtype='f4')
        for i in range(20):
            f.attrs['a%02d' % i] = np.arange(50, dtype='f8')
    with h5py.File(os.path.join(OUT, 'unsupported_fixed_array.h5'), 'w', libver='latest') as f:
        f.create_dataset('chunks', data=np.arange(64, dtype='f4').reshape(8, 8), chunks=(2, 2))

    # what h5py reads back
    arrays, meta = {}, {}

    def norm(v):
        a = np.asarray(v)
        if a.dtype == object:
            return {'str': [x.decode() if isinstance(x, bytes) else str(x) for x in a.ravel()], 'shape': list(a.shape)}
        if a.dtype.kind == 'S':
            return {'bytes': [x.decode('latin1') for x in a.ravel()], 'shape': list(a.shape)}
        if a.dtype.kind == 'U':
            return {'str': [str(x) for x in a.ravel()], 'shape': list(a.shape)}
        return {'num': a.ravel().tolist(), 'dtype': a.dtype.newbyteorder('=').str, 'shape': list(a.shape)}

    def walk(name, fname, obj, rec):
        rec[name] = {'attrs': {k: norm(v) for k, v in obj.attrs.items()}}
        if isinstance(obj, h5py.Group):
            rec[name]['keys'] = list(obj.keys())
            for k in obj.keys():
                walk((name.rstrip('/') + '/' + k), fname, obj[k], rec)
        else:
            rec[name]['shape'] = list(obj.shape)
            a = np.asarray(obj)
            if a.dtype.kind in 'OSU':
                rec[name]['value'] = norm(a)
            else:
                rec[name]['dtype'] = a.dtype.newbyteorder('=').str
                arrays['%s|%s' % (fname, name)] = a.astype(a.dtype.newbyteorder('='))
    for fname in sorted(os.listdir(OUT)):
        if fname.endswith(('.h5', '.mat')) and not fname.startswith('unsupported_'):
            with h5py.File(os.path.join(OUT, fname), 'r') as f:
                meta[fname] = {}
                walk('/', fname, f, meta[fname])
    for k, v in flats.items():
        arrays['%s|flat' % k] = v
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **arrays)
    json.dump({'arch': [CIN, COUT, D, F], 'files': meta, 'made_with': 'h5py %s, HDF5 %s' % (h5py.__version__, h5py.version.hdf5_version)},
              open(os.path.join(OUT, 'expected.json'), 'w'), indent=0, sort_keys=True)
    for fname in sorted(os.listdir(OUT)):
        print('%8d  %s' % (os.path.getsize(os.path.join(OUT, fname)), fname))


if __name__ == '__main__':
    main()
