"""dsen2_amd/hdf5_min.py — the dependency-free HDF5 reader behind `model.load_weights('<...>.hdf5')` (testing/supres.py:63) and the
MATLAB v7.3 tiles (testing/demoDSen2.py:14-28) — against files written by the real HDF5 library and what h5py itself read
back from them (tests/golden/hdf5/, made by tests/golden/make_golden_hdf5.py under the build image's conda python).  Runs
in the interpreter WITHOUT h5py: that is the point."""
import json
import os
import shutil

import numpy as np
import pytest

from dsen2_amd import hdf5_min

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, 'golden', 'hdf5')
META = json.load(open(os.path.join(DIR, 'expected.json')))
EXPECTED = np.load(os.path.join(DIR, 'expected.npz'))


def _same_value(got, want, where):
    if 'num' in want:
        a = np.asarray(got)
        assert a.dtype.str == want['dtype'] and list(a.shape) == want['shape'], (where, a.dtype, a.shape, want)
        assert a.ravel().tolist() == want['num'], where
    else:
        a = np.asarray(got)
        assert list(a.shape) == want['shape'], (where, a.shape, want)
        if 'bytes' in want:
            assert [x.decode('latin1') for x in a.ravel()] == want['bytes'], where
        else:
            assert [x.decode() if isinstance(x, bytes) else str(x) for x in a.ravel()] == want['str'], where


@pytest.mark.parametrize('fname', sorted(META['files']))
def test_every_object_reads_like_h5py_read_it(fname):
    rec = META['files'][fname]
    seen = 0
    with hdf5_min.File(os.path.join(DIR, fname)) as f:
        for name, want in rec.items():
            obj = f if name == '/' else f[name]
            assert set(obj.attrs) == set(want['attrs']), (name, sorted(obj.attrs), sorted(want['attrs']))
            for k, v in want['attrs'].items():
                _same_value(obj.attrs[k], v, (fname, name, k))
            if 'keys' in want:
                assert isinstance(obj, hdf5_min.Group) and obj.keys() == want['keys'] and len(obj) == len(want['keys'])
                assert all(k in obj for k in want['keys']) and 'no such thing' not in obj
            else:
                assert isinstance(obj, hdf5_min.Dataset) and list(obj.shape) == want['shape']
                if 'value' in want:
                    _same_value(obj.read(), want['value'], (fname, name))
                else:
                    a = np.asarray(obj)
                    ref = EXPECTED['%s|%s' % (fname, name)]
                    assert a.dtype.str == want['dtype'] and a.dtype == obj.dtype
                    assert a.shape == ref.shape and a.tobytes() == ref.tobytes(), (fname, name)     # bit for bit, NaN-safe
                    seen += 1
        with pytest.raises(KeyError):
            f['no such thing']
    assert seen > 0 or fname.endswith('.h5')
    if fname == 'features.h5':                     # the fixture does exercise the two-level B-trees it was made for
        blob = open(os.path.join(DIR, fname), 'rb').read()
        import re
        nodes = [(blob[m.start() + 4], blob[m.start() + 5]) for m in re.finditer(b'TREE', blob)]       # (node type, level)
        assert (0, 1) in nodes and (1, 1) in nodes and rec['/many_links']['keys'][-1] == 'd199'


@pytest.mark.parametrize('fname', ['keras_full_model.h5', 'keras_weights_only.h5', 'keras_split_attrs.h5'])
def test_keras_checkpoints_load_without_h5py(fname, tmp_path):
    """load_flat() on checkpoint layouts written by the real library: the flat vector = the kernels and biases in graph order."""
    from dsen2_amd import weights as W
    cin, cout, d, f = META['arch']
    want = EXPECTED['%s|flat' % fname]
    got = W.load_flat(os.path.join(DIR, fname), cin, cout, d, f)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    # asked for by the reference's file name (testing/supres.py:55-60)
    p = str(tmp_path / 's2_032_lr_1e-04.hdf5')
    shutil.copy(os.path.join(DIR, fname), p)
    assert np.array_equal(W.load_flat(p, cin, cout, d, f), want)
    with pytest.raises(ValueError):
        W.load_flat(p, cin + 2, 2, d, f)


def test_matlab_v73_tile_reads_as_the_reference_readh5_does():
    """cli._load(.mat): datasets are CHW on disk (MATLAB is column-major), transposed to HWC (testing/demoDSen2.py:14-28)."""
    from dsen2_amd import cli
    d10, d20, d60 = cli._load(os.path.join(DIR, 'matlab_v73.mat'))
    for a, key in ((d10, 'im10'), (d20, 'im20'), (d60, 'im60')):
        assert np.array_equal(a, EXPECTED['matlab_v73.mat|/%s' % key].transpose())
    with hdf5_min.File(os.path.join(DIR, 'matlab_v73.mat')) as f:
        assert f.userblock_size == 512 and f['im10'].attrs['MATLAB_class'] == b'single'


@pytest.mark.parametrize('tile', ['T33UUB', 'T49JGM'])
def test_the_reference_tiles_themselves(tile):
    """The two tiles the reference ships, where its checkout is present (this container; never the GPU box): the reader's
    arrays = the committed captures of them that h5py made (tests/golden/tile_*_600.npz)."""
    import glob
    hits = glob.glob('/root/reference/data/*%s.mat' % tile)
    if not hits:
        pytest.skip('no reference checkout here')
    from dsen2_amd import cli
    g = np.load(os.path.join(HERE, 'golden', 'tile_%s_600.npz' % tile))
    got = cli._load(hits[0])
    for a, key in zip(got, ('d10', 'd20', 'd60')):
        assert a.shape == g[key].shape and np.array_equal(a.astype(g[key].dtype), g[key]), key


@pytest.mark.parametrize('fname,path,word', [('unsupported_compound.h5', 'table', 'compound'),
                                             ('unsupported_dense_attrs.h5', None, 'dense'),
                                             ('unsupported_fixed_array.h5', 'chunks', 'fixed array')])
def test_what_is_not_implemented_is_named_not_guessed(fname, path, word):
    with hdf5_min.File(os.path.join(DIR, fname)) as f:
        if 'ok' in f.keys():
            assert np.array_equal(np.asarray(f['ok']), np.arange(3, dtype='f4'))      # the rest of the file still reads
        with pytest.raises(hdf5_min.UnsupportedHDF5) as e:
            if path is None:
                f.attrs
            else:
                np.asarray(f[path])
        assert word in str(e.value)
    # through read_with(): no h5py here, so the error also says what to do
    import importlib.util
    if importlib.util.find_spec('h5py') is None:
        with pytest.raises(hdf5_min.UnsupportedHDF5) as e:
            hdf5_min.read_with(os.path.join(DIR, fname), lambda f: f.attrs if path is None else np.asarray(f[path]), 'ADVICE')
        assert word in str(e.value) and 'h5py is not installed' in str(e.value) and 'ADVICE' in str(e.value)


def test_not_hdf5_and_missing_files_raise_oserror_like_keras(tmp_path):
    p = tmp_path / 'junk.hdf5'
    p.write_bytes(b'not an hdf5 file' * 100)
    with pytest.raises(OSError):
        hdf5_min.File(str(p))
    (tmp_path / 'empty.hdf5').write_bytes(b'')
    with pytest.raises(OSError):
        hdf5_min.File(str(tmp_path / 'empty.hdf5'))
    with pytest.raises(OSError):
        hdf5_min.File(str(tmp_path / 'missing.hdf5'))
    from dsen2_amd import weights as W
    with pytest.raises(OSError):
        W.load_flat(str(p), 10, 6, 2, 8)


def test_damaged_files_fail_with_an_exception_never_a_hang_or_wrong_shape(tmp_path):
    """Truncations and 400 random byte flips of a checkpoint: every outcome is either the right answer or a Python exception
    (all loops of the reader are bounded by the file's own sizes)."""
    from dsen2_amd import weights as W
    cin, cout, d, f = META['arch']
    blob = open(os.path.join(DIR, 'keras_full_model.h5'), 'rb').read()
    want = EXPECTED['keras_full_model.h5|flat']
    p = str(tmp_path / 'x.hdf5')
    for cut in (9, 100, 2000, len(blob) // 2, len(blob) - 100):
        open(p, 'wb').write(blob[:cut])
        with pytest.raises(Exception):
            W.load_flat(p, cin, cout, d, f)
    rng = np.random.default_rng(0)
    outcomes = {'same': 0, 'different values': 0, 'exception': 0}
    for _ in range(400):
        b = bytearray(blob)
        for pos in rng.integers(0, len(b), 3):
            b[pos] ^= 1 << int(rng.integers(0, 8))
        open(p, 'wb').write(bytes(b))
        try:
            got = W.load_flat(p, cin, cout, d, f)
            assert got.shape == want.shape
            outcomes['same' if np.array_equal(got, want) else 'different values'] += 1
        except RecursionError:
            raise
        except Exception:
            outcomes['exception'] += 1
    print(outcomes)
    assert outcomes['same'] + outcomes['different values'] + outcomes['exception'] == 400


def test_tree_listing_of_a_checkpoint():
    """python -m dsen2_amd.hdf5_min FILE: what a user looks at when a checkpoint does not load (layer names, weight names,
    shapes)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, '-m', 'dsen2_amd.hdf5_min', os.path.join(DIR, 'keras_full_model.h5')], cwd=root,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1000:]
    assert 'model_weights/' in p.stdout and '@layer_names' in p.stdout and 'kernel:0  float32 (3, 3, 10, 8)' in p.stdout
    assert 'optimizer_weights/' in p.stdout


def test_fletcher32_is_verified_and_short_reads_are_errors(tmp_path):
    """the checksum the library stores after a fletcher32 chunk is CHECKED (the real library's file passes; a
    stored checksum with one bit flipped is a ValueError naming the dataset), and no read past the end of the map comes
    back short."""
    from dsen2_amd import hdf5_min as H
    blob = bytearray(open(os.path.join(DIR, 'features.h5'), 'rb').read())
    with H.File(os.path.join(DIR, 'features.h5')) as f:
        want = f['shuffled'][()]
        addrs = []
        f['shuffled']._chunks_for_test(addrs) if hasattr(f['shuffled'], '_chunks_for_test') else None
    assert want.shape == (50, 70)
    # odd length + end-around carry, against the definition written out byte by byte
    def slow(d):
        s1 = s2 = 0
        for i in range(0, len(d) - 1, 2):
            s1 += (d[i] << 8) | d[i + 1]
            s2 += s1
        if len(d) & 1:
            s1 += d[-1] << 8
            s2 += s1
        f16 = lambda s: (s % 65535) or (65535 if s else 0)      # noqa: E731
        return (f16(s2) << 16) | f16(s1)
    rng = np.random.default_rng(2)
    for n in (0, 1, 2, 3, 126, 127, 4096, 4097):
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert H._fletcher32(d) == slow(d), n
    assert H._fletcher32(b'\xff\xff' * 5) == slow(b'\xff\xff' * 5)
    # a flipped bit anywhere INSIDE a stored chunk of 'shuffled' (gzip + shuffle + fletcher32) is an exception — caught by zlib
    # or by the checksum — never different values.  (Bits of the unchecksummed metadata around it can still change what is
    # read: version-1 B-tree nodes carry no checksum; that is the format, and test_damaged_files_... covers it.)
    chunks = []
    real = H.Dataset._unfilter

    def spy(self, chunk, mask, filters, itemsize):
        raw = bytes(chunk)
        chunks.append((bytes(blob).find(raw), len(raw)))
        return real(self, chunk, mask, filters, itemsize)
    H.Dataset._unfilter = spy
    try:
        with H.File(os.path.join(DIR, 'features.h5')) as f:
            f['shuffled'][()]
    finally:
        H.Dataset._unfilter = real
    assert len(chunks) == 8 * 8 and all(pos > 0 and n > 8 for pos, n in chunks)       # 50 x 70 in chunks of 7 x 9
    p = str(tmp_path / 'x.h5')
    for pos, n in chunks[::3]:
        for off in (0, n // 2, n - 1):                                           # the last 4 bytes are the stored checksum
            b = bytearray(blob)
            b[pos + off] ^= 1 << int(rng.integers(0, 8))
            open(p, 'wb').write(bytes(b))
            with pytest.raises(Exception):
                with H.File(p) as f:
                    f['shuffled'][()]
    m = H._Map(b'0123456789')
    assert m[2:4] == b'23' and m[9] == ord('9') and len(m) == 10
    for bad in (slice(8, 12), slice(-1, 3), slice(5, 2), slice(None, 11)):
        with pytest.raises(ValueError):
            m[bad]
    with pytest.raises(ValueError):
        m[10]
    with pytest.raises(ValueError):
        m[None:4] if False else m[slice('a', 4)]
