"""Weight containers for the DSen2 network.

"keras flat" order (what dsen2_model_load_weights takes): for every Conv2D in graph order
(utils/DSen2Net.py:29-35: conv_in, d x (convA, convB), conv_out) the kernel in keras' HWIO layout
(3, 3, Cin, Cout) followed by the bias (Cout), all float32, concatenated.

Files accepted by load_flat():
  *.npy / *.npz['flat']   a flat array in that order (what tools/convert_keras_hdf5.py writes)
  *.hdf5 / *.h5           a keras checkpoint as saved by training/supres_train.py:195-201 (read by hdf5_min.py; no h5py needed)
"""
import os

import numpy as np


def layer_shapes(cin, cout, num_layers, feature_size):
    shapes = [(cin, feature_size)]
    for _ in range(num_layers):
        shapes += [(feature_size, feature_size)] * 2
    shapes.append((feature_size, cout))
    return shapes


def num_params(cin, cout, num_layers, feature_size):
    return sum(9 * a * b + b for a, b in layer_shapes(cin, cout, num_layers, feature_size))


def random_he_uniform(cin, cout, num_layers, feature_size, seed=1, bias_scale=0.0):
    """Random-init weights of the reference architecture (keras he_uniform: U(+-sqrt(6/(9*Cin))), zero bias)
    for benchmarks and tests — the trained checkpoints are not redistributable with this repo."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    for a, b in layer_shapes(cin, cout, num_layers, feature_size):
        limit = np.sqrt(6.0 / (9 * a))
        parts.append(rng.uniform(-limit, limit, size=(3, 3, a, b)).astype(np.float32).ravel())
        parts.append(rng.uniform(-bias_scale, bias_scale, size=(b,)).astype(np.float32) if bias_scale > 0
                     else np.zeros((b,), np.float32))
    return np.concatenate(parts)


def _from_keras_hdf5(path, shapes):
    from . import hdf5_min
    return hdf5_min.read_with(path, lambda f: _from_keras_group(f, path, shapes),
                              'convert it once with tools/convert_keras_hdf5.py where h5py is')


def _from_keras_group(f, path, shapes):
    import re
    parts = []
    root = f['model_weights'] if 'model_weights' in f else f
    _s = lambda n: n.decode() if isinstance(n, bytes) else str(n)

    def attr_list(obj, name):
        # keras' load_attributes_from_hdf5_group: a list too long for one object-header message (64 KB) is saved in
        # pieces name0, name1, ...
        if name in obj.attrs:
            return [_s(n) for n in obj.attrs[name]]
        out, i = [], 0
        while '%s%d' % (name, i) in obj.attrs:
            out += [_s(n) for n in obj.attrs['%s%d' % (name, i)]]
            i += 1
        return out if i else None
    names = attr_list(root, 'layer_names')
    if names is None:
        names = list(root.keys())
    convs = []
    for lname in names:
        grp = root[lname]
        wn = attr_list(grp, 'weight_names') or []
        if not wn:
            continue                       # Input / Concatenate / Activation / Lambda / Add carry no weights
        # a Conv2D holds exactly '<layer>/kernel:0' and '<layer>/bias:0' — picked by NAME, not by position
        kern = [n for n in wn if n.split('/')[-1].startswith('kernel')]
        bias = [n for n in wn if n.split('/')[-1].startswith('bias')]
        if len(wn) != 2 or len(kern) != 1 or len(bias) != 1:
            raise ValueError('%s: layer %r holds weights %r, expected one kernel and one bias' % (path, lname, wn))
        # (the weight's own path is what is read: a name scope TensorFlow made unique — 'conv2d_1_1/kernel:0' under
        # layer 'conv2d_1' — is legitimate)
        m = re.search(r'(\d+)$', lname)
        # multi-backend keras numbers from conv2d_1; tf.keras calls the session's first layer plain 'conv2d'
        convs.append((int(m.group(1)) if m else 0, lname, np.asarray(grp[kern[0]]), np.asarray(grp[bias[0]])))
    if len(convs) != len(shapes):
        raise ValueError('%s holds %d conv layers, the architecture has %d' % (path, len(convs), len(shapes)))

    def mismatch(order):
        for (_, lname, k, b), (a, o) in zip(order, shapes):
            if k.shape != (3, 3, a, o) or b.shape != (o,):
                return 'layer %s: shape %s/%s does not match (3,3,%d,%d)' % (lname, k.shape, b.shape, a, o)
        return None
    # keras' load_weights (testing/supres.py:63) pairs the k-th weighted layer of the FILE's layer_names with the k-th weighted
    # layer of the model; for s2model's chain that is graph order (utils/DSen2Net.py:29-35) = the order the Conv2D layers were
    # created = their numeric suffix (conv2d_7, conv2d_8, ... when other models were built in the same session).  Both orders
    # are formed; the one used must chain (Cin of the first layer, F -> F through the body, Cout of the last):
    #   both chain and agree      -> that order (every checkpoint s2model can have written)
    #   only one chains           -> that one (a file whose layer_names were reordered still loads by its numbering)
    #   both chain but differ     -> the file's order, as keras would, with a warning that names both
    #   no usable numbering       -> the file's order, with a warning: the shape chain cannot tell the 2d body layers apart
    import warnings
    by_file = convs
    by_number = sorted(convs, key=lambda c: c[0]) if len(set(c[0] for c in convs)) == len(convs) else None
    err_file, err_number = mismatch(by_file), (mismatch(by_number) if by_number is not None else 'no usable numbering')
    if err_file and err_number:
        raise ValueError('%s: %s' % (path, err_file if by_number is None else err_number))
    listed = lambda order: ', '.join(c[1] for c in order)
    if err_file is None:
        chosen = by_file
        if by_number is None:
            warnings.warn('%s: conv layer names carry no usable numbering; trusting the file\'s layer_names order: %s'
                          % (path, listed(chosen)), RuntimeWarning, stacklevel=4)
        elif err_number is None and [c[1] for c in by_number] != [c[1] for c in by_file]:
            warnings.warn('%s: layer_names order (%s) differs from the layers\' numbering (%s) and both fit the architecture; '
                          'using layer_names order like keras\' load_weights' % (path, listed(by_file), listed(by_number)),
                          RuntimeWarning, stacklevel=4)
    else:
        chosen = by_number
    for _, _, k, b in chosen:
        parts += [k.astype(np.float32).ravel(), b.astype(np.float32)]
    return np.concatenate(parts)


def load_flat(path, cin, cout, num_layers, feature_size):
    """Read a weight file into keras-flat order.  A missing file raises OSError like keras' load_weights."""
    if not os.path.exists(path):
        stem = os.path.splitext(path)[0]
        for alt in (stem + '.npy', stem + '.npz'):
            if os.path.exists(alt):
                path = alt
                break
        else:
            raise OSError('Unable to open file (name = %r)' % path)
    shapes = layer_shapes(cin, cout, num_layers, feature_size)
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npy':
        flat = np.load(path)
    elif ext == '.npz':
        flat = np.load(path)['flat']
    else:
        flat = _from_keras_hdf5(path, shapes)
    flat = np.ascontiguousarray(flat, np.float32).ravel()
    want = num_params(cin, cout, num_layers, feature_size)
    if flat.size != want:
        raise ValueError('%s has %d parameters, the architecture needs %d' % (path, flat.size, want))
    return flat
