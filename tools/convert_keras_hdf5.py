#!/usr/bin/env python3
"""Convert the reference's keras checkpoints (models/s2_03x_lr_*.hdf5, saved by training/supres_train.py:195-201
as FULL models, so the weights sit under /model_weights/<layer>/<layer>/{kernel:0,bias:0}) into the flat .npy
(optional: dsen2_amd reads the .hdf5 itself through dsen2_amd/hdf5_min.py; the .npy loads faster and is
what to ship when the checkpoint uses an HDF5 feature that reader does not implement).

    python tools/convert_keras_hdf5.py models/s2_032_lr_1e-04.hdf5 [more.hdf5 ...]
writes models/s2_032_lr_1e-04.npy next to each input; dsen2_amd.weights.load_flat() picks it up when asked for
the .hdf5 name (testing/supres.py:55-60 file naming is kept).
The architecture is inferred from the file name exactly as supres.py selects it:
    s2_032 -> DSen2_20 (10->6, d=6, F=128)    s2_030 -> DSen2_60 (12->2, d=6, F=128)
    s2_033 -> VDSen2_20 (10->6, d=32, F=256)  s2_034 -> VDSen2_60 (12->2, d=32, F=256)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import weights      # noqa: E402

ARCH = {'s2_032': (10, 6, 6, 128), 's2_030': (12, 2, 6, 128), 's2_033': (10, 6, 32, 256), 's2_034': (12, 2, 32, 256)}


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 1
    for path in sys.argv[1:]:
        key = os.path.basename(path)[:6]
        if key not in ARCH:
            print('%s: cannot infer the architecture from the file name' % path)
            return 1
        cin, cout, d, f = ARCH[key]
        flat = weights._from_keras_hdf5(path, weights.layer_shapes(cin, cout, d, f))
        assert flat.size == weights.num_params(cin, cout, d, f)
        out = os.path.splitext(path)[0] + '.npy'
        np.save(out, flat)
        print('%s -> %s (%d parameters)' % (path, out, flat.size))
    return 0


if __name__ == '__main__':
    sys.exit(main())
