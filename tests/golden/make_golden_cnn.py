"""Write tests/golden/cnn_*.npz: inputs and float64 outputs of oracle/dsen2_oracle.py for seeded
synthetic weights (weights are NOT stored: regenerate with he_uniform_weights(seed)).

These are NOT reference outputs (keras is unavailable — the CNN oracle is "parity unpinned"); they
freeze the oracle so a later edit cannot silently change what the HIP path is compared with, and they
let the C oracle and the numpy oracle be checked against a common answer.
    python tests/golden/make_golden_cnn.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dsen2_oracle as do  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (bands, n, h, w, num_layers, feature_size, weight_seed, input_seed)
    'cnn_20_d6_f128': ((4, 6), 2, 16, 16, 6, 128, 1, 0),
    'cnn_60_d6_f128': ((4, 6, 2), 2, 16, 16, 6, 128, 2, 3),
    'cnn_20_d2_f256': ((4, 6), 1, 16, 16, 2, 256, 4, 5),
    'cnn_20_d6_f128_ragged': ((4, 6), 1, 21, 37, 6, 128, 6, 7),   # H, W not multiples of the 16x16 tile
}

if __name__ == '__main__':
    for name, (bands, n, h, w, d, f, ws, xs) in CASES.items():
        cin, cout = sum(bands), bands[-1]
        flat = do.he_uniform_weights(cin, cout, d, f, seed=ws, bias_scale=0.05)
        inputs = do.synthetic_inputs(n, h, w, bands, seed=xs)
        out = do.forward(inputs, flat, d, f)
        kw = {'x%d' % i: a for i, a in enumerate(inputs)}
        np.savez_compressed(os.path.join(HERE, name + '.npz'), out=out, bands=np.array(bands), num_layers=d,
                            feature_size=f, weight_seed=ws, bias_scale=0.05,
                            weights_checksum=float(flat.astype(np.float64).sum()), **kw)
        print(name, out.shape, float(np.abs(out).max()))
