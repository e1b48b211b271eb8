"""dsen2_amd.supres held to what the REFERENCE's own testing/supres.py did: tests/golden/supres_reference_runs.{npz,json} are
recordings of its DSen2_20 / DSen2_60 / _predict, unmodified, over its own utils/patches.py, with keras' s2model replaced by
a stand-in "network" (a fixed elementwise function of all inputs; tests/golden/make_golden_supres.py, run in the build
container under /opt/conda/bin/python3.9).  Here the SAME stand-in sits behind dsen2_amd.supres's s2model, so everything
around the network — symmetric padding, tiling, per-patch up-sampling, /2000, which architecture and checkpoint are asked
for, recomposition with clamped tiles, *2000, the printed lines — must reproduce the reference's images, bit for bit."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from bits import assert_same_bits

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
META = json.load(open(os.path.join(GOLDEN, 'supres_reference_runs.json')))


class StandInModel(object):
    """The stand-in of make_golden_supres.py behind the surface supres._run uses (S2Model's): same arithmetic, same order."""
    def __init__(self, input_shape, num_layers, feature_size, device=None, precision='fp32'):
        self.seen = {'input_shape': [list(s) for s in input_shape], 'num_layers': num_layers, 'feature_size': feature_size}
        self.bands = tuple(int(s[0]) for s in input_shape)
        self.cin, self.cout = sum(self.bands), self.bands[-1]
        self.device = device

    def load_weights(self, path):
        self.seen['load_weights'] = path

    def preferred_batch(self, h, w):
        return 3                                   # several batches per call: results must not depend on it

    def forward_device(self, xs, out=None):
        p10, last = xs[0], xs[-1]
        m = (p10[:, 0] + p10[:, 1]) + (p10[:, 2] + p10[:, 3])
        y = 0.5 * last + 0.25 * m[:, None]
        if out is not None:
            out.copy_(y)
            return out
        return y


@pytest.mark.parametrize('name', sorted(META['runs']))
def test_same_rasters_same_image_as_the_reference_supres(name, monkeypatch):
    from dsen2_amd import supres
    rec = META['runs'][name]
    z = np.load(os.path.join(GOLDEN, 'supres_reference_runs.npz'))
    args = [z['%s|in%d' % (name, i)] for i in range(2 if rec['kind'] == '20' else 3)]
    want = z['%s|out' % name]
    made = []

    def factory(input_shape, num_layers=32, feature_size=256, device=None, precision='fp32'):
        made.append(StandInModel(input_shape, num_layers, feature_size, device, precision))
        return made[-1]
    monkeypatch.setattr(supres, 's2model', factory)
    assert supres.SCALE == META['SCALE'] and supres.MDL_PATH == META['MDL_PATH']
    supres.clear_model_cache()
    keep = [a.copy() for a in args]
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        img = (supres.DSen2_20 if rec['kind'] == '20' else supres.DSen2_60)(*args, deep=rec['deep'])
    supres.clear_model_cache()
    assert all(np.array_equal(a, b) for a, b in zip(args, keep))                 # the caller's arrays are not touched
    assert str(img.dtype) == rec['out_dtype'] and list(img.shape) == rec['out_shape']
    # what _predict asked of the network: architecture and checkpoint (testing/supres.py:55-60)
    for k in ('input_shape', 'num_layers', 'feature_size', 'load_weights'):
        assert made[0].seen[k] == rec['model'][k], (k, made[0].seen[k], rec['model'][k])
    # the printed lines (keras' progress bar aside: the stand-in prints none on either side)
    assert out.getvalue().splitlines() == [ln for ln in rec['stdout'].splitlines() if ln.strip()]
    # the image: the stand-in is elementwise and computed in the same order on both sides, the tiling and the recomposition are
    # copies, and the up-sampler follows scikit-image 0.18.3's float32 arithmetic operation by operation — so everything
    # around the network reproduces the reference's run BIT FOR BIT
    assert_same_bits(img, want, name)
