import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def oracle_dsen2_tile():
    """fn(name, flat, run_60=False) -> the float64 oracle pipeline's DSen2_20 (DSen2_60) image (raw units) of the bundled tile
    tests/golden/<name> for the keras-flat weights `flat`: oracle tiling + up-sampling, /2000, C oracle CNN on all 36 (16)
    patches, oracle recomposition, *2000 (testing/supres.py:15-50).  ~20 s of CPU, so it is computed once per session and
    shared by the tests that compare different arithmetic modes of the same call against it."""
    import contextlib
    import io

    import numpy as np
    cache = {}

    def get(name, flat, run_60=False):
        from oracle import c_oracle, patches_oracle as po       # checker only
        key = (name, bool(run_60), hash(np.asarray(flat).tobytes()))
        if key not in cache:
            g = np.load(os.path.join(GOLDEN, name))
            d = [g[k].astype(np.float32) for k in ('d10', 'd20', 'd60')]
            if run_60:
                p, border = po.get_test_patches60(d[0], d[1], d[2], patchSize=192, border=12, f32_coords=True), 12
            else:
                p, border = po.get_test_patches(d[0], d[1], patchSize=128, border=8, f32_coords=True), 8
            pred = c_oracle.forward([a / np.float32(2000) for a in p], flat, 6, 128)
            with contextlib.redirect_stdout(io.StringIO()):
                img = po.recompose_images(pred, border=border, size=d[0].shape)
            cache[key] = img.astype(np.float64) * 2000
        return cache[key]
    return get
