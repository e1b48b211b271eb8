"""The HDF5-dependent tests (keras checkpoint reader, MATLAB v7.3 tile reader) must not be silently skipped: the
system python of the build image has no h5py, the image's conda python has.  When h5py is missing here and that
interpreter exists, run tests/test_keras_hdf5_reader.py under it."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA_PY = os.environ.get('DSEN2_H5PY_PYTHON', '/opt/conda/bin/python3.9')


@pytest.mark.skipif(importlib.util.find_spec('h5py') is not None, reason='h5py importable here: the tests run directly')
def test_hdf5_readers_under_the_interpreter_that_has_h5py():
    if not os.path.exists(CONDA_PY):
        pytest.skip('no interpreter with h5py on this machine (%s)' % CONDA_PY)
    probe = subprocess.run([CONDA_PY, '-c', 'import h5py, pytest, numpy'], capture_output=True)
    if probe.returncode != 0:
        pytest.skip('%s lacks h5py / pytest / numpy' % CONDA_PY)
    p = subprocess.run([CONDA_PY, '-m', 'pytest', '-q', '-p', 'no:cacheprovider', os.path.join(ROOT, 'tests', 'test_keras_hdf5_reader.py')],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert ' passed' in p.stdout and 'skipped' not in p.stdout.splitlines()[-1]
