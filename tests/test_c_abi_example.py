"""include/dsen2_hip.h from plain C: examples/c_abi_forward.c (C99, gcc, no Python / torch in the process) builds against the
header and the shared library on the CPU box, and on the GPU runs the create -> load_weights -> workspace -> forward sequence
on raw hipMalloc'ed buffers, giving the Python host's result for the same weights and inputs bit for bit."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'examples', 'c_abi_forward.c')
SRC_TILE = os.path.join(ROOT, 'examples', 'c_abi_dsen2_20.c')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def build_example(out, src=SRC):
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I' + os.path.join(ROCM, 'include'),
           '-I' + os.path.join(ROOT, 'include'), src, '-L' + os.path.join(ROOT, 'dsen2_amd'), '-ldsen2_hip',
           '-L' + os.path.join(ROCM, 'lib'), '-lamdhip64', '-Wl,-rpath,' + os.path.join(ROOT, 'dsen2_amd'),
           '-Wl,-rpath,' + os.path.join(ROCM, 'lib'), '-o', out]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


@pytest.mark.skipif(shutil.which('gcc') is None, reason='gcc not available')
def test_the_header_is_plain_c_and_the_example_links(tmp_path):
    """No GPU needed: the header compiles as C99 with -Wall -Werror and every symbol the example uses resolves against the
    product library."""
    from dsen2_amd import build
    build.build()
    exe = build_example(str(tmp_path / 'c_abi_forward'))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and 'usage:' in p.stderr
    exe = build_example(str(tmp_path / 'c_abi_dsen2_20'), SRC_TILE)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and 'usage:' in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize('precision,d,feat,n,h,w', [(0, 6, 128, 3, 32, 32), (2, 2, 128, 2, 21, 37), (1, 1, 256, 2, 16, 32)])
def test_c_host_gives_the_python_hosts_bits(tmp_path, precision, d, feat, n, h, w):
    import torch
    from dsen2_amd import weights
    from dsen2_amd.DSen2Net import PRECISIONS, s2model
    exe = build_example(str(tmp_path / 'c_abi_forward'))
    flat = weights.random_he_uniform(10, 6, d, feat, seed=77 + precision, bias_scale=0.05)
    rng = np.random.default_rng(precision)
    x10 = (rng.random((n, 4, h, w), dtype=np.float32) * 5).astype(np.float32)
    x20 = (rng.random((n, 6, h, w), dtype=np.float32) * 5).astype(np.float32)
    files = {k: str(tmp_path / (k + '.f32')) for k in ('w', 'x10', 'x20', 'out')}
    flat.astype('<f4').tofile(files['w']); x10.astype('<f4').tofile(files['x10']); x20.astype('<f4').tofile(files['x20'])
    env = dict(os.environ)
    env.pop('LD_PRELOAD', None)
    p = subprocess.run([exe, files['w'], files['x10'], files['x20'], files['out'], str(n), str(h), str(w), str(d), str(feat),
                        str(precision)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    assert 'short workspace refused' in p.stdout and 'dsen2_hip' in p.stdout
    got = np.fromfile(files['out'], dtype='<f4').reshape(n, 6, h, w)
    name = [k for k, v in PRECISIONS.items() if v == precision][0]
    m = s2model(((4, None, None), (6, None, None)), num_layers=d, feature_size=feat, precision=name)
    m.set_weights_flat(flat)
    want = m.forward_device([torch.from_numpy(x10).cuda(), torch.from_numpy(x20).cuda()]).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('precision,x,y', [(0, 288, 252), (2, 224, 336), (0, 112, 224)])
def test_whole_dsen2_20_call_from_c_gives_the_python_hosts_image(tmp_path, monkeypatch, precision, x, y):
    """examples/c_abi_dsen2_20.c: tiling arithmetic in C, every kernel through the C ABI (tile_gather x 2, upsample, forward,
    recompose) — the image of dsen2_amd.supres.DSen2_20 for the same rasters and weights, bit for bit (divisible and clamped
    sizes; fp32 and bf16x3)."""
    import contextlib
    import io
    from dsen2_amd import supres, weights
    from dsen2_amd.DSen2Net import PRECISIONS
    exe = build_example(str(tmp_path / 'c_abi_dsen2_20'), SRC_TILE)
    flat = weights.random_he_uniform(10, 6, 6, 128, seed=5, bias_scale=0.05)
    rng = np.random.default_rng(x + y)
    d10 = rng.integers(35, 9000, size=(x, y, 4)).astype(np.float32)
    d20 = rng.integers(35, 9000, size=(x // 2, y // 2, 6)).astype(np.float32)
    files = {k: str(tmp_path / (k + '.f32')) for k in ('w', 'd10', 'd20', 'out')}
    flat.astype('<f4').tofile(files['w']); d10.astype('<f4').tofile(files['d10']); d20.astype('<f4').tofile(files['d20'])
    env = dict(os.environ)
    env.pop('LD_PRELOAD', None)
    p = subprocess.run([exe, files['w'], files['d10'], files['d20'], files['out'], str(x), str(y), '6', '128', str(precision)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    got = np.fromfile(files['out'], dtype='<f4').reshape(x, y, 6)
    np.save(str(tmp_path / 's2_032_lr_1e-04.npy'), flat)
    monkeypatch.setattr(supres, 'MDL_PATH', str(tmp_path) + os.sep)
    monkeypatch.setattr(supres, 'PRECISION', [k for k, v in PRECISIONS.items() if v == precision][0])
    supres.clear_model_cache()
    with contextlib.redirect_stdout(io.StringIO()):
        want = supres.DSen2_20(d10, d20)
    supres.clear_model_cache()
    assert want.shape == got.shape and np.array_equal(got, want)
