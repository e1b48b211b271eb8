"""Build libdsen2_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m dsen2_amd.build [--force]
"""
import concurrent.futures
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['conv3x3_bodyd.hip', 'conv3x3_body32.hip', 'conv3x3_body16.hip', 'conv3x3_body.hip', 'conv3x3_out.hip', 'conv3x3_mfma.hip', 'patch_ops.hip', 'capi.hip']
HEADERS = [os.path.join(CSRC, "dsen2_internal.h"), os.path.join(CSRC, "conv3x3_dma.h"), os.path.join(os.path.dirname(HERE), 'include', 'dsen2_hip.h')]
LIB = os.path.join(HERE, 'libdsen2_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: HIP's __fmul_rn/__fadd_rn are plain operators, so the default contraction would fuse the
# up-sampler's `scale*dst + offset` (skimage rounds twice) and the residual epilogue's `x + 0.1*t` into FMAs.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-ffp-contract=off',
         '-Wall', '-Wno-unused-function']
JOBS = int(os.environ.get('DSEN2_BUILD_JOBS', '4'))      # translation units compiled side by side


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    # one hipcc process per translation unit (objects in a scratch directory, nothing but the .so is left in-tree)
    with tempfile.TemporaryDirectory(prefix='dsen2_build_') as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + '.o')
            cmd = [HIPCC] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, JOBS)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-fno-gpu-rdc'] + objs + ['-o', LIB + '.tmp']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
