"""Build libdsen2_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m dsen2_amd.build [--force]
    python -m dsen2_amd.build --diag      -> build/libdsen2_hip_diag.so (-DDSEN2_DIAG: kernel-structure A/B switches and
                                             timing-only ablations for tools/; never loaded by the product unless
                                             DSEN2_HIP_LIB points at it)

Every product build also checks the compile-time contract of the inline-asm LDS-DMA kernels on the code object it
has just produced (dsen2_amd/asm_contract.py): a toolchain that breaks it fails the BUILD, not just a test.
"""
import concurrent.futures
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['conv3x3_body32.hip', 'conv3x3_body16w.hip', 'conv3x3_out.hip', 'conv3x3_out_mfma.hip', 'conv3x3_mfma.hip', 'conv3x3_first.hip', 'conv3x3_first16.hip', 'patch_ops.hip', 'capi.hip']
HEADERS = [os.path.join(CSRC, "dsen2_internal.h"), os.path.join(CSRC, "conv3x3_dma.h"), os.path.join(CSRC, "conv3x3_bf16_common.h"), os.path.join(os.path.dirname(HERE), 'include', 'dsen2_hip.h')]
LIB = os.path.join(HERE, 'libdsen2_hip.so')
DIAG_LIB = os.path.join(os.path.dirname(HERE), 'build', 'libdsen2_hip_diag.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: HIP's __fmul_rn/__fadd_rn are plain operators, so the default contraction would fuse the
# up-sampler's `scale*dst + offset` (skimage rounds twice) and the residual epilogue's `x + 0.1*t` into FMAs.
# -Werror=array-bounds: an out-of-range constant index into a register array is silently "undefined" (the archived
# single-copy fp32 chain computed with whatever the registers held: profiles/r04_ablation.md §1).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-ffp-contract=off',
         '-Wall', '-Wno-unused-function', '-Werror=array-bounds']
JOBS = int(os.environ.get('DSEN2_BUILD_JOBS', '4'))      # translation units compiled side by side


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__),
                                                                os.path.join(HERE, 'asm_contract.py')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, diag=False, variant=None, extra_flags=()):
    """variant = NAME (tools/ only): build/lib_NAME.so from the same sources with `extra_flags` (-D switches of single
    translation units, for same-box A/B runs through DSEN2_HIP_LIB); never the product library, no ISA contract check."""
    lib = os.path.join(os.path.dirname(DIAG_LIB), 'lib_%s.so' % variant) if variant else DIAG_LIB if diag else LIB
    if not force and not variant and not needs_build(lib):
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    flags = FLAGS + (['-DDSEN2_DIAG'] if diag else []) + list(extra_flags)
    # one hipcc process per translation unit (objects in a scratch directory, nothing but the .so is left in-tree)
    with tempfile.TemporaryDirectory(prefix='dsen2_build_') as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + '.o')
            cmd = [HIPCC] + flags + ['-c', os.path.join(CSRC, src), '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, JOBS)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-fno-gpu-rdc'] + objs + ['-o', lib + '.tmp']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        if not diag and not variant:
            from . import asm_contract
            try:
                asm_contract.check_sources(HIPCC, FLAGS, verbose=verbose, isa_json=asm_contract.ISA_JSON)
            except Exception:
                os.unlink(lib + '.tmp')
                raise
    os.replace(lib + '.tmp', lib)
    return lib


if __name__ == '__main__':
    if '--variant' in sys.argv:        # python -m dsen2_amd.build --variant NAME -DFOO=1 ...
        i = sys.argv.index('--variant')
        print(build(force=True, verbose=True, variant=sys.argv[i + 1], extra_flags=[a for a in sys.argv[i + 2:] if a.startswith('-D')]))
    else:
        print(build(force='--force' in sys.argv, verbose=True, diag='--diag' in sys.argv))
