"""Compile-time contract of the LDS-DMA kernels (no GPU needed: hipcc cross-compiles gfx950).

The DMAs of conv3x3_body32.hip / conv3x3_body16.hip are issued from inline asm that writes M0 without telling the
compiler, and their synchronisation is hand-counted; both rest on properties of the generated code that a toolchain
update could silently change.  This test pins them:
  * hipcc itself never reads or writes M0 in these kernels (every mention is the asm's own `s_mov_b32 m0`);
  * every `buffer_load_dwordx4 ... lds` is the asm's (preceded by its M0 write and the wait state);
  * no kernel spills (a scratch reload is a vector-memory operation hipcc would wait for with vmcnt(0), draining
    the DMA queue every time);
  * every kernel ends its DMA stream with an explicit vmcnt(0) before s_endpgm.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('src', ['conv3x3_body32.hip', 'conv3x3_body16.hip'])
def test_dma_kernels_asm_contract(tmp_path, src):
    out = tmp_path / (src + '.s')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-gpu-rdc', '-ffp-contract=off',
                           '-Wno-unused-function', '-S', '--cuda-device-only',
                           os.path.join(ROOT, 'dsen2_amd', 'csrc', src), '-o', str(out)],
                          stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    code = [ln.strip() for ln in lines if ln.strip() and not ln.strip().startswith((';', '.', '//'))]
    m0 = [i for i, ln in enumerate(code) if re.search(r'\bm0\b', ln)]
    assert m0
    for i in m0:       # the asm's own triple: M0 write, wait state, DMA
        assert code[i].startswith('s_mov_b32 m0, ') and code[i + 1] == 's_nop 0' and \
            code[i + 2].startswith('buffer_load_dwordx4') and code[i + 2].endswith('lds'), code[i:i + 3]
    dma = [i for i, ln in enumerate(code) if ln.startswith('buffer_load_dwordx4') and ln.endswith('lds')]
    assert dma, 'no LDS-DMA instruction found'
    for i in dma:
        assert code[i - 1] == 's_nop 0' and code[i - 2].startswith('s_mov_b32 m0,'), code[i - 2:i + 1]
    assert not any('scratch_' in ln for ln in code), 'a DMA kernel spills registers'
    # every kernel body: the last s_waitcnt before s_endpgm that mentions vmcnt is vmcnt(0)
    ends = [i for i, ln in enumerate(code) if ln == 's_endpgm']
    assert ends
    for e in ends:
        back = [ln for ln in code[max(0, e - 40):e] if ln.startswith('s_waitcnt') and 'vmcnt' in ln]
        assert back and back[-1].replace(' ', '') in ('s_waitcntvmcnt(0)', 's_waitcntvmcnt(0)lgkmcnt(0)'), back[-3:]
