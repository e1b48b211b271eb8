"""Compile-time contract of the LDS-DMA kernels (no GPU needed: hipcc cross-compiles gfx950).

The checks live in dsen2_amd/asm_contract.py and also run inside every product build (dsen2_amd.build), so a
toolchain that breaks them fails the build; here they run in the CPU suite, together with negative cases that
prove the checker would notice."""
import os
import shutil

import pytest

from dsen2_amd import asm_contract, build

HIPCC = build.HIPCC


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_dma_kernels_asm_contract():
    assert asm_contract.check_sources(HIPCC, build.FLAGS)


GOOD = """
_Z4kernILi0EEvv:
	s_mov_b32 s5, m0
	s_mov_b32 m0, s4
	s_nop 0
	buffer_load_dwordx4 v1, s[8:11], s2 offen lds
	s_mov_b32 m0, s5
	s_waitcnt vmcnt(0)
	s_endpgm
"""


def test_checker_accepts_the_pattern_and_rejects_breakage():
    asm_contract.check_listing(GOOD, 'x.hip')
    with pytest.raises(asm_contract.AsmContractError):       # compiler-made M0 use
        asm_contract.check_listing(GOOD.replace('s_waitcnt vmcnt(0)', 's_mov_b32 m0, s9\n\ts_waitcnt vmcnt(0)'), 'x.hip')
    with pytest.raises(asm_contract.AsmContractError):       # a spill
        asm_contract.check_listing(GOOD.replace('s_waitcnt vmcnt(0)', 'scratch_load_dword v0, off, off\n\ts_waitcnt vmcnt(0)'), 'x.hip')
    with pytest.raises(asm_contract.AsmContractError):       # DMA still in flight at exit
        asm_contract.check_listing(GOOD.replace('s_waitcnt vmcnt(0)', 's_waitcnt vmcnt(1)'), 'x.hip')
    with pytest.raises(asm_contract.AsmContractError):       # DMA without the wait state
        asm_contract.check_listing(GOOD.replace('\ts_nop 0\n', ''), 'x.hip')
    with pytest.raises(asm_contract.AsmContractError):       # M0 not saved
        asm_contract.check_listing(GOOD.replace('\ts_mov_b32 s5, m0\n', '').replace('\ts_mov_b32 m0, s5\n', ''), 'x.hip')
