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


OUT_GOOD = ''.join("""
_ZN5dsen223conv3x3_out_mfma_kernelILi%dELi%dEEEvNS_10ConvParamsENS_11OutMfmaGeomE:
	buffer_load_dwordx4 v[0:3], v9, s[8:11], 0 offen
	v_mfma_f32_32x32x2_f32 v[16:31], v4, v0, v[16:31]
""" % fc + """	s_waitcnt vmcnt(15)
	v_mfma_f32_32x32x2_f32 v[16:31], v4, v0, v[16:31]
	buffer_load_dwordx4 v[0:3], v9, s[8:11], 0 offen
""" * 8 + """	v_mfma_f32_32x32x2_f32 v[16:31], v4, v0, v[16:31]
	s_endpgm
""" for fc in ((128, 1), (128, 3), (256, 1), (256, 3)))


def test_output_kernel_checker():
    """conv3x3_out_mfma.hip: the operand waits between MFMAs must leave the next unit's loads in flight, nothing spills."""
    asm_contract.check_out_mfma_listing(OUT_GOOD)
    with pytest.raises(asm_contract.AsmContractError):       # a wait for the loads issued just before (conditional fetch)
        asm_contract.check_out_mfma_listing(OUT_GOOD.replace('vmcnt(15)', 'vmcnt(3)', 1))
    with pytest.raises(asm_contract.AsmContractError):       # a spill
        asm_contract.check_out_mfma_listing(OUT_GOOD.replace('s_endpgm', 'scratch_load_dword v0, off, off\n\ts_endpgm', 1))
    with pytest.raises(asm_contract.AsmContractError):       # an instantiation missing
        asm_contract.check_out_mfma_listing(OUT_GOOD.replace('ILi256ELi3E', 'ILi256ELi1E'))
