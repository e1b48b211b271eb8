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
    """... and the ISA hashes of the kernels whose PMC traffic bench.py quotes: kernel_isa.json (written by the build next
    to the library it describes) must be what the sources compile to now, so a library built from other sources cannot
    carry a current-looking hash."""
    import json
    hashes = asm_contract.check_sources(HIPCC, build.FLAGS)
    assert set(hashes) == set(asm_contract.TRAFFIC_KERNELS)
    if not build.needs_build():
        assert json.load(open(asm_contract.ISA_JSON)) == hashes


def test_traffic_figures_carry_the_hash_of_the_kernel_they_were_measured_on():
    """profiles/body_conv_traffic.json: every entry that bench.py may quote names the ISA it was measured on (an entry
    without a hash, or with another build's hash, is reported as stale by bench.py, never quoted)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'body_conv_traffic.json')
    data = json.load(open(path))
    assert 'dsen2_20_fp32' in data
    for cfg, d in data.items():
        assert cfg in asm_contract.TRAFFIC_KERNELS
        if 'isa_sha256' in d:
            assert len(d['isa_sha256']) == 64 and d['traffic_bytes'] >= d['algorithmic_bytes'] > 0


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


CHAIN_GOOD = """
_ZN5dsen228conv3x3_body16w_chain_kernelILi64ELi128ELi0EEEvNS_10ConvParamsENS_9ChainArgsE:
	s_mov_b32 s5, m0
	s_mov_b32 m0, s4
	s_nop 0
	buffer_load_dwordx4 v1, s[8:11], s2 offen lds
	s_mov_b32 m0, s5
	scratch_load_dword v0, off, off
	s_waitcnt vmcnt(0)
""" + "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n" * 288 + """	s_waitcnt vmcnt(0)
	s_endpgm
"""


def test_chain_kernel_scratch_accesses_must_sit_in_drained_regions():
    """A spill outside the item bodies is a vector-memory operation the hand-counted waits do not know: allowed only
    where the next vmcnt wait is vmcnt(0) (and before the next MFMA)."""
    def kernel_rules(text):          # check_listing up to the per-kernel rules (the source-specific counts need the real file)
        try:
            asm_contract.check_listing(text, 'x.hip')
        except asm_contract.AsmContractError as e:
            return str(e)
        return ''
    assert kernel_rules(CHAIN_GOOD) == ''
    bad = CHAIN_GOOD.replace('scratch_load_dword v0, off, off\n\ts_waitcnt vmcnt(0)', 'scratch_load_dword v0, off, off\n\ts_waitcnt vmcnt(3)')
    assert 'scratch access outside a drained region' in kernel_rules(bad)
    bad = CHAIN_GOOD.replace('scratch_load_dword v0, off, off\n\ts_waitcnt vmcnt(0)\n', 'scratch_load_dword v0, off, off\n')
    assert 'scratch access outside a drained region' in kernel_rules(bad)


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


FIRST16_GOOD = '\n'.join(
    '''_ZN5dsen222conv3x3_first16_kernelILi%dELi%dELb%dEEEvNS_10ConvParamsENS_11FirstInputsEi:
	buffer_load_dword v1, v2, s[20:23], 0 offen
	s_waitcnt vmcnt(0)
	ds_write_b16 v3, v1 offset:36864
	s_barrier
	buffer_load_dword v1, v2, s[20:23], 0 offen
	v_mfma_f32_32x32x16_bf16 v[96:111], v[64:67], v[68:71], v[96:111]
	buffer_store_dwordx4 v[68:71], v78, s[28:31], 0 offen
	s_waitcnt vmcnt(%d)
	ds_write_b16 v3, v1 offset:52416
	s_barrier
	s_endpgm
''' % (c, f, x, 24 if x else 16) for c in (10, 12) for f in (128, 256) for x in (0, 1))


def test_first16_contract_catches_what_it_is_for():
    """conv3x3_first16.hip: exact counted waits inside the tile loop (no drain of the deferred stores), 128-bit buffer stores
    only with the immediate soffset 0 (gfx950's late data read with a register soffset: experiments/README.md), no spills."""
    asm_contract.check_first16_listing(FIRST16_GOOD)
    with pytest.raises(asm_contract.AsmContractError):            # a register soffset on a 128-bit store
        asm_contract.check_first16_listing(FIRST16_GOOD.replace('s[28:31], 0 offen', 's[28:31], s33 offen', 1))
    with pytest.raises(asm_contract.AsmContractError):            # a wait in the loop that drains the stores
        asm_contract.check_first16_listing(FIRST16_GOOD.replace('vmcnt(16)', 'vmcnt(6)', 1))
    with pytest.raises(asm_contract.AsmContractError):            # a spill
        asm_contract.check_first16_listing(FIRST16_GOOD.replace('s_endpgm', 'scratch_load_dword v0, off, off\n\ts_endpgm', 1))
    with pytest.raises(asm_contract.AsmContractError):            # a store that bypasses the per-image descriptors
        asm_contract.check_first16_listing(FIRST16_GOOD.replace('s_endpgm', 'global_store_dwordx4 v[0:1], v[2:5], off\n\ts_endpgm', 1))
    with pytest.raises(asm_contract.AsmContractError):            # an instantiation missing
        asm_contract.check_first16_listing(FIRST16_GOOD.replace('ILi12ELi256ELb1E', 'ILi12ELi256ELb0E'))
