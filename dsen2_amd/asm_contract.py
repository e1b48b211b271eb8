"""Compile-time contract of the inline-asm LDS-DMA kernels (conv3x3_body32.hip, conv3x3_body16w.hip).

The DMAs are issued from inline asm and synchronised by hand-counted `s_waitcnt vmcnt(N)`; both rest on
properties of the generated code that a toolchain or flag change could silently break.  `check_sources` compiles
the two translation units to ISA with exactly the product's flags and fails (AsmContractError) unless

  * every mention of M0 belongs to a DMA statement, which saves M0, sets it, waits one state, issues
    `buffer_load_dwordx4 ... lds` and restores M0 (hipcc reserves M0 and refuses it in a clobber list, so the
    statements preserve it themselves);
  * every `buffer_load_dword[x4] ... lds` is one of those statements;
  * no kernel spills (a scratch access is a vector-memory operation: hipcc waits for a reload with vmcnt(0), which
    drains the DMA queue, and it is not in the hand counts) — the chain kernel (one launch over all body layers, three
    epilogue forms in one kernel) may spill a few values ACROSS its item loops, i.e. at a layer boundary where the
    workgroup drains everything anyway, but never between the first and the last MFMA of a nine-step item body;
  * every kernel ends with `s_waitcnt vmcnt(0)` before `s_endpgm` (no DMA may still be writing LDS that already
    belongs to the next workgroup);
  * the bf16 kernel's epilogues contain exactly the number of vector-memory operations its first-chunk waits count
    as younger than their target (conv3x3_body16w.hip, E_OPS): an over-count there would be a weaker wait;
  * nothing is called out of line (an epilogue that is not inlined passes 128 accumulators through memory).

The matrix-core output convolution (conv3x3_out_mfma.hip) has no inline-asm DMA, but its speed rests on where hipcc
puts the waits for its operand loads (one unit = 8 loads ahead, issued in two halves): `check_out_mfma_listing` fails unless
no instantiation spills and every vector-memory wait between a kernel's first and last MFMA is `vmcnt(8)` or weaker (group J
of a unit has 11 - J or 15 - J younger loads) — a smaller count means the MFMAs wait for loads issued just before them
(what a conditional fetch produced).

dsen2_amd.build runs it on every product build; tests/test_dma_asm_contract.py runs it in the CPU suite.
"""
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
DMA_SOURCES = ['conv3x3_body32.hip', 'conv3x3_body16w.hip']


class AsmContractError(RuntimeError):
    pass


def _code_lines(text):
    return [ln.strip() for ln in text.splitlines() if ln.strip() and not ln.strip().startswith((';', '.', '//'))]


def _kernels(text):
    """name -> list of instruction lines, for every kernel in an ISA listing."""
    out, name, body = {}, None, []
    for ln in text.splitlines():
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            t = ln.strip()
            if t and not t.startswith((';', '.', '//')):
                body.append(t.split(';')[0].strip())
            if t.startswith('s_endpgm'):
                out[name] = body
                name = None
    return out


def check_listing(text, src):
    code = _code_lines(text)
    code = [ln.split(';')[0].strip() for ln in code]

    def need(cond, msg, ctx=()):
        if not cond:
            raise AsmContractError('%s: %s %s' % (src, msg, list(ctx)))

    m0 = [i for i, ln in enumerate(code) if re.search(r'\bm0\b', ln)]
    need(m0, 'no M0 use found (the DMA statements are gone?)')
    dma = [i for i, ln in enumerate(code) if ln.startswith('buffer_load_dword') and ln.endswith('lds')]    # dwordx4, and dword (the chain's bias)
    need(dma, 'no LDS-DMA instruction found')
    allowed = set()
    for i in dma:
        need(code[i - 1] == 's_nop 0' and re.match(r's_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$', code[i - 2]), 'DMA without its M0 write + wait state',
             code[i - 2:i + 1])
        allowed.update((i - 2,))
        # the statement's save (before the first M0 write; the EXEC-narrowing form has two more scalar moves in
        # between) and restore (after the last DMA)
        for j in range(i - 3, max(i - 7, -1), -1):
            if re.match(r's_mov_b32 (s\d+|vcc_lo|vcc_hi), m0$', code[j]):
                allowed.add(j)
                break
        k = i + 1
        if re.match(r's_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$', code[k]):
            allowed.add(k)
    for i in m0:
        need(i in allowed, 'M0 touched outside a DMA statement', code[max(0, i - 2):i + 3])
    saves = sum(1 for ln in code if re.match(r's_mov_b32 (s\d+|vcc_lo|vcc_hi), m0$', ln))
    need(saves > 0, 'the DMA statements no longer save M0')
    STEP_MFMAS = 9 * 32                 # bf16 kernel: MFMAs of one copy of the nine-step body (4 x 8 per step)
    need(not any(ln.startswith(('s_swappc', 's_call')) for ln in code), 'a DMA kernel calls a function (an epilogue was not inlined)')
    kernels = _kernels(text)
    need(kernels, 'no kernel found')
    n_dma_kernels = 0
    for name, body in kernels.items():
        if not any(ln.startswith('buffer_load_dword') and ln.endswith('lds') for ln in body):
            continue                       # a kernel without LDS-DMA (the split / join helpers)
        n_dma_kernels += 1
        chain = '_chain_kernel' in name
        if not chain:
            need(not any('scratch_' in ln for ln in body), 'kernel %s spills registers' % name)
        else:
            mf = [i for i, ln in enumerate(body) if ln.startswith('v_mfma')]
            need(mf and len(mf) % STEP_MFMAS == 0, 'chain kernel %s: %d MFMAs is not a whole number of item bodies' % (name, len(mf)))
            for g in range(0, len(mf), STEP_MFMAS):
                inner = body[mf[g]:mf[g + STEP_MFMAS - 1]]
                need(not any('scratch_' in ln for ln in inner), 'chain kernel %s spills inside an item body' % name)
            # ... and a scratch access outside the item bodies must sit in a DRAINED region: it is a vector-memory
            # operation the hand counts do not know, so a counted wait (vmcnt(k > 0)) that follows it would retire one
            # operation fewer of the epilogue it targets.  Required: after every scratch access the next vmcnt wait in the
            # listing is vmcnt(0), and it comes before the next MFMA.
            for i, ln in enumerate(body):
                if 'scratch_' not in ln:
                    continue
                nxt = None
                for ln2 in body[i + 1:]:
                    if ln2.startswith('v_mfma'):
                        break
                    mm = re.search(r'vmcnt\((\d+)\)', ln2) if ln2.startswith('s_waitcnt') else None
                    if mm:
                        nxt = int(mm.group(1))
                        break
                need(nxt == 0, 'chain kernel %s: scratch access outside a drained region (next vmcnt wait: %r)' % (name, nxt), [ln])
        back = [ln for ln in body[-40:] if ln.startswith('s_waitcnt') and 'vmcnt' in ln]
        need(back and back[-1].replace(' ', '') in ('s_waitcntvmcnt(0)', 's_waitcntvmcnt(0)lgkmcnt(0)'),
             'kernel %s does not drain its DMAs before s_endpgm' % name, back[-3:])
    need(n_dma_kernels > 0, 'no LDS-DMA kernel found')
    if src == 'conv3x3_body16w.hip':
        # E_OPS of the kernel: epilogue 0 = 16 stores; 1 and 3 = 32 loads + 32 stores (compiler-visible buffer
        # operations only: the DMAs are `... lds`), once in each of the two copies of the item loop (issuing waves
        # 0-3 / worker waves 4-7)
        expect = {0: (0, 32), 1: (64, 64), 3: (64, 64)}
        found = 0
        for name, body in _kernels(text).items():
            m = re.search(r'conv3x3_body16[wx]_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', name)
            if not m or int(m.group(4)) != 0:
                continue
            found += 1
            loads = sum(1 for ln in body if ln.startswith('buffer_load_dwordx4') and not ln.endswith('lds'))
            stores = sum(1 for ln in body if ln.startswith('buffer_store_dwordx4'))
            other = [ln for ln in body if re.match(r'(buffer|global|flat)_(load|store|atomic)', ln)
                     and not ln.startswith(('buffer_load_dwordx4', 'buffer_store_dwordx4'))]
            # bias preload: global_load_dword in the prologue (before the first vmcnt(0)) is the only other access
            need(len(other) <= 1, 'unexpected vector-memory instructions in the bf16 body kernel', other[:4])
            need((loads, stores) == expect[int(m.group(3))],
                 'epilogue %s has %d loads / %d stores, the waits count %r' % (m.group(3), loads, stores, expect[int(m.group(3))]))
        need(found >= 6, 'expected the 6 product instantiations of the bf16 body kernel, found %d' % found)
        # the bf16x3 form (precision 2): conv-A stores two planes (32 per copy of the item loop), conv-B stores hi, xl and lo16 (48)
        expect3 = {0: (0, 64), 1: (64, 96), 3: (64, 64)}
        found3 = 0
        for name, body in _kernels(text).items():
            m = re.search(r'conv3x3_body16w_x3_kernelILi(\d+)ELi(\d+)ELi(\d+)E', name)
            if not m:
                continue
            found3 += 1
            loads = sum(1 for ln in body if ln.startswith('buffer_load_dwordx4') and not ln.endswith('lds'))
            stores = sum(1 for ln in body if ln.startswith('buffer_store_dwordx4'))
            need((loads, stores) == expect3[int(m.group(3))],
                 'bf16x3 epilogue %s has %d loads / %d stores, the waits count %r' % (m.group(3), loads, stores, expect3[int(m.group(3))]))
        need(found3 == 6, 'expected the 6 instantiations of the bf16x3 body kernel, found %d' % found3)
        # the chain kernel holds all three epilogues, each in both copies of the item loop
        chains = 0
        for name, body in _kernels(text).items():
            m = re.search(r'conv3x3_body16w_chain_kernelILi(\d+)ELi(\d+)ELi(\d+)E', name)
            if not m or int(m.group(3)) != 0:
                continue
            chains += 1
            loads = sum(1 for ln in body if ln.startswith('buffer_load_dwordx4') and not ln.endswith('lds'))
            stores = sum(1 for ln in body if ln.startswith('buffer_store_dwordx4'))
            want = (sum(v[0] for v in expect.values()), sum(v[1] for v in expect.values()))
            need((loads, stores) == want, 'chain kernel has %d loads / %d stores in its epilogues, the waits count %r' % (loads, stores, want))
        need(chains == 2, 'expected the 2 product instantiations of the chain kernel, found %d' % chains)
        chains3 = 0
        for name, body in _kernels(text).items():
            if not re.search(r'conv3x3_body16w_x3_chain_kernelILi(\d+)ELi(\d+)E', name):
                continue
            chains3 += 1
            loads = sum(1 for ln in body if ln.startswith('buffer_load_dwordx4') and not ln.endswith('lds'))
            stores = sum(1 for ln in body if ln.startswith('buffer_store_dwordx4'))
            want = (sum(v[0] for v in expect3.values()), sum(v[1] for v in expect3.values()))
            need((loads, stores) == want, 'bf16x3 chain kernel has %d loads / %d stores in its epilogues, the waits count %r' % (loads, stores, want))
        need(chains3 == 2, 'expected the 2 instantiations of the bf16x3 chain kernel, found %d' % chains3)


def check_out_mfma_listing(text, src='conv3x3_out_mfma.hip'):
    def need(cond, msg, ctx=()):
        if not cond:
            raise AsmContractError('%s: %s %s' % (src, msg, list(ctx)))

    kernels = {k: v for k, v in _kernels(text).items() if 'conv3x3_out_mfma_kernel' in k}
    need(len(kernels) == 4, 'expected 4 instantiations of the output kernel, found %d' % len(kernels))
    for name, body in kernels.items():
        need(not any('scratch_' in ln for ln in body), 'kernel %s spills registers' % name)
        need(not any(ln.startswith(('s_swappc', 's_call')) for ln in body), 'kernel %s calls a function' % name)
        mf = [i for i, ln in enumerate(body) if ln.startswith('v_mfma')]
        need(mf, 'kernel %s has no MFMA' % name)
        waits = [ln for ln in body[mf[0]:mf[-1]] if ln.startswith('s_waitcnt') and 'vmcnt' in ln]
        need(len(waits) >= 8, 'kernel %s: no operand waits between its MFMAs' % name)
        bad = [ln for ln in waits if int(re.search(r'vmcnt\((\d+)\)', ln).group(1)) < 8]
        need(not bad, 'kernel %s waits for the loads of the NEXT unit' % name, bad[:4])
    return True


def check_first16_listing(text, src='conv3x3_first16.hip'):
    """conv3x3_first16.hip relies on hipcc's own wait counts, which are only exact while the tile loop's body has no divergent
    branch, and on 128-bit buffer stores whose soffset is the IMMEDIATE 0 (with a register soffset gfx950 reads the store data
    late and hipcc does not pad the hazard: experiments/README.md)."""
    def need(cond, msg, ctx=()):
        if not cond:
            raise AsmContractError('%s: %s %s' % (src, msg, list(ctx)))

    kernels = {k: v for k, v in _kernels(text).items() if 'conv3x3_first16_kernel' in k}
    need(len(kernels) == 8, 'expected 8 instantiations of the kernel, found %d' % len(kernels))
    for name, body in kernels.items():
        need(not any('scratch_' in ln for ln in body), 'kernel %s spills registers' % name)
        need(not any(ln.startswith(('s_swappc', 's_call')) for ln in body), 'kernel %s calls a function' % name)
        stores = [ln for ln in body if ln.startswith('buffer_store_dwordx4')]
        need(stores, 'kernel %s has no 128-bit buffer store' % name)
        bad = [ln for ln in stores if not re.search(r'\], 0 offen', ln)]
        need(not bad, 'kernel %s: 128-bit buffer store with a register soffset (late data read, unpadded hazard)' % name, bad[:2])
        need(not any(ln.startswith('global_store') for ln in body), 'kernel %s stores outside its buffer descriptors' % name)
        # the halo scatter (ds_write_b16) inside the tile loop may only wait for its own loads: with the 16 (bf16x3: 24) stores of
        # the previous tile issued after them, every counted wait in front of a scatter write must leave at least that many
        # operations in flight.  (The prologue's scatter, before the first MFMA, drains: nothing else is in flight there.)
        mf = [i for i, ln in enumerate(body) if ln.startswith('v_mfma')]
        need(mf, 'kernel %s has no MFMA' % name)
        x3 = 'ELb1EEE' in name                      # template argument X3 = true
        floor = 24 if x3 else 16
        loop = body[mf[0]:]
        scat = [i for i, ln in enumerate(loop) if ln.startswith('ds_write_b16')]
        need(scat, 'kernel %s: no halo scatter after the first MFMA' % name)
        waits = [ln for ln in loop[:scat[-1]] if ln.startswith('s_waitcnt') and 'vmcnt' in ln]
        low = [ln for ln in waits if int(re.search(r'vmcnt\((\d+)\)', ln).group(1)) < floor]
        need(not low, 'kernel %s: a wait inside the tile loop drains the deferred stores (divergent branch in the loop body?)' % name, low[:4])
    return True


# The kernels whose measured HBM traffic bench.py quotes from profiles/body_conv_traffic.json (PMC counters cannot be read
# in-process): config -> (source file, regular expression on the mangled name).  Their ISA hash is written next to the
# library at build time (kernel_isa.json) and next to the traffic figure when tools/update_traffic_json.py records it, so
# bench.py can refuse a figure measured on another instruction stream.
TRAFFIC_KERNELS = {
    'dsen2_20_fp32': ('conv3x3_body32.hip', r'conv3x3_body32_kernelILi128ELi128ELi0ELi0ELi0ELb1ELb1E'),
    'vdsen2_20_bf16': ('conv3x3_body16w.hip', r'conv3x3_body16w_chain_kernelILi128ELi256ELi0E'),
    'dsen2_20_bf16x3': ('conv3x3_body16w.hip', r'conv3x3_body16w_x3_chain_kernelILi64ELi128E'),
    'vdsen2_20_fp32': ('conv3x3_body32.hip', r'conv3x3_body32_kernelILi256ELi256ELi0ELi0ELi0ELb1ELb1E'),
    'dsen2_20_bf16': ('conv3x3_body16w.hip', r'conv3x3_body16w_chain_kernelILi64ELi128ELi0E'),
    'vdsen2_20_bf16x3': ('conv3x3_body16w.hip', r'conv3x3_body16w_x3_chain_kernelILi128ELi256E'),
}
ISA_JSON = os.path.join(HERE, 'kernel_isa.json')


def isa_hashes(text):
    """mangled kernel name -> sha256 of its instruction list (local labels and whitespace normalised)."""
    import hashlib
    out = {}
    for name, body in _kernels(text).items():
        norm = [re.sub(r'\.L\w+', 'L', re.sub(r'\s+', ' ', ln)) for ln in body]
        out[name] = hashlib.sha256('\n'.join(norm).encode()).hexdigest()
    return out


def traffic_kernel_hashes(listings):
    """listings: source file -> ISA text.  config -> {'kernel': mangled name, 'isa_sha256': ...} for TRAFFIC_KERNELS."""
    out = {}
    for cfg, (src, pat) in TRAFFIC_KERNELS.items():
        hits = {k: v for k, v in isa_hashes(listings[src]).items() if re.search(pat, k)}
        if len(hits) != 1:
            raise AsmContractError('%s: expected one kernel matching %s, found %d' % (src, pat, len(hits)))
        (k, v), = hits.items()
        out[cfg] = {'kernel': k, 'isa_sha256': v}
    return out


def check_sources(hipcc, flags, verbose=False, isa_json=None):
    """Compiles the contract's translation units to ISA and checks them; returns traffic_kernel_hashes() of what it has
    compiled (and writes it to `isa_json` when given: dsen2_amd.build does, next to the library)."""
    listings = {}
    with tempfile.TemporaryDirectory(prefix='dsen2_asm_') as tmp:
        for src in DMA_SOURCES + ['conv3x3_out_mfma.hip', 'conv3x3_first16.hip']:
            out = os.path.join(tmp, src + '.s')
            cmd = [hipcc] + [f for f in flags if f not in ('-fPIC',)] + ['-S', '--cuda-device-only', os.path.join(CSRC, src), '-o', out]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            with open(out) as f:
                listings[src] = f.read()
            if src in DMA_SOURCES:
                check_listing(listings[src], src)
            elif src == 'conv3x3_first16.hip':
                check_first16_listing(listings[src], src)
            else:
                check_out_mfma_listing(listings[src], src)
    hashes = traffic_kernel_hashes(listings)
    if isa_json:
        import json
        with open(isa_json, 'w') as f:
            json.dump(hashes, f, indent=1, sort_keys=True)
            f.write('\n')
    return hashes
