#!/usr/bin/env python3
"""Record the measured HBM traffic of a bench config's dominant kernel in profiles/body_conv_traffic.json, together with
the hash of the instruction stream it was measured on.

    python tools/update_traffic_json.py <config> <pmc summary .md> "<source note>" [output.json]

(output.json: default profiles/body_conv_traffic.json, updated in place; tools/profile_round.sh on the GPU box writes
gpurun_out/<tag>/body_conv_traffic.json — the repo's file with this config's entry replaced — to be copied over it.)

<pmc summary> = what tools/summarize_pmc.py wrote from the separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes of tools/profile_round.sh (mean per dispatch, KiB).  Corrections as MI355X_MICROARCH.md §HBM
prescribes for gfx950: FETCH_SIZE is doubled (it under-reports 16-byte-per-lane reads by 2), WRITE_SIZE is taken as it is;
both are in KiB.  `isa_sha256` comes from dsen2_amd/kernel_isa.json (written by the product build from the ISA it has just
compiled): bench.py quotes `traffic` only while that hash is the hash of the library it runs.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsen2_amd import asm_contract       # noqa: E402

# config -> (regular expression on the demangled kernel name in the PMC summary, algorithmic bytes per launch, description)
ROWS = {
    'dsen2_20_fp32': (r'conv3x3_body32_kernel<128, 128, 0, ', 512 * 32 * 32 * 128 * 4 * 2 + 9 * 128 * 128 * 4,
                      'conv3x3_body32_kernel<128,128,relu,...,STG,DEFER> (conv-A): 256 MiB in + 256 MiB out + weights'),
    'vdsen2_20_bf16': (r'conv3x3_body16w_chain_kernel<128, 256, 0>', 32 * (256 * 32 * 32 * 256 * (2 + 2) + 256 * 32 * 32 * 256 * (2 + 2 + 2 + 2 + 2)),
                       'conv3x3_body16w_chain_kernel<128,256,0>: ONE launch = all 64 body convolutions; per block hi read + t '
                       'written (conv-A), t + hi + lo read + hi + lo written (conv-B), x 32 blocks'),
    # per value and block: conv-A reads hi + xl (4 B), writes t's hi + lo (4 B); conv-B reads t (4 B) and the stream's hi + lo16
    # (4 B), writes hi, xl, lo16 (6 B) = 22 B; the last block writes fp32 (4 B) instead of the three planes: -2 B
    'dsen2_20_bf16x3': (r'conv3x3_body16w_x3_chain_kernel<64, 128>', 512 * 32 * 32 * 128 * (6 * 22 - 2),
                        'conv3x3_body16w_x3_chain_kernel<64,128>: ONE launch = all 12 body convolutions in bf16x3 (the second staging '
                        'of the hi plane, 2 B per value and convolution, is not counted as algorithmic)'),
    'vdsen2_20_fp32': (r'conv3x3_body32_kernel<256, 256, 0, ', 256 * 32 * 32 * 256 * 4 * 2 + 9 * 256 * 256 * 4,
                       'conv3x3_body32_kernel<256,256,relu,...,STG,DEFER> (conv-A): 256 MiB in + 256 MiB out + weights'),
    'dsen2_20_bf16': (r'conv3x3_body16w_chain_kernel<64, 128, 0>', 6 * (512 * 32 * 32 * 128 * (2 + 2) + 512 * 32 * 32 * 128 * (2 + 2 + 2 + 2 + 2)),
                      'conv3x3_body16w_chain_kernel<64,128,0>: ONE launch = all 12 body convolutions; per block hi read + t written '
                      '(conv-A), t + hi + lo read + hi + lo written (conv-B), x 6 blocks'),
    'vdsen2_20_bf16x3': (r'conv3x3_body16w_x3_chain_kernel<128, 256>', 256 * 32 * 32 * 256 * (32 * 22 - 2),
                         'conv3x3_body16w_x3_chain_kernel<128,256>: ONE launch = all 64 body convolutions in bf16x3'),
}


def main():
    cfg, md, note = sys.argv[1], sys.argv[2], sys.argv[3]
    dst = sys.argv[4] if len(sys.argv) > 4 else None
    pat, algorithmic, desc = ROWS[cfg]
    header, row = None, None
    for ln in open(md):
        cells = [c.strip() for c in ln.strip().strip('|').split('|')]
        if len(cells) > 2 and cells[0] == 'kernel':
            header = cells
        elif header and re.search(pat, cells[0]):
            row = dict(zip(header, cells))
    if row is None:
        sys.exit('no row matching %r in %s' % (pat, md))
    fetch_kib, write_kib = float(row['FETCH_SIZE']), float(row['WRITE_SIZE'])
    isa = json.load(open(asm_contract.ISA_JSON))[cfg]
    path = os.path.join(ROOT, 'profiles', 'body_conv_traffic.json')
    data = json.load(open(path)) if os.path.exists(path) else {}
    read_b, write_b = 2.0 * fetch_kib * 1024.0, write_kib * 1024.0
    data[cfg] = {'kernel': desc, 'mangled': isa['kernel'], 'isa_sha256': isa['isa_sha256'],
                 'fetch_size_kib': fetch_kib, 'write_size_kib': write_kib,
                 'read_bytes_corrected': read_b, 'write_bytes': write_b, 'traffic_bytes': read_b + write_b,
                 'algorithmic_bytes': algorithmic, 'dispatches': int(row['dispatches']), 'source': note}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in row and 'GRBM_GUI_ACTIVE' in row:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (MI355X_MICROARCH.md, SQ PMC
        # units): busy fraction of the matrix pipes = MFMA_BUSY / (1024 x GRBM / 8)
        data[cfg]['mfma_busy'] = round(float(row['SQ_VALU_MFMA_BUSY_CYCLES']) / (128.0 * float(row['GRBM_GUI_ACTIVE'])), 4)
        data[cfg]['gui_active_cycles_per_xcd'] = float(row['GRBM_GUI_ACTIVE']) / 8.0
    with open(dst or path, 'w') as f:
        json.dump(data, f, indent=1)
        f.write('\n')
    print('%s: %.1f MB per launch for %.1f MB algorithmic (%.2fx), isa %s' % (
        cfg, (read_b + write_b) / 1e6, algorithmic / 1e6, (read_b + write_b) / algorithmic, isa['isa_sha256'][:12]))


if __name__ == '__main__':
    main()
