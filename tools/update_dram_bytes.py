#!/usr/bin/env python3
"""Add the EXACT request bytes at the L2's memory side to profiles/body_conv_traffic.json.

    python tools/update_dram_bytes.py <config> <pmc summary .md of tools/dram_request_bytes.sh> "<source note>"

TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B count 32-byte units of the requests the L2 sends towards DRAM-addressed
memory (a 64-byte request counts 2, a 128-byte one 4; GMI and IO requests have counters of their own, both 0 here): a byte
count that needs no correction factor.  They sit on the same EA interface as FETCH_SIZE / WRITE_SIZE — BEFORE the Infinity
Cache — so they confirm `traffic_bytes` (and the guide's x 2 for FETCH_SIZE) but cannot say how much of it the Infinity Cache
absorbs: `rocprofv3 -L` on this image (profiles/r05_rocprofv3_counter_names.txt) lists no counter beyond the EA.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from update_traffic_json import ROWS      # noqa: E402


def main():
    cfg, md, note = sys.argv[1], sys.argv[2], sys.argv[3]
    pat = ROWS[cfg][0]
    header = row = None
    for ln in open(md):
        cells = [c.strip() for c in ln.strip().strip('|').split('|')]
        if len(cells) > 2 and cells[0] == 'kernel':
            header = cells
        elif header and re.search(pat, cells[0].strip('`')):
            row = dict(zip(header, cells))
    if row is None:
        sys.exit('no row matching %r in %s' % (pat, md))
    f = lambda k: float(row[k])      # noqa: E731
    path = os.path.join(ROOT, 'profiles', 'body_conv_traffic.json')
    data = json.load(open(path))
    e = data[cfg]
    rd, wr = f('TCC_EA0_RDREQ_DRAM_32B_sum') * 32, f('TCC_EA0_WRREQ_WRITE_DRAM_32B_sum') * 32
    e['dram_bytes'] = {
        'read': rd, 'write': wr, 'total': rd + wr,
        'vs_traffic_bytes': round((rd + wr) / e['traffic_bytes'], 4),
        'read_requests': f('TCC_EA0_RDREQ_sum'), 'read_requests_128B': f('TCC_EA0_RDREQ_128B_sum'),
        'read_requests_64B': f('TCC_EA0_RDREQ_64B_sum'), 'read_requests_32B': f('TCC_EA0_RDREQ_32B_sum'),
        'gmi_32B': f('TCC_EA0_RDREQ_GMI_32B'), 'io_32B': f('TCC_EA0_RDREQ_IO_32B'),
        'l2_read_sectors_32B': f('TCC_READ_SECTORS_sum'),
        'dispatches': int(row['dispatches']),
        'meaning': 'exact bytes of the requests L2 -> DRAM-addressed memory (32-byte units, no correction factor); measured '
                   'at the EA, i.e. BEFORE the Infinity Cache: an upper bound of what reaches HBM, not a split',
        'source': note,
    }
    json.dump(data, open(path, 'w'), indent=1)
    print(cfg, json.dumps(e['dram_bytes'], indent=1))


if __name__ == '__main__':
    main()
