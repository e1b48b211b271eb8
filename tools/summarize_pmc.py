#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel.

    python tools/summarize_pmc.py out.md gpurun_out/pmc_*        (directories)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=70):
    name = name.replace('void ', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def main():
    dst, dirs = sys.argv[1], sys.argv[2:]
    table = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [values per dispatch]
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            for row in csv.DictReader(open(f)):
                key = (row['Dispatch_Id'], row['Counter_Name'])
                per_dispatch[key] += float(row['Counter_Value'])
                names[row['Dispatch_Id']] = row['Kernel_Name']
            for (disp, ctr), v in per_dispatch.items():
                table[names[disp]][ctr].append(v)
    counters = sorted({c for k in table.values() for c in k})
    lines = ['# rocprofv3 PMC summary (mean per dispatch; each counter group collected in its own pass)', '',
             '| kernel | dispatches | ' + ' | '.join(counters) + ' |', '|---|---|' + '---|' * len(counters)]
    for k, cs in table.items():
        if 'dsen2' not in k:
            continue
        n = max(len(v) for v in cs.values())
        lines.append('| `%s` | %d | ' % (short(k), n) + ' | '.join(
            ('%.4g' % (sum(cs[c]) / len(cs[c]))) if c in cs else '-' for c in counters) + ' |')
    open(dst, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
