#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a short markdown summary
(kernel names truncated) suitable for committing under profiles/.

    python tools/summarize_rocprof.py gpurun_out/prof1 profiles/r01_bench_kernel_stats.md "command line"
"""
import csv
import glob
import os
import sys


def short(name, n=110):
    name = name.replace('void ', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def main():
    src, dst = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ''
    stats = glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True)
    trace = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    lines = ['# rocprofv3 kernel summary', '', 'command: `%s`' % cmd, '',
             '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for f in stats:
        for row in csv.DictReader(open(f)):
            lines.append('| `%s` | %s | %.3f | %.1f | %.1f | %.1f | %s |' % (
                short(row['Name']), row['Calls'], float(row['TotalDurationNs']) / 1e6, float(row['AverageNs']) / 1e3,
                float(row['MinNs']) / 1e3, float(row['MaxNs']) / 1e3, row['Percentage']))
    if trace:
        seen = {}
        for row in csv.DictReader(open(trace[0])):
            k = row['Kernel_Name']
            if k not in seen:
                seen[k] = row
        lines += ['', '| kernel | VGPR | AGPR | SGPR | LDS bytes | scratch | workgroup | grid |', '|---|---|---|---|---|---|---|---|']
        for k, r in seen.items():
            lines.append('| `%s` | %s | %s | %s | %s | %s | %s | %s |' % (
                short(k, 80), r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'],
                r['Scratch_Size'], r['Workgroup_Size_X'], r['Grid_Size_X']))
    open(dst, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
