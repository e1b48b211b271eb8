#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a short markdown summary
(kernel names truncated) suitable for committing under profiles/.

    python tools/summarize_rocprof.py gpurun_out/prof1 profiles/archive/r01_bench_kernel_stats.md "command line"
"""
import csv
import glob
import os
import sys


def short(name, n=110):
    name = name.replace('void ', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def in_network_table(rows):
    """Split every kernel's dispatches by WHERE they ran: inside a forward pass of the network — a `first` convolution,
    then body-convolution launches only, then an output convolution, in dispatch order — or standalone (bench.py's
    per-epilogue timing on dense random operands, kernel-level tests).  The first forward of the process (cold: code
    object load, first touch of every buffer) is listed apart.  Also the gaps between consecutive kernels of a forward
    (start of one minus end of the one before): what the events around the body convolutions see beside the kernels."""
    rows = sorted(rows, key=lambda r: int(r['Start_Timestamp']))
    is_first = lambda n: 'conv3x3_first_kernel' in n or 'pack_inputs_kernel' in n
    is_out = lambda n: 'conv3x3_out_' in n
    is_body = lambda n: 'conv3x3_body' in n or ('conv3x3_mfma_kernel' in n)
    forwards, cur = [], None
    tagged = {}                                    # dispatch id -> 'cold' | 'net' | 'alone'
    for r in rows:
        n, d = r['Kernel_Name'], r['Dispatch_Id']
        if is_first(n):
            cur = [r]
        elif cur is not None and is_out(n):
            cur.append(r)
            forwards.append(cur)
            cur = None
        elif cur is not None and is_body(n):
            cur.append(r)
        else:
            cur = None
            tagged[d] = 'alone'
    # The chip's clock after an idle stretch: for ~25 ms after the GPU sat idle for more than a millisecond (process start,
    # a host-side allocation or synchronisation followed by host work) the same launch is up to 16 % slower
    # (profiles/r04_ablation.md §2).  A forward pass that STARTS inside such a window is tagged `ramp`, not `net`.
    IDLE_NS, RAMP_NS = 1_000_000, 30_000_000
    idle_ends = [int(rows[0]['Start_Timestamp'])] if rows else []
    for a_, b_ in zip(rows, rows[1:]):
        if int(b_['Start_Timestamp']) - int(a_['End_Timestamp']) > IDLE_NS:
            idle_ends.append(int(b_['Start_Timestamp']))

    def in_ramp(t):
        return any(0 <= t - e < RAMP_NS for e in idle_ends)
    steady = []
    for i, f in enumerate(forwards):
        ramp = in_ramp(int(f[0]['Start_Timestamp']))
        if not ramp:
            steady.append(f)
        for r in f:
            tagged[r['Dispatch_Id']] = 'ramp' if ramp else 'net'
    for r in rows:
        if tagged.get(r['Dispatch_Id']) is None and in_ramp(int(r['Start_Timestamp'])):
            tagged[r['Dispatch_Id']] = 'alone-ramp'
    for r in rows:
        tagged.setdefault(r['Dispatch_Id'], 'alone')
    agg = {}
    for r in rows:
        if 'dsen2' not in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'], tagged[r['Dispatch_Id']])
        agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    lines = ['', '## by place of the launch (from the kernel trace, in dispatch order)', '',
             '`net` = inside a forward pass (first convolution ... output convolution) that started at least 30 ms after the GPU '
             'last sat idle for more than 1 ms; `ramp` = inside a forward pass that started sooner after such an idle stretch '
             '(process start, host-side allocation or synchronisation + host work): the clock is still coming back; `alone` / '
             '`alone-ramp` = outside a forward pass (bench.py times each epilogue alone on dense random operands, after 24 '
             'untimed launches).  The `net` average is the one `roofline.ms_per_launch` is to be compared with.', '',
             '| kernel | where | calls | avg us | min us | max us |', '|---|---|---|---|---|---|']
    for (k, where), v in sorted(agg.items()):
        lines.append('| `%s` | %s | %d | %.1f | %.1f | %.1f |' % (short(k, 90), where, len(v), sum(v) / len(v), min(v), max(v)))
    warm = steady
    if warm:
        dur = [(int(f[-1]['End_Timestamp']) - int(f[0]['Start_Timestamp'])) / 1e3 for f in warm]
        ksum = [sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in f) for f in warm]
        gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for f in warm for a, b in zip(f, f[1:])]
        body = [[r for r in f if is_body(r['Kernel_Name'])] for f in warm]
        bspan = [(int(b[-1]['End_Timestamp']) - int(b[0]['Start_Timestamp'])) / 1e3 for b in body if b]
        lines += ['', '%d steady-clock (`net`) forward passes of %d kernels: first start to last end %.1f us on average, of which kernels %.1f us '
                  'and %d gaps between consecutive kernels %.2f us each on average (max %.2f); the body convolutions alone span '
                  '%.1f us per pass = %.2f us per launch INCLUDING the gaps between them.'
                  % (len(warm), len(warm[0]), sum(dur) / len(dur), sum(ksum) / len(ksum), len(warm[0]) - 1,
                     sum(gaps) / max(1, len(gaps)), max(gaps) if gaps else 0.0, sum(bspan) / max(1, len(bspan)),
                     sum(bspan) / max(1, len(bspan)) / max(1, len(body[0])))]
    return lines


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
        lines += in_network_table(list(csv.DictReader(open(trace[0]))))
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
