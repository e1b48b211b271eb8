#!/usr/bin/env python3
"""Samples the GPU's power, power cap and shader clock from sysfs (hwmon) while a bench.py run keeps one kernel family
busy — evidence for HISTORY.md §3.2b / profiles/archive/r02_ablation.md §2 (is the bf16 body convolution power-limited?).

    python tools/power_probe.py [config ...]          # default: dsen2_20_fp32 vdsen2_20_bf16
    python tools/power_probe.py --mfma                # build/mfma_peak (tools/mfma_peak.hip): the bare matrix pipe

The parent process never touches the GPU: it starts `bench.py --config C --steps K --no-cpu-baseline` as a child and
reads /sys/class/drm/card*/device/hwmon/hwmon*/{power1_average,power1_input,power1_cap,freq1_input} every 50 ms.
Prints one JSON line per config with the median / max over the child's steady phase for every card that has sensors
(on a multi-GPU host the busy card is the one whose power rises)."""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sensors():
    out = []
    for hw in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
        card = hw.split('/')[4]
        s = {'card': card}
        for key, names in (('power', ('power1_average', 'power1_input')), ('cap', ('power1_cap',)),
                           ('sclk', ('freq1_input',)), ('mclk', ('freq2_input',)), ('temp', ('temp1_input',))):
            for n in names:
                p = os.path.join(hw, n)
                if os.path.exists(p):
                    s[key] = p
                    break
        if 'power' in s or 'sclk' in s:
            out.append(s)
    return out


def read(p):
    try:
        return float(open(p).read().split()[0])
    except (OSError, ValueError, IndexError):
        return None


def sample(sens):
    row = {}
    for s in sens:
        row[s['card']] = {k: read(s[k]) for k in ('power', 'cap', 'sclk', 'mclk', 'temp') if k in s}
    return row


def med(v):
    v = sorted(x for x in v if x is not None)
    return v[len(v) // 2] if v else None


def summarize(rows, sens):
    out = {}
    for s in sens:
        c = s['card']
        col = lambda k: [r[c].get(k) for r in rows if c in r]
        pw, ck = col('power'), col('sclk')
        out[c] = {
            'power_w_median': None if med(pw) is None else round(med(pw) / 1e6, 1),
            'power_w_max': None if med(pw) is None else round(max(x for x in pw if x is not None) / 1e6, 1),
            'power_cap_w': None if med(col('cap')) is None else round(med(col('cap')) / 1e6, 1),
            'sclk_mhz_median': None if med(ck) is None else round(med(ck) / 1e6, 1),
            'sclk_mhz_max': None if med(ck) is None else round(max(x for x in ck if x is not None) / 1e6, 1),
            'sclk_mhz_min': None if med(ck) is None else round(min(x for x in ck if x is not None) / 1e6, 1),
            'mclk_mhz_median': None if med(col('mclk')) is None else round(med(col('mclk')) / 1e6, 1),
            'temp_c_max': None if med(col('temp')) is None else round(max(x for x in col('temp') if x is not None) / 1e3, 1),
        }
    return out


def busiest(rows, sens):
    busy = None
    for s in sens:
        pw = [r[s['card']].get('power') for r in rows]
        pw = [x for x in pw if x is not None]
        if pw and (busy is None or max(pw) > busy[1]):
            busy = (s['card'], max(pw))
    return busy


def main():
    configs = sys.argv[1:] or ['dsen2_20_fp32', 'vdsen2_20_bf16']
    sens = sensors()
    if not sens:
        print(json.dumps({'error': 'no hwmon power / clock sensors readable under /sys/class/drm'}))
        return 0
    idle = [sample(sens) for _ in range(20) if not time.sleep(0.05)]
    print(json.dumps({'phase': 'idle', 'cards_with_sensors': len(sens),
                      'median_idle_power_w': med([v['power_w_median'] for v in summarize(idle, sens).values()])}), flush=True)
    if configs and configs[0] == '--mfma':
        # calibration: tools/mfma_peak.hip (built to build/mfma_peak), one variant per child so each gets its own samples
        exe = os.path.join(ROOT, 'build', 'mfma_peak')
        for var in ('bf16_16x16x32_random', 'bf16_32x32x16_random', 'bf16_16x16x32_zeros', 'bf16_lds_reads_12_per_32', 'bf16_lds_reads_7_per_32',
                    'bf16_lds_reads_4_per_32', 'f32_32x32x2_random', 'f32_32x32x2_zeros'):
            child = subprocess.Popen([exe, '4', var], stdout=subprocess.PIPE, text=True)
            rows = []
            while child.poll() is None:
                rows.append(sample(sens))
                time.sleep(0.05)
            out = [l for l in child.stdout.read().splitlines() if l.startswith('{')]
            res = json.loads(out[-1]) if out else {}
            busy = busiest(rows, sens)
            steady = [r for r in rows if busy and (r[busy[0]].get('power') or 0) >= 0.9 * busy[1]] or rows
            res.update({'phase': 'mfma_peak', 'busy_card': busy and busy[0],
                        'busy': summarize(steady, sens).get(busy[0]) if busy else None})
            print(json.dumps(res), flush=True)
            time.sleep(2.0)
        return 0
    for cfg in configs:
        steps = '300' if 'fp32' in cfg and 'vdsen2' not in cfg else '200'
        if cfg == 'vdsen2_20_fp32':
            steps = '40'
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', cfg, '--steps', steps,
                                  '--warmup', '3', '--no-cpu-baseline'], stdout=subprocess.PIPE, text=True)
        rows, t0 = [], time.time()
        while child.poll() is None:
            rows.append((time.time() - t0, sample(sens)))
            time.sleep(0.05)
        line = child.stdout.read().strip().splitlines()
        bench = json.loads(line[-1]) if line and line[-1].startswith('{') else None
        # steady phase: the busiest card's samples above 80 % of its maximum power
        busy = None
        for s in sens:
            pw = [r[s['card']].get('power') for _, r in rows]
            pw = [x for x in pw if x is not None]
            if pw and (busy is None or max(pw) > busy[1]):
                busy = (s['card'], max(pw))
        steady = [r for _, r in rows if busy and (r[busy[0]].get('power') or 0) >= 0.8 * busy[1]] or [r for _, r in rows]
        print(json.dumps({'phase': cfg, 'samples': len(rows), 'steady_samples': len(steady),
                          'busy_card': busy[0] if busy else None, 'busy': summarize(steady, sens).get(busy[0]) if busy else None,
                          'bench_value': bench and bench.get('value'), 'bench_ms_per_step': bench and bench.get('ms_per_step'),
                          'roofline_frac': bench and bench.get('roofline', {}).get('frac')}), flush=True)
        time.sleep(2.0)
    return 0


if __name__ == '__main__':
    sys.exit(main())
