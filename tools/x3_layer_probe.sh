#!/bin/bash
# One bf16x3 layer of each kind on FIXED dense random operands under rocprofv3, once per build of the library (timing-only
# variants must be compared on the same operand data: profiles/r04_bf16x3.md §4).  Run via gpurun.
#   tools/x3_layer_probe.sh [lib.so ...]        (paths relative to the repo; default: the product library alone)
# The probe builds of round 4 (build/lib_x3probe{,_a,_b}.so) were made by hand from edited copies of conv3x3_body16w.hip
# (`hipcc ... -shared`, not by dsen2_amd.build) and are not in the tree: pass whatever builds you want compared; a path
# that does not exist is skipped.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -eq 0 ] && set -- dsen2_amd/libdsen2_hip.so
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  [ -f "$R/$l" ] || { echo "skip $l (not built)"; continue; }
  n=$(basename $l .so)
  export DSEN2_HIP_LIB=$R/$l
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x3lp_$n -- python3 $R/tools/x3_layer_probe.py > $R/gpurun_out/x3lp_$n.log 2>&1
  python3 - <<PY
import csv, glob, statistics
f = glob.glob('$R/gpurun_out/x3lp_$n/**/*kernel_trace.csv', recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    if 'x3_kernel' in r['Kernel_Name']:
        d.setdefault(r['Kernel_Name'].split('(')[0][-40:], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k_, v in d.items():
    v = v[len(v) // 2:]            # second half: steady clock
    print('$l', k_, 'median %.1f us  min %.1f' % (statistics.median(v), min(v)))
PY
done
