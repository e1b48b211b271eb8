cd /tmp && export TMPDIR=/tmp
for l in dsen2_amd/libdsen2_hip.so build/lib_x3probe_a.so build/lib_x3probe_b.so build/lib_x3probe.so; do
  n=$(basename $l .so)
  export DSEN2_HIP_LIB=$GRAFT_REPO_ROOT/$l
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/x3lp_$n -- python3 $GRAFT_REPO_ROOT/tools/x3_layer_probe.py > $GRAFT_REPO_ROOT/gpurun_out/x3lp_$n.log 2>&1
  python3 - <<PY
import csv, glob, statistics
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/x3lp_$n/**/*kernel_trace.csv', recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    if 'x3_kernel' in r['Kernel_Name']:
        d.setdefault(r['Kernel_Name'].split('(')[0][-40:], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k_, v in d.items():
    v = v[len(v) // 2:]            # second half: steady clock
    print('$l', k_, 'median %.1f us  min %.1f' % (statistics.median(v), min(v)))
PY
done
