#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in ${@:-5=0 5=1 5=2 5=4 5=3}; do
  export DSEN2_HIP_LIB=$R/build/libdsen2_hip_diag.so DSEN2_DIAG_SET=$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fa_$c -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/fa_$c.log 2>&1
  python3 $R/tools/summarize_rocprof.py $R/gpurun_out/fa_$c $R/gpurun_out/fa_$c.md "$c" > /dev/null
  echo "== $c"; grep "first" $R/gpurun_out/fa_$c.md | head -1 | cut -c1-160
done
