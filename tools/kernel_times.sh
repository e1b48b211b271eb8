#!/bin/bash
# rocprofv3 kernel-trace averages of one bench config for several builds of the library (run via gpurun).
#   tools/kernel_times.sh <config> <pattern> <lib.so> [<lib.so> ...]     prints the lines of the summary matching <pattern>
set -u
CFG=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  n=$(basename $l .so)
  export DSEN2_HIP_LIB=$R/$l
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$n -- python3 $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/kt_$n.log 2>&1
  python3 $R/tools/summarize_rocprof.py $R/gpurun_out/kt_$n $R/gpurun_out/kt_$n.md "$l" > /dev/null
  echo "== $l"; grep "$PAT" $R/gpurun_out/kt_$n.md | cut -c1-160
done
