#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (run via gpurun).
# Usage: tools/pmc_traffic.sh <tag> [bench.py arguments]
set -u
TAG=${1:-traffic}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_$c.log 2>&1; echo "$c exit=$?"
done
python3 $R/tools/summarize_pmc.py $OUT/pmc.md $OUT/pmc_* > /dev/null; cat $OUT/pmc.md
