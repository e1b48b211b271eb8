#!/bin/bash
# Same-box comparison of several builds of libdsen2_hip.so (run via gpurun): alternating bench.py runs.
#   tools/ab_many.sh <config> <rounds> <lib.so> [<lib.so> ...]       (libraries relative to the repo root)
set -u
CFG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $N); do
  for l in "$@"; do
    DSEN2_HIP_LIB=$R/$l timeout -k 10 300 python3 $R/bench.py --config $CFG --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$l', d['value'], d['ms_per_step'], r['ms_per_launch'], r['ms_relu_randn'], r['ms_residual_randn'])"
  done
done
