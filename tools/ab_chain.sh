#!/bin/bash
# Same-box A/B (run via gpurun) of the bf16 body convolutions as ONE chain launch vs layer by layer, on the diagnostic
# build (python -m dsen2_amd.build --diag): alternating bench.py runs with DSEN2_DIAG_SET=4=1 / 4=0 (+ optional
# timing-only ablation masks of the chain kernel: 1024 no layer boundary, 3 no epilogue traffic, 1027 both).
#   tools/ab_chain.sh [config=vdsen2_20_bf16] [rounds=3] [extra DSEN2_DIAG_SET settings to time as well, e.g. 1=1024 1=3]
set -u
CFG=${1:-vdsen2_20_bf16}; N=${2:-3}; shift 2 2>/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $N); do
  for c in 4=1 4=0 "$@"; do
    DSEN2_HIP_LIB=$R/build/libdsen2_hip_diag.so DSEN2_DIAG_SET=$c timeout -k 10 300 python3 $R/bench.py --config $CFG --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('diag $c', d['value'], d['ms_per_step'], r['ms_per_launch'], r['frac'])"
  done
done
