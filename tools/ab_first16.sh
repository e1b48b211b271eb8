#!/bin/bash
# Same-box A/B of conv3x3_first16.hip variant builds (python -m dsen2_amd.build --variant NAME -D...): the first
# convolution's in-network duration (bench.py's roofline.first_ms, HIP events) per library, for the bf16 and bf16x3 lines.
#   tools/ab_first16.sh [lib.so ...]     (paths relative to the repo; missing files are skipped)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -eq 0 ] && set -- dsen2_amd/libdsen2_hip.so build/lib_f16abl1.so build/lib_f16abl2.so build/lib_f16abl4.so build/lib_f16abl3.so
for round in 1 2; do
for l in "$@"; do
  [ -f "$R/$l" ] || { echo "skip $l"; continue; }
  for cfg in dsen2_20_bf16 dsen2_20_bf16x3; do
    DSEN2_HIP_LIB=$R/$l python3 $R/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --roofline-seconds 1 --sustain-seconds 0 2>/dev/null \
      | python3 -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round  %-28s %-16s first_ms %.4f  out_ms %.4f  step %.4f ms' % ('$l', '$cfg', r['roofline']['first_ms'], r['roofline']['out_ms'], r['ms_per_step']))"
  done
done
done
