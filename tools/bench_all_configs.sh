#!/bin/bash
# Every bench config on one box, one after the other (DESIGN §5's table): gpurun_out/<tag>_bench_<config>.json
#   tools/bench_all_configs.sh <tag>
set -u
TAG=${1:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for c in dsen2_20_fp32 dsen2_60_fp32 vdsen2_20_fp32 dsen2_20_bf16 vdsen2_20_bf16 dsen2_20_bf16x3 vdsen2_20_bf16x3; do
  python3 $R/bench.py --config $c --other-seconds 0 > $R/gpurun_out/${TAG}_bench_$c.json 2> $R/gpurun_out/${TAG}_bench_$c.err
  python3 -c "import json,sys; r=json.loads(open('$R/gpurun_out/${TAG}_bench_$c.json').read().strip().splitlines()[-1]); rl=r['roofline']; print('%-18s %10.1f patches/s  %8.4f ms  frac %.4f  first %.4f  out %.4f  launches %d x %.4f ms' % ('$c', r['value'], r['ms_per_step'], rl['frac'], rl['first_ms'], rl['out_ms'], rl['launches_per_forward'], rl['ms_per_launch']))"
done
