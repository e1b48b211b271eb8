#!/bin/bash
# Run on the GPU box (via gpurun): BASELINE configs[3]'s geometry on one GPU — tools/bench_full_tile.py in fp32 and bf16,
# then rocprofv3 --kernel-trace --stats of the fp32 run (body kernel at the REAL patch sizes 128^2 / 192^2).
#   tools/profile_full_tile.sh <tag> [size=10980]
set -u
TAG=${1:-r03}; SIZE=${2:-10980}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_full_tile.py --size $SIZE > $OUT/full_tile_fp32.json 2> $OUT/full_tile_fp32.err; echo "fp32 exit=$?"
python3 $R/tools/bench_full_tile.py --size $SIZE --precision bf16 > $OUT/full_tile_bf16.json 2> $OUT/full_tile_bf16.err; echo "bf16 exit=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full_tile -- python3 $R/tools/bench_full_tile.py --size $SIZE > $OUT/trace_full_tile.log 2>&1; echo "trace exit=$?"
cd $R
python3 tools/summarize_rocprof.py $OUT/trace_full_tile $OUT/full_tile_fp32_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/bench_full_tile.py --size $SIZE" > /dev/null
cat $OUT/full_tile_fp32.json $OUT/full_tile_bf16.json
