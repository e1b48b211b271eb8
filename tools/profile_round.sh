#!/bin/bash
# Run on the GPU box (via gpurun): bench, rocprofv3 kernel trace, PMC passes. Usage: tools/profile_round.sh <tag>
# Every rocprofv3 invocation puts python3 itself after `--` and uses --pmc only together with --kernel-trace.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace exit=$?"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" GRBM_GUI_ACTIVE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_$n.log 2>&1; echo "$n exit=$?"
done
cat $OUT/bench.json
