#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the tiling / up-sampling / recomposition kernels.
#   bash tools/profile_patch_ops.sh <tag>          -> gpurun_out/<tag>/{kernel_stats.md,pmc.md}
# Every rocprofv3 invocation puts python3 itself after `--` and uses --pmc only together with --kernel-trace.
set -u
TAG=${1:?usage: tools/profile_patch_ops.sh <tag>}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_patch_ops.py > $OUT/trace.log 2>&1; echo "trace exit=$?"
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/tools/bench_patch_ops.py > $OUT/pmc_$n.log 2>&1; echo "$n exit=$?"
done
cd $R
python3 tools/summarize_rocprof.py $OUT/trace $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/bench_patch_ops.py" > /dev/null
python3 tools/summarize_pmc.py $OUT/pmc.md $OUT/pmc_* > /dev/null
grep -E "upsample|gather|recompose" $OUT/kernel_stats.md | head -8 | cut -c1-160
grep -E "^\| kernel|upsample|gather|recompose" $OUT/pmc.md | cut -c1-330
