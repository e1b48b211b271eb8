#!/bin/bash
# PMC passes over tools/ab_bf16.py (run on the GPU box via gpurun). Usage: tools/pmc_ab_bf16.sh <tag> <variants> [batch]
set -u
TAG=${1:-pmc16}; VAR=${2:-2,4}; export AB_BATCH=${3:-1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in GRBM_GUI_ACTIVE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/tools/ab_bf16.py $VAR > $OUT/pmc_$n.log 2>&1; echo "$n exit=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/ab_bf16.py $VAR > $OUT/trace.log 2>&1; echo "trace exit=$?"
python3 $R/tools/summarize_pmc.py $OUT/pmc.md $OUT/pmc_* > /dev/null; cat $OUT/pmc.md
tail -4 $OUT/trace.log
