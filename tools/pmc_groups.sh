#!/bin/bash
# PMC passes (one per ';'-separated counter group) over a python script.  Run via gpurun.
#   tools/pmc_groups.sh <tag> "<group1>;<group2>;..." <script.py> [script args]
set -u
TAG=$1; GROUPS_=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GS <<< "$GROUPS_"
i=0
for c in "${GS[@]}"; do
  i=$((i+1))
  # a counter group the hardware cannot collect makes rocprofv3 abort and then hang: bound every pass
  timeout -k 5 ${PMC_PASS_TIMEOUT:-180} rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$i -- python3 $R/"$@" > $OUT/pmc_$i.log 2>&1; echo "group $i ($c) exit=$?"
done
python3 $R/tools/summarize_pmc.py $OUT/pmc.md $OUT/pmc_* > /dev/null; cat $OUT/pmc.md
