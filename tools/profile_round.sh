#!/bin/bash
# Run on the GPU box (via gpurun): bench, rocprofv3 kernel trace, PMC passes for one bench config.
#   tools/profile_round.sh <tag> [config]          (config: dsen2_20_fp32 (default), vdsen2_20_bf16, ...)
# Every rocprofv3 invocation puts python3 itself after `--` and uses --pmc only together with --kernel-trace.
set -u
TAG=${1:-r02}; CFG=${2:-dsen2_20_fp32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config $CFG > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --other-seconds 0 > $OUT/trace.log 2>&1; echo "trace exit=$?"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" GRBM_GUI_ACTIVE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --roofline-seconds 1 --sustain-seconds 0 --other-seconds 0 > $OUT/pmc_$n.log 2>&1; echo "$n exit=$?"
done
cd $R
python3 tools/summarize_rocprof.py $OUT/trace $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --other-seconds 0 (the driver's command without the CPU leg and without the other_configs legs)" > /dev/null
python3 tools/summarize_pmc.py $OUT/pmc.md $OUT/pmc_* > /dev/null
# the traffic figure bench.py quotes, together with the hash of the ISA it was just measured on (copy over profiles/body_conv_traffic.json)
case $CFG in
  dsen2_20_fp32|vdsen2_20_bf16|dsen2_20_bf16x3|vdsen2_20_fp32|dsen2_20_bf16|vdsen2_20_bf16x3)
    python3 tools/update_traffic_json.py $CFG $OUT/pmc.md "profiles/${TAG}_pmc.md = gpurun_out/$TAG/pmc.md (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of tools/profile_round.sh $TAG $CFG; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM for 16-B/lane reads)" $OUT/body_conv_traffic.json ;;
esac
cat $OUT/bench.json
