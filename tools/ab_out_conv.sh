#!/bin/bash
# Output-layer kernels side by side on one box (run via gpurun): rocprofv3 kernel-trace averages of the bench config with
# the diagnostic library, output variant 2 (conv3x3_out_mfma.hip where it fits) and 3 (conv3x3_out.hip always).
#   tools/ab_out_conv.sh [config ...]        default: dsen2_20_fp32 dsen2_60_fp32
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export DSEN2_HIP_LIB=$R/build/libdsen2_hip_diag.so
for cfg in ${@:-dsen2_20_fp32 dsen2_60_fp32}; do
  for v in 2 3; do
    export DSEN2_DIAG_SET=2=$v
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/oc_${cfg}_$v -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/oc_${cfg}_$v.log 2>&1
    python3 $R/tools/summarize_rocprof.py $R/gpurun_out/oc_${cfg}_$v $R/gpurun_out/oc_${cfg}_$v.md "$cfg out_variant $v" > /dev/null
    echo "== $cfg out_variant $v"; grep "conv3x3_out" $R/gpurun_out/oc_${cfg}_$v.md | cut -c1-170
    tail -1 $R/gpurun_out/oc_${cfg}_$v.log | cut -c1-200
  done
done
