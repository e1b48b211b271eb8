#!/bin/bash
# Timing-only ablations of the matrix-core output convolution (conv3x3_out_mfma.hip; diagnostic library, key 6), run via
# gpurun: 1 no second stage, 2 no fetches, 4 no MFMAs, 8 no dx sum / Q writes, 16 no barriers.  Outputs are WRONG.
#   tools/ablate_out_conv.sh [mask ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export DSEN2_HIP_LIB=$R/build/libdsen2_hip_diag.so
for m in ${@:-0 1 2 4 8 16 6 14 15 31}; do
  export DSEN2_DIAG_SET=6=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/oa_$m -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/oa_$m.log 2>&1
  python3 $R/tools/summarize_rocprof.py $R/gpurun_out/oa_$m $R/gpurun_out/oa_$m.md "mask $m" > /dev/null
  echo "== mask $m"; grep "conv3x3_out" $R/gpurun_out/oa_$m.md | head -1 | cut -c1-150
done
