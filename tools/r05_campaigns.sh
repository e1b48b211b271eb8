#!/bin/bash
# Round 5's measurement campaigns, one gpurun call each (what produced profiles/r05_b_*, r05_d_*, r05_e_*, r05_scaling_projection.json):
#   tools/r05_campaigns.sh b                      headline: profile_round.sh r05_b dsen2_20_fp32 + project_scaling.py
#   tools/r05_campaigns.sh d                      configs[4]: profile_round.sh r05_d vdsen2_20_bf16 + the full tile in three precisions
#   tools/r05_campaigns.sh e <config> [...]       profile_round.sh r05_e_<config> <config> for each config given
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
case ${1:-} in
  b) bash tools/profile_round.sh r05_b dsen2_20_fp32 > gpurun_out/r05_b_log.txt 2>&1; echo "fp32 campaign exit $?"
     python3 tools/project_scaling.py > gpurun_out/r05_scaling_projection.json 2> gpurun_out/r05_scaling_projection.err; echo "projection exit $?"
     tail -c 600 gpurun_out/r05_scaling_projection.json ;;
  d) bash tools/profile_round.sh r05_d vdsen2_20_bf16 > gpurun_out/r05_d_log.txt 2>&1; echo "vdsen2 bf16 campaign exit $?"
     for p in fp32 bf16x3 bf16; do
       python3 tools/bench_full_tile.py --precision $p --repeat 2 > gpurun_out/r05_d_full_tile_$p.json 2> gpurun_out/r05_d_full_tile_$p.err; echo "full tile $p exit $?"
       tail -c 400 gpurun_out/r05_d_full_tile_$p.json
     done ;;
  e) shift
     for c in "$@"; do
       bash tools/profile_round.sh r05_e_$c $c > gpurun_out/r05_e_${c}_log.txt 2>&1; echo "$c campaign exit $?"; tail -2 gpurun_out/r05_e_${c}_log.txt | cut -c1-200
     done ;;
  *) echo "usage: $0 b | d | e <config> ..."; exit 2 ;;
esac
