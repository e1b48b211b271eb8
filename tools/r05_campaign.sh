set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile_round.sh r05_b dsen2_20_fp32 > gpurun_out/r05_b_log.txt 2>&1; echo "fp32 campaign exit $?"
python3 tools/project_scaling.py > gpurun_out/r05_scaling_projection.json 2> gpurun_out/r05_scaling_projection.err; echo "projection exit $?"
tail -c 600 gpurun_out/r05_scaling_projection.json
