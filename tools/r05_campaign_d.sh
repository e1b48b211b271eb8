set -u
bash tools/profile_round.sh r05_d vdsen2_20_bf16 > gpurun_out/r05_d_log.txt 2>&1; echo "vdsen2 bf16 campaign exit $?"
for p in fp32 bf16x3 bf16; do python3 tools/bench_full_tile.py --precision $p --repeat 2 > gpurun_out/r05_d_full_tile_$p.json 2> gpurun_out/r05_d_full_tile_$p.err; echo "full tile $p exit $?"; tail -c 400 gpurun_out/r05_d_full_tile_$p.json; done
