#!/bin/bash
# ONE command for the first box with more than one GPU (no N > 1 hardware run exists yet): pre-flight, the weak-scaling
# bench at every N with and without DSEN2_RCCL_HIGH_PRIORITY, the full tile with the one-shot and the chunked gather.
#   tools/first_contact_ab.sh [N ...]            (default: 2 4 8, capped at the GPUs visible)
#   BACKEND=gloo tools/first_contact_ab.sh 2     (rehearsal of the script itself on a one-GPU box: the numbers mean nothing)
# Everything goes to gpurun_out/first_contact/ ; summary.txt is the file to read.  Every run is bounded by `timeout`, and a run
# that fails does not stop the next one (each is a fresh set of processes: nothing is retried inside a process).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=${BACKEND:-nccl}
OUT=$R/gpurun_out/first_contact
mkdir -p $OUT
GPUS=$(python3 -c "import torch; print(torch.cuda.device_count())")
[ $# -eq 0 ] && set -- 2 4 8
PORT=29700
launch() { PORT=$((PORT + 1)); n=$1; shift; timeout -k 10 ${RUN_LIMIT:-420} python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT "$@"; }
line() { python3 -c "
import sys, json
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
except Exception as e:
    print('   NO JSON LINE (%s) — see %s and its .err' % (e, sys.argv[1])); sys.exit(0)
keys = sys.argv[2:]
print('   ' + '  '.join('%s=%s' % (k, r.get(k)) for k in keys))" "$@"; }
{
echo "first contact A/B — backend $B, $GPUS GPU(s) visible, $(date -u +%FT%TZ)"
python3 $R/bench.py --no-cpu-baseline --other-seconds 0 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench N=1:"; line $OUT/bench_n1.json value ms_per_step
for n in "$@"; do
  if [ "$B" = nccl ] && [ $n -gt $GPUS ]; then echo "N=$n: skipped ($GPUS GPU(s) visible)"; continue; fi
  echo "== N=$n"
  launch $n -m dsen2_amd.dist --backend $B > $OUT/preflight_n$n.json 2> $OUT/preflight_n$n.err; echo " pre-flight rc=$?"
  line $OUT/preflight_n$n.json world gather_12p6MB_ms gather_GBps_per_peer_link chunked_gather_s chunked_gather_GBps_per_peer_link gather_payload_ok chunked_payload_ok
  for hp in 0 1; do
    DSEN2_RCCL_HIGH_PRIORITY=$hp launch $n $R/bench.py --gpus $n --backend $B > $OUT/bench_n${n}_hp$hp.json 2> $OUT/bench_n${n}_hp$hp.err; echo " bench high_priority=$hp rc=$?"
    line $OUT/bench_n${n}_hp$hp.json value ms_per_step ms_per_step_no_gather value_no_gather ranks_in_collective per_rank_ms_per_step gather_wait_ms_per_step
  done
  for cg in 0 1; do
    DSEN2_CHUNKED_GATHER=$cg launch $n $R/tools/bench_full_tile.py --skip60 --backend $B --repeat 2 > $OUT/full_tile_n${n}_chunked$cg.json 2> $OUT/full_tile_n${n}_chunked$cg.err; echo " full tile chunked=$cg rc=$?"
    line $OUT/full_tile_n${n}_chunked$cg.json n_gpus chunked_gather dsen2_20_s dsen2_20_s_runs dsen2_20_equiv_32x32_patches_per_s
  done
done
} 2>&1 | tee $OUT/summary.txt
