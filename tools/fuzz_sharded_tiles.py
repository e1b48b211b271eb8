#!/usr/bin/env python3
"""Randomised cases of the patch-sharded full-tile path on ONE GPU (gloo rehearsal): tile size, number of ranks (2-4), gather form
(one-shot / chunked with a random number of pieces), with and without DSen2_60 — every case must give the single-rank image bit
for bit (tools/bench_full_tile.py --check).  A screen to run after touching dist.ChunkedGather / supres._run; the fixed cases live
in tests/test_gpu_bench_rehearsal.py.
    python tools/fuzz_sharded_tiles.py [--cases 12] [--seed 0]"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--cases', type=int, default=12)
ap.add_argument('--seed', type=int, default=0)
args = ap.parse_args()
rng = random.Random(args.seed)
bad = 0
for i in range(args.cases):
    size = 6 * rng.randint(40, 260)                  # 240 ... 1560: 9 ... 196 patches of 128 (DSen2_20), 4 ... 100 of 192 (DSen2_60)
    ranks = rng.randint(2, 4)
    chunked = rng.random() < 0.7
    chunks = rng.choice([1, 2, 3, 5, 8, 13, 50])
    skip60 = rng.random() < 0.5 or size < 384        # DSen2_60 needs at least one 192-patch of image
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DSEN2_CHUNKED_GATHER='1' if chunked else '0', DSEN2_GATHER_CHUNKS=str(chunks))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tools', 'bench_full_tile.py'), '--size', str(size), '--backend', 'gloo',
           '--check'] + (['--skip60'] if skip60 else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    ok = False
    if p.returncode == 0:
        r = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
        ok = r.get('matches_single_rank') is True and r.get('chunked_gather') is chunked
    bad += int(not ok)
    print('case %2d: size %4d ranks %d %s skip60=%s -> %s' % (i, size, ranks, 'chunked/%d' % chunks if chunked else 'one-shot', skip60,
                                                               'ok' if ok else 'FAILED rc=%d %s' % (p.returncode, p.stderr[-300:])), flush=True)
print('fuzz_sharded_tiles: %d cases, %d failed' % (args.cases, bad))
sys.exit(1 if bad else 0)
