"""RCCL with more than one rank — runs on any box that shows at least two GPUs and skips itself on the one-GPU boxes the
builder has (VERDICT r4 #1c).  On such a box these are the first N > 1 hardware runs of the product: bench.py's weak-scaling
step over backend "nccl" (= RCCL over xGMI) and the patch-sharded full tile against the single-rank image with both forms
of the gather.  (The same box also un-skips test_gpu_forward.py::test_a_handle_belongs_to_the_device_it_was_created_on.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two visible GPUs (RCCL with N > 1)')]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _launch(n, *args, **kw):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', _free_port()] + list(args)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, **kw)


def _json_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_ranks_over_rccl():
    r = _json_line(_launch(2, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '3',
                           '--roofline-seconds', '0.5', '--sustain-seconds', '0.5', cwd=ROOT))
    assert r['n_gpus'] == 2 and r['config']['backend'] == 'rccl' and r['config']['output_gather'] is True
    assert r['ranks_in_collective'] == 2 and len(r['per_rank_ms_per_step']['ranks']) == 2
    # What this test asserts is that the path WORKS and reports; how fast it is belongs to the record, not to a gate that a
    # first contact with new hardware could trip for reasons outside the code (the expectation table is DESIGN §6): printed
    # for the log, and only a catastrophic slowdown (the gather serialising whole steps) fails.
    print('N=2 over RCCL: %s patches/s, %.3f ms/step (no gather: %.3f), per rank %s, gather wait %s'
          % (r['value'], r['ms_per_step'], r['ms_per_step_no_gather'], r['per_rank_ms_per_step'], r['gather_wait_ms_per_step']))
    assert r['ms_per_step'] < 2.0 * r['ms_per_step_no_gather'], r
    assert r['value'] > 512 / (r['ms_per_step_no_gather'] * 1e-3), r          # two GPUs beat one


def test_full_tile_two_ranks_over_rccl_equals_the_single_rank_image():
    r = _json_line(_launch(2, os.path.join(ROOT, 'tools', 'bench_full_tile.py'), '--size', '10980', '--skip60', '--backend',
                           'nccl', '--check', cwd=ROOT))
    assert r['n_gpus'] == 2 and r['patches20'] == 9801 and r['matches_single_rank'] is True


def test_full_tile_chunked_gather_over_rccl_equals_the_single_rank_image():
    env = dict(os.environ, DSEN2_CHUNKED_GATHER='1')
    r = _json_line(_launch(2, os.path.join(ROOT, 'tools', 'bench_full_tile.py'), '--size', '10980', '--skip60', '--backend',
                           'nccl', '--check', cwd=ROOT, env=env))
    assert r['n_gpus'] == 2 and r['matches_single_rank'] is True and r.get('chunked_gather') is True
