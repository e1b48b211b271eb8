"""The N>1 path on CPU: world_size-2 gloo processes exercise the sharding arithmetic, the weight
broadcast (C1) and the output all-gather (C2) of dsen2_amd/dist.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_every_count():
    from dsen2_amd import dist
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 7, 8, 9, 36, 512, 9801):
            spans = [dist.shard_range(n, r, world) for r in range(world)]
            covered = []
            for first, count in spans:
                covered += list(range(first, first + count))
            assert covered == list(range(n)), (world, n)
            assert max(c for _, c in spans) == dist.per_rank(n, world)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dsen2_amd import dist
        # C1: only rank 0 holds the weights
        flat = np.arange(1000, dtype=np.float32) * 0.5 if rank == 0 else None
        got = dist.broadcast_weights(flat, 1000)
        ok = np.array_equal(got, np.arange(1000, dtype=np.float32) * 0.5)
        # C2: each rank "predicts" its shard with a stand-in computation (patch index encoded in the data)
        first, count = dist.shard_range(total)
        local = torch.stack([torch.full((2, 4, 4), float(first + i)) for i in range(count)]) if count else \
            torch.zeros((0, 2, 4, 4))
        full = dist.gather_patches(local, total)
        ok = ok and full.shape == (total, 2, 4, 4)
        ok = ok and all(float(full[i, 0, 0, 0]) == float(i) and float(full[i, 1, 3, 3]) == float(i) for i in range(total))
        q.put((rank, bool(ok)))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('total', [9, 2, 1])
def test_broadcast_and_gather_world2(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}
