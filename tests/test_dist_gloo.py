"""The N>1 path on CPU: world_size-2 and -3 gloo processes exercise the sharding arithmetic, the weight
broadcast (C1) and the gather of outputs to rank 0 (C2) of dsen2_amd/dist.py, including uneven shards and a rank
with nothing to do."""
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
        # C2: each rank "predicts" its shard with a stand-in computation (patch index encoded in the data) straight
        # into the [per, ...] buffer the gather sends; only rank 0 receives
        first, count = dist.shard_range(total)
        per = dist.per_rank(total, world)
        send = torch.full((per, 2, 4, 4), -1.0)
        for i in range(count):
            send[i] = float(first + i)
        full = dist.gather_to_root(send, total)
        if rank == 0:
            ok = ok and full is not None and full.shape == (total, 2, 4, 4)
            ok = ok and all(float(full[i, 0, 0, 0]) == float(i) and float(full[i, 1, 3, 3]) == float(i) for i in range(total))
        else:
            ok = ok and full is None
        q.put((rank, bool(ok)))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('world,total', [(2, 9), (2, 2), (2, 1), (3, 7), (3, 2), (4, 9), (8, 13)])
def test_broadcast_and_gather_to_root(world, total):
    """(3, 7): uneven shards 3 + 3 + 1; (3, 2) and (2, 1): the last rank has no patch at all; (8, 13): the node size the
    north star names — six ranks with two patches, one with one, one with none."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {r: True for r in range(world)}


def _c1_worker(rank, world, port, root_dir, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dsen2_amd import dist, weights
        # every rank has its OWN model directory (like supres.MDL_PATH on ranks with different working directories);
        # only rank 0's holds the file
        mine = os.path.join(root_dir, 'rank%d' % rank)
        got = dist.load_weights_on_root(os.path.join(mine, 's2_032_lr_1e-04.hdf5'), 10, 6, 1, 128)
        want = weights.random_he_uniform(10, 6, 1, 128, seed=21)
        ok = got.dtype == np.float32 and np.array_equal(got, want)
        # a checkpoint nobody has: the SAME failure on every rank (nobody left waiting in the broadcast), OSError like keras
        try:
            dist.load_weights_on_root(os.path.join(mine, 's2_030_lr_1e-05.hdf5'), 12, 2, 1, 128)
            ok = False
        except OSError as e:
            ok = ok and ('s2_030' in str(e))
        # ... and the group still works afterwards
        t = torch.tensor([float(rank)])
        td.all_reduce(t)
        ok = ok and float(t.item()) == float(sum(range(world)))
        q.put((rank, bool(ok)))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_only_rank_0_needs_the_checkpoint(world, tmp_path):
    """C1 in the product path (supres._get_model -> dist.load_weights_on_root): rank 0 reads the file, the others receive
    the flat vector by broadcast although THEIR model directory is empty; a missing file fails on all ranks alike."""
    from dsen2_amd import weights
    for r in range(world):
        os.makedirs(str(tmp_path / ('rank%d' % r)))
    np.save(str(tmp_path / 'rank0' / 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 1, 128, seed=21))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c1_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {r: True for r in range(world)}


def test_launch_environment_is_read_like_torch_distributed_run_sets_it(monkeypatch):
    """dist.launched_world(): RANK / LOCAL_RANK / WORLD_SIZE of a torch.distributed.run launch; a plain start is
    (0, 0, 1).  init_from_env() refuses an unknown backend before touching any device."""
    from dsen2_amd import dist
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    assert dist.launched_world() == (0, 0, 1)
    monkeypatch.setenv('RANK', '5'); monkeypatch.setenv('LOCAL_RANK', '1'); monkeypatch.setenv('WORLD_SIZE', '8')
    assert dist.launched_world() == (5, 1, 8)
    with pytest.raises(ValueError):
        dist.init_from_env('mpi')
    if not torch.cuda.is_available():               # the product has no CPU fallback: the N-GPU entry says so
        monkeypatch.setenv('WORLD_SIZE', '1')
        with pytest.raises(RuntimeError):
            dist.init_from_env('gloo')
