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


# ---- first contact fails fast and says where (VERDICT r4 #1a) ----

def _run_py(code, env=None, timeout=120):
    import subprocess
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    e.update(env or {})
    return subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout, env=e)


def test_guarded_step_names_a_failing_step_and_exits_non_zero():
    p = _run_py("from dsen2_amd import dist\n"
                "with dist.guarded_step('unit step', 30):\n"
                "    raise OSError('no such peer')\n"
                "print('not reached')\n", env={'RANK': '3', 'WORLD_SIZE': '8'})
    assert p.returncode == 70 and 'not reached' not in p.stdout
    assert "rank 3/8" in p.stderr and "step 'unit step' FAILED" in p.stderr and 'no such peer' in p.stderr


def test_guarded_step_ends_a_hung_step_at_its_limit():
    import time
    t0 = time.time()
    p = _run_py("import time\nfrom dsen2_amd import dist\n"
                "with dist.guarded_step('sleepy step', 2):\n"
                "    time.sleep(600)\n")
    assert p.returncode == 71 and time.time() - t0 < 60
    assert "step 'sleepy step' did not finish within 2 s" in p.stderr


def test_guarded_step_has_a_watchdog_that_needs_no_gil():
    """A step hung inside a C call that keeps the GIL never lets the timer thread run: faulthandler's own thread ends the
    process (exit 1) with every thread's stack, 10 s after the limit."""
    import time
    t0 = time.time()
    p = _run_py("import ctypes\nfrom dsen2_amd import dist\n"
                "libc = ctypes.PyDLL(None)\n"                       # PyDLL: calls keep the GIL
                "with dist.guarded_step('gil-holding step', 1):\n"
                "    libc.sleep(600)\n")
    assert p.returncode == 1 and time.time() - t0 < 90
    assert 'Timeout' in p.stderr and "-> step 'gil-holding step' (limit 1 s)" in p.stderr


def test_a_passing_step_leaves_nothing_armed():
    p = _run_py("import time\nfrom dsen2_amd import dist\n"
                "with dist.guarded_step('quick', 1):\n    pass\n"
                "time.sleep(13)\nprint('alive', dist.first_contact()['seconds']['quick'] < 1)\n")
    assert p.returncode == 0 and 'alive True' in p.stdout


_CONNECT = r'''
import os, sys
from dsen2_amd import dist
rank, _, world = dist.launched_world()
dist.connect('gloo', rank, world)
fc = dist.first_contact()
assert fc['ranks_in_collective'] == world and fc['backend'] == 'gloo'
assert set(fc['seconds']) == {'init_process_group(gloo)', 'first all_reduce'}
print('connected', rank, fc['ranks_in_collective'])
dist.finalize()
'''


def test_connect_counts_the_ranks_in_the_first_collective():
    import subprocess
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, '-c', _CONNECT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(r),
                                       WORLD_SIZE='3', DSEN2_DIST_TIMEOUT='60')) for r in range(3)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        assert 'connected %d 3' % r in out


def test_connect_gives_up_inside_the_timeout_when_a_rank_never_arrives():
    """WORLD_SIZE says 2, only rank 0 starts: it must not sit in the rendezvous for torch's default 10 minutes — it names
    the step and exits non-zero within DSEN2_DIST_TIMEOUT (+ the watchdog's 10 s)."""
    import time
    t0 = time.time()
    p = _run_py(_CONNECT, env={'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port()), 'RANK': '0', 'WORLD_SIZE': '2',
                               'DSEN2_DIST_TIMEOUT': '5'}, timeout=120)
    assert p.returncode in (70, 71, 1) and time.time() - t0 < 60
    assert "init_process_group(gloo)" in p.stderr and 'connected' not in p.stdout


# ---- C2 in pieces (DSEN2_CHUNKED_GATHER) ----

def test_chunk_bounds_cover_the_shard_once():
    from dsen2_amd import dist
    for per in (0, 1, 2, 7, 8, 9, 154, 1226):
        for chunks in (1, 2, 3, 8, 50):
            b = dist.chunk_bounds(per, chunks)
            assert len(b) <= chunks and [x for c0, c1 in b for x in range(c0, c1)] == list(range(per))
            assert all(c1 - c0 == b[0][1] - b[0][0] for c0, c1 in b[:-1])


def test_final_row_runs_release_every_row_exactly_once_and_only_when_its_patches_are_there():
    """Against the definition: image row y is decided by tile row (y >= H - inner ? last : y // inner), patches.py:394-403."""
    from dsen2_amd import patches
    rng = np.random.default_rng(5)
    for H, W, inner in ((600, 600, 112), (570, 333, 112), (10980, 10980, 112), (1008, 504, 168), (112, 300, 112), (113, 112 * 3, 112)):
        x_tiles, y_tiles = -(-W // inner), -(-H // inner)
        n = x_tiles * y_tiles
        order = rng.permutation(n)
        have, done = np.zeros(n, bool), np.zeros(y_tiles, bool)
        owner = np.where(np.arange(H) >= H - inner, y_tiles - 1, np.arange(H) // inner)
        seen = np.zeros(H, int)
        for cut in np.array_split(order, 7):
            have[cut] = True
            for r0, r1 in patches.final_row_runs(have, done, (H, W), inner):
                assert 0 <= r0 < r1 <= H
                assert have.reshape(y_tiles, x_tiles)[owner[r0:r1]].all()
                seen[r0:r1] += 1
        assert (seen == 1).all() and done.all()


def _chunked_worker(rank, world, port, total, chunks, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dsen2_amd import dist
        first, count = dist.shard_range(total)
        per = dist.per_rank(total, world)
        g = torch.Generator().manual_seed(100 + rank)
        send = torch.rand((per, 2, 3, 3), generator=g)
        whole = dist.gather_to_root(send.clone(), total)               # the one-shot form: what the pieces must add up to
        cg = dist.ChunkedGather(send, total, chunks)
        ok = cg.n_chunks == len(dist.chunk_bounds(per, chunks))
        # a rank issues a piece once it has "written" it; ranks run at different speeds — the order is what matters
        for c in range(cg.n_chunks):
            cg.issue(c)
        arrived = 0
        for c in range(cg.n_chunks):
            c0, c1 = cg.complete(c)
            if rank == 0:
                arrived = cg.slots_done(c)
                for r in range(world):      # everything up to this piece is in place for every rank, bit for bit
                    lo, hi = r * per, min(total, r * per + arrived)
                    ok = ok and (hi <= lo or torch.equal(cg.recv[lo:hi], whole[lo:hi]))
        if rank == 0:
            ok = ok and arrived == per and torch.equal(cg.recv[:total], whole)
        else:
            ok = ok and cg.recv is None and whole is None
        q.put((rank, bool(ok)))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('world,total,chunks', [(2, 9, 3), (2, 1, 8), (3, 7, 2), (3, 2, 8), (4, 36, 8), (8, 13, 4), (2, 16, 1)])
def test_chunked_gather_delivers_what_the_single_gather_delivers(world, total, chunks):
    """Uneven shards, a rank with nothing ((2, 1), (3, 2), (8, 13)), more pieces asked for than slots, one piece."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunked_worker, args=(r, world, port, total, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {r: True for r in range(world)}
