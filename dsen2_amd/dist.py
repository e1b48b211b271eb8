"""Patch sharding across the GPUs of one node: one process per GPU, torch.distributed over RCCL.

The forward pass has no exchange step (patches are independent), so there are exactly two
collectives and neither is on the per-layer path:
  C1  broadcast_weights : rank 0 reads the checkpoint once, everyone receives the flat float32 vector
                          (7.16 MB DSen2, 151 MB VDSen2) with one broadcast
  C2  gather_to_root    : each rank's (inner-cropped) predictions go to rank 0 with ONE gather into views of one
                          pre-sized buffer (the "gather of outputs over xGMI"); no other rank receives anything
Partitioning: contiguous ranges of the row-major patch index, ceil(N/R) per rank (last ranks may be
short or empty).  With backend "nccl" (= RCCL on ROCm) tensors stay on the GPU; with "gloo" (CPU tests)
they are staged through host memory.
"""
import contextlib
import datetime
import faulthandler
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as td

BACKENDS = ('nccl', 'gloo')
# First contact with RCCL must fail FAST and say where: the process-group timeout (torch's default is 10 minutes — longer
# than the 600 s a driver gives the whole bench) and the limit of every guarded set-up step.  DSEN2_DIST_TIMEOUT overrides.
# It stays the group's timeout for every later collective too: a rank that arrives at a gather more than this long after
# the first one (a very slow reader of its rows, say) needs a larger value.
DIST_TIMEOUT_DEFAULT_S = 120.0


def timeout_s():
    return float(os.environ.get('DSEN2_DIST_TIMEOUT', DIST_TIMEOUT_DEFAULT_S))


EXIT_STEP_FAILED = 70          # a guarded step raised
EXIT_STEP_HUNG = 71            # a guarded step did not return within its limit
_first_contact = {}            # what init_from_env saw: ranks_in_collective, seconds per set-up step


def _say(msg):
    rank, _, world = launched_world()
    sys.stderr.write('dsen2_amd.dist[rank %d/%d pid %d]: %s\n' % (rank, world, os.getpid(), msg))
    sys.stderr.flush()


@contextlib.contextmanager
def guarded_step(name, limit_s=None):
    """Run one multi-process set-up step (rendezvous, RCCL communicator set-up, the first collectives) so that it either
    returns, or this rank says WHICH step failed and leaves with a non-zero code inside `limit_s` — torch.distributed.run then
    ends the other ranks.  Never a retry: a process that has touched the GPU is not re-used (a fresh start is the only
    retry this pool tolerates).
      * the step raises            -> one line naming the step and the error, exit EXIT_STEP_FAILED
      * the step hangs             -> a timer thread prints the step and exits EXIT_STEP_HUNG at `limit_s`; if that thread
                                      cannot run (the hung call holds the GIL) faulthandler's GIL-free watchdog dumps every
                                      thread's stack and exits 10 s later
    Only used where nothing is worth saving (before any result exists)."""
    limit_s = timeout_s() if limit_s is None else float(limit_s)
    t0 = time.perf_counter()
    # said up front, so that the GIL-free watchdog's bare stack dump can be read against it
    _say("-> step '%s' (limit %.0f s)" % (name, limit_s))

    def _expired():
        _say("step '%s' did not finish within %.0f s — giving up (exit %d); a hang here is RCCL / rendezvous set-up, not the "
             'kernels: check HSA_ENABLE_IPC_MODE_LEGACY=0, MASTER_ADDR=127.0.0.1, one visible GPU per rank'
             % (name, limit_s, EXIT_STEP_HUNG))
        os._exit(EXIT_STEP_HUNG)
    timer = threading.Timer(limit_s, _expired)
    timer.daemon = True
    timer.start()
    faulthandler.dump_traceback_later(limit_s + 10.0, exit=True)
    try:
        yield
    except SystemExit:
        raise
    except BaseException as e:           # noqa: BLE001 — reported, then the process ends
        _say("step '%s' FAILED after %.1f s: %s: %s (exit %d)" % (name, time.perf_counter() - t0, type(e).__name__, e,
                                                                 EXIT_STEP_FAILED))
        sys.stderr.flush()
        os._exit(EXIT_STEP_FAILED)
    finally:
        timer.cancel()
        faulthandler.cancel_dump_traceback_later()
    _first_contact.setdefault('seconds', {})[name] = round(time.perf_counter() - t0, 3)


def first_contact():
    """What init_from_env() established at N > 1 (a copy): `ranks_in_collective` = an all-reduce of 1 over the new group
    (proof that the backend's collectives saw every rank), `backend`, and the seconds each guarded set-up step took.
    Empty for a single process."""
    return dict(_first_contact)


def launched_world():
    """(rank, local_rank, world) of a `python -m torch.distributed.run` launch (RANK / LOCAL_RANK / WORLD_SIZE in the
    environment); (0, 0, 1) for a plain `python` start."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init_from_env(backend='nccl'):
    """The N-GPU entry of every front end (dsen2_amd.cli, bench.py, tools/bench_full_tile.py): one process per GPU,
    launched by torch.distributed.run.  Selects this rank's GPU (LOCAL_RANK) and, when WORLD_SIZE > 1, initialises
    torch.distributed — backend 'nccl' = RCCL over xGMI; 'gloo' is a rehearsal of the control flow on a box with fewer
    GPUs than ranks (ranks share devices, collectives are staged through host memory).  Returns (rank, world, device).

    Must run before anything initialises HIP: RCCL's intra-node transport exchanges buffer handles between the rank
    processes, and this driver stack only supports the dmabuf form of that (HSA_ENABLE_IPC_MODE_LEGACY=0; with the
    legacy mode hipIpcGetMemHandle fails with "invalid argument").  The runtime reads the variable once, when it
    starts, so it is set here — not by the user, not at import time — and only for a multi-process run."""
    if backend not in BACKENDS:
        raise ValueError('backend %r (one of %s)' % (backend, ', '.join(BACKENDS)))
    rank, local_rank, world = launched_world()
    if world > 1:
        if 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ:
            if torch.cuda.is_initialized():
                sys.stderr.write('dsen2_amd.dist: HIP was initialised before init_from_env(); '
                                 'HSA_ENABLE_IPC_MODE_LEGACY=0 can no longer take effect\n')
            os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    n_dev = torch.cuda.device_count()            # counting devices does not initialise HIP
    if n_dev == 0:
        raise RuntimeError('dsen2_amd needs a ROCm GPU (gfx950); there is no CPU fallback')
    pinned = n_dev == 1 and any(os.environ.get(k) for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'))
    if backend == 'nccl' and local_rank >= n_dev and not pinned:
        raise RuntimeError('LOCAL_RANK %d but %d GPU(s) visible: RCCL needs one GPU per rank '
                           '(backend gloo rehearses with shared devices)' % (local_rank, n_dev))
    # (`pinned`: a launcher that gives every rank its own *_VISIBLE_DEVICES shows each process ONE device, number 0)
    dev = torch.device('cuda', local_rank % n_dev)
    torch.cuda.set_device(dev)
    if world > 1 and not td.is_initialized():
        connect(backend, rank, world, dev)
    return rank, world, dev


def connect(backend, rank, world, dev=None):
    """Join the process group and make first contact, both under guarded_step (fail fast, say where).  `dev` is this rank's
    GPU for backend 'nccl' (RCCL binds the communicator to it); gloo needs none — which is how the CPU tests drive this."""
    timeout = datetime.timedelta(seconds=timeout_s())
    with guarded_step('init_process_group(%s)' % backend):
        if backend == 'nccl':
            # DSEN2_RCCL_HIGH_PRIORITY=1: RCCL's kernels on a high-priority stream — both body kernels fill every CU, so a
            # gather in flight is dispatched at a launch boundary; with priority it is the first thing dispatched there.  Off by
            # default: an A/B for the first N > 1 box (bench.py's gather_wait_ms_per_step / ms_per_step_no_gather show it).
            opts = None
            if os.environ.get('DSEN2_RCCL_HIGH_PRIORITY', '0') not in ('', '0'):
                opts = td.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            td.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=timeout, pg_options=opts)
        else:
            td.init_process_group('gloo', rank=rank, world_size=world, timeout=timeout)
    # the first collective is where RCCL builds its rings / exchanges IPC handles: do it HERE, guarded, with a payload
    # that also proves every rank takes part (sum of ones = world), instead of inside the first real transfer
    with guarded_step('first all_reduce'):
        cdev = dev if backend == 'nccl' else torch.device('cpu')
        ones = torch.ones(1, dtype=torch.int64, device=cdev)
        td.all_reduce(ones)
        seen = int(ones.item())              # .item() synchronises: the collective has completed
        if seen != world:
            raise RuntimeError('all_reduce of 1 over the group gave %d, WORLD_SIZE is %d' % (seen, world))
    _first_contact.update(ranks_in_collective=seen, backend=backend, timeout_s=timeout_s())


def finalize():
    """Leave the process group (if any) after a last barrier, so no rank tears RCCL down under a peer's collective."""
    if td.is_available() and td.is_initialized():
        td.barrier()
        td.destroy_process_group()


def rank_world():
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def per_rank(n, world):
    return (n + world - 1) // world if n > 0 else 0


def shard_range(n, rank=None, world=None):
    """(first, count) of this rank's contiguous share of n patches."""
    if rank is None or world is None:
        rank, world = rank_world()
    per = per_rank(n, world)
    first = min(n, rank * per)
    return first, max(0, min(n, first + per) - first)


def _collective_device(t):
    backend = td.get_backend()
    return t.device if backend == 'nccl' else torch.device('cpu')


def gather_to_root(send, total, dst=0):
    """C2.  Every rank passes `send` = a [per_rank(total, world), ...] tensor whose first shard_range(total)[1]
    rows are its results (the rest is padding that is never read: allocate the buffer at that size and fill it in
    place, no copy is made here).  Rank `dst` returns a [total, ...] tensor (a view of the one receive buffer, in
    patch order); every other rank returns None and allocates nothing."""
    rank, world = rank_world()
    if world == 1:
        return send[:total]
    per = per_rank(total, world)
    assert send.shape[0] == per, (send.shape, per)
    cdev = _collective_device(send)
    src = send if send.device == cdev else send.to(cdev)
    if rank == dst:
        recv = torch.empty((world * per,) + tuple(send.shape[1:]), dtype=send.dtype, device=cdev)
        td.gather(src, list(recv.chunk(world)), dst=dst)
        out = recv[:total]
        return out if out.device == send.device else out.to(send.device)
    td.gather(src, None, dst=dst)
    return None


def chunk_bounds(per, chunks):
    """[(c0, c1)] — `per` slots of a rank's send buffer cut into at most `chunks` contiguous pieces of equal size (the last
    may be short); the same on every rank, so that piece c of every rank is one gather."""
    if per <= 0:
        return []
    size = (per + max(1, int(chunks)) - 1) // max(1, int(chunks))
    return [(c0, min(per, c0 + size)) for c0 in range(0, per, size)]


class ChunkedGather(object):
    """C2 in pieces (DSEN2_CHUNKED_GATHER): the same [per_rank(total, world), ...] send buffer and the same [world * per, ...]
    receive buffer on rank `dst` as gather_to_root, but slots [c0, c1) of every rank travel as gather number c — issued
    asynchronously as soon as a rank has written them, while it goes on computing — so rank `dst` can recompose and download
    what has arrived under the shards' remaining work instead of after one gather at the very end.

    Every rank must call issue(0 .. n_chunks-1) in order (they are collectives), whatever its share of the work.  RCCL:
    issue() only enqueues (the collective waits for the compute stream's work up to this point on its own stream);
    complete(c) makes the CURRENT stream wait for gather c.  gloo (rehearsal): issue() stages the slots through host memory
    (synchronising with the compute stream), complete(c) blocks the host and uploads what arrived."""

    def __init__(self, send, total, chunks, dst=0):
        self.rank, self.world = rank_world()
        self.per = per_rank(total, self.world)
        assert self.world > 1 and send.shape[0] == self.per, (self.world, send.shape, self.per)
        self.send, self.total, self.dst = send, total, dst
        self.bounds = chunk_bounds(self.per, chunks)
        self.n_chunks = len(self.bounds)
        self.work = [None] * self.n_chunks
        self.on_device = td.get_backend() == 'nccl'
        self.recv = self._staged = None
        if self.rank == dst:
            self.recv = torch.empty((self.world * self.per,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
            self._staged = [None] * self.n_chunks

    def _views(self, buf, c0, c1):
        return [buf[r * self.per + c0:r * self.per + c1] for r in range(self.world)]

    def issue(self, c):
        assert self.work[c] is None and (c == 0 or self.work[c - 1] is not None), 'gathers are issued once, in order'
        c0, c1 = self.bounds[c]
        src = self.send[c0:c1]
        into = None
        if self.on_device:
            if self.rank == self.dst:
                into = self._views(self.recv, c0, c1)
        else:
            src = src.cpu()
            if self.rank == self.dst:
                self._staged[c] = torch.empty((self.world, c1 - c0) + tuple(src.shape[1:]), dtype=src.dtype)
                into = list(self._staged[c].unbind(0))
        self.work[c] = td.gather(src, into, dst=self.dst, async_op=True)

    def complete(self, c):
        """Root: everything slots [c0, c1) of every rank hold is in `recv` as far as the current stream is concerned.
        Other ranks: gather c no longer needs the send buffer."""
        self.work[c].wait()
        if self.rank == self.dst and not self.on_device:
            c0, c1 = self.bounds[c]
            for r, v in enumerate(self._views(self.recv, c0, c1)):
                v.copy_(self._staged[c][r])
            self._staged[c] = None
        return self.bounds[c]

    def slots_done(self, c):
        """Number of leading slots of every rank's shard that have arrived once gathers 0..c are complete."""
        return self.bounds[c][1]


def load_weights_on_root(path, cin, cout, num_layers, feature_size, device=None, src=0):
    """C1 as the product path uses it (supres._get_model under an initialised process group): rank `src` alone opens and
    parses the checkpoint (the stand-in for `model.load_weights(predict_file)`, testing/supres.py:63), every rank gets the
    keras-flat float32 vector by ONE broadcast — the file is needed on rank `src` only.
    A failure on the root is a failure everywhere: a one-element status goes first, so that no rank is left waiting in the
    data broadcast; the root re-raises its own exception (OSError for a missing file, like keras), the others an OSError
    naming the file and the root."""
    from . import weights as _weights
    rank, world = rank_world()
    if world == 1:
        return _weights.load_flat(path, cin, cout, num_layers, feature_size)
    count = _weights.num_params(cin, cout, num_layers, feature_size)
    backend = td.get_backend()
    dev = torch.device('cpu') if backend != 'nccl' else (device or torch.device('cuda', torch.cuda.current_device()))
    flat, err = None, None
    if rank == src:
        try:
            flat = _weights.load_flat(path, cin, cout, num_layers, feature_size)
        except Exception as e:          # reported to every rank below, then re-raised here
            err = e
    status = torch.tensor([0 if (rank == src and err is not None) else 1], dtype=torch.int64, device=dev)
    td.broadcast(status, src=src)
    if int(status.item()) == 0:
        if err is not None:
            raise err
        raise OSError('rank %d could not read the weights (name = %r); see its error' % (src, path))
    return broadcast_weights(flat, count, device=device, src=src)


def broadcast_weights(flat, count, device=None, src=0):
    """C1: `flat` (float32 ndarray) is only read on rank `src`; returns the ndarray on every rank."""
    rank, world = rank_world()
    if world == 1:
        return np.ascontiguousarray(flat, np.float32)
    backend = td.get_backend()
    dev = torch.device('cpu') if backend != 'nccl' else (device or torch.device('cuda', torch.cuda.current_device()))
    if rank == src:
        t = torch.from_numpy(np.ascontiguousarray(flat, np.float32)).to(dev)
        assert t.numel() == count
    else:
        t = torch.empty(count, dtype=torch.float32, device=dev)
    td.broadcast(t, src=src)
    return t.cpu().numpy()


def _preflight(argv=None):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m dsen2_amd.dist [--backend gloo]

    Pre-flight of the N-GPU path WITHOUT the network: first contact (guarded: fails fast, says where), then the two
    collectives of the product with their real sizes — C1, one broadcast of the DSen2 weight vector (7.16 MB), and C2, the
    bench's per-step gather of 512 x 6 x 32 x 32 float32 outputs per rank (12.6 MB) — timed over 20 repetitions, plus one
    chunked gather of a full-tile shard's size.  Rank 0 prints one JSON line: what a node delivers per xGMI link before any
    kernel of ours runs (DESIGN §6 prices the gather at 48 GB/s per link)."""
    import argparse
    import json
    ap = argparse.ArgumentParser(prog='python -m dsen2_amd.dist')
    ap.add_argument('--backend', default='nccl', choices=list(BACKENDS))
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--shard-patches', type=int, default=1226, help='patches of 6 x 112 x 112 float32 per rank for the chunked gather (1226 = a 10980^2 tile over 8 ranks)')
    args = ap.parse_args(argv)
    rank, world, dev = init_from_env(args.backend)
    cdev = dev if args.backend == 'nccl' else torch.device('cpu')

    def sync():
        torch.cuda.synchronize(dev)

    out = {'world': world, 'backend': 'rccl' if args.backend == 'nccl' else 'gloo', 'first_contact': first_contact()}
    if world > 1:
        with guarded_step('broadcast 7.16 MB (C1)'):
            w = torch.zeros(1790000, dtype=torch.float32, device=cdev)
            td.broadcast(w, src=0)
            sync()
        with guarded_step('gather 12.6 MB per rank (C2), %d repetitions' % args.reps):
            send = torch.full((512, 6, 32, 32), float(rank), dtype=torch.float32, device=cdev)
            recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
            td.gather(send, recv, dst=0)            # warm: RCCL sets its channels up here
            sync()
            td.barrier()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                td.gather(send, recv, dst=0)
            sync()
            td.barrier()
            dt = (time.perf_counter() - t0) / args.reps
            ok = rank != 0 or all(float(recv[r][0, 0, 0, 0]) == float(r) for r in range(world))
            out['gather_12p6MB_ms'] = round(dt * 1e3, 3)
            out['gather_GBps_per_peer_link'] = round(send.numel() * 4 / dt / 1e9, 2)       # every peer sends its 12.6 MB in that time
            out['gather_payload_ok'] = bool(ok)
        with guarded_step('chunked gather of a full-tile shard'):
            per = args.shard_patches
            shard = torch.full((per, 6, 112, 112), float(rank), dtype=torch.float32, device=dev)
            sync()
            td.barrier()
            t0 = time.perf_counter()
            cg = ChunkedGather(shard, per * world, 8)
            for c in range(cg.n_chunks):
                cg.issue(c)
            for c in range(cg.n_chunks):
                cg.complete(c)
            sync()
            td.barrier()
            dt = time.perf_counter() - t0
            out['chunked_gather_s'] = round(dt, 4)
            out['chunked_gather_GBps_per_peer_link'] = round(shard.numel() * 4 / dt / 1e9, 2)
            if rank == 0:
                out['chunked_payload_ok'] = bool(all(float(cg.recv[r * per, 0, 0, 0]) == float(r) and float(cg.recv[r * per + per - 1, 5, 111, 111]) == float(r)
                                                     for r in range(world)))
    if rank == 0:
        print(json.dumps(out), flush=True)
    finalize()


if __name__ == '__main__':
    _preflight()
