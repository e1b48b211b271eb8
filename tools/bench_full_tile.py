#!/usr/bin/env python3
"""BASELINE configs[3]: DSen2_20 + DSen2_60 over a synthetic full-size Sentinel-2 tile (10980 x 10980 @10 m),
end to end through the drop-in surface (host ndarray in, host ndarray out), on one GPU or patch-sharded over the
GPUs of a node (one process per GPU; the inner crops of the predictions are gathered to rank 0 over RCCL, which
recomposes and returns the image — every other rank gets None: dsen2_amd/dist.py).

    python tools/bench_full_tile.py [--size 10980] [--skip60] [--precision fp32|bf16|bf16x3] [--deep]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_full_tile.py
    (--backend gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks)
Rank 0 prints one JSON line with wall times.  Random-init weights; the data is synthetic.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsen2_amd import supres, weights        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=10980)
ap.add_argument('--skip60', action='store_true')
ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'])
ap.add_argument('--check', action='store_true', help='multi-rank: also verify the result against a single-rank run')
ap.add_argument('--precision', default=None, choices=['fp32', 'bf16', 'bf16x3'], help="supres.PRECISION (default: DSEN2_PRECISION or fp32)")
ap.add_argument('--deep', action='store_true', help='VDSen2 (d=32, F=256) instead of DSen2')
ap.add_argument('--lazy', type=int, default=-1, metavar='MARGIN', help='hand the images over as cli.LazyRows (rows read on demand, as the GDAL branch of the command line does under torch.distributed) with this margin of 10 m rows; reports the largest share of rows a rank read')
ap.add_argument('--plain-batches', action='store_true', help='A/B: cut the patches into memory-bound batches (batch_limit) instead of preferred_batch (bf16 modes: multiples of the CU count, which run the one-launch chain)')
ap.add_argument('--repeat', type=int, default=1, help='run DSen2_20 this many times in the process and report every time (the first call also pays for the workspace, the prediction buffer and the page-locked output buffer; later calls reuse them)')
ap.add_argument('--layout', default='c', choices=['c', 'rollaxis', 'transpose'], help="how the caller's arrays lie in memory: C-contiguous HWC, np.rollaxis(chw, 0, 3) as testing/s2_tiles_supres.py builds them from GDAL's ReadAsArray, or chw.transpose() as testing/demoDSen2.py's readh5 returns them")
ap.add_argument('--host-contiguous', action='store_true', help='A/B: make the views C-contiguous on the host inside the timed call (np.ascontiguousarray: what the upload did before it followed the storage order)')
ap.add_argument('--port', type=int, default=0, help=argparse.SUPPRESS)
args = ap.parse_args()

import torch.distributed as td      # noqa: E402
from dsen2_amd import dist          # noqa: E402
rank, world, _ = dist.init_from_env(args.backend)       # LOCAL_RANK's GPU, dmabuf-IPC environment, process group
if args.precision:
    supres.PRECISION = args.precision
if args.plain_batches:
    from dsen2_amd.DSen2Net import S2Model
    S2Model.preferred_batch = S2Model.batch_limit

n = args.size - args.size % 6
rng = np.random.default_rng(0)
d10 = rng.integers(35, 13110, size=(n, n, 4), dtype=np.uint16)
d20 = rng.integers(35, 13110, size=(n // 2, n // 2, 6), dtype=np.uint16)
d60 = rng.integers(35, 13110, size=(n // 6, n // 6, 2), dtype=np.uint16)
if args.layout != 'c':
    def _as_view(a):
        if args.layout == 'rollaxis':
            return np.rollaxis(np.ascontiguousarray(np.rollaxis(a, 2, 0)), 0, 3)       # HWC view of a CHW array
        return np.asfortranarray(a)                                                    # = (c, y, x)-ordered storage, transposed
    d10, d20, d60 = _as_view(d10), _as_view(d20), _as_view(d60)
    assert not d10.flags.c_contiguous
a10, a20, a60 = d10, d20, d60         # the arrays themselves: warm-up crops
lazy = []
if args.lazy >= 0:
    from dsen2_amd.cli import LazyRows

    def _lazy(a, div):
        z = LazyRows(lambda r0, r1, a=a: a[r0:r1], a.shape, a.dtype, margin=args.lazy // div)
        lazy.append(z)
        return z
    d10, d20, d60 = _lazy(d10, 1), _lazy(d20, 2), _lazy(d60, 6)
tmp = tempfile.mkdtemp()
if rank != 0:
    pass        # the weight files exist in rank 0's directory ONLY: every other rank receives them by broadcast (C1, dist.load_weights_on_root)
elif args.deep:                       # testing/supres.py:55-57: VDSen2 reads s2_033 / s2_034
    np.save(os.path.join(tmp, 's2_033_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 32, 256, seed=13))
    np.save(os.path.join(tmp, 's2_034_lr_1e-04.npy'), weights.random_he_uniform(12, 2, 32, 256, seed=14))
else:
    np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11))
    np.save(os.path.join(tmp, 's2_030_lr_1e-05.npy'), weights.random_he_uniform(12, 2, 6, 128, seed=12))
supres.MDL_PATH = os.path.join(tmp, '')
out = {'tile': [n, n], 'data': 'synthetic', 'n_gpus': world, 'precision': supres.PRECISION, 'batches': 'batch_limit' if args.plain_batches else 'preferred_batch', 'layout': args.layout, 'host_contiguous': bool(args.host_contiguous), 'deep': bool(args.deep), 'patches20': int(np.ceil(n / 112.0) ** 2), 'patches60': int(np.ceil(n / 168.0) ** 2)}


def timed(fn, *a):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        y = fn(*a)
    torch.cuda.synchronize()
    return y, time.perf_counter() - t0


with contextlib.redirect_stdout(io.StringIO()):
    supres.DSen2_20(a10[:240, :240], a20[:120, :120], args.deep)       # warm-up: library load, model build, weight upload
run20 = supres.DSen2_20
if args.host_contiguous:
    run20 = lambda a, b, deep: supres.DSen2_20(np.ascontiguousarray(a), np.ascontiguousarray(b), deep)      # noqa: E731
y20, t20 = timed(run20, d10, d20, args.deep)
if world > 1:                       # whole-job wall time: the slowest rank's
    tt = torch.tensor([t20], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
    td.all_reduce(tt, op=td.ReduceOp.MAX)
    t20 = float(tt.item())
out['dsen2_20_s'] = round(t20, 3)
if args.repeat > 1:
    runs = [round(t20, 3)]
    for _ in range(args.repeat - 1):
        y20 = None
        y20, t = timed(run20, d10, d20, args.deep)
        runs.append(round(t, 3))
    out['dsen2_20_s_runs'] = runs
out['dsen2_20_patches_per_s_128'] = round(out['patches20'] / t20, 1)
out['dsen2_20_equiv_32x32_patches_per_s'] = round(out['patches20'] * 16 / t20, 1)
if rank == 0:
    assert y20.shape == (n, n, 6) and np.isfinite(y20[::97, ::89]).all()
else:
    assert y20 is None              # only rank 0 receives, recomposes and downloads
if not args.skip60:
    with contextlib.redirect_stdout(io.StringIO()):
        supres.DSen2_60(a10[:384, :384], a20[:192, :192], a60[:64, :64], args.deep)
    y60, t60 = timed(supres.DSen2_60, d10, d20, d60, args.deep)
    if world > 1:
        tt = torch.tensor([t60], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        t60 = float(tt.item())
    out['dsen2_60_s'] = round(t60, 3)
    out['dsen2_60_equiv_32x32_patches_per_s'] = round(out['patches60'] * 36 / t60, 1)
    assert (y60.shape == (n, n, 2)) if rank == 0 else (y60 is None)
out['peak_gpu_mem_gib'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
if lazy:
    share = max(z.rows_read / float(z.shape[0]) for z in lazy)      # both networks' windows (the second mostly from the kept one)
    if world > 1:
        tt = torch.tensor([share], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        share = float(tt.item())
    out['largest_share_of_rows_read_by_a_rank'] = round(share, 3)
if world > 1:
    out['chunked_gather'] = supres._chunked_gather_wanted()
if world > 1 and args.check:
    td.barrier()
    td.destroy_process_group()            # dist.rank_world() now reports (0, 1): every rank computes everything
    if rank == 0:
        with contextlib.redirect_stdout(io.StringIO()):
            ref = supres.DSen2_20(d10, d20, args.deep)
        out['matches_single_rank'] = bool(np.array_equal(ref, y20))
if rank == 0:
    print(json.dumps(out))
dist.finalize()
