"""Drop-in for the reference's testing/supres.py: same names, signatures, constants, prints and returns.

    from supres import DSen2_20, DSen2_60          (testing/s2_tiles_supres.py:9, testing/demoDSen2.py:5)

DSen2_20(d10, d20, deep=False)      -> [x, y, 6] float32     testing/supres.py:15-30
DSen2_60(d10, d20, d60, deep=False) -> [x, y, 2] float32     testing/supres.py:33-50
_predict(test, input_shape, deep=False, run_60=False)        testing/supres.py:53-66

The whole chain — symmetric pad + overlapped tiling, mirror-bilinear up-sampling, /2000, the CNN,
recomposition, *2000 — runs on one MI355X without touching the host between stages; with
torch.distributed initialised (one process per GPU, every rank called with the same arrays) the patches are
sharded across ranks, each rank uploads only the rows its patches read, and the inner crops of the predictions are
gathered to rank 0, which recomposes and returns the image; every other rank returns None (dsen2_amd/dist.py).  Models are cached per (device, architecture, weight file) instead of being rebuilt
and re-read on every call as the reference does (supres.py:59,63) — observable behaviour is unchanged.
"""
from __future__ import division

import os

import numpy as np
import torch

from . import dist as _dist
from . import patches as _patches
from .DSen2Net import s2model

SCALE = 2000
MDL_PATH = '../models/'
# Not in the reference: arithmetic of the residual-block convolutions.  'fp32' (default, what keras computes),
# 'bf16' (bf16 operands, fp32 accumulate and residual stream; ~7x faster for deep=True, ~1e-3 relative error) or
# 'bf16x3' (every fp32 operand as two bf16 numbers, three bf16 MFMAs per product: ~1e-5 rmse in the normalised domain —
# inside the 1e-4 gate — at ~3x the fp32 rate).
PRECISION = os.environ.get('DSEN2_PRECISION', 'fp32')

_MODEL_CACHE = {}


def _weight_file(deep, run_60):
    # testing/supres.py:55-60
    if deep:
        return MDL_PATH + 's2_034_lr_1e-04.hdf5' if run_60 else MDL_PATH + 's2_033_lr_1e-04.hdf5'
    return MDL_PATH + 's2_030_lr_1e-05.hdf5' if run_60 else MDL_PATH + 's2_032_lr_1e-04.hdf5'


def _get_model(input_shape, deep, run_60):
    if deep:
        num_layers, feature_size = 32, 256      # supres.py:56
    else:
        num_layers, feature_size = 6, 128       # supres.py:59
    predict_file = _weight_file(deep, run_60)
    dev = _patches.default_device()
    # (under torch.distributed the ranks' MDL_PATH may differ — only rank 0's is read — so every rank keys its cache on its
    # own path string: all ranks hit or miss together as long as they make the same calls, which the collectives need anyway)
    key = (str(dev), tuple(s[0] for s in input_shape), num_layers, feature_size, os.path.abspath(predict_file), PRECISION)
    model = _MODEL_CACHE.get(key)
    if model is None:
        model = s2model(input_shape, num_layers=num_layers, feature_size=feature_size, device=dev, precision=PRECISION)
        print('Symbolic Model Created.')
        if _dist.rank_world()[1] > 1:
            # one process per GPU: rank 0 alone reads the checkpoint, one RCCL broadcast delivers it (C1, dsen2_amd/dist.py)
            # — the file need not exist on any other rank.  Collective: every rank
            # gets here on its first call for this architecture, all of them make the same calls.
            model.set_weights_flat(_dist.load_weights_on_root(predict_file, model.cin, model.cout, num_layers, feature_size,
                                                              device=dev))
        else:
            model.load_weights(predict_file)
        _MODEL_CACHE[key] = model
    else:
        print('Symbolic Model Created.')
    print("Predicting using file: {}".format(predict_file))
    return model


def clear_model_cache():
    _MODEL_CACHE.clear()


def _predict(test, input_shape, deep=False, run_60=False):
    """testing/supres.py:53-66 — list of NCHW float32 arrays in, [N, Cout, H, W] float32 array out."""
    model = _get_model(input_shape, deep, run_60)
    return model.predict(test, verbose=1)


def _check_sizes(dsets, scales):
    """The reference fails with a broadcasting ValueError when the images do not have the 10 m : 20 m (: 60 m) size
    ratio its crop arithmetic assumes (patches.py:67,136-137); say so before anything reaches the GPU."""
    lo = dsets[-1].shape
    for d, s in zip(dsets, scales):
        if len(d.shape) != 3:
            raise ValueError('expected HWC images, got shape %r' % (tuple(d.shape),))
        if d.shape[0] < s * lo[0] or d.shape[1] < s * lo[1]:
            raise ValueError('image of shape %r does not cover %d x the lowest-resolution image %r'
                             % (tuple(d.shape), s, tuple(lo)))


def _row_slab(org_lr, scale, patch, border, height):
    """Rows [r0, r1) of one resolution's image that the patches with low-res origins `org_lr` read (the symmetric
    padding only ever reflects at the true image edges, which a slab touching them shares)."""
    rows = org_lr[:, 0].astype(np.int64) * scale
    r0 = max(0, int(rows.min()) - border)
    r1 = min(int(height), int(rows.max()) - border + patch)
    return r0, r1


def _run(dsets, scales, patch, border, deep, run_60):
    _check_sizes(dsets, scales)
    dev = _patches.default_device()
    rank, world = _dist.rank_world()
    patch_sizes = [patch // (scales[0] // s) for s in scales]          # P, P//2(, P//6)
    borders = [border // (scales[0] // s) for s in scales]             # b, b//2(, b//6)
    org, n_alloc = _patches.tile_origins(dsets[-1].shape, patch_sizes[-1], borders[-1])
    used = org.shape[0]
    input_shape = tuple((int(d.shape[2]), None, None) for d in dsets)
    model = _get_model(input_shape, deep, run_60)
    cout = model.cout
    size = dsets[0].shape
    # this rank's contiguous share of the USED patches (the reference's trailing all-zero patches are
    # never read by recompose_images, so they are not computed)
    first, count = _dist.shard_range(used)
    per = _dist.per_rank(used, world)
    inner = patch - 2 * border
    single = n_alloc == 1            # recompose_images' single-patch shortcut (patches.py:375-376): a[0] uncropped
    # The buffer this rank's predictions go to.  One rank (or the single-patch case): whole patches.  Several ranks:
    # the inner crops, in the [per, ...] buffer the gather sends as it is (2.95 GB for a 10980^2 tile over all ranks
    # instead of 3.85 GB of whole patches to EVERY rank).
    if world == 1 or single:
        send = None
        pred_local = torch.empty((count, cout, patch, patch), dtype=torch.float32, device=dev)
    else:
        send = torch.empty((per, cout, inner, inner), dtype=torch.float32, device=dev)
        pred_local = None
    # Several ranks, DSEN2_CHUNKED_GATHER=1: the crops travel in DSEN2_GATHER_CHUNKS (8) pieces while the shards are still computing
    # (dist.ChunkedGather) and rank 0 recomposes + downloads what has arrived under the remaining work, instead of one
    # gather at the very end followed by recomposition and 51 ms of D2H on rank 0 alone.  Off by default until
    # an N > 1 box has measured RCCL's kernels next to two CU-filling persistent ones; the result is the same image.
    cg, next_chunk, cg_img, cg_ready = None, 0, None, None
    if send is not None and _chunked_gather_wanted():
        cg = _dist.ChunkedGather(send, used, _gather_chunks())
        if rank == 0:
            # the image rank 0 recomposes into, allocated NOW and fenced by an event: its tail stream may then start on the
            # first piece while the shard is still computing — waiting for the compute stream later (wait_stream) would wait
            # for the whole shard and put recomposition + download back after it
            cg_img = torch.empty((int(size[0]), int(size[1]), cout), dtype=torch.float32, device=dev)
            cg_ready = torch.cuda.Event()
            cg_ready.record(torch.cuda.current_stream(dev))
    bands = None                     # one rank, large image: the bands of rows already recomposed (below)
    if count > 0:
        # upload only the rows this rank's patches read (1/world of the tile), origins shifted into the slab
        my_org = org[first:first + count]
        imgs, org_dev = [], []
        for d, s, ps, b in zip(dsets, scales, patch_sizes, borders):
            r0, r1 = _row_slab(my_org, s, ps, b, d.shape[0]) if world > 1 else (0, d.shape[0])
            imgs.append(_patches._to_device_f32(d[r0:r1], dev))
            shifted = (my_org * s).astype(np.int32)
            shifted[:, 0] -= r0
            org_dev.append(torch.from_numpy(np.ascontiguousarray(shifted)).to(dev))      # uploaded once: the loop only enqueues
        bs = model.preferred_batch(patch, patch)
        # One rank, a large image: rows that are final are recomposed right after the batch that completes them — rows below
        # min(t * inner, H - inner) once the first t tile rows of patches are done (the last `inner` rows belong to the clamped
        # last tile row, patches.py:396-401) — and an event marks each band, so that the download can later run band by band
        # on a copy stream UNDER the batches still computing (a D2H takes 5-40 % of its own time away from the kernels,
        # profiles/r04_ablation.md §3).  This loop still only enqueues.
        if world == 1 and not single and _pinned_wanted((int(size[0]), int(size[1]), cout)) and \
                os.environ.get('DSEN2_BANDED_OUTPUT', '1') != '0':
            bands = dict(x_tiles=_patches.recompose_grid(size, patch, border)[0], rows=0, list=[],
                         img=torch.empty((int(size[0]), int(size[1]), cout), dtype=torch.float32, device=dev))
        for i0 in range(0, count, bs):
            n = min(bs, count - i0)
            xs = []
            for k, (img, s, ps, b) in enumerate(zip(imgs, scales, patch_sizes, borders)):
                if k == 0:
                    # `p10 /= SCALE` (supres.py:23) folded into the gather (IEEE divide, bit-identical)
                    xs.append(_patches.gather_patches_device(img, my_org, s, b, ps, n_alloc, divisor=SCALE,
                                                             first=i0, count=n, origins_dev=org_dev[k]))
                else:
                    lr = _patches.gather_patches_device(img, my_org, s, b, ps, n_alloc, first=i0, count=n,
                                                        origins_dev=org_dev[k])
                    # up-sample raw values, then `/= SCALE` (supres.py:24,43-44)
                    xs.append(_patches.interp_patches_device(lr, (patch, patch), post_divisor=SCALE))
            if send is None:
                model.forward_device(xs, out=pred_local[i0:i0 + n])
            else:
                y = model.forward_device(xs)
                send[i0:i0 + n].copy_(y[:, :, border:patch - border, border:patch - border])
                # a piece whose slots this rank has all written (a short shard: all it will ever write) goes out now
                while cg is not None and next_chunk < cg.n_chunks and i0 + n >= min(cg.bounds[next_chunk][1], count):
                    cg.issue(next_chunk)
                    next_chunk += 1
            if bands is not None:
                done = i0 + n
                final = int(size[0]) if done == count else min((done // bands['x_tiles']) * inner, int(size[0]) - inner)
                if final > bands['rows']:
                    # `images *= SCALE` (supres.py:29) folded into the recomposition
                    _patches.recompose_rows_device(pred_local, border, bands['img'], bands['rows'], final, scale=SCALE)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    bands['list'].append((bands['rows'], final, ev))
                    bands['rows'] = final
    if bands is not None and count > 0:
        # everything is enqueued; the page-locked buffer is allocated now, under that work (as in the one-shot path below),
        # and the bands whose events have fired by then start downloading at once
        print((cout, size[0], size[1]))                                # patches.py:392
        host = _host_output((int(size[0]), int(size[1]), cout))
        if host is None:               # page-locked memory exhausted: the image is complete on the device
            return bands['img'].cpu().numpy()
        copy_stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(copy_stream):
            for r0, r1, ev in bands['list']:
                copy_stream.wait_event(ev)
                host[r0:r1].copy_(bands['img'][r0:r1], non_blocking=True)
        copy_stream.synchronize()
        return host.numpy()            # ndarray view of the page-locked tensor (kept alive by the array)
    if single:
        if rank != 0:
            return None
        images = pred_local[0].permute(1, 2, 0).contiguous() * SCALE
        return images.cpu().numpy()
    # Everything above is only ENQUEUED (seconds of GPU work for a full tile): rank 0 allocates the page-locked
    # buffer the result is downloaded into now, under that work (0.18 s for a 10980^2 x 6 image; the download itself
    # then runs at 57 GB/s instead of 11 GB/s from pageable memory: 0.05 s instead of 0.26 s).
    if cg is not None:
        return _finish_chunked(cg, next_chunk, rank, dev, size, cout, inner, cg_img, cg_ready)
    host = _host_output((int(size[0]), int(size[1]), cout)) if rank == 0 else None
    if world == 1:
        print((cout, size[0], size[1]))                                # patches.py:392
        # `images *= SCALE` (supres.py:29) folded into the recomposition
        images = _patches.recompose_device(pred_local, border, size, scale=SCALE)
    else:
        # gather to root: rank 0 receives every rank's inner crops into views of one buffer and recomposes them
        # (cropped patches of `inner` with border 0 tile exactly like whole patches with their border, patches.py:380-403);
        # the other ranks return None — no page-locked buffer, no download, nothing received.
        crops = _dist.gather_to_root(send, used)
        if rank != 0:
            return None
        print((cout, size[0], size[1]))
        images = _patches.recompose_device(crops, 0, size, scale=SCALE)
    if host is None:
        return images.cpu().numpy()
    host.copy_(images, non_blocking=True)
    torch.cuda.synchronize(dev)
    return host.numpy()            # ndarray view of the page-locked tensor (kept alive by the array)


def _chunked_gather_wanted():
    return os.environ.get('DSEN2_CHUNKED_GATHER', '0') not in ('', '0')


def _gather_chunks():
    return max(1, int(os.environ.get('DSEN2_GATHER_CHUNKS', '8')))


def _finish_chunked(cg, next_chunk, rank, dev, size, cout, inner, img, ready):
    """The tail of a sharded run with the chunked gather.  Every rank: the pieces it has not issued yet (a rank whose shard
    is short or empty still takes part in every gather).  Rank 0: on a stream of its own — so that its compute stream never
    waits for RCCL — piece by piece: wait for the gather, recompose the image rows whose patches have all arrived (the crops
    tile like whole patches with border 0, patches.py:380-403), download them into the page-locked buffer.  All of it is
    enqueued at once; only the final synchronise blocks."""
    while next_chunk < cg.n_chunks:
        cg.issue(next_chunk)
        next_chunk += 1
    if rank != 0:
        for c in range(cg.n_chunks):
            cg.complete(c)             # the send buffer stays alive until every piece has left
        return None
    print((cout, size[0], size[1]))                                    # patches.py:392
    H, W = int(size[0]), int(size[1])
    host = _host_output((H, W, cout))          # page-locked, allocated under the work already enqueued
    x_tiles, y_tiles = int(np.ceil(W / float(inner))), int(np.ceil(H / float(inner)))
    slot = np.arange(x_tiles * y_tiles) % cg.per                       # a patch's slot in its rank's shard
    done_rows = np.zeros(y_tiles, bool)
    tail = torch.cuda.Stream(dev)
    tail.wait_event(ready)                     # `img` / `recv` exist as far as the compute stream is concerned (NOT: the shard is done)
    with torch.cuda.stream(tail):
        for c in range(cg.n_chunks):
            cg.complete(c)
            for r0, r1 in _patches.final_row_runs(slot < cg.slots_done(c), done_rows, size, inner):
                # `images *= SCALE` (supres.py:29) folded into the recomposition
                _patches.recompose_rows_device(cg.recv, 0, img, r0, r1, scale=SCALE)
                if host is not None:
                    host[r0:r1].copy_(img[r0:r1], non_blocking=True)
    tail.synchronize()
    assert done_rows.all()
    return host.numpy() if host is not None else img.cpu().numpy()


PINNED_OUTPUT_MIN_BYTES = 64 << 20     # DSEN2_PINNED_OUTPUT=0 disables; torch caches page-locked blocks for reuse


def _pinned_wanted(shape):
    nbytes = 4 * shape[0] * shape[1] * shape[2]
    return os.environ.get('DSEN2_PINNED_OUTPUT', '1') != '0' and nbytes >= PINNED_OUTPUT_MIN_BYTES


def _host_output(shape):
    if not _pinned_wanted(shape):
        return None
    try:
        return torch.empty(shape, dtype=torch.float32, pin_memory=True)
    except RuntimeError:           # page-locked memory exhausted: the pageable path still works
        return None


def DSen2_20(d10, d20, deep=False):
    """20 m -> 10 m.  d10 [x, y, 4] (B2 B3 B4 B8), d20 [x/2, y/2, 6] (B5 B6 B7 B8A B11 B12), any real dtype, HWC.
    deep=True selects VDSen2 (d=32, F=256).  Returns [x, y, 6] float32.  Geometry of testing/supres.py:21-22:
    patches of 128 with an 8-pixel border.

    With torch.distributed initialised (one process per GPU; every rank must make the same call with the same
    arrays — the call contains a collective) the patches are sharded over the ranks and ONLY RANK 0 returns the
    image; every other rank returns None.  The reference script's `if sr20 is None: exit` branch
    (testing/s2_tiles_supres.py:346-348) is what a non-root rank then takes."""
    return _run([d10, d20], [2, 1], patch=128, border=8, deep=deep, run_60=False)


def DSen2_60(d10, d20, d60, deep=False):
    """60 m -> 10 m.  As DSen2_20 plus d60 [x/6, y/6, 2] (B1 B9; B10 is not super-resolved).  Returns [x, y, 2]
    float32.  Geometry of testing/supres.py:40-41: patches of 192 with a 12-pixel border.  Under torch.distributed: the
    image on rank 0, None on every other rank (see DSen2_20)."""
    return _run([d10, d20, d60], [6, 3, 1], patch=192, border=12, deep=deep, run_60=True)
