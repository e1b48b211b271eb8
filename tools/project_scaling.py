#!/usr/bin/env python3
"""Projection of the patch-sharded full tile (BASELINE configs[3]: DSen2_20 over 10980 x 10980, supres._run) for
N = 1, 2, 4, 8 GPUs from stages MEASURED on one GPU — an expectation for the first hardware run to be compared with, NOT a
scaling measurement (no N > 1 run on hardware exists).

What a rank does in supres._run, and how each part scales:
  1/N   host uint16 -> float32 of the row slab its patches read, H2D of the slab (measured here for rank 0's slab at each N:
        the slabs overlap by the patches' borders, so it is slightly more than 1/N)
  1/N   tiling, up-sampling, the network and the crop of the predictions into the send buffer for ceil(9801 / N) patches
        (measured: the full run's GPU time / 9801 patches, and the time of rank 0's shard at each N)
  C2    gather of the inner crops to rank 0: (N-1) peers x 2.95 GB / N each over their own xGMI link — not measurable on
        one GPU; priced at 48 GB/s per link (a third of the 153 GB/s link peak: RCCL gather = point-to-point sends)
  1     on rank 0 only: page-locked buffer for the 2.9 GB result (allocated under the GPU work: hidden while that work
        is longer than the allocation), recomposition of all 9801 crops, D2H of the image
Two forms of C2 are priced (supres._run):
  one-shot  (default)                  work + gather + recomposition + D2H, one after the other on rank 0
  chunked   (DSEN2_CHUNKED_GATHER=1)   the crops travel in K pieces under the shards' work; rank 0 recomposes and downloads
            what has arrived on a stream of its own.  MEASURED here on one GPU: rank 0's shard with the recomposition + D2H of
            the first K-1 pieces of the WHOLE image running beside it (what the overlap costs the kernels), then the last
            piece alone; the transfers themselves are priced: only the last piece's gather is exposed.
Prints one JSON object; profiles/r05_scaling_projection.json is a committed run, DESIGN.md §6 quotes it.
"""
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
from dsen2_amd import dist as D, patches as P, supres, weights      # noqa: E402

XGMI_GBPS = 48.0
n = 10980
rng = np.random.default_rng(0)
d10 = rng.integers(35, 13110, size=(n, n, 4), dtype=np.uint16)
d20 = rng.integers(35, 13110, size=(n // 2, n // 2, 6), dtype=np.uint16)
tmp = tempfile.mkdtemp()
np.save(os.path.join(tmp, 's2_032_lr_1e-04.npy'), weights.random_he_uniform(10, 6, 6, 128, seed=11))
supres.MDL_PATH = os.path.join(tmp, '')


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


quiet(supres.DSen2_20, d10[:240, :240], d20[:120, :120])            # library load, model build, weight upload
dev = P.default_device()
model = quiet(supres._get_model, ((4, None, None), (6, None, None)), False, False)
org, n_alloc = P.tile_origins(d20.shape, 64, 4)
used = org.shape[0]
patch, border, inner = 128, 8, 112
out = {'tile': [n, n], 'patches': int(used), 'xgmi_gbps_assumed_per_link': XGMI_GBPS}

# ---- the whole single-rank call (what N = 1 is) ----
t0 = T(); y = quiet(supres.DSen2_20, d10, d20); t1 = T()
out['n1_first_call_s'] = round(t1 - t0, 3)          # includes the first page-locked allocation of the process (2.9 GB)
del y
t0 = T(); y = quiet(supres.DSen2_20, d10, d20); t1 = T()
out['n1_measured_s'] = round(t1 - t0, 3)
del y

# ---- rank 0's shard at each N: slab conversion + upload, GPU work on its patches, crop into the send buffer ----
bs = model.preferred_batch(patch, patch)
per_n = {}
for world in (1, 2, 4, 8):
    per = D.per_rank(used, world)
    my = org[:per]
    r = {}
    t0 = T()
    imgs, orgs = [], []
    for d, s, ps, b in ((d10, 2, 128, 8), (d20, 1, 64, 4)):
        r0, r1 = supres._row_slab(my, s, ps, b, d.shape[0]) if world > 1 else (0, d.shape[0])
        imgs.append(P._to_device_f32(d[r0:r1], dev))
        sh = (my * s).astype(np.int32); sh[:, 0] -= r0
        orgs.append(torch.from_numpy(np.ascontiguousarray(sh)).to(dev))
    t1 = T()
    r['slab_to_f32_and_h2d_s'] = round(t1 - t0, 4)
    send = torch.empty((per, 6, inner, inner), dtype=torch.float32, device=dev)
    t1 = T()
    for i0 in range(0, per, bs):
        nb = min(bs, per - i0)
        p10 = P.gather_patches_device(imgs[0], my, 2, 8, 128, n_alloc, divisor=2000, first=i0, count=nb, origins_dev=orgs[0])
        lr = P.gather_patches_device(imgs[1], my, 1, 4, 64, n_alloc, first=i0, count=nb, origins_dev=orgs[1])
        p20 = P.interp_patches_device(lr, (patch, patch), post_divisor=2000)
        yb = model.forward_device([p10, p20])
        send[i0:i0 + nb].copy_(yb[:, :, border:patch - border, border:patch - border])
    t2 = T()
    r['gpu_shard_s'] = round(t2 - t1, 4)
    r['patches_per_rank'] = int(per)
    r['gather_in_s_at_assumed_link_rate'] = round((per * 6 * inner * inner * 4) / (XGMI_GBPS * 1e9), 4) if world > 1 else 0.0
    per_n[world] = r
    del imgs, orgs
    if world < 8:
        del send

# ---- rank 0's serial tail: pinned allocation, recomposition of all crops, D2H ----
crops = torch.empty((used, 6, inner, inner), dtype=torch.float32, device=dev)
crops[:send.shape[0]] = send
del send
t0 = T(); host = torch.empty((n, n, 6), dtype=torch.float32, pin_memory=True); t1 = T()
out['pinned_alloc_s'] = round(t1 - t0, 4)
img = quiet(P.recompose_device, crops, 0, (n, n, 4), scale=2000); t2 = T()
out['recompose_s'] = round(t2 - t1, 4)
host.copy_(img, non_blocking=True); t3 = T()
out['d2h_pinned_s'] = round(t3 - t2, 4)
tail = out['recompose_s'] + out['d2h_pinned_s']

# ---- the chunked form, emulated on this one GPU: rank 0 computes its shard while its tail stream recomposes and downloads
# the rows whose crops "have arrived" (they are all in `crops` already; the transfers are priced, not run) ----
K = int(os.environ.get('DSEN2_GATHER_CHUNKS', '8'))
out['gather_chunks'] = K
chunked = {}
for world in (2, 4, 8):
    per = D.per_rank(used, world)
    my = org[:per]
    imgs, orgs = [], []
    for d, s, ps, b in ((d10, 2, 128, 8), (d20, 1, 64, 4)):
        r0, r1 = supres._row_slab(my, s, ps, b, d.shape[0])
        imgs.append(P._to_device_f32(d[r0:r1], dev))
        sh = (my * s).astype(np.int32); sh[:, 0] -= r0
        orgs.append(torch.from_numpy(np.ascontiguousarray(sh)).to(dev))
    send = torch.empty((per, 6, inner, inner), dtype=torch.float32, device=dev)
    bounds = D.chunk_bounds(per, K)
    slot = np.arange(used) % per
    done_rows = np.zeros(-(-n // inner), bool)
    tail_stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nxt = 0
    events = []
    for i0 in range(0, per, bs):
        nb = min(bs, per - i0)
        p10 = P.gather_patches_device(imgs[0], my, 2, 8, 128, n_alloc, divisor=2000, first=i0, count=nb, origins_dev=orgs[0])
        lr = P.gather_patches_device(imgs[1], my, 1, 4, 64, n_alloc, first=i0, count=nb, origins_dev=orgs[1])
        p20 = P.interp_patches_device(lr, (patch, patch), post_divisor=2000)
        yb = model.forward_device([p10, p20])
        send[i0:i0 + nb].copy_(yb[:, :, border:patch - border, border:patch - border])
        while nxt < len(bounds) and i0 + nb >= bounds[nxt][1]:
            ev = torch.cuda.Event(); ev.record(); events.append(ev); nxt += 1
    with torch.cuda.stream(tail_stream):
        for c, ev in enumerate(events):
            tail_stream.wait_event(ev)                          # piece c of every rank is "there" when rank 0 has computed its own
            for r0, r1 in P.final_row_runs(slot < bounds[c][1], done_rows, (n, n), inner):
                P.recompose_rows_device(crops, 0, img, r0, r1, scale=2000)
                host[r0:r1].copy_(img[r0:r1], non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    assert done_rows.all()
    last_gather = (bounds[-1][1] - bounds[-1][0]) * 6 * inner * inner * 4 / (XGMI_GBPS * 1e9)
    chunked[world] = {'shard_with_overlapped_tail_s': round(t1 - t0, 4), 'last_piece_gather_s': round(last_gather, 4)}
    del imgs, orgs, send

proj = {}
for world, r in per_n.items():
    work = r['slab_to_f32_and_h2d_s'] + r['gpu_shard_s']
    exposed_alloc = max(0.0, out['pinned_alloc_s'] - r['gpu_shard_s'])     # allocated while the GPU work runs
    total = work + r['gather_in_s_at_assumed_link_rate'] + exposed_alloc + tail
    proj[world] = {'projected_s': round(total, 3), 'per_rank_work_s': round(work, 3), 'rank0_tail_s': round(tail + exposed_alloc, 3),
                   'gather_s': r['gather_in_s_at_assumed_link_rate'], **r}
    if world in chunked:
        c = chunked[world]
        # rank 0: slab upload, then its shard with the tail of the arrived pieces beside it (measured), plus the last piece's
        # transfer (priced; the earlier pieces travel under the work)
        proj[world]['chunked'] = dict(c, projected_s=round(r['slab_to_f32_and_h2d_s'] + c['shard_with_overlapped_tail_s'] + c['last_piece_gather_s'] + exposed_alloc, 3))
# N = 1 is the MEASURED whole call (which hides its download under the batches: profiles/r04_ablation.md §3); N > 1 cannot (rank 0
# receives the crops at the end), so the speed-ups are against the measured single-GPU time
base = out['n1_measured_s']
for world in proj:
    proj[world]['speedup_vs_measured_n1'] = round(base / (proj[world]['projected_s'] if world > 1 else base), 2)
    if 'chunked' in proj[world]:
        proj[world]['chunked']['speedup_vs_measured_n1'] = round(base / proj[world]['chunked']['projected_s'], 2)
out['projection'] = proj
print(json.dumps(out))
