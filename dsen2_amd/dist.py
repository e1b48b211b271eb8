"""Patch sharding across the GPUs of one node: one process per GPU, torch.distributed over RCCL.

The forward pass has no exchange step (patches are independent), so there are exactly two
collectives and neither is on the per-layer path:
  C1  broadcast_weights : rank 0 reads the checkpoint once, everyone receives the flat float32 vector
                          (7.16 MB DSen2, 151 MB VDSen2) with one broadcast
  C2  gather_patches    : each rank's predictions are collected with one all-gather into a
                          pre-sized buffer (the "gather of outputs over xGMI")
Partitioning: contiguous ranges of the row-major patch index, ceil(N/R) per rank (last ranks may be
short or empty).  With backend "nccl" (= RCCL on ROCm) tensors stay on the GPU; with "gloo" (CPU tests)
they are staged through host memory.
"""
import numpy as np
import torch
import torch.distributed as td


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


def gather_patches(local, total):
    """All ranks contribute their [count, ...] slice (shard_range order); all ranks get [total, ...]."""
    rank, world = rank_world()
    if world == 1:
        assert local.shape[0] == total
        return local
    per = per_rank(total, world)
    cdev = _collective_device(local)
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
    padded[:local.shape[0]] = local.to(cdev)
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
    td.all_gather_into_tensor(out, padded)
    return out[:total].to(local.device)


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
