"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl").

The reference wraps the model in apex DistributedDataParallel (bin/vc_train.py:423-431), i.e. an averaged
gradient all-reduce per step plus an initial parameter broadcast; BatchNorm statistics stay rank-local.
Here the gradients already live in ONE flat fp32 buffer (optim.FlatAdam), so the exchange is a few large
collectives instead of one per tensor.  xGMI is point-to-point (7 links per GPU), so large messages are
what reaches link bandwidth; `chunk_numel` keeps each collective big (default 32 Mi elements = 128 MiB)
while letting the first chunks start before the last are issued.
"""
import os

import torch


def init_from_env(backend=None):
    """env:// initialisation used by the launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return dist, dist.get_rank(), dist.get_world_size()


def allreduce_mean_(flat, dist, world, chunk_numel=32 * 1024 * 1024, group=None, force=False):
    """In-place mean all-reduce of a flat buffer in large chunks (works on cuda/RCCL and cpu/gloo).
    force: issue the collectives even at world size 1 (exercises the backend on a single device)."""
    if world <= 1 and not force:
        return flat
    n = flat.numel()
    handles = []
    for o in range(0, n, chunk_numel):
        handles.append(dist.all_reduce(flat[o:o + chunk_numel], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for h in handles:
        h.wait()
    flat.mul_(1.0 / world)
    return flat


def allreduce_sum_begin(flat, dist, world, chunk_numel=32 * 1024 * 1024, group=None, force=False):
    """Start the in-place SUM all-reduce of a flat buffer (or a slice of one) and return the pending handles: the
    collectives run on the backend's own stream while the caller keeps launching compute; allreduce_end() joins."""
    if world <= 1 and not force:
        return []
    return [dist.all_reduce(flat[o:o + chunk_numel], op=dist.ReduceOp.SUM, group=group, async_op=True)
            for o in range(0, flat.numel(), chunk_numel)]


def allreduce_end(handles):
    for h in handles:
        h.wait()


def broadcast_(flat, dist, world, src=0, group=None):
    """Initial parameter broadcast from rank 0 (what DDP does at wrap time)."""
    if world > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat
