"""Probe (GPU box): can two processes share ONE GPU and all-reduce CUDA tensors over gloo / over nccl?"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, backend, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group(backend, rank=rank, world_size=world)
        t = torch.full((1 << 20,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"[{backend}] rank {rank}: all_reduce of a CUDA tensor -> {t[0].item()} (expect 3)", flush=True)
        dist.broadcast(t, src=0)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"[{backend}] rank {rank}: FAILED {type(e).__name__}: {e}", flush=True)


if __name__ == "__main__":
    for i, backend in enumerate(sys.argv[1:] or ["gloo"]):
        mp.spawn(worker, args=(2, backend, 29650 + i), nprocs=2, join=True)
