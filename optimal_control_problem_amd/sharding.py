"""Batch sharding across GPUs: one process per GPU, contiguous shards, no data-path collective.

QP instances are independent, so the batch is partitioned statically and the only
collectives are the timing reduction and the final gather of solutions (RCCL when on GPUs, gloo on CPU)."""
import os

import numpy as np


def shard_range(total, rank, world):
    """Contiguous [start, stop) of `total` instances owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_distributed(n_gpus, backend=None):
    """Returns (rank, world, local_rank, dist_or_None).  world == 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world == 1 and n_gpus <= 1:
        return 0, 1, 0, None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local, dist


def _dev(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist is not None and dist.get_backend() == "nccl" else torch.device("cpu")


def barrier(dist):
    if dist is None:
        return
    if dist.get_backend() == "nccl":
        import torch
        dist.barrier(device_ids=[torch.cuda.current_device()])     # name the device: no guessing, no hang on a wrong one
    else:
        dist.barrier()


def max_over_ranks(value, dist):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_dev(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_dev(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_rows(local, dist, dst=0):
    """Final gather of per-rank result rows (equal row count per rank) onto rank `dst`; returns the
    concatenated array there and None elsewhere.  `local` is a NumPy array or a torch tensor."""
    if dist is None:
        return local
    import torch
    t = local if hasattr(local, "data_ptr") else torch.from_numpy(np.ascontiguousarray(local))
    t = t.to(_dev(dist))
    world = dist.get_world_size()
    out = [torch.empty_like(t) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(t, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    cat = torch.cat(out, dim=0)
    return cat if hasattr(local, "data_ptr") else cat.cpu().numpy()
