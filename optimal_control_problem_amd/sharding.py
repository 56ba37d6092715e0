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


INIT_TIMEOUT_S = 180       # rendezvous + communicator set-up + the first collective; a rank that cannot join ends with a one-line reason instead of hanging the job
GROUP_TIMEOUT_S = 1800     # every later collective (the group's own timeout): ranks may be minutes apart, e.g. one of them rebuilding the library


def init_distributed(n_gpus, backend=None):
    """Returns (rank, world, local_rank, dist_or_None).  world == 1 needs no process group.  A rank that fails to join (rendezvous, RCCL
    communicator) exits non-zero with one line on stderr; the launcher then ends its siblings."""
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
        import datetime
        import sys
        import threading

        def give_up():      # the join watchdog: only the rendezvous and the first collective are under the short limit, not the group
            print("sharding: rank %d of %d could not join the %s group on %s:%s within %d s" % (rank, world, backend, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"), INIT_TIMEOUT_S),
                  file=sys.stderr, flush=True)
            os._exit(3)
        watchdog = threading.Timer(INIT_TIMEOUT_S, give_up); watchdog.daemon = True; watchdog.start()
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=GROUP_TIMEOUT_S))
            # the first collective is where a broken RCCL set-up shows (communicators are created lazily): do it here, under the watchdog
            t = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", local) if backend == "nccl" else torch.device("cpu"))
            dist.all_reduce(t)
            if backend == "nccl":
                torch.cuda.synchronize()
            watchdog.cancel()
        except Exception as e:      # noqa: BLE001 -- whatever the backend raises
            watchdog.cancel()
            print("sharding: rank %d of %d could not join the %s group on %s:%s: %s" % (rank, world, backend, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"),
                                                                                       str(e).splitlines()[0] if str(e) else type(e).__name__), file=sys.stderr, flush=True)
            raise SystemExit(3)
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


def all_values(value, dist):
    """one float per rank, on every rank (diagnostics of a multi-rank run: per-rank kernel times, device ordinals)"""
    if dist is None:
        return [float(value)]
    import torch
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=_dev(dist))
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def collective_record(dist, kernel_ms, gather_bytes, gather_ms, local_device, share_gpu=False):
    """what the collective layer saw in a multi-rank run -- backend, ranks, per-rank kernel time and device, size and time of the final
    gather -- so that the record of an N-GPU run answers "did RCCL see N ranks" by itself; None for a single process.  Collective call:
    every rank must make it."""
    if dist is None:
        return None
    return {"backend": dist.get_backend(), "world": dist.get_world_size(), "kernel_ms_per_rank": all_values(kernel_ms, dist),
            "devices": [int(v) for v in all_values(float(local_device), dist)], "gather_bytes": int(gather_bytes), "gather_ms": float(gather_ms),
            "share_gpu": bool(share_gpu)}


def gather_rows(local, dist, dst=0):
    """Final gather of per-rank result rows onto rank `dst`; returns the concatenated array there and None
    elsewhere.  Row counts may differ by rank (shard_range gives shards that differ by one row when the batch is
    not a multiple of the world size): the counts are exchanged first and short shards are padded to the longest
    for the collective, then trimmed.  `local` is a NumPy array or a torch tensor."""
    if dist is None:
        return local
    import torch
    t = local if hasattr(local, "data_ptr") else torch.from_numpy(np.ascontiguousarray(local))
    t = t.to(_dev(dist))
    world = dist.get_world_size()
    counts = torch.zeros(world, dtype=torch.int64, device=t.device)
    counts[dist.get_rank()] = t.shape[0]
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    counts = [int(c) for c in counts.tolist()]
    rows = max(counts)
    if t.shape[0] < rows:
        pad = torch.zeros((rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    t = t.contiguous()
    out = [torch.empty_like(t) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(t, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    cat = torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)
    return cat if hasattr(local, "data_ptr") else cat.cpu().numpy()


def launch_ranks(n_ranks, argv, extra_env=None):
    """Start `n_ranks` fresh processes of the running script, one per GPU, with the torch.distributed rendezvous
    variables set (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR = 127.0.0.1 / MASTER_PORT = a free port), wait for all of
    them and return the worst exit code.  Called BEFORE anything touches the GPU (a process that has initialised HIP must
    not be replaced or forked); the children are ordinary child processes, rank 0's stdout passes through.  A child that
    fails ends the others."""
    import socket
    import subprocess
    import sys
    import time
    timeout_s = float(os.environ.get("MPCQP_LAUNCH_TIMEOUT_S", "1500"))
    procs = []

    def reap(grace=5.0):
        """end whatever is still running: terminate, then kill after a grace period"""
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_end = time.time() + grace
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    rc = 0
    try:
        for attempt in range(3):
            # a free port is found by bind / close, which another job may grab in between: a rank that cannot rendezvous exits with code 3
            # within INIT_TIMEOUT_S, and the launch is retried on a new port (fresh child processes every time)
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            del procs[:]
            for r in range(n_ranks):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                env.update(extra_env or {})
                procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
            rc = 0
            t_end = time.time() + timeout_s
            alive = list(procs)
            while alive:
                time.sleep(0.2)
                if time.time() > t_end:
                    print("sharding.launch_ranks: %d rank(s) still running after %.0f s, ending them" % (len(alive), timeout_s), file=sys.stderr, flush=True)
                    rc = rc or 124
                    reap()
                    break
                for p in list(alive):
                    code = p.poll()
                    if code is None:
                        continue
                    alive.remove(p)
                    if code != 0:
                        rc = rc or code
                        reap()
                        alive = []
                        break
            if rc != 3:
                break
    finally:
        reap()
    return rc
