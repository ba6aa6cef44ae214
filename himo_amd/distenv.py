"""One process per GPU: what the command-line entry points (``save``, ``save_zip``, ``eval``) do when they are started by
``torchrun`` / ``python -m torch.distributed.run`` -- the shape of the reference's 4-GPU job (assets/slurm/ssl-train-av2.sh:3)
applied to its serial per-frame loops (save_zip.py:112, eval.py:281): frame i belongs to rank i % world, every rank drives
the GPU of its LOCAL_RANK, and the only exchange is the final gather / barrier.

Without this, ``_dist()`` would see no process group, every rank would walk the whole dataset on cuda:0 and all of them
would race on the same result files.  The group is RCCL (``"nccl"``) when a HIP device is visible, gloo otherwise (the
CPU tests); ``HIMO_DIST_BACKEND`` overrides.
"""
from __future__ import annotations

import os
from contextlib import contextmanager


def launched_world() -> int:
    """WORLD_SIZE as exported by torchrun (1 when started as a plain process)."""
    return int(os.environ.get("WORLD_SIZE", "1"))


@contextmanager
def process_group():
    """Join the job's process group for the duration of a command-line ``main``: yields (rank, world).
    A plain single process (WORLD_SIZE unset or 1) yields (0, 1) and touches nothing; a group some caller has already
    initialised is used as is and left alone."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        yield dist.get_rank(), dist.get_world_size()
        return
    world = launched_world()
    if world <= 1 and os.environ.get("HIMO_DIST_FORCE") != "1":      # HIMO_DIST_FORCE=1: a one-rank group (exercises RCCL on one GPU)
        yield 0, 1
        return
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this host driver
    backend = os.environ.get("HIMO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        if torch.cuda.device_count() <= local:
            raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        yield rank, world
    finally:
        dist.destroy_process_group()


def all_ranks_ok(ok: bool) -> bool:
    """The job's rendezvous after the sharded loop: True only if EVERY rank got through its share.  A rank that failed
    still arrives here (callers wrap their loop in try/except and pass ok=False) so the others are never left waiting
    at a barrier for a process that has already died; afterwards the failing rank re-raises and the rest stop cleanly."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return ok
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())
