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
from datetime import timedelta


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
    # Two clocks.  DATA collectives (the metric gather, the gradient all-reduce) start when every rank is already there, so
    # a short timeout only ever catches a dead peer.  The RENDEZVOUS after the sharded loop is different: it completes when
    # the SLOWEST rank has finished its shard, so its timeout bounds the finish-time skew between ranks (whole-scene shards,
    # slow h5 reads) -- a rank that is merely early must wait, for hours if need be, not fail the job after all the work is
    # done.  A peer that was hard-killed is the launcher's business (torchrun takes the whole job down).
    timeout = timedelta(seconds=float(os.environ.get("HIMO_DIST_TIMEOUT_S", "300")))
    rendezvous_timeout = timedelta(seconds=float(os.environ.get("HIMO_RENDEZVOUS_TIMEOUT_S", str(12 * 3600))))
    global _FLAG_GROUP
    if backend == "nccl":
        if torch.cuda.device_count() <= local:
            raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=timeout)
    else:
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    # the "did every rank get through" flag travels over a host-side gloo group of its own: a rank whose GPU faulted would
    # fail again inside an RCCL collective and bury the original error, and this group carries the long rendezvous timeout;
    # data (the metric gather, gradients) stays on the job's main group
    _FLAG_GROUP = dist.new_group(backend="gloo", timeout=rendezvous_timeout)
    try:
        yield rank, world
    finally:
        _FLAG_GROUP = None
        dist.destroy_process_group()


_FLAG_GROUP = None


def all_ranks_ok(ok: bool) -> bool:
    """The job's rendezvous after the sharded loop: True only if EVERY rank got through its share.  A rank whose loop raised
    an ``Exception`` still arrives here (callers pass ok=False) so the others are never left waiting for a process that has
    already given up; afterwards the failing rank re-raises and the rest stop cleanly."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return ok
    if _FLAG_GROUP is not None:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_FLAG_GROUP)
        return bool(flag.item())
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def rendezvous(err: Exception | None, what: str = "its share of the frames") -> None:
    """``all_ranks_ok`` + the raising that follows it in every entry point.  ``err``: the ``Exception`` this rank's loop
    raised, or None.  KeyboardInterrupt / SystemExit are never routed here (the mains catch ``Exception`` only): an
    interrupted rank leaves at once and the launcher takes the job down.  If the rendezvous collective itself fails on a
    rank that already carries an error, the ORIGINAL error is raised, with the collective's failure as its cause."""
    try:
        everyone = all_ranks_ok(err is None)
    except Exception as coll:
        if err is not None:
            raise err from coll
        raise
    if err is not None:
        raise err
    if not everyone:
        raise RuntimeError(f"another rank failed; this rank finished {what}")
