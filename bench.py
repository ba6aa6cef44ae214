#!/usr/bin/env python3
"""bench.py -- LiDAR frames/s of the HiMo motion-compensation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pipeline|compdis]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (synthetic 120k-point sweeps already resident in HBM when the clock starts):
  pipeline (default)  north_star's per-frame path: voxelise 3 sweeps -> SeFlow++-style network forward (random-init,
                      float32 on the matrix cores) -> per-point flow -> ego-motion removal, dt0, flow2compDis ->
                      comp_dis.  A step = one batch of B frames (B network forwards + one fused comp_dis launch).
  compdis             only stages a1-a4 (the part of the path the reference tree contains) over a ragged batch of
                      B sweeps: the HBM-bound kernel on its own.
  train               BASELINE config 5: the self-supervised training step, data parallel (one flat all-reduce per step).
  fastnsf             BASELINE config 4: optimisation-based flow -- a step fits the per-scene coordinate MLP to one 120k-point
                      sweep pair (--fastnsf-iters Adam iterations); frames shard over the ranks like the pipeline's.
``--gpus N`` without a torchrun environment re-launches this file as N ranks (one per GPU) under
``python -m torch.distributed.run`` on 127.0.0.1; with one (WORLD_SIZE set) it must agree with WORLD_SIZE.
``--dry-run-cpu`` replaces ONLY the device work by a host stand-in (gloo instead of RCCL): it exists so that the launch,
barrier, max-over-ranks and gather logic is covered by CPU tests; its numbers mean nothing and the line says so.
Every rank owns its own frames (frames shard embarrassingly; weak scaling); the only collective is the final
gather of per-rank counts after the timed region.  Rank 0 prints ONE JSON line that also carries
  "roofline":     dominant kernel: algorithmic flops|bytes per launch / HIP-event-timed launch duration vs peak,
  "cpu_baseline": the CPU restatement (oracle) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as himo_amd/__init__.py (streams that share a hardware queue serialise); before any device call

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TF = 157.3         # dense float32-input MFMA peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0       # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)
POINTS_PER_FRAME = 120_000       # BASELINE.json metric


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "compdis", "train", "fastnsf"])
    ap.add_argument("--fastnsf-iters", type=int, default=100, help="fastnsf workload: optimiser iterations per frame")
    ap.add_argument("--frames-per-step", type=int, default=None, help="frames per rank per step")
    ap.add_argument("--points", type=int, default=POINTS_PER_FRAME)
    ap.add_argument("--precision", default="f16x2", choices=["bf16x3", "f16x2", "f32"],
                    help="network matrix arithmetic: split-bf16 (float32-class accuracy) or float32 MFMA")
    ap.add_argument("--train-precision", default="mixed", choices=["mixed", "bf16x3", "f32"],
                    help="train workload: forward fp16-split + data-gradient split-bf16 (mixed), all split-bf16, or float32 MFMA")
    ap.add_argument("--train-batchnorm", default="batch", choices=["batch", "frozen"],
                    help="train workload / leg: BatchNorm in training mode (batch statistics, gamma / beta trained, running statistics "
                         "updated: the reference's from-scratch job) or frozen (fine-tuning convention)")
    ap.add_argument("--float32-activations", action="store_true",
                    help="pipeline, f16x2: keep the backbone's maps float32 in HBM instead of the split activation format (A/B switch)")
    ap.add_argument("--refined", action="store_true", help="also write refined points (+12 B/pt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=110.0,
                    help="CPU work budget of the all-cores leg of the baseline: BASELINE.md section 3's protocol (5 warm-ups + 50 frames, "
                         "median) runs in full when 55 frames fit this budget at the probed per-frame time, otherwise as many frames as "
                         "fit (at least 3) and `cpu_baseline.sample` states the shortfall; the 1-thread leg gets a third of it")
    ap.add_argument("--cpu-frames", type=int, default=0,
                    help="time exactly this many frames per CPU leg (both legs) after 5 warm-ups, ignoring --cpu-seconds")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0 = torch's default)")
    ap.add_argument("--no-extra-precisions", action="store_true",
                    help="pipeline, N=1: skip the bf16x3 / f32 legs that follow the timed f16x2 region")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="pipeline, N=1: skip the short training-step (config 5) and FastNSF (config 4) legs that follow the timed region")
    ap.add_argument("--leg-train-steps", type=int, default=50, help="timed optimiser steps of the `leg_train` leg (after 2 warm-ups)")
    ap.add_argument("--train-batch", type=int, default=1,
                    help="train workload: samples per forward / backward pass and optimiser step (the launcher's batch_size=8 on one process: 8); "
                         "the default line's `leg_train` is 1, `leg_train_b8` is 8")
    ap.add_argument("--leg-fit-steps", type=int, default=200,
                    help="optimiser steps of the `leg_fit_h5` leg (seflow.fit.fit over .h5 scene files, batch_size 8); 0 skips the leg")
    ap.add_argument("--leg-fastnsf-fits", type=int, default=6, help="timed fits of the `leg_fastnsf` leg (after 1 warm-up)")
    ap.add_argument("--fit-workers", type=int, default=1, help="`leg_fit_h5`: reader threads of the training feeder (0: samples built inside the step loop)")
    ap.add_argument("--cloud", default="uniform", choices=["uniform", "rings"],
                    help="synthetic sweeps: SURVEY 8(d) uniform cloud with instances (default) or the LiDAR-like ring cloud")
    ap.add_argument("--sample-sets", type=int, default=3, help="distinct input batches rotated through the timed steps")
    ap.add_argument("--batches-in-flight", type=int, default=3, help="pipeline workload: networks / HIP streams fed in turn (default 3, the product's default)")
    ap.add_argument("--fits-in-flight", type=int, default=2, help="fastnsf workload: engines / HIP streams fed in turn (default 2)")
    ap.add_argument("--single-stream", action="store_true",
                    help="pipeline: one batch in flight in the timed region (default: two, pipeline.OverlappedPipeline; the roofline "
                         "kernel is then timed in a second, single-stream region of the same K steps)")
    ap.add_argument("--no-hostfed-leg", action="store_true", help="pipeline, N=1: skip the host-fed leg (`leg_hostfed`)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="join a process group even at N = 1 (a one-rank RCCL communicator): lets a single-GPU box execute the "
                         "nccl branch -- init, barrier, max all-reduce, count all-gather -- that N > 1 runs use")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test switch for boxes with fewer GPUs than ranks: rank r drives device r %% device_count and the collectives "
                         "run over gloo (RCCL refuses two ranks on one device); the N > 1 code path with real device work, "
                         "the rate itself is meaningless and the line says so")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="CPU test switch: host stand-in for the device work, gloo for RCCL; the reported numbers are meaningless")
    ap.add_argument("--traffic-json", default=str(REPO / "profiles" / "traffic_latest.json"))
    a = ap.parse_args()
    if a.workload == "fastnsf":
        a.steps = 5 if a.steps is None else a.steps
        a.warmup = 1 if a.warmup is None else a.warmup
        a.frames_per_step = 1
    elif a.workload == "train":
        a.steps = 10 if a.steps is None else a.steps
        a.warmup = 2 if a.warmup is None else a.warmup
        a.frames_per_step = 1 if a.frames_per_step is None else a.frames_per_step
        if a.train_batch > 1:
            a.frames_per_step = a.train_batch                   # a step = one pass over train_batch samples
    elif a.workload == "pipeline":
        a.steps = 100 if a.steps is None else a.steps           # 1600 frames: ~2.4 s timed -- long enough for an outside observer's
                                                                # utilisation samples to see the region; 20 steps give the same rate
        a.warmup = 3 if a.warmup is None else a.warmup
        # 16 samples per backbone launch (16 GB of activation buffers of the 288): +3 % over 8 -- the low-resolution
        # layers and the batched head get whole rounds of blocks; 24 / 32 add under 1 % more
        a.frames_per_step = 16 if a.frames_per_step is None else a.frames_per_step
    else:
        a.steps = 50 if a.steps is None else a.steps
        a.warmup = 5 if a.warmup is None else a.warmup
        a.frames_per_step = 256 if a.frames_per_step is None else a.frames_per_step
    return a


# ------------------------------------------------------------------------------------------------------
# synthetic inputs, generated on the device with the distributions of SURVEY.md 8(d)
# ------------------------------------------------------------------------------------------------------
def _poses(n, rng):
    pose0 = np.tile(np.eye(4), (n, 1, 1))
    pose1 = np.tile(np.eye(4), (n, 1, 1))
    yaw = np.deg2rad(rng.uniform(-2, 2, n))
    pose1[:, 0, 0], pose1[:, 0, 1], pose1[:, 1, 0], pose1[:, 1, 1] = np.cos(yaw), -np.sin(yaw), np.sin(yaw), np.cos(yaw)
    pose1[:, 0, 3], pose1[:, 1, 3] = rng.uniform(-3, 3, n), rng.uniform(-0.5, 0.5, n)
    return pose0, pose1


def _sweep(n, g, device):
    import torch
    lo = torch.tensor([-51.2, -51.2, -3.0, 0.0], device=device)
    hi = torch.tensor([51.2, 51.2, 3.0, 1.0], device=device)
    return torch.rand((n, 4), generator=g, device=device, dtype=torch.float32) * (hi - lo) + lo


def synthetic_batch(n_frames: int, n_points: int, device, seed: int):
    """Ragged-batch container for the comp_dis workload (uniform xyz in the network range, intensity U[0,1],
    lidar_dt U[0,0.1], yaw <= 2 deg + translation <= 3 m ego motion, flow ~ N(0,1) m per sweep)."""
    import torch
    from himo_amd.compdis import FrameBatch
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    T = n_frames * n_points
    pc0 = _sweep(T, g, device)
    flow = torch.randn((T, 3), generator=g, device=device, dtype=torch.float32)
    lidar_dt = torch.rand(T, generator=g, device=device, dtype=torch.float32) * 0.1
    pose0, pose1 = _poses(n_frames, np.random.default_rng(seed))
    offsets = np.arange(n_frames + 1, dtype=np.int64) * n_points
    return FrameBatch(offsets_host=offsets, offsets=torch.from_numpy(offsets).to(device),
                      pose0=torch.from_numpy(pose0).to(device), pose1=torch.from_numpy(pose1).to(device),
                      pc0=pc0, lidar_dt=lidar_dt, flow=flow)


def synthetic_sample_sets(n_sets: int, n_frames: int, n_points: int, device, seed: int, cloud: str = "uniform"):
    """``n_sets`` distinct batches of B network inputs.  Every sweep is a ``himo_amd.synthetic.make_frame`` frame (SURVEY.md
    8(d): seeded by its frame index, ~30 box-shaped moving instances, yaw <= 2 deg + <= 3 m ego motion, lidar_dt U[0,0.1]);
    sample j of a set is three consecutive frames (history, pc0, pc1) of a sliding window, so a set costs B + 2 frames.
    Returns (sets of Samples on ``device``, the host frames of set 0's first sample for the parity / CPU legs)."""
    from himo_amd.pipeline import Sample
    from himo_amd.synthetic import make_frame
    sets, first = [], None
    for k in range(n_sets):
        base = 100_000 * seed + 1_000 * k
        frames = [make_frame(base + i, n_points=n_points, cloud=cloud) for i in range(n_frames + 2)]
        if first is None:
            first = frames
        sets.append([Sample.from_frames(frames[j], frames[j + 1], frames[j + 2], device=device) for j in range(n_frames)])
    return sets, first


def frame_to_host(batch, k: int) -> dict:
    o = batch.offsets_host
    s = slice(int(o[k]), int(o[k + 1]))
    return {"pc0": batch.pc0[s].cpu().numpy(), "seflowpp_best": batch.flow[s].cpu().numpy(),
            "lidar_dt": batch.lidar_dt[s].cpu().numpy(), "pose0": batch.pose0[k].cpu().numpy(),
            "pose1": batch.pose1[k].cpu().numpy()}


# ------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle is the thing timed here, never the thing shipped)
# ------------------------------------------------------------------------------------------------------
def cpu_baseline_compdis(frames: list[dict], budget_s: float, exact_frames: int = 0) -> dict:
    """numpy port of save_zip.py:113-121 incl. the f32 cast (pinned against the reference's own output), timed per
    BASELINE.md section 3: median after warm-up, BLAS pool at its default size and limited to 1 thread."""
    sys.path.insert(0, str(REPO / "oracle"))
    import himo_oracle as oracle
    run_one = lambda i: oracle.comp_dis_frame_f32(frames[i], "seflowpp_best")
    legs = {"all_cores": _median_rate(run_one, len(frames), budget_s / 2, exact_frames, est_s_per_frame=0.02)}
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            legs["one_thread"] = _median_rate(run_one, len(frames), budget_s / 2, exact_frames, est_s_per_frame=0.02)
    except ImportError:                                   # pragma: no cover
        legs["one_thread"] = legs["all_cores"]
    a, o = legs["all_cores"], legs["one_thread"]
    return {"value": a["frames_per_s"], "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
            "value_1_thread": o["frames_per_s"], "all_cores": a, "one_thread": o, "host_logical_cores": os.cpu_count(), "cpu_quota_cores": cpu_quota_cores(),
            "sample": f"median of {a['frames_timed']} x {len(frames[0]['pc0'])}-pt frames (default BLAS threads) and of "
                      f"{o['frames_timed']} (1 thread) after warm-up: numpy oracle/himo_oracle.py comp_dis_frame_f32 "
                      f"(the arithmetic is element-wise numpy: the thread count barely matters)"}


PROTOCOL_WARMUPS, PROTOCOL_FRAMES = 5, 50


def _median_rate(run_one, n_frames_avail: int, budget_s: float, exact_frames: int, est_s_per_frame: float | None = None):
    """warm-up frames, then the MEDIAN per-frame time of the measured ones.  ``exact_frames`` > 0:
    5 warm-ups + exactly that many frames.  Otherwise the full protocol (5 + 50) when ``est_s_per_frame`` says it fits into
    ``budget_s``; else 1 warm-up + as many frames as fit (at least 3) -- the returned dict says which."""
    full = exact_frames > 0 or (est_s_per_frame is not None and
                                est_s_per_frame * (PROTOCOL_WARMUPS + PROTOCOL_FRAMES) <= budget_s)
    n_exact = exact_frames if exact_frames > 0 else (PROTOCOL_FRAMES if full else 0)
    warm = PROTOCOL_WARMUPS if full else 1
    for i in range(warm):
        run_one(i % n_frames_avail)
    times, t_start = [], time.perf_counter()
    while True:
        t0 = time.perf_counter()
        run_one((warm + len(times)) % n_frames_avail)
        times.append(time.perf_counter() - t0)
        if n_exact > 0:
            if len(times) >= n_exact:
                break
        elif len(times) >= 3 and (time.perf_counter() - t_start + times[-1] > budget_s or len(times) >= PROTOCOL_FRAMES):
            break
    med = float(np.median(times))
    return {"frames_per_s": 1.0 / med, "ms_per_frame": med * 1e3, "frames_timed": len(times), "warmup_frames": warm,
            "ms_min": float(np.min(times)) * 1e3, "ms_max": float(np.max(times)) * 1e3,
            "protocol": ("BASELINE.md section 3 in full (5 warm-ups, median of >= 50 frames)" if warm >= PROTOCOL_WARMUPS and len(times) >= PROTOCOL_FRAMES
                         else f"SHORT of BASELINE.md section 3 (asks 5 warm-ups + 50 frames): {warm} warm-up(s) + {len(times)} frames fitted the "
                              f"{budget_s:.0f} s budget of this leg")}


def cpu_baseline_pipeline(host_samples, params, budget_s: float, exact_frames: int = 0, threads_all: int = 0) -> dict:
    """PyTorch-CPU float32 restatement of the network (oracle/seflow_oracle.py) + numpy comp_dis on 120k-point frames, timed
    per BASELINE.md section 3 at all cores AND at 1 thread (median after warm-up; core count stated).  ``host_samples``:
    [(history frame, frame, next frame)] of host dicts.  PARITY UNPINNED: this is the build's own restatement, not the
    reference's code (absent)."""
    import torch
    sys.path.insert(0, str(REPO / "oracle"))
    import himo_oracle as oracle
    import seflow_oracle as so

    def run_one(i):
        fh, f0, f1 = host_samples[i]
        flow = so.forward(params, fh["pc0"], f0["pc0"], f1["pc0"], fh["pose0"], f0["pose0"], f0["pose1"])
        oracle.comp_dis_frame_f32(dict(f0, seflowpp_best=flow), "seflowpp_best")

    default_threads = torch.get_num_threads()
    legs, probe = {}, {}
    try:
        if not threads_all:
            # "all cores" = the thread count at which this host runs the restatement FASTEST, found by one frame each: on the
            # 2 x 64-core GPU box torch's default (128 threads) is 3x slower than 16-32 threads -- the element-wise ops between
            # the convolutions are memory-bound and oversubscribed -- and a baseline slowed by its own thread pool is no baseline
            for thr in sorted({default_threads, 64, 32, 16} & set(range(1, default_threads + 1)), reverse=True):
                torch.set_num_threads(thr)
                run_one(0)
                t0 = time.perf_counter()
                run_one(1 % len(host_samples))
                probe[thr] = time.perf_counter() - t0
            threads_all = min(probe, key=probe.get)
        est = probe.get(threads_all)
        for name, thr, budget in (("all_cores", threads_all, budget_s), ("one_thread", 1, budget_s / 3.0)):
            torch.set_num_threads(thr)
            legs[name] = dict(_median_rate(run_one, len(host_samples), budget, exact_frames, est if thr == threads_all else None), threads=thr)
    finally:
        torch.set_num_threads(default_threads)
    a, o = legs["all_cores"], legs["one_thread"]
    return {"value": a["frames_per_s"], "unit": "frames/s", "cores": a["threads"], "kind": "port",
            "value_1_thread": o["frames_per_s"], "all_cores": a, "one_thread": o, "host_logical_cores": os.cpu_count(), "cpu_quota_cores": cpu_quota_cores(),
            "thread_probe_s_per_frame": {str(k): round(v, 3) for k, v in probe.items()},
            "sample": f"median of {a['frames_timed']} frame(s) after {a['warmup_frames']} warm-up(s) at {a['threads']} torch threads "
                      f"(the fastest of {sorted(probe) or [a['threads']]} on this host) "
                      f"({a['ms_per_frame']:.0f} ms/frame) and of {o['frames_timed']} at 1 thread ({o['ms_per_frame']:.0f} ms/frame); "
                      f"a frame = 3 x {len(host_samples[0][1]['pc0'])}-point sweeps through the PyTorch-CPU fp32 restatement "
                      f"(oracle/seflow_oracle.py) + numpy comp_dis; host has {os.cpu_count()} logical cores.  All-cores leg: {a['protocol']}; "
                      f"1-thread leg: {o['protocol']}"}


def reduce_job(elapsed: float, frames_done: int, device, world: int, rank: int):
    """Max-over-ranks wall time and the whole-job frame count.  The frame counts travel through the path's only
    exchange, the final gather to rank 0 (RCCL on GPUs; gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        device = torch.device("cpu")                    # gloo gathers host tensors (CPU tests, --share-gpu)
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    done = torch.tensor([frames_done], device=device, dtype=torch.int64)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros_like(done) for _ in range(world)]
        dist.all_gather(gathered, done)                 # every backend implements all_gather; rank 0 reports
        per_rank = [int(g.item()) for g in gathered]
    else:
        per_rank = [int(done.item())]
    return float(el.item()), sum(per_rank), per_rank




# ------------------------------------------------------------------------------------------------------
# launch: one process per GPU
# ------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    """``python bench.py --gpus N`` outside a torchrun environment: start N ranks of this file, one per GPU, under
    ``python -m torch.distributed.run`` on 127.0.0.1 (the driver's own multi-GPU command line) and pass their exit code
    on.  Rank 0's JSON line goes to this process's stdout unchanged."""
    import socket
    import subprocess
    if not args.dry_run_cpu and not args.share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _dry_run_step(args, rank: int, world: int):
    """Host stand-in for the device work of --dry-run-cpu (tests only): a few microseconds of numpy per frame, and for the
    train workload the real data-parallel exchange -- ``allreduce_mean_`` on a flat float32 buffer of the parameter count."""
    import torch
    flat = torch.full((1 << 16,), float(rank + 1)) if args.workload == "train" else None
    sink = np.zeros(8)

    def step():
        sink[:] = np.sqrt(np.arange(8.0) + args.frames_per_step)
        if flat is not None:
            from himo_amd.seflow.train import allreduce_mean_
            flat.fill_(float(rank + 1))
            allreduce_mean_(flat)
            want = (world + 1) / 2.0
            if abs(float(flat[0]) - want) > 1e-6:
                raise RuntimeError(f"all-reduce mean {float(flat[0])} != {want}")
    return step


def cpu_quota_cores():
    """CPUs this process may use at once according to its control group (cgroup v2 ``cpu.max`` / v1 ``cpu.cfs_quota_us``): the pool's
    boxes show 256 logical cores and allow 16 -- which is where the CPU baseline's thread probe peaks and what caps every host-side
    stage that runs in parallel (reader processes, staging threads).  None when unlimited or unreadable."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else round(quota / period, 2)
    except Exception:
        return None


def main() -> int:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); "
                         "they must agree (value is the whole-job rate over n_gpus)")
    import torch
    import torch.distributed as dist

    dry = args.dry_run_cpu
    if dry:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device; there is no CPU path to benchmark")
        if args.share_gpu:
            local_rank %= torch.cuda.device_count()
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} HIP device(s) visible")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    grouped = world > 1 or args.force_process_group
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver
        if dry or args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)        # "nccl" IS RCCL on ROCm
    try:
        line = run_rank(args, rank, world, device, sync)
        if line is not None:
            line["config"]["collectives"] = (dist.get_backend() + f" x{dist.get_world_size()}") if grouped else "none (single process)"
            if args.share_gpu:
                line["config"]["shared_gpu"] = "TEST MODE: the ranks share HIP devices and exchange over gloo; `value` is not a scaling figure"
    finally:
        if grouped:
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)
    return 0


def make_fastnsf_step(args, rank: int, device, result: dict):
    """BASELINE config 4: a step fits the per-scene coordinate MLP to one 120k-point sweep pair."""
    import torch
    from himo_amd.fastnsf import FastNSF, OverlappedFastNSF
    from himo_amd.synthetic import make_frame
    P = args.points
    fr = [make_frame(100_000 * rank + 77 + i, n_points=P, cloud=args.cloud) for i in range(2)]
    p0 = torch.from_numpy(np.ascontiguousarray(fr[0]["pc0"][:, :3])).to(device)
    p1 = torch.from_numpy((fr[0]["pc0"][:, :3] + fr[0]["flow"]).astype(np.float32)).to(device)     # the sweep one step later
    # the product's way to run a stream of sweep pairs: two fits in flight on two HIP streams (fastnsf.OverlappedFastNSF; fits of
    # different pairs are independent); --single-stream keeps one engine
    nsf = None if args.single_stream else OverlappedFastNSF(device=device, engines=max(2, args.fits_in_flight), iters=args.fastnsf_iters)
    fitter = FastNSF(device=device, iters=args.fastnsf_iters) if nsf is None else nsf.engines[0]

    def step_single():
        result["flow"] = fitter.fit(p0, p1, fr[0]["pose0"], fr[0]["pose1"])

    def step():
        if nsf is None:
            return step_single()
        result["flow"] = nsf.submit(p0, p1, fr[0]["pose0"], fr[0]["pose1"])[1]      # (valid once the engine's fit is collected)
    return step, fitter, fr, nsf, step_single


def make_train_step(args, rank: int, device, result: dict, n_sets: int = 2):
    """BASELINE config 5: self-supervised training.  ``--train-batch 1`` (default): ``frames_per_step`` samples per rank per step, one
    optimiser step each; ``--train-batch B``: a step is ONE forward / backward pass over B samples (every encoder layer one launch over
    B x 3 images, BatchNorm statistics over the batch) and one optimiser step -- the reference launcher's ``batch_size=8`` on one
    process (assets/slurm/ssl-train-av2.sh:32-34).  ONE flat-gradient all-reduce over RCCL per optimiser step, Adam.  Labels: ~10 % of
    the points in 30 dynamic clusters.  ``n_sets`` distinct sample sets rotate through the steps, so the incremental pillar images meet
    cells that emptied and cells that filled (a single repeated sample would skip every empty cell: a best case)."""
    import torch
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import SeFlowTrainer
    TB = max(1, getattr(args, "train_batch", 1))
    B, P = (TB if TB > 1 else args.frames_per_step), args.points
    params = spec.init_params(0)
    trainer = SeFlowTrainer(params, device=device, max_points=P, precision=args.train_precision, batchnorm=args.train_batchnorm, batch=TB)
    sets, _ = synthetic_sample_sets(n_sets, B, P, device, seed=rank, cloud=args.cloud)
    g = torch.Generator(device=device); g.manual_seed(99 + rank)
    labels = []
    for _ in range(B):
        pick = torch.rand(P, generator=g, device=device) < 0.1
        lab = torch.randint(1, 31, (P,), generator=g, device=device, dtype=torch.int32) * pick.to(torch.int32)
        labels.append((lab, lab.clone()))
    turn = [0]

    def step():
        samples = sets[turn[0] % len(sets)]
        turn[0] += 1
        if TB > 1:
            result["loss"] = trainer.train_batch([(smp.pch1, smp.pc0, smp.pc1, smp.pose_h1, smp.pose0, smp.pose1, l0, l1, 31)
                                                  for smp, (l0, l1) in zip(samples, labels)], bucketed=True)      # (every rank: TB samples, one pass)
            return
        for smp, (l0, l1) in zip(samples, labels):
            _, total = trainer.train_step(smp.pch1, smp.pc0, smp.pc1, smp.pose_h1, smp.pose0, smp.pose1, l0, l1, n_labels=31)
            result["loss"] = total
    return step, trainer


def run_rank(args, rank: int, world: int, device, sync) -> dict | None:
    import torch
    import torch.distributed as dist
    dry = args.dry_run_cpu
    B, P = args.frames_per_step, args.points
    sys.path.insert(0, str(REPO / "oracle"))
    _lib = None
    if not dry:
        from himo_amd import _lib

    out, result, host_frames, params, pipe, batch, sets, overlapped = {}, {}, None, None, None, None, None, None
    turn = [0]
    if dry:
        step = _dry_run_step(args, rank, world)
    elif args.workload == "compdis":
        from himo_amd.compdis import CompDisEngine
        batch = synthetic_batch(B, P, device, seed=rank)
        eng = CompDisEngine(device=device, max_frames=B)

        def step():
            eng.run(batch, sensor_dt=0.1, refined=args.refined, out=out)
    elif args.workload == "fastnsf":
        step, fitter, fr, overlapped, step_single = make_fastnsf_step(args, rank, device, result)
    elif args.workload == "train":
        step, trainer = make_train_step(args, rank, device, result)
    else:
        from himo_amd.pipeline import HiMoPipeline
        from himo_amd.seflow import spec
        from himo_amd.seflow.model import SeFlowNet
        params = spec.init_params(0)
        net = SeFlowNet(params, device=device, max_points=P, precision=args.precision, max_batch=B)
        if args.float32_activations:
            net.split_acts = False
        pipe = HiMoPipeline(net, device=device)
        sets, host_frames = synthetic_sample_sets(max(1, args.sample_sets), B, P, device, seed=rank, cloud=args.cloud)
        overlapped = None
        if not args.single_stream:
            # the product's default way to run a stream of batches: several in flight, each on its own HIP stream with its own network buffers
            from himo_amd.pipeline import OverlappedPipeline
            more = [SeFlowNet(params, device=device, max_points=P, precision=args.precision, max_batch=B) for _ in range(max(2, args.batches_in_flight) - 1)]
            for nb in more:
                nb.split_acts = net.split_acts
            overlapped = OverlappedPipeline(nets=[net] + more, device=device)
            pipe = overlapped.pipes[0]                          # (the single-stream region below runs this very pipeline)

        def step_single():
            result.update(pipe.run(sets[turn[0] % len(sets)], sensor_dt=0.1, refined=args.refined))
            turn[0] += 1

        def step():
            # a DIFFERENT batch every step (the sets rotate): the ragged batch container -- point / lidar_dt concatenation,
            # offsets + poses upload -- is rebuilt inside the timed region, as a stream of fresh frames would make it
            if overlapped is None:
                return step_single()
            result.update(overlapped.run(sets[turn[0] % len(sets)], sensor_dt=0.1, refined=args.refined))
            turn[0] += 1

    step()                                      # priming pass on every rank (one-off tile autotune, operator-list recording,
    if overlapped is not None:                  # workspace growth): never inside the timed region, whatever --warmup is
        for _ in range(len(getattr(overlapped, "pipes", getattr(overlapped, "engines", [0, 0]))) - 1):
            step()                              # (the other network(s) of the batches-in-flight pipeline / the other FastNSF engine)
    sync()
    for _ in range(args.warmup):
        step()
    sync()

    # parity spot-check on rank 0 (not timed): EPE / max abs vs the CPU oracle on one frame
    parity, ref_flow = None, None
    if rank == 0 and not dry:
        import himo_oracle as oracle
        if args.workload == "compdis":
            step()
            sync()
            f = frame_to_host(batch, 0)
            ref = oracle.comp_dis_frame_f32(f, "seflowpp_best")
            got = out["comp_dis"][:P].cpu().numpy()
            d = got.astype(np.float64) - ref
            parity = {"comp_dis_mean_epe_vs_ref": float(np.linalg.norm(d, axis=1).mean()),
                      "comp_dis_max_abs_vs_ref": float(np.abs(d).max()), "bit_exact_fraction": float((got == ref).mean())}
        elif args.workload == "fastnsf":
            if overlapped is not None:
                overlapped.sync_check()                         # the fits in flight: loss histories read, flows final
            gt = fr[0]["flow"]
            got = result["flow"].cpu().numpy()
            parity = {"loss_first_to_last_iteration": [fitter.loss_history[0][1], fitter.loss_history[-1][1]],
                      "flow_mean_epe_vs_generating_flow": float(np.linalg.norm(got - gt, axis=1).mean()),
                      "note": "gradient / trajectory parity vs the CPU restatement (oracle/fastnsf_oracle.py, unpinned): tests/test_fastnsf_gpu.py"}
        elif args.workload == "train":
            parity = {"loss_after_warmup": float(result["loss"].item()),
                      "note": "gradient parity vs CPU autograd through the oracle network: tests/test_train_gpu.py"}
        elif not args.no_cpu_baseline:
            import seflow_oracle as so
            fh, f0, f1 = host_frames[0], host_frames[1], host_frames[2]
            ref_flow = so.forward(params, fh["pc0"], f0["pc0"], f1["pc0"], fh["pose0"], f0["pose0"], f0["pose1"])
            ref_cd = oracle.comp_dis_frame_f32(dict(f0, seflowpp_best=ref_flow), "seflowpp_best")
            res = pipe.run(sets[0], sensor_dt=0.1, refined=args.refined)
            sync()
            got_flow, got_cd = res["flow"][:P].cpu().numpy(), res["comp_dis"][:P].cpu().numpy()
            parity = {"flow_mean_epe_vs_cpu_restatement": float(np.linalg.norm(got_flow - ref_flow, axis=1).mean()),
                      "flow_max_abs_vs_cpu_restatement": float(np.abs(got_flow - ref_flow).max()),
                      "comp_dis_max_abs_vs_cpu_restatement": float(np.abs(got_cd.astype(np.float64) - ref_cd).max()),
                      "note": "network parity is against this build's own CPU restatement (reference source absent)"}

    # The roofline kernel is timed live inside the timed region (HIP events around each of ITS launches, on the launch
    # stream); the other kernels are left alone there -- two event records per launch on ~45 launches per frame cost
    # ~12 % of the frame rate -- and get their table from one extra, untimed, fully profiled step afterwards.
    # train: the 3x3 weight gradients are the largest kernel family of the step in every precision mode (mixed: split-bf16
    # operands on the stride-1 layers; bf16x3 / f32: float32 matrix instructions)
    dominant = {"compdis": "compdis_kernel", "train": TRAIN_DOMINANT, "fastnsf": FASTNSF_DOMINANT}.get(
        args.workload, {"bf16x3": "conv3x3_bf16x3_kernel", "f16x2": "conv3x3_f16x2_kernel"}.get(args.precision, "conv3x3_mfma_kernel"))
    grouped = dist.is_available() and dist.is_initialized()
    if grouped:
        dist.barrier()
    sync()
    import gc
    gc.collect()
    gc.disable()                                # no collector pauses inside the timed region
    n_threads = torch.get_num_threads()
    if parity is not None:
        # the CPU restatement of the parity check just ran on every host core: let its worker threads park before the
        # launch thread is timed (spinning OpenMP workers otherwise cost ~5 % of the frame rate)
        torch.set_num_threads(1)
        time.sleep(1.0)
    train_overlap = args.workload == "train" and not dry and trainer.overlap_wgrad
    if _lib is not None and overlapped is None and not train_overlap:
        _lib.prof_start(only=dominant)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if overlapped is not None:
        overlapped.sync_check()
    elif pipe is not None:
        pipe.sync_check()                       # the last batch's finite-flow flag (fp16-split precision)
    sync()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    single = None
    if overlapped is not None:
        # the roofline kernel alone: the SAME K steps with one batch in flight (with two, a launch's HIP-event time includes
        # whatever the other stream co-runs, and `roofline` is a statement about that kernel)
        for _ in range(2):
            step_single()
        sync()
        _lib.prof_start(only=dominant)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_single()
        if pipe is not None:
            pipe.sync_check()
        sync()
        single = time.perf_counter() - t1
    elif train_overlap:
        # the training step's roofline kernel alone: the SAME K steps with the weight gradients back on the main stream
        trainer.set_side_streams(False)
        for _ in range(2):
            step()
        sync()
        _lib.prof_start(only=dominant)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        single = time.perf_counter() - t1
        trainer.set_side_streams(True)
    gc.enable()
    torch.set_num_threads(n_threads)
    prof = _lib.prof_stop() if _lib is not None else {}
    all_kernels = {}
    if _lib is not None:
        # one extra, untimed step with every kernel timed (rank 0's table).  EVERY rank runs it: the train workload's step
        # contains the gradient all-reduce, and a collective only rank 0 entered would never complete
        if rank == 0:
            _lib.prof_start()
        (step_single if overlapped is not None else step)()
        sync()
        if rank == 0:
            all_kernels = _lib.prof_stop()

    elapsed, total_frames, per_rank = reduce_job(elapsed, B * args.steps, device, world, rank)
    if rank != 0:
        return None

    traffic = {}
    try:
        traffic = json.loads(Path(args.traffic_json).read_text())
    except Exception:
        pass
    per_kernel = {n: {"avg_ms": v["avg_ms"], "launches_per_step": v["count"], "ms_per_step": v["total_ms"]}
                  for n, v in all_kernels.items()}
    extra = {}
    if dry:
        roofline = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        workload, dtype = f"DRY RUN on the CPU ({args.workload}): no device work was done, the numbers are meaningless", "none"
    elif args.workload == "compdis":
        bytes_per_pt = 44 + (12 if args.refined else 0)     # xyzi 16 + flow 12 + dt 4 + comp_dis 12 [+ refined 12]
        k = prof.get("compdis_kernel", {"avg_ms": float("nan"), "count": 0})
        achieved = bytes_per_pt * B * P / (k["avg_ms"] * 1e-3) / 1e9 if k["count"] else float("nan")
        roofline = {"bound": "hbm", "kernel": "compdis_kernel<4,f64>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic.get("compdis_kernel", {}).get("hbm_bytes_per_launch"),
                    "algorithmic_bytes_per_launch": bytes_per_pt * B * P, "avg_launch_ms": k["avg_ms"],
                    "launches_timed": k["count"]}
        workload = ("flow->comp_dis fused path only (a1-a4: ego-motion removal, dt0, flow2compDis; f64 chain, f32 I/O) "
                    "over a ragged HBM-resident batch; network forward NOT included")
        dtype = "f64"
    elif args.workload == "fastnsf":
        roofline, workload, dtype = fastnsf_roofline(args, prof, args.steps, single if single is not None else elapsed)
        if single is not None:
            roofline["measured_in"] = ("a second region of the same K fits on ONE engine (un-overlapped launches), right after the timed "
                                       "region; `value` is the two-fits-in-flight rate")
            extra["value_single_stream"] = args.steps / single
            extra["leg_single_stream"] = {"frames_per_s": args.steps / single, "ms_per_step": single / args.steps * 1e3, "steps": args.steps,
                                          "note": "FastNSF.fit, one fit at a time: the configuration of the earlier rounds' figures; the "
                                                  "roofline kernel's launches are timed here"}
    elif args.workload == "train":
        roofline, workload, dtype = train_roofline(args, prof, B * args.steps, single if single is not None else elapsed)
        # the parameter BITS this rank ends with (every rank holds the same): two runs that must agree bit for bit compare this
        extra["param_bits_sum_at_end"] = int(trainer.flat_p.view(torch.int32).to(torch.int64).sum().item())
        extra["gradient_exchange"] = ("bucket by bucket under the backward pass (BucketedAllReduce)" if (args.train_batch > 1 and trainer.overlap_allreduce)
                                      else "one flat all-reduce after the backward pass")
        if single is not None:
            roofline["measured_in"] = ("a second region of the same K steps with every weight gradient on the main stream (un-overlapped "
                                       "launches), right after the timed region; `value` is the rate with the two side streams")
            extra["value_single_stream"] = B * args.steps / single
    else:
        roofline, dtype = conv_roofline(args.precision, prof, B, args.steps, single if single is not None else elapsed, traffic,
                                        folded=pipe.net.fold_decoder)
        if single is not None:
            roofline["measured_in"] = ("a second region of the same K steps with ONE batch in flight (un-overlapped launches), right after the "
                                       f"timed region; `value` is the rate with {len(overlapped.pipes)} batches in flight")
            extra["value_single_stream"] = B * args.steps / single
            extra["leg_single_stream"] = {"frames_per_s": B * args.steps / single, "ms_per_step": single / args.steps * 1e3, "steps": args.steps,
                                          "note": "HiMoPipeline, one batch in flight: the round-3 headline configuration; the roofline "
                                                  "kernel's launches are timed here"}
        if rank == 0 and _lib is not None:
            # what the matrix pipes of THIS box sustain on their own (register-resident chains, no memory traffic; ~0.4 s per
            # leg, outside the timed region): the clock is power-managed and the power of a matrix instruction depends on its
            # operand bits, so the nameplate `peak` is a zero-operand figure
            kind = {"f16x2": "f16", "bf16x3": "bf16", "f32": "f32"}[args.precision]
            rnd, zer = _lib.mfma_sustained_tflops(kind, False), _lib.mfma_sustained_tflops(kind, True)
            issued = roofline["issued_matrix_tflops"]
            roofline["sustained_matrix_rate"] = {
                "random_operands_tflops": rnd, "zero_operands_tflops": zer, "issued_over_random_operand_rate": issued / rnd,
                "note": "himo_mfma_sustained_tflops on this device right after the timed region: independent v_mfma chains from "
                        "registers on every SIMD; `issued_matrix_tflops` / random-operand rate = the share of the chip's "
                        "data-carrying matrix rate the convolution kernels reach while also moving their operands"}
        workload = ("per-frame pipeline: pillarise 3 sweeps (512x512 grid) -> SeFlow++-style encoder/decoder + GRU head "
                    "(random-init, self-specified: reference network source absent) -> per-point flow -> ego-motion "
                    "removal + dt0 + flow2compDis -> comp_dis")
        # the whole step against the HBM roofline (BASELINE.json's metric asks for the achieved fraction): bytes every operator
        # of the executed graph must move once (himo_amd/seflow/spec.py algorithmic_bytes_per_frame) / measured step time
        from himo_amd.seflow import spec as _spec
        parts = _spec.algorithmic_bytes_per_frame(P, folded=pipe.net.fold_decoder)
        step_s = elapsed / args.steps
        gbs = parts["total"] * B / step_s / 1e9
        roofline["step_hbm"] = {"algorithmic_bytes_per_step": parts["total"] * B, "achieved": gbs, "GB/s": gbs, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "algorithmic_MB_per_frame": {k: round(v / 1e6, 1) for k, v in parts.items()},
                                "note": "every operator of the executed graph reads its inputs and writes its outputs once, 4 B per feature-map "
                                        "value, weights L2-resident; the step is bound by the matrix rate of the 3x3 convolutions (`roofline`), "
                                        "not by HBM: at 8 TB/s these bytes would take "
                                        f"{parts['total'] * B / HBM_PEAK_GBS / 1e6:.1f} ms of the {step_s * 1e3:.1f} ms step"}
        roofline["traffic_source"] = (f"{Path(args.traffic_json).name}: builder-run rocprofv3 PMC passes of this workload (scripts/collect_traffic.sh: "
                                      "FETCH_SIZE x2 per the gfx950 note + WRITE_SIZE, separate passes, per launch of the tuned kernels) -- "
                                      "a cross-reference read from profiles/, NOT measured in this run") if roofline.get("traffic") else None
        if world == 1 and not args.no_extra_precisions:
            extra.update(extra_precision_legs(args, params, sets, device, ref_flow, exclude=args.precision))
        if world == 1 and not args.no_extra_workloads:
            extra.update(extra_workload_legs(args, device))
        if world == 1 and not args.no_hostfed_leg:
            extra.update(hostfed_leg(args, overlapped if overlapped is not None else pipe, host_frames, device))
            extra.update(h5fed_leg(args, overlapped if overlapped is not None else pipe, host_frames, device))
        if world == 1 and not args.no_extra_workloads and args.leg_fit_steps > 0:
            extra.update(fit_h5_leg(args, device))
        if world == 1 and not args.no_hostfed_leg:
            extra.update(eval_h5_leg(args, device))
    line = {
        "metric": "lidar_frames_per_sec_120k", "value": total_frames / elapsed, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic" if not dry else "none (dry run on the CPU: INVALID as a measurement)",
        "config": {"workload": workload, "frames_per_step_per_gpu": B, "points_per_frame": P,
                   "sweeps_per_frame": 3 if args.workload == "pipeline" else 1,
                   "parallelism": f"frames sharded x{world}", "refined_output": bool(args.refined),
                   "frames_per_rank": per_rank},
        "roofline": roofline, "kernels": per_kernel,
        "kernels_note": "one extra untimed step with every kernel timed; the timed region times only the roofline kernel",
        "parity": parity,
    }
    line.update(extra)
    if args.workload == "pipeline":
        line["config"]["batches_in_flight"] = 1 if overlapped is None else len(overlapped.pipes)
        line["config"]["matrix_arithmetic"] = args.precision
        line["config"]["samples_per_backbone_launch"] = B
        line["config"]["input"] = (f"himo_amd.synthetic.make_frame sweeps (SURVEY 8(d) seeded frames, cloud={args.cloud}); "
                                   f"{max(1, args.sample_sets)} distinct batches rotate through the steps, so the ragged batch "
                                   "container is rebuilt from different samples inside every timed step")
    if args.workload == "fastnsf":
        line["metric"] = "fastnsf_frames_per_sec_120k"
        line["config"]["iterations_per_frame"] = args.fastnsf_iters
        line["config"]["fits_in_flight"] = 1 if overlapped is None else len(overlapped.engines)
    if args.workload == "train":
        line["metric"] = "train_frames_per_sec_120k"
        line["config"]["parallelism"] = f"data parallel x{world}, one flat all-reduce per step"
        line["config"]["matrix_arithmetic"] = args.train_precision
        line["config"]["batchnorm"] = args.train_batchnorm
        line["config"]["samples_per_pass_and_optimiser_step"] = args.train_batch
    if not dry and not args.no_cpu_baseline and args.workload in ("pipeline", "compdis") and world == 1:     # CPU leg: rank 0 at N = 1 only
        if args.workload == "compdis":
            frames = [frame_to_host(batch, i) for i in range(min(8, B))]
            line["cpu_baseline"] = cpu_baseline_compdis(frames, args.cpu_seconds, args.cpu_frames)
        else:
            host_samples = [(host_frames[j], host_frames[j + 1], host_frames[j + 2]) for j in range(min(B, len(host_frames) - 2))]
            line["cpu_baseline"] = cpu_baseline_pipeline(host_samples, params, args.cpu_seconds, args.cpu_frames, args.cpu_threads)
        line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    line["summary"] = line_summary(line)
    return line


def line_summary(line: dict) -> dict:
    """The line's figures once more, compact, as its LAST key: a reader that keeps only the end of a long line (the driver stores the last
    2000 characters beside the fields it parses) still sees every leg's number.  Nothing here is new: each entry repeats a value above."""
    r3 = lambda v: None if v is None else round(float(v), 3)
    g = lambda leg, key="frames_per_s": r3((line.get(leg) or {}).get(key))
    out = {"value": r3(line.get("value")), "roofline_frac": r3((line.get("roofline") or {}).get("frac")),
           "roofline_avg_launch_ms": r3((line.get("roofline") or {}).get("avg_launch_ms")),
           "value_single_stream": r3(line.get("value_single_stream")), "value_bf16x3": r3(line.get("value_bf16x3")), "value_f32": r3(line.get("value_f32")),
           "hostfed": g("leg_hostfed"), "h5fed": g("leg_h5fed"),
           "train": g("leg_train"), "train_b8": g("leg_train_b8"), "train_bf16x3": g("leg_train_bf16x3"), "train_rings": g("leg_train_rings"),
           "train_wgrad_roofline_frac": r3(((line.get("leg_train") or {}).get("roofline") or {}).get("frac")),
           "train_b8_wgrad_roofline_frac": r3(((line.get("leg_train_b8") or {}).get("roofline") or {}).get("frac")),
           "fit_h5": g("leg_fit_h5"), "fit_h5_by_epoch": (line.get("leg_fit_h5") or {}).get("frames_per_s_by_epoch"),
           "eval_h5_sweeps_per_s": g("leg_eval_h5", "sweeps_per_s"), "eval_h5_whole_loop_sweeps_per_s": g("leg_eval_h5", "whole_loop_sweeps_per_s"),
           "eval_resident_sweeps_per_s": g("leg_eval_h5", "resident_sweeps_per_s"),
           "fastnsf": g("leg_fastnsf"), "fastnsf_one_fit_at_a_time": g("leg_fastnsf", "frames_per_s_one_fit_at_a_time"),
           "fastnsf_roofline_frac": r3(((line.get("leg_fastnsf") or {}).get("roofline") or {}).get("frac")),
           "cpu_baseline": r3((line.get("cpu_baseline") or {}).get("value")), "cpu_cores": (line.get("cpu_baseline") or {}).get("cores"),
           "gpu_over_cpu": r3(line.get("gpu_over_cpu"))}
    errors = {k: v["error"] for k, v in line.items() if k.startswith("leg_") and isinstance(v, dict) and "error" in v}
    if errors:
        out["leg_errors"] = errors
    return {k: v for k, v in out.items() if v is not None}


FASTNSF_DOMINANT = "nsf_"          # nsf_forward / nsf_backward / nsf_update_kernel (csrc/nsffused.hip): substring filter of himo_prof_filter


def fastnsf_roofline(args, prof: dict, n_fits: int, elapsed: float, traffic: dict | None = None):
    """roofline object / workload / dtype of the FastNSF fit (csrc/nsffused.hip: three launches per iteration).  The dominant kernel
    is nsf_backward_kernel -- the chain of input gradients AND all weight gradients: 14 of the iteration's 21 products of
    128 x 128 per point -- which is matrix-bound; the forward kernel's figure (an HBM stream: it spills the activations) and the
    update kernel's ride along."""
    from himo_amd.fastnsf import HIDDEN, N_HIDDEN
    P = args.points
    if traffic is None:
        try:
            traffic = json.loads(Path(args.traffic_json).read_text())
        except Exception:
            traffic = {}
    nil = {"avg_ms": float("nan"), "count": 0, "total_ms": float("nan")}
    f, b, u = prof.get("nsf_forward_kernel", nil), prof.get("nsf_backward_kernel", nil), prof.get("nsf_update_kernel", nil)
    map_bytes = 4.0 * P * HIDDEN                                                           # one H_k: two bf16 planes
    blocks = -(-P // 256)
    params = 2 * (4 * HIDDEN + HIDDEN) + (N_HIDDEN - 1) * (HIDDEN * HIDDEN + HIDDEN) + 4 - HIDDEN
    bytes_fwd = (N_HIDDEN - 1) * map_bytes + N_HIDDEN * 4.0 * P + 3 * 16.0 * P             # H_0..H_6 spilled + mask bits; x0 read, out + dout written
    bytes_bwd = (N_HIDDEN - 1) * map_bytes + N_HIDDEN * 4.0 * P + 2 * 16.0 * P + 4.0 * blocks * params    # H_0..H_6 read (both planes) + mask bits; partials written
    bytes_upd = 4.0 * blocks * params + 7 * 4.0 * params
    flops_bwd = 2.0 * P * HIDDEN * HIDDEN * 2 * (N_HIDDEN - 1)                              # input gradients + weight gradients of the 7 hidden products
    flops_fwd = 2.0 * P * HIDDEN * HIDDEN * (N_HIDDEN - 1)
    tf = lambda fl, k: fl / (k["avg_ms"] * 1e-3) / 1e12 if k["count"] else float("nan")
    gbs = lambda by, k: by / (k["avg_ms"] * 1e-3) / 1e9 if k["count"] else float("nan")
    peak = MFMA_BF16_PEAK_TF / 3.0
    total_ms = f["total_ms"] + b["total_ms"] + u["total_ms"]
    roofline = {"bound": "mfma", "kernel": "nsf_backward_kernel (csrc/nsffused.hip: per block of 256 points the whole chain of input gradients and every "
                                           "weight gradient; v_mfma_f32_32x32x16_bf16 with two-term bf16 operands, 3 per float32 product block; "
                                           "weight-gradient operands straight from accumulator-layout registers / spilled fragments)",
                "achieved": tf(flops_bwd, b), "peak": peak, "peak_note": f"{MFMA_BF16_PEAK_TF:.0f} TFLOP/s dense bf16 MFMA peak / 3 matrix products per float32 product",
                "unit": "TFLOP/s", "frac": tf(flops_bwd, b) / peak, "traffic": traffic.get("nsf_backward_kernel", {}).get("hbm_bytes_per_launch"),
                "algorithmic_flops_per_launch": flops_bwd, "avg_launch_ms": b["avg_ms"], "launches_timed": b["count"],
                "hbm_side": {"algorithmic_bytes_per_launch": bytes_bwd, "GB/s": gbs(bytes_bwd, b), "frac_of_hbm_peak": gbs(bytes_bwd, b) / HBM_PEAK_GBS},
                "traffic_source": ("traffic_latest.json: builder-run rocprofv3 PMC passes (scripts/collect_traffic.sh), a cross-reference read from "
                                   "profiles/, NOT measured in this run") if traffic.get("nsf_backward_kernel") else None,
                "forward_kernel": {"kernel": "nsf_forward_kernel (whole forward pass + distance-transform objective; spills H_k as bf16 fragments)",
                                   "traffic": traffic.get("nsf_forward_kernel", {}).get("hbm_bytes_per_launch"),
                                   "avg_launch_ms": f["avg_ms"], "launches_timed": f["count"], "algorithmic_bytes_per_launch": bytes_fwd,
                                   "GB/s": gbs(bytes_fwd, f), "frac_of_hbm_peak": gbs(bytes_fwd, f) / HBM_PEAK_GBS,
                                   "matrix_tflops_f32_equivalent": tf(flops_fwd, f)},
                "update_kernel": {"kernel": "nsf_update_kernel (fixed-order sum of the block partials, Adam, re-pack)", "avg_launch_ms": u["avg_ms"],
                                  "launches_timed": u["count"], "algorithmic_bytes_per_launch": bytes_upd, "GB/s": gbs(bytes_upd, u)},
                "share_of_step_time": total_ms / (elapsed * 1e3),
                "launches_per_iteration": 3,
                "note": "an iteration = nsf_forward_kernel, nsf_backward_kernel, nsf_update_kernel; the distance transform of pc1 is built once per pair "
                        "(round 3: 29 launches per iteration, 49 % of the fit in eight split-K weight-gradient products)"}
    workload = (f"FastNSF (BASELINE config 4): fit the per-scene coordinate MLP (3 -> 8 x 128 -> 3) to one pair of {P}-point sweeps, "
                f"{args.fastnsf_iters} Adam iterations per frame, distance-transform objective (pc1 -> 0.1 m distance volume once per pair, "
                "trilinear lookup per iteration)")
    dtype = ("mixed: forward fp16 split (two-term, 22-bit products), input / weight gradients two-term bf16 (16 significant bits, float32 range), "
             "float32 sums and optimiser")
    return roofline, workload, dtype


TRAIN_DOMINANT = "conv_wgrad_tiled_kernel"       # the 3x3 weight gradients: largest kernel family of the step in every precision mode


def train_roofline(args, prof: dict, n_steps: int, elapsed: float):
    """roofline object / workload / dtype of the training step (``n_steps`` optimiser steps inside ``elapsed`` seconds)."""
    from himo_amd.seflow import spec
    H, W = spec.GRID
    # 2*M*N*K of the 23 3x3 layers of one sample (20 stride-1 + the 3 stride-2 ones at their output resolution): the
    # algorithmic work of the weight gradients AND of the data gradients (a transposed convolution of the same size)
    flops_w = spec.conv3x3_flops() + sum(spec.NUM_FRAMES * 2.0 * (H // d) * (W // d) * ci * co * 9
                                         for d, ci, co in ((2, 32, 64), (4, 64, 128), (8, 128, 256)))
    k = prof.get(TRAIN_DOMINANT, {"avg_ms": float("nan"), "count": 0, "total_ms": float("nan")})
    alg_tf = flops_w * n_steps / (k["total_ms"] * 1e-3) / 1e12 if k["count"] else float("nan")
    if args.train_precision == "mixed":
        kdesc = ("conv_wgrad_tiled_kernel family: 3x3 weight gradients -- the 20 stride-1 layers on v_mfma_f32_32x32x16_bf16 with two-term "
                 "split-bf16 operands (3 per float32 product block), the 3 stride-2 layers on v_mfma_f32_32x32x2_f32")
        peak, pnote = MFMA_BF16_PEAK_TF / 3.0, (f"{MFMA_BF16_PEAK_TF:.0f} TFLOP/s dense bf16 MFMA peak / 3 matrix products per float32 product "
                                                "(the stride-2 launches, 10 % of the family's flops, run float32 matrix instructions)")
    else:
        kdesc, peak, pnote = "conv_wgrad_tiled_kernel (v_mfma_f32_32x32x2_f32; 3x3 weight gradients)", MFMA_F32_PEAK_TF, "dense float32 MFMA peak"
    roofline = {"bound": "mfma", "kernel": kdesc, "achieved": alg_tf, "peak": peak, "peak_note": pnote, "unit": "TFLOP/s",
                "frac": alg_tf / peak, "traffic": None, "avg_launch_ms": k["avg_ms"], "launches_timed": k["count"],
                "algorithmic_flops_per_step": flops_w, "share_of_step_time": k["total_ms"] / (elapsed * 1e3),
                "note": "split-K over pixel tiles; the fixed-order reduction of the partials is a separate (small) kernel"}
    bn = ("BatchNorm in training mode (batch statistics per forward, gamma / beta trained, running statistics updated)"
          if args.train_batchnorm == "batch" else "BatchNorm frozen (fine-tuning convention)")
    workload = ("self-supervised TRAINING step (BASELINE config 5): pillarise 3 sweeps -> network forward with saved "
                "activations -> 4-term NN/Chamfer loss -> full backward -> flat-gradient all-reduce -> Adam; "
                f"one {args.points}-point sample per GPU per step; {bn}")
    dtype = {"mixed": "forward f16x2 (two-term fp16 split); every data-gradient product (3x3, 1x1, the head's GRU sweep) and every weight-gradient "
                      "product two-term bf16 split (16-bit operands, float32 range and sums); everything else and the optimiser f32",
             "bf16x3": "forward + data-gradient convolutions bf16x3 (split bf16, float32-class); weight gradients / optimiser f32",
             "f32": "f32"}[args.train_precision]
    return roofline, workload, dtype


def conv_roofline(precision: str, prof: dict, B: int, steps: int, elapsed: float, traffic: dict, folded: bool = True):
    """roofline object + dtype string of the pipeline workload's dominant kernel (the stride-1 3x3 convolutions) from the
    HIP-event timings of ITS launches inside the timed region."""
    from himo_amd.seflow import spec
    bf, f16 = precision == "bf16x3", precision == "f16x2"
    kname = "conv3x3_bf16x3_kernel" if bf else "conv3x3_f16x2_kernel" if f16 else "conv3x3_mfma_kernel"
    k = prof.get(kname, {"avg_ms": float("nan"), "count": 0, "total_ms": float("nan")})
    n_fwd = B * steps
    # algorithmic flops of the 20 stride-1 3x3 convolutions of one forward (2*M*N*K each), see DESIGN.md -- of the graph that
    # RUNS: with the decoder joints folded (dec1.u5 / dec2.u5 absorb the next 1x1) 362.4 GFLOP per sample, not the spec's 381.7
    flops3 = spec.conv3x3_flops(folded=folded)
    launches_per_fwd = k["count"] / max(n_fwd, 1)
    alg_tf = flops3 * n_fwd / (k["total_ms"] * 1e-3) / 1e12 if k["count"] else float("nan")
    # `achieved` = ALGORITHMIC flops (2*M*N*K float32 multiply-adds of the 20 layers) / measured kernel time.  The peak it is
    # priced against is the dense MFMA peak of the arithmetic the kernel runs in, in the same unit: a split-precision
    # kernel spends `per` matrix multiply-adds per float32 multiply-add, so its ceiling is (2.5 PF dense fp16|bf16) / per.
    if bf:
        per, note = 6.0, "v_mfma_f32_32x32x16_bf16, 6 per float32 product block (h*h, h*m, m*h, m*m, h*l, l*h)"
    elif f16:
        per, note = 3.0, "v_mfma_f32_32x32x16_f16, 3 per float32 product block (h*h, h*l, l*h)"
    else:
        per, note = 1.0, "v_mfma_f32_32x32x2_f32"
    peak = MFMA_F32_PEAK_TF if per == 1.0 else MFMA_BF16_PEAK_TF / per
    roofline = {"bound": "mfma", "kernel": f"{kname} ({note})", "achieved": alg_tf, "peak": peak, "unit": "TFLOP/s",
                "frac": alg_tf / peak,
                "peak_note": ("dense float32 MFMA peak" if per == 1.0 else
                              f"{MFMA_BF16_PEAK_TF:.0f} TFLOP/s dense 16-bit MFMA peak / {per:.0f} matrix products per float32 product"),
                "issued_matrix_tflops": per * alg_tf,
                "traffic": traffic.get("conv3x3_mfma_kernel" if precision == "f32" else "conv3x3_split_kernel", {}).get("hbm_bytes_per_launch"),
                "vs_f32_mfma_peak_157_3": alg_tf / MFMA_F32_PEAK_TF,
                "algorithmic_flops_per_launch": flops3 / max(launches_per_fwd, 1e-9), "avg_launch_ms": k["avg_ms"],
                "graph": ("executed graph: dec1.u5 / dec2.u5 folded with the next block's 1x1 conv (linear, nothing between them) "
                          "-> 362.4 GFLOP of 3x3 convolutions per sample; the unfolded specification has 381.7") if folded else "specification graph",
                "launches_timed": k["count"], "launches_per_frame": launches_per_fwd, "samples_per_launch": B,
                "share_of_step_time": k["total_ms"] / (elapsed * 1e3)}
    dtype = ("bf16x3 (three-term split bf16 on the matrix cores, float32 accumulate; float32-class accuracy)" if bf else
             "f16x2 (two-term split fp16, x = h + l with exact subnormals, on the matrix cores; float32 accumulate; ~22-bit products)"
             if f16 else "f32")
    return roofline, dtype


def extra_workload_legs(args, device) -> dict:
    """N = 1 only, after the timed pipeline region: SHORT legs of BASELINE configs 5 (training step) and 4 (FastNSF fit) at the
    same 120k points, each with the protocol of its own workload (priming pass, warm-up, K timed steps between
    synchronisations, the leg's dominant kernel timed live by HIP events) and its own `roofline` object, so the driver-run
    line carries figures for them too: ``leg_train`` / ``leg_fastnsf`` (+ ``value_train`` / ``value_fastnsf`` frames/s).
    `python bench.py --workload train|fastnsf` runs the same code as the main workload."""
    import copy
    import torch
    from himo_amd import _lib
    out = {}
    legs = [("train", args.leg_train_steps, 2), ("fastnsf", args.leg_fastnsf_fits, 1)]
    if args.train_precision == "mixed" and args.leg_train_steps > 0:
        # the same step in the float32-class arithmetic (three-term bf16 split forward and data gradients, float32 MFMA weight gradients):
        # the figure that stands beside `leg_train`'s the way `value_bf16x3` stands beside `value`
        legs.insert(1, ("train_bf16x3", max(10, args.leg_train_steps // 2), 2))
    if args.leg_train_steps > 0:
        # ... and as the launcher batches it: 8 samples per pass and optimiser step (`leg_train_b8`; frames/s = samples/s)
        legs.insert(1, ("train_b8", max(6, args.leg_train_steps // 5), 2))
    if args.cloud == "uniform":         # the training step again on LiDAR-shaped sweeps (himo_amd.synthetic.lidar_rings: crowded cells near the
        legs.insert(1, ("train_rings", args.leg_train_steps, 2))       # sensor, surfaces the other sweep lacks): `leg_train_rings` must stay near `leg_train`
    for leg_name, steps, warm in legs:
        if steps <= 0:
            continue
        name = "train" if leg_name.startswith("train") else leg_name
        a = copy.copy(args)
        a.workload, a.frames_per_step = name, 1
        if leg_name == "train_rings":
            a.cloud = "rings"
        if leg_name == "train_bf16x3":
            a.train_precision = "bf16x3"
        a.train_batch = 8 if leg_name == "train_b8" else 1
        per_step = a.train_batch                                   # samples per step of this leg
        result = {}
        try:
            torch.cuda.reset_peak_memory_stats(device)
            mem0 = torch.cuda.memory_allocated(device)             # (what the main workload's objects still hold)
            if name == "train":
                step, obj = make_train_step(a, 0, device, result)
                dominant, nsf, step_single = TRAIN_DOMINANT, None, None
            else:
                step, obj, fr, nsf, step_single = make_fastnsf_step(a, 0, device, result)
                dominant = FASTNSF_DOMINANT
            step()                                                  # priming pass (workspace growth, one-off autotune)
            for _ in range(len(nsf.engines) - 1 if nsf is not None else 0):
                step()                                              # (the other engine(s))
            for _ in range(warm):
                step()
            if nsf is not None:
                nsf.sync_check()
            torch.cuda.synchronize()
            side = name == "train" and obj.overlap_wgrad
            if nsf is None and not side:
                _lib.prof_start(only=dominant)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            if nsf is not None:
                nsf.sync_check()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            el_single = None
            if side:                                                # the roofline kernel's launches un-overlapped: weight gradients on the main stream
                obj.set_side_streams(False)
                step()
                torch.cuda.synchronize()
                _lib.prof_start(only=dominant)
                t1 = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                el_train_single = (time.perf_counter() - t1) / 3
                obj.set_side_streams(True)
            if nsf is not None:                                     # the roofline kernel's launches un-overlapped: two fits on one engine
                step_single()                                       # (one untimed fit first: the engine's own buffers and the device clock settle --
                torch.cuda.synchronize()                            #  the two timed fits alone read 16.6 once against 20.3 in every other run)
                _lib.prof_start(only=dominant)
                t1 = time.perf_counter()
                for _ in range(2):
                    step_single()
                torch.cuda.synchronize()
                el_single = (time.perf_counter() - t1) / 2
            prof = _lib.prof_stop()
            if name == "train":
                roof, workload, dtype = train_roofline(a, prof, (3 if side else steps) * per_step, 3 * el_train_single if side else el)
                if side:
                    roof["measured_in"] = "three steps with every weight gradient on the main stream, right after the timed steps (un-overlapped launches)"
                parity = {"loss_after_leg": float(result["loss"].item()),
                          "note": "gradient parity vs CPU autograd through the oracle network (unpinned): tests/test_train_gpu.py"}
            else:
                roof, workload, dtype = fastnsf_roofline(a, prof, steps if el_single is None else 2, el if el_single is None else 2 * el_single)
                if el_single is not None:
                    roof["measured_in"] = "two fits on ONE engine right after the timed fits (un-overlapped launches)"
                got = result["flow"].cpu().numpy()
                parity = {"loss_first_to_last_iteration": [obj.loss_history[0][1], obj.loss_history[-1][1]],
                          "flow_mean_epe_vs_generating_flow": float(np.linalg.norm(got - fr[0]["flow"], axis=1).mean()),
                          "note": "gradient / trajectory parity vs the CPU restatement (oracle/fastnsf_oracle.py, unpinned): tests/test_fastnsf_gpu.py"}
            leg = {"frames_per_s": steps * per_step / el, "ms_per_step": el / steps * 1e3, "steps": steps, "samples_per_step": per_step, "warmup": warm,
                   "points_per_frame": a.points, "cloud": a.cloud, "workload": workload, "dtype": dtype, "roofline": roof, "parity": parity}
            if name == "train":
                leg["device_memory_GB"] = round((torch.cuda.max_memory_allocated(device) - mem0) / 1e9, 1)      # this leg's trainer, samples and workspaces (peak)
            if name == "train" and side:
                leg["frames_per_s_without_side_streams"] = per_step / el_train_single
            if name == "fastnsf" and el_single is not None:
                leg["fits_in_flight"] = len(nsf.engines)
                leg["frames_per_s_one_fit_at_a_time"] = 1.0 / el_single
        except Exception as e:                                      # a leg must never cost the main line
            leg = {"error": f"{type(e).__name__}: {e}"}
        out[f"leg_{leg_name}"] = leg
        if "frames_per_s" in leg:
            out[f"value_{leg_name}"] = leg["frames_per_s"]
        step = obj = result = None
        torch.cuda.empty_cache()
    return out


def hostfed_leg(args, pipe, host_frames, device) -> dict:
    """N = 1 only, after the timed region: the SAME pipeline fed from HOST memory -- every step's 3 x B sweeps are staged into
    pinned memory and copied to the device inside the loop (``feeder.SampleFeeder``: a background thread, a pinned ring, a copy
    stream, two batches ahead) -- the PCIe-inclusive rate the driver line did not carry before round 4.  ``value`` keeps the
    contract's definition (inputs resident in HBM when the timed region starts); this leg is reported beside it."""
    import torch
    from himo_amd.feeder import SampleFeeder
    B = args.frames_per_step
    steps = max(6, min(args.steps, 40))                        # (12 until the end of round 4: the fill and drain of the batches in flight were 8 % of that)
    fr = host_frames                                           # B + 2 host frames: sample j = (fr[j], fr[j + 1], fr[j + 2])

    def source(n_batches):
        k = 0
        for _ in range(n_batches):
            for j in range(B):
                yield (k, fr[j], fr[j + 1], fr[j + 2])
                k += 1

    def run(n_batches):
        done = 0
        for batch in SampleFeeder(source(n_batches), device=device, batch=B, depth=3):       # as many batches ahead as the pipeline keeps in flight
            pipe.run([smp for _, _, smp in batch], sensor_dt=0.1, refined=args.refined)
            done += len(batch)
        pipe.sync_check()
        torch.cuda.synchronize()
        return done

    run(3)
    t0 = time.perf_counter()
    frames = run(steps)
    el = time.perf_counter() - t0
    mb = sum(fr[j]["pc0"].nbytes + fr[j + 1]["pc0"].nbytes + fr[j + 2]["pc0"].nbytes + fr[j + 1]["lidar_dt"].nbytes for j in range(B)) / 1e6
    leg = {"frames_per_s": frames / el, "ms_per_step": el / steps * 1e3, "steps": steps, "host_MB_per_step": round(mb, 1),
           "note": "inputs start in pageable host memory every step: staged into a pinned ring and copied over PCIe by a feeder thread three "
                   "batches ahead of the launches (himo_amd/feeder.py); same kernels, same results"}
    return {"value_hostfed": leg["frames_per_s"], "leg_hostfed": leg}


def h5fed_leg(args, pipe, host_frames, device) -> dict:
    """N = 1 only, after the timed region: the reference's loop as a user runs it -- ``dataset[i]`` -> network -> write
    (save_zip.py:111-123; ``python save.py checkpoint=... dataset_path=...``, README.md:50) -- END TO END over 120k-point ``.h5``
    scene files: ``himo_amd.save.run`` reads the scenes (``dataset.HDF5Dataset``: files kept open, only the sweeps / poses / time
    stamps, views of the file mappings), stages them through the pinned feeder, runs the flow network with the batches in flight
    and writes every sweep's (N,3) flow back under ``<timestamp>/<res_name>`` (``save.H5ResultSink``: libhdf5 in place when one
    can be loaded, else a result file beside the scene).  The scene files are written once with ``h5lite.write_file`` into a
    temporary directory (every dataset the reference's extractors write, dataprocess/extract_sca.py:76-93, so that the loader
    has labels and masks to skip); the first pass is the warm-up (page cache, tile choices), the second is timed."""
    import pickle
    import shutil
    import tempfile
    import warnings
    import torch
    from himo_amd import h5lite, save
    from himo_amd.dataset import SAVE_FIELDS, HDF5Dataset
    B = args.frames_per_step
    n_scenes, per_scene = 8, int(os.environ.get("HIMO_BENCH_H5FED_BATCHES_PER_SCENE", "4")) * B + 9   # 8 x 72 items of 16 = 36 batches (the last scene's results are written after the last batch: a tail the length of a batch or two; 20 batches until round 6)
    root = Path(tempfile.mkdtemp(prefix="himo_h5fed_"))
    try:
        index = []
        for sc in range(n_scenes):
            scene, tree = f"bench{sc:02d}", {}
            for k in range(per_scene):
                f = host_frames[(7 * sc + k) % len(host_frames)]
                ts = str(315_965_785_000_000_000 + (1000 * sc + k) * 100_000_000)
                tree[ts] = {"lidar": f["pc0"], "lidar_dt": f["lidar_dt"], "lidar_id": f["lidar_id"], "pose": f["pose0"], "ground_mask": f["gm0"],
                            "flow": f["flow"], "flow_is_valid": f["flow_is_valid"], "flow_category_indices": f["flow_category_indices"],
                            "flow_instance_id": f["flow_instance_id"]}
                index.append([scene, ts])
            h5lite.write_file(root / f"{scene}.h5", tree)
        with open(root / "index_total.pkl", "wb") as fh:
            pickle.dump(index, fh)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                  # (the last sweep of a scene has no successor; no HDF5 library -> side files)
            ds = HDF5Dataset(root, fields=SAVE_FIELDS, zero_copy=True)
            t0 = time.perf_counter()
            for i in range(len(ds)):
                ds[i]
            cold = len(ds) / (time.perf_counter() - t0)
            t0 = time.perf_counter()
            for i in range(len(ds)):
                ds[i]
            loader = len(ds) / (time.perf_counter() - t0)
            # `loader_views_per_s` above is the rate of CREATING items whose sweeps are views of the file mapping (no byte of a sweep is
            # touched); the rate with pc0 / pc1 / lidar_dt actually copied out of the mapping -- what a consumer pays once, into its
            # pinned staging memory -- is this one (the two were reported under one name)
            sink_buf = [np.empty_like(np.asarray(ds[0][k])) for k in ("pc0", "pc1", "lidar_dt")]
            t0 = time.perf_counter()
            for i in range(len(ds)):
                item = ds[i]
                for buf, k in zip(sink_buf, ("pc0", "pc1", "lidar_dt")):
                    if item[k].shape == buf.shape:
                        np.copyto(buf, item[k])
            copied = len(ds) / (time.perf_counter() - t0)
            how = None
            for rep in range(2):
                sink = save.H5ResultSink(root, "seflowpp_bench", before_write=ds.forget)
                how = sink.how
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done = save.run(ds, "seflowpp_bench", sink=sink, pipeline=pipe, batch_frames=B, by_scene=True)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            ds.close()
        leg = {"frames_per_s": done / el, "frames": done, "seconds": el, "scenes": n_scenes, "sweeps_per_scene": per_scene,
               "points_per_sweep": int(host_frames[0]["pc0"].shape[0]), "loader_views_per_s": loader, "loader_views_per_s_first_pass": cold, "loader_items_copied_out_per_s": copied,
               "result_writer": how, "scene_file_MB": round((root / "bench00.h5").stat().st_size / 1e6, 1),
               "note": "himo_amd.save.run end to end: read .h5 scenes -> pinned feeder -> network (batches in flight) -> flow written back per sweep; "
                       "second pass over the files (page cache warm)"}
        return {"value_h5fed": leg["frames_per_s"], "leg_h5fed": leg}
    except Exception as e:                                       # a leg must never cost the main line
        return {"leg_h5fed": {"error": f"{type(e).__name__}: {e}"}}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def eval_h5_leg(args, device) -> dict:
    """N = 1 only, after the timed region: the evaluator as the PROGRAM a user starts -- ``python eval.py --data_dir ... --res_name
    seflowpp_best`` (eval.py:270-313: ``dataset[i]`` -> ego-motion removal, eval mask, per-instance MPE / Chamfer -> the tables) --
    END TO END over 120k-point ``.h5`` scene files holding the estimate beside the ground truth, run as ``python -m himo_amd.eval`` in
    its own interpreter: forked reader processes (the reference loop's ``DataLoader`` workers) read the scenes (``fields=`` / views of
    the file mapping) and pack the batches into shared slots, the feeder thread issues one DMA per batch, the launch thread scores
    them on the device and writes ``res-av2.json``.  Beside it the same program on reader threads, ``eval.main`` called inside
    this process (reader threads: a process that holds device state does not fork), and the same sweeps as device-RESIDENT batches
    (``InstanceMetrics.step_batch``: kernels + the pipelined record read-back + the host-side bucket bookkeeping), the figure of
    profiles/r0x_evaluator_throughput.txt."""
    import contextlib
    import io
    import shutil
    import tempfile
    import warnings
    import torch
    from himo_amd import eval as ev
    from himo_amd.synthetic import make_frame, write_h5_scenes
    n_scenes, per_scene, distinct, P = 8, 257, 33, args.points    # 2048 scored sweeps = 128 batches (33 distinct sweeps per scene, repeated under new time stamps)
    root = Path(tempfile.mkdtemp(prefix="himo_eval_av2_"))
    try:
        frames = []
        for sc in range(n_scenes):
            made = [make_frame(9000 + 40 * sc + k, n_points=P, scene_id=f"eval{sc:02d}", cloud=args.cloud) for k in range(distinct)]
            fr = []
            for k in range(per_scene):
                f = dict(made[k % distinct])
                f["timestamp"] = int(made[0]["timestamp"]) + k * 100_000_000
                fr.append(f)
            frames.append(fr)
        # (write_h5_scenes writes the reference extractors' datasets; the estimate goes in beside them, as the reference's save.py leaves it)
        import pickle
        from himo_amd import h5lite
        index = []
        for fr in frames:
            tree = {}
            for f in fr:
                tree[str(f["timestamp"])] = {"lidar": f["pc0"], "lidar_dt": f["lidar_dt"], "lidar_id": f["lidar_id"], "pose": f["pose0"],
                                             "ground_mask": f["gm0"], "flow": f["flow"], "flow_is_valid": f["flow_is_valid"],
                                             "flow_category_indices": f["flow_category_indices"], "flow_instance_id": f["flow_instance_id"],
                                             "seflowpp_best": f["seflowpp_best"]}
                index.append([f["scene_id"], str(f["timestamp"])])
            h5lite.write_file(root / f"{fr[0]['scene_id']}.h5", tree)
        with open(root / "index_total.pkl", "wb") as fh:
            pickle.dump(index, fh)
        B = 16
        # the PROGRAM, in its own interpreter, as a user starts it: reader processes are forked before the HIP runtime starts (a fork from
        # a process that holds device state -- this one -- is paid for at its next device call: profiles/r06_exp_fork_cost.txt)
        import re
        import subprocess

        def program(workers):
            t0 = time.perf_counter()
            out = subprocess.run([sys.executable, "-W", "ignore", "-m", "himo_amd.eval", "--data_dir", str(root), "--res_name", "seflowpp_best",
                                  "--batch_frames", str(B)] + ([] if workers is None else ["--num_workers", str(workers)]),
                                 cwd=str(root), capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=str(REPO)))
            wall = time.perf_counter() - t0
            if out.returncode != 0:
                raise RuntimeError(out.stderr[-1500:])
            line = [l for l in out.stdout.splitlines() if l.startswith("Scoring loop")][-1]
            got = re.match(r"Scoring loop: (\d+) sweeps/s \((\d+) sweeps in ([\d.]+) s, (\d+) reader (\w+)\)(?:; (\d+) sweeps/s after)?", line)
            return {"sweeps_per_s": float(got.group(1)), "sweeps": int(got.group(2)), "seconds": float(got.group(3)), "readers": int(got.group(4)),
                    "reader_kind": got.group(5), "sweeps_per_s_after_warm_up": float(got.group(6)) if got.group(6) else None,
                    "process_wall_s": round(wall, 2)}
        program(None)                                              # first pass: page cache
        prog = program(None)                                       # the default: 4 reader processes for an evaluation of this length
        prog_threads = program(0)
        sink = io.StringIO()
        with warnings.catch_warnings(), contextlib.redirect_stdout(sink):
            warnings.simplefilter("ignore")
            for rep in range(2):                                   # ... and inside THIS process (device state: reader threads), as rounds 5-6 timed it
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m = ev.main(str(root), res_name="seflowpp_best", batch_frames=B, file_name=str(root / "res.json"))
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            sweeps = m.frame_cnt
            # the same work on device-resident batches
            flat = [f for fr in frames for f in fr[:-1]][:4 * B]
            for f, nxt in zip(flat, [f for fr in frames for f in fr[1:]][:4 * B]):
                f["pose1"] = nxt["pose0"]
            ebs = [ev.EvalBatch.from_frames(flat[k * B:(k + 1) * B], "seflowpp_best", device=device) for k in range(4)]
            res = ev.InstanceMetrics("av2")
            for eb in ebs:
                res.step_batch(eb)
            res.flush(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(12):
                res.step_batch(ebs[k % 4])
            res.flush(); torch.cuda.synchronize()
            resident = 12 * B / (time.perf_counter() - t0)
        mb = sum(f[k].nbytes for f in frames[0][:1] for k in ("pc0", "lidar_dt", "gm0", "flow", "flow_is_valid", "flow_category_indices",
                                                              "flow_instance_id", "seflowpp_best")) / 1e6
        steady = prog["sweeps_per_s_after_warm_up"] or prog["sweeps_per_s"]
        leg = {"sweeps_per_s": steady, "whole_loop_sweeps_per_s": prog["sweeps_per_s"], "sweeps": prog["sweeps"], "seconds": prog["seconds"],
               "readers": f"{prog['readers']} reader {prog['reader_kind']}", "process_wall_s": prog["process_wall_s"],
               "reader_threads_program": prog_threads, "in_this_process_reader_threads_sweeps_per_s": sweeps / el,
               "resident_sweeps_per_s": resident, "fraction_of_resident": steady / resident,
               "scenes": n_scenes, "sweeps_per_scene": per_scene, "points_per_sweep": P, "batch_frames": B, "host_MB_read_per_sweep": round(mb, 2),
               "note": "python -m himo_amd.eval in its own interpreter, end to end: forked reader processes (h5lite views of the scene files, batches "
                       "packed into shared slots registered with the runtime) -> one DMA per batch on the feeder's stream -> eval mask, per-instance "
                       "MPE / Chamfer on the device -> tables + res-av2.json; page cache warm.  sweeps_per_s = the scoring loop after its first 8 "
                       "batches (they pay for the runtime's start, workspaces and slot registration; whole_loop_sweeps_per_s includes them, "
                       "process_wall_s also the interpreter's start and imports).  reader_threads_program = the same program with --num_workers 0; "
                       "in_this_process_... = eval.main called here, where it reads on threads.  resident = the same scoring on batches already in HBM"}
        return {"value_eval_h5": leg["sweeps_per_s"], "leg_eval_h5": leg}
    except Exception as e:                                           # a leg must never cost the main line
        return {"leg_eval_h5": {"error": f"{type(e).__name__}: {e}"}}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def fit_h5_leg(args, device) -> dict:
    """N = 1 only, after the timed region: BASELINE config 5 as the PROGRAM a user starts -- ``python -m himo_amd.seflow.fit`` =
    ``seflow.fit.fit`` over ``.h5`` scene files with the reference launcher's numbers (assets/slurm/ssl-train-av2.sh:31-34:
    ``batch_size=8``, ``+ssl_label=seflow_auto``, dataloader workers ahead of the step): 8 scenes x 41 consecutive 120k-point sweeps
    (``synthetic.make_scene``: one coherent drive per scene, written once by ``h5lite.write_file`` with every dataset the
    reference's extractors write), samples read with ``fields=`` on reader threads, staged in pinned memory, copied and LABELLED
    on the device (two exact nearest-neighbour passes + two DBSCANs per pair) ahead of the optimiser step
    (``feeder.TrainFeeder``), ``--leg-fit-steps`` optimiser steps of 8 samples each, BatchNorm in training mode, Adam, StepLR
    bookkeeping, the epoch's loss read back.  ``leg_train_b8`` beside it is the bare step (8 samples per pass) on resident samples
    with pre-made labels."""
    import shutil
    import tempfile
    import warnings
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from himo_amd.dataset import HDF5Dataset
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import fit, train_fields, triplets
    from himo_amd.seflow.train import SeFlowTrainer
    from himo_amd.synthetic import make_scene, write_h5_scenes
    n_scenes, per_scene, bs = 8, 41, 8
    root = Path(tempfile.mkdtemp(prefix="himo_fit_h5_"))
    try:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=n_scenes) as pool:       # (numpy's generators release the GIL)
            scenes = list(pool.map(lambda sc: make_scene(500 + sc, per_scene, n_points=args.points, scene_id=f"drive{sc:02d}", cloud=args.cloud),
                                   range(n_scenes)))
        write_h5_scenes(root, scenes)
        scenes = None
        made = time.perf_counter() - t0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                          # (the last sweep of a scene has no successor)
            ds = HDF5Dataset(root, fields=train_fields("seflow_auto"), zero_copy=True)
        n_trip = len(triplets(ds))
        steps_per_epoch = -(-n_trip // bs)
        epochs = max(1, -(-args.leg_fit_steps // steps_per_epoch))
        tr = SeFlowTrainer(spec.init_params(0), device=device, max_points=int(args.points * 1.02), precision=args.train_precision,
                           batchnorm=args.train_batchnorm, batch=bs)      # a step's 8 samples in ONE pass, as `leg_train_b8`
        fit(ds, trainer=tr, epochs=1, batch_size=bs, max_steps=3, log=None, num_workers=args.fit_workers)      # warm-up: tile choices, buffers, page cache
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fit(ds, trainer=tr, epochs=epochs, batch_size=bs, max_steps=args.leg_fit_steps, log=None, num_workers=args.fit_workers)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ds.close()
        hist = out["history"]
        steps = sum(h["steps"] for h in hist)
        samples = sum(h["samples"] for h in hist)
        feeder = {k: sum(h["feeder"][k] for h in hist if h.get("feeder")) for k in ("read", "upload", "labels", "samples")}
        leg = {"frames_per_s": samples / el, "optimiser_steps": steps, "samples": samples, "batch_size": bs, "epochs": len(hist),
               "seconds": el, "ms_per_sample": el / max(samples, 1) * 1e3, "scenes": n_scenes, "sweeps_per_scene": per_scene,
               "points_per_sweep": args.points, "ssl_label": "seflow_auto", "reader_threads": args.fit_workers,
               "train_loss_first_last_epoch": [hist[0]["train_loss"], hist[-1]["train_loss"]],
               "frames_per_s_by_epoch": [round(h["samples"] / h["train_seconds"], 1) for h in hist],
               "labels": "generated on the device in the first epoch (two nearest-neighbour passes + two DBSCANs per pair), kept on the host and "
                         "uploaded in the later ones (the reference's job reads labels an offline pass wrote once)",
               "feeder_thread_host_ms_per_sample": {k: round(feeder[k] / max(feeder["samples"], 1) * 1e3, 3) for k in ("read", "upload", "labels")},
               "scene_files_written_in_s": round(made, 1), "dtype": args.train_precision,
               "note": "seflow.fit.fit end to end over .h5 scenes: read (fields=, views of the file mapping) -> pinned -> HBM -> labels generated "
                       "on a side stream -> train_batch of 8 samples (one Adam step) -> per-epoch loss read-back; feeder host times are per "
                       "sample on their own threads (reads summed over the reader threads), beside the step, not in series with it"}
        return {"value_fit_h5": leg["frames_per_s"], "leg_fit_h5": leg}
    except Exception as e:                                           # a leg must never cost the main line
        return {"leg_fit_h5": {"error": f"{type(e).__name__}: {e}"}}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def extra_precision_legs(args, params, sets, device, ref_flow, exclude: str) -> dict:
    """N = 1 only, after the timed region: the SAME workload (same batches, same steps protocol: priming pass, warm-up,
    K timed steps between synchronisations) in the other two matrix arithmetics, so the float32-range figures are measured
    in the same run as ``value``: ``value_bf16x3`` / ``value_f32`` (+ their parity against the CPU restatement)."""
    import torch
    from himo_amd.pipeline import HiMoPipeline
    from himo_amd.seflow.model import SeFlowNet
    out = {}
    B, P = args.frames_per_step, args.points
    for prec, steps in (("f16x2", args.steps), ("bf16x3", max(3, args.steps // 2)), ("f32", max(3, args.steps // 4))):
        if prec == exclude:
            continue
        net = SeFlowNet(params, device=device, max_points=P, precision=prec, max_batch=B)
        pipe = HiMoPipeline(net, device=device)
        res = pipe.run(sets[0], sensor_dt=0.1, refined=args.refined)                        # priming pass
        torch.cuda.synchronize()
        leg = {}
        if ref_flow is not None:
            got = res["flow"][:P].cpu().numpy()
            leg["flow_max_abs_vs_cpu_restatement"] = float(np.abs(got - ref_flow).max())
            leg["flow_mean_epe_vs_cpu_restatement"] = float(np.linalg.norm(got - ref_flow, axis=1).mean())
        for k in range(2):
            pipe.run(sets[k % len(sets)], sensor_dt=0.1, refined=args.refined)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            pipe.run(sets[k % len(sets)], sensor_dt=0.1, refined=args.refined)
        pipe.sync_check()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        leg.update({"frames_per_s": B * steps / el, "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": 2})
        out[f"value_{prec}"] = leg["frames_per_s"]
        out[f"leg_{prec}"] = leg
        del pipe, net, res
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    sys.exit(main())
