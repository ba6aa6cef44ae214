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

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TF = 157.3         # dense float32-input MFMA peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0       # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)
POINTS_PER_FRAME = 120_000       # BASELINE.json metric


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "compdis", "train"])
    ap.add_argument("--frames-per-step", type=int, default=None, help="frames per rank per step")
    ap.add_argument("--points", type=int, default=POINTS_PER_FRAME)
    ap.add_argument("--precision", default="f16x2", choices=["bf16x3", "f16x2", "f32"],
                    help="network matrix arithmetic: split-bf16 (float32-class accuracy) or float32 MFMA")
    ap.add_argument("--train-precision", default="mixed", choices=["mixed", "bf16x3", "f32"],
                    help="train workload: forward fp16-split + data-gradient split-bf16 (mixed), all split-bf16, or float32 MFMA")
    ap.add_argument("--float32-activations", action="store_true",
                    help="pipeline, f16x2: keep the backbone's maps float32 in HBM instead of the split activation format (A/B switch)")
    ap.add_argument("--refined", action="store_true", help="also write refined points (+12 B/pt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work budget for the baseline leg")
    ap.add_argument("--traffic-json", default=str(REPO / "profiles" / "traffic_latest.json"))
    a = ap.parse_args()
    if a.workload == "train":
        a.steps = 10 if a.steps is None else a.steps
        a.warmup = 2 if a.warmup is None else a.warmup
        a.frames_per_step = 1 if a.frames_per_step is None else a.frames_per_step
    elif a.workload == "pipeline":
        a.steps = 20 if a.steps is None else a.steps            # 320 frames: ~0.55 s timed, host hiccups average out
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


def synthetic_samples(n_frames: int, n_points: int, device, seed: int):
    """B network inputs: history sweep, pc0, pc1 (each n_points x 4), poses, lidar_dt."""
    import torch
    from himo_amd.pipeline import Sample
    g = torch.Generator(device=device)
    g.manual_seed(4321 + seed)
    rng = np.random.default_rng(seed)
    pose0, pose1 = _poses(n_frames, rng)
    _, pose_h = _poses(n_frames, rng)
    out = []
    for k in range(n_frames):
        out.append(Sample(_sweep(n_points, g, device), _sweep(n_points, g, device), _sweep(n_points, g, device),
                          np.linalg.inv(pose_h[k]), pose0[k], pose1[k],
                          torch.rand(n_points, generator=g, device=device, dtype=torch.float32) * 0.1,
                          scene_id=f"bench-{seed}", timestamp=k))
    return out


def frame_to_host(batch, k: int) -> dict:
    o = batch.offsets_host
    s = slice(int(o[k]), int(o[k + 1]))
    return {"pc0": batch.pc0[s].cpu().numpy(), "seflowpp_best": batch.flow[s].cpu().numpy(),
            "lidar_dt": batch.lidar_dt[s].cpu().numpy(), "pose0": batch.pose0[k].cpu().numpy(),
            "pose1": batch.pose1[k].cpu().numpy()}


# ------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle is the thing timed here, never the thing shipped)
# ------------------------------------------------------------------------------------------------------
def cpu_baseline_compdis(frames: list[dict], budget_s: float) -> dict:
    """numpy port of save_zip.py:113-121 incl. the f32 cast, single thread."""
    sys.path.insert(0, str(REPO / "oracle"))
    import himo_oracle as oracle
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:                                   # pragma: no cover
        pass
    for f in frames[:2]:
        oracle.comp_dis_frame_f32(f, "seflowpp_best")   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.comp_dis_frame_f32(frames[n % len(frames)], "seflowpp_best")
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 20000:
            break
    return {"value": n / el, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} x {len(frames[0]['pc0'])}-pt frames in {el:.1f}s, numpy oracle/himo_oracle.py "
                      f"comp_dis_frame_f32 (host has {os.cpu_count()} cores, 1 used)"}


def cpu_baseline_pipeline(samples, params, budget_s: float) -> dict:
    """PyTorch-CPU float32 restatement of the network (oracle/seflow_oracle.py) + numpy comp_dis, all host cores
    torch wants.  PARITY UNPINNED: this is the build's own restatement, not the reference's code (absent)."""
    import torch
    sys.path.insert(0, str(REPO / "oracle"))
    import himo_oracle as oracle
    import seflow_oracle as so
    threads = torch.get_num_threads()
    n, t0, el = 0, time.perf_counter(), 0.0
    while True:
        s = samples[n % len(samples)]
        flow = so.forward(params, s.pch1.cpu().numpy(), s.pc0.cpu().numpy(), s.pc1.cpu().numpy(), s.pose_h1, s.pose0, s.pose1)
        frame = {"pc0": s.pc0.cpu().numpy(), "seflowpp_best": flow, "lidar_dt": s.lidar_dt.cpu().numpy(),
                 "pose0": s.pose0, "pose1": s.pose1}
        oracle.comp_dis_frame_f32(frame, "seflowpp_best")
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    return {"value": n / el, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frame(s) of 3 x {len(samples[0].pc0)} pts in {el:.1f}s: PyTorch-CPU fp32 restatement "
                      f"(oracle/seflow_oracle.py) + numpy comp_dis; torch threads={threads}, host cores={os.cpu_count()}"}


def reduce_job(elapsed: float, frames_done: int, device, world: int, rank: int):
    """Max-over-ranks wall time and the whole-job frame count.  The frame counts travel through the path's only
    exchange, the final gather to rank 0 (RCCL on GPUs; gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    done = torch.tensor([frames_done], device=device, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros_like(done) for _ in range(world)]
        dist.all_gather(gathered, done)                 # every backend implements all_gather; rank 0 reports
        total = int(sum(int(g.item()) for g in gathered)) if rank == 0 else 0
    else:
        total = int(done.item())
    return float(el.item()), total


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path to benchmark")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver
        dist.init_process_group("nccl", device_id=device)

    from himo_amd import _lib
    B, P = args.frames_per_step, args.points
    sys.path.insert(0, str(REPO / "oracle"))

    if args.workload == "compdis":
        from himo_amd.compdis import CompDisEngine
        batch = synthetic_batch(B, P, device, seed=rank)
        eng = CompDisEngine(device=device, max_frames=B)
        out = {}

        def step():
            eng.run(batch, sensor_dt=0.1, refined=args.refined, out=out)
    elif args.workload == "train":
        # BASELINE config 5: self-supervised training, one sample per rank per optimiser step, ONE flat-gradient
        # all-reduce over RCCL, Adam.  Labels: ~10 % of the points in 30 dynamic clusters.
        from himo_amd.seflow import spec
        from himo_amd.seflow.train import SeFlowTrainer
        params = spec.init_params(0)
        trainer = SeFlowTrainer(params, device=device, max_points=P, precision=args.train_precision)
        samples = synthetic_samples(B, P, device, seed=rank)
        g = torch.Generator(device=device); g.manual_seed(99 + rank)
        labels = []
        for _ in range(B):
            pick = torch.rand(P, generator=g, device=device) < 0.1
            lab = torch.randint(1, 31, (P,), generator=g, device=device, dtype=torch.int32) * pick.to(torch.int32)
            labels.append((lab, lab.clone()))
        result = {}

        def step():
            for smp, (l0, l1) in zip(samples, labels):
                _, total = trainer.train_step(smp.pch1, smp.pc0, smp.pc1, smp.pose_h1, smp.pose0, smp.pose1, l0, l1, n_labels=31)
                result["loss"] = total
    else:
        from himo_amd.pipeline import HiMoPipeline
        from himo_amd.seflow import spec
        from himo_amd.seflow.model import SeFlowNet
        params = spec.init_params(0)
        net = SeFlowNet(params, device=device, max_points=P, precision=args.precision, max_batch=B)
        if args.float32_activations:
            net.split_acts = False
        pipe = HiMoPipeline(net, device=device)
        samples = synthetic_samples(B, P, device, seed=rank)
        result = {}

        def step():
            result.update(pipe.run(samples, sensor_dt=0.1, refined=args.refined))

    step()                                      # priming pass on every rank (one-off tile autotune, operator-list recording,
    torch.cuda.synchronize()                    # workspace growth): never inside the timed region, whatever --warmup is
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # parity spot-check on rank 0 (not timed): EPE / max abs vs the CPU oracle on one frame
    parity = None
    if rank == 0:
        import himo_oracle as oracle
        step()
        torch.cuda.synchronize()
        if args.workload == "compdis":
            f = frame_to_host(batch, 0)
            ref = oracle.comp_dis_frame_f32(f, "seflowpp_best")
            got = out["comp_dis"][:P].cpu().numpy()
            d = got.astype(np.float64) - ref
            parity = {"comp_dis_mean_epe_vs_ref": float(np.linalg.norm(d, axis=1).mean()),
                      "comp_dis_max_abs_vs_ref": float(np.abs(d).max()), "bit_exact_fraction": float((got == ref).mean())}
        elif args.workload == "train":
            parity = {"loss_after_warmup": float(result["loss"].item()),
                      "note": "gradient parity vs CPU autograd through the oracle network: tests/test_train_gpu.py"}
        elif not args.no_cpu_baseline:
            import seflow_oracle as so
            s = samples[0]
            ref_flow = so.forward(params, s.pch1.cpu().numpy(), s.pc0.cpu().numpy(), s.pc1.cpu().numpy(), s.pose_h1, s.pose0, s.pose1)
            got_flow = result["flow"][:P].cpu().numpy()
            frame = {"pc0": s.pc0.cpu().numpy(), "seflowpp_best": ref_flow, "lidar_dt": s.lidar_dt.cpu().numpy(),
                     "pose0": s.pose0, "pose1": s.pose1}
            ref_cd = oracle.comp_dis_frame_f32(frame, "seflowpp_best")
            got_cd = result["comp_dis"][:P].cpu().numpy()
            # the same frame through the float32-MFMA kernels (no split arithmetic anywhere): what the split costs
            net32 = SeFlowNet(params, device=device, max_points=P, precision="f32", autotune=False)
            flow32 = net32.forward_device(s.pch1, s.pc0, s.pc1, s.pose_h1, s.pose0, s.pose1).cpu().numpy()
            del net32
            torch.cuda.empty_cache()
            parity = {"flow_mean_epe_vs_cpu_restatement": float(np.linalg.norm(got_flow - ref_flow, axis=1).mean()),
                      "flow_max_abs_vs_float32_mfma_kernels": float(np.abs(got_flow - flow32).max()),
                      "float32_mfma_kernels_max_abs_vs_cpu_restatement": float(np.abs(flow32 - ref_flow).max()),
                      "flow_max_abs_vs_cpu_restatement": float(np.abs(got_flow - ref_flow).max()),
                      "comp_dis_max_abs_vs_cpu_restatement": float(np.abs(got_cd.astype(np.float64) - ref_cd).max()),
                      "note": "network parity is against this build's own CPU restatement (reference source absent)"}

    # The roofline kernel is timed live inside the timed region (HIP events around each of ITS launches, on the launch
    # stream); the other kernels are left alone there -- two event records per launch on ~45 launches per frame cost
    # ~12 % of the frame rate -- and get their table from one extra, untimed, fully profiled step afterwards.
    dominant = {"compdis": "compdis_kernel", "train": "conv_wgrad_tiled_kernel"}.get(
        args.workload, {"bf16x3": "conv3x3_bf16x3_kernel", "f16x2": "conv3x3_f16x2_kernel"}.get(args.precision, "conv3x3_mfma_kernel"))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.disable()                                # no collector pauses inside the timed region
    if parity is not None:
        # the CPU restatement of the parity check just ran on every host core: let its worker threads park before the
        # launch thread is timed (spinning OpenMP workers otherwise cost ~5 % of the frame rate)
        n_threads = torch.get_num_threads()
        torch.set_num_threads(1)
        time.sleep(1.0)
    _lib.prof_start(only=dominant)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if args.workload == "pipeline":
        pipe.sync_check()                       # the last batch's finite-flow flag (fp16-split precision)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if parity is not None:
        torch.set_num_threads(n_threads)
    prof = _lib.prof_stop()
    all_kernels = {}
    if rank == 0:
        _lib.prof_start()
        step()
        torch.cuda.synchronize()
        all_kernels = _lib.prof_stop()

    elapsed, total_frames = reduce_job(elapsed, B * args.steps, device, world, rank)

    if rank == 0:
        traffic = None
        try:
            traffic = json.loads(Path(args.traffic_json).read_text())
        except Exception:
            traffic = {}
        per_kernel = {n: {"avg_ms": v["avg_ms"], "launches_per_step": v["count"], "ms_per_step": v["total_ms"]}
                      for n, v in all_kernels.items()}
        if args.workload == "compdis":
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
        elif args.workload == "train":
            from himo_amd.seflow import spec
            # dominant kernel of the step: the 3x3 weight gradients (float32 MFMA, LDS-tiled split-K), 19 launches per step
            k = prof.get("conv_wgrad_tiled_kernel", {"avg_ms": float("nan"), "count": 0, "total_ms": float("nan")})
            n_steps = B * args.steps
            H, W = spec.GRID
            flops_w = spec.conv3x3_flops() + sum(spec.NUM_FRAMES * 2.0 * (H // d) * (W // d) * ci * co * 9
                                                 for d, ci, co in ((2, 32, 64), (4, 64, 128), (8, 128, 256)))
            alg_tf = flops_w * n_steps / (k["total_ms"] * 1e-3) / 1e12 if k["count"] else float("nan")
            roofline = {"bound": "mfma", "kernel": "conv_wgrad_tiled_kernel (v_mfma_f32_32x32x2_f32; 3x3 weight gradients)",
                        "achieved": alg_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": alg_tf / MFMA_F32_PEAK_TF,
                        "traffic": None, "avg_launch_ms": k["avg_ms"], "launches_timed": k["count"],
                        "algorithmic_flops_per_step": flops_w, "share_of_step_time": k["total_ms"] / (elapsed * 1e3)}
            workload = ("self-supervised TRAINING step (BASELINE config 5): pillarise 3 sweeps -> network forward with saved "
                        "activations -> 4-term NN/Chamfer loss -> full backward -> flat-gradient all-reduce -> Adam; "
                        "one 120k-point sample per GPU per step")
            dtype = "f32 weight gradients / optimiser; bf16x3 (split bf16, float32-class) forward + data-gradient convolutions"
        else:
            from himo_amd.seflow import spec
            bf, f16 = args.precision == "bf16x3", args.precision == "f16x2"
            kname = "conv3x3_bf16x3_kernel" if bf else "conv3x3_f16x2_kernel" if f16 else "conv3x3_mfma_kernel"
            k = prof.get(kname, {"avg_ms": float("nan"), "count": 0, "total_ms": float("nan")})
            n_fwd = B * args.steps
            # algorithmic flops of the 20 stride-1 3x3 convolutions of one forward (2*M*N*K each), see DESIGN.md
            flops3 = spec.conv3x3_flops()
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
                        "traffic": traffic.get("conv3x3_mfma_kernel" if args.precision == "f32" else "conv3x3_split_kernel", {}).get("hbm_bytes_per_launch"),
                        "vs_f32_mfma_peak_157_3": alg_tf / MFMA_F32_PEAK_TF,
                        "algorithmic_flops_per_launch": flops3 / max(launches_per_fwd, 1e-9), "avg_launch_ms": k["avg_ms"],
                        "launches_timed": k["count"], "launches_per_frame": launches_per_fwd, "samples_per_launch": B,
                        "share_of_step_time": k["total_ms"] / (elapsed * 1e3)}
            workload = ("per-frame pipeline: pillarise 3 sweeps (512x512 grid) -> SeFlow++-style encoder/decoder + GRU head "
                        "(random-init, self-specified: reference network source absent) -> per-point flow -> ego-motion "
                        "removal + dt0 + flow2compDis -> comp_dis")
            dtype = ("bf16x3 (three-term split bf16 on the matrix cores, float32 accumulate; float32-class accuracy)" if bf else
                     "f16x2 (two-term split fp16, x = h + l with exact subnormals, on the matrix cores; float32 accumulate; ~22-bit products)"
                     if f16 else "f32")
        line = {
            "metric": "lidar_frames_per_sec_120k", "value": total_frames / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "frames_per_step_per_gpu": B, "points_per_frame": P,
                       "sweeps_per_frame": 3 if args.workload == "pipeline" else 1,
                       "parallelism": f"frames sharded x{world}", "refined_output": bool(args.refined)},
            "roofline": roofline, "kernels": per_kernel,
            "kernels_note": "one extra untimed step with every kernel timed; the timed region times only the roofline kernel",
            "parity": parity,
        }
        if args.workload == "pipeline":
            line["config"]["matrix_arithmetic"] = args.precision
            line["config"]["samples_per_backbone_launch"] = B
        if args.workload == "train":
            line["metric"] = "train_frames_per_sec_120k"
            line["config"]["parallelism"] = f"data parallel x{world}, one flat all-reduce per step"
            line["config"]["matrix_arithmetic"] = args.train_precision
        if not args.no_cpu_baseline and args.workload != "train" and world == 1:     # CPU leg: rank 0 at N = 1 only
            if args.workload == "compdis":
                frames = [frame_to_host(batch, i) for i in range(min(8, B))]
                line["cpu_baseline"] = cpu_baseline_compdis(frames, args.cpu_seconds)
            else:
                line["cpu_baseline"] = cpu_baseline_pipeline(samples, params, args.cpu_seconds)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
