"""Two independent pipelines (own network buffers) fed alternately on two HIP streams from one launch thread, against one
pipeline on one stream: does overlapping one batch's latency-bound stages (pillar stage, head, 1x1 / upsampling) with the other
batch's power-limited 3x3 convolutions raise the frame rate?  usage: python scripts/exp_two_streams.py [frames_per_batch]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from himo_amd.pipeline import HiMoPipeline
from himo_amd.seflow import spec
from himo_amd.seflow.model import SeFlowNet

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = 120_000
params = spec.init_params(0)
sets, _ = bench.synthetic_sample_sets(3, B, P, dev, seed=0, cloud="uniform")


def make():
    net = SeFlowNet(params, device=dev, max_points=P, precision="f16x2", max_batch=B)
    return HiMoPipeline(net, device=dev)


def run(pipes, streams, steps):
    for i in range(steps):
        k = i % len(pipes)
        with torch.cuda.stream(streams[k]):
            pipes[k].run(sets[i % len(sets)])
    for p in pipes:
        p.sync_check()
    torch.cuda.synchronize()


for n_pipes in (1, 2, 1, 2, 1, 2):
    pipes = [make() for _ in range(n_pipes)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_pipes)]
    run(pipes, streams, 3 * n_pipes)          # priming: autotune, plans, graphs
    run(pipes, streams, 4)
    steps = 96
    t0 = time.perf_counter()
    run(pipes, streams, steps)
    dt = time.perf_counter() - t0
    print(f"{n_pipes} pipeline(s) / stream(s): {steps * B / dt:.1f} frames/s ({dt / steps * 1e3:.2f} ms per {B}-frame batch)", flush=True)
    del pipes
    torch.cuda.empty_cache()
