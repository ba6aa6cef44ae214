"""GPU experiment: pipeline step time with the backbone replayed as plain launches / one call / one hipGraph."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.argv = ["bench.py"]
import torch
import bench
from himo_amd.pipeline import HiMoPipeline
from himo_amd.seflow import spec
from himo_amd.seflow.model import SeFlowNet

dev = torch.device("cuda", 0)
PREC = "f16x2"
for mode in ("launches", "plan", "graph"):
    net = SeFlowNet(spec.init_params(0), device=dev, max_points=120_000, precision=PREC)
    net.use_plan = mode != "launches"
    net.use_graph = mode == "graph"
    pipe = HiMoPipeline(net, device=dev)
    samples = bench.synthetic_samples(8, 120_000, dev, seed=0)
    for _ in range(2):
        out = pipe.run(samples, sensor_dt=0.1)
    torch.cuda.synchronize()
    ref = out["flow"].clone()
    t0 = time.perf_counter()
    for _ in range(10):
        out = pipe.run(samples, sensor_dt=0.1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 80
    print(f"{mode:9s} {dt*1e3:.3f} ms/frame  {1/dt:.1f} frames/s   flow checksum {float(out['flow'].double().sum()):.6f} same={bool((out['flow']==ref).all())}")
