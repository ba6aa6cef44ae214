"""GPU experiment: per-kernel timing of one SeFlowNet forward at 3 x 120k points."""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from himo_amd import _lib
from himo_amd.seflow.model import SeFlowNet
from himo_amd.seflow import spec
from himo_amd.synthetic import make_frame

dev = torch.device("cuda", 0)
net = SeFlowNet(device=dev)
fr = [make_frame(i, n_points=120_000) for i in range(3)]
pts = [torch.from_numpy(f["pc0"]).to(dev) for f in fr]
args = (pts[0], pts[1], pts[2], fr[0]["pose0"], fr[1]["pose0"], fr[1]["pose1"])
for _ in range(2): net.forward(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): net.forward(*args)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5
_lib.prof_start()
for _ in range(3): net.forward(*args)
torch.cuda.synchronize()
prof = _lib.prof_stop()
tot = sum(v["total_ms"] for v in prof.values()) / 3
print(f"wall per forward {wall*1e3:.3f} ms; sum of kernels {tot:.3f} ms; conv GFLOP {spec.conv_flops()/1e9:.1f}")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{k:28s} n={v['count']//3:3d}/fwd total {v['total_ms']/3:8.3f} ms/fwd avg {v['avg_ms']*1e3:9.1f} us")
conv_ms = sum(v["total_ms"] for k, v in prof.items() if k.startswith("conv")) / 3
print(f"conv kernels: {conv_ms:.3f} ms -> {(spec.conv_flops() + spec.head_flops_per_point()*120000)/conv_ms/1e9:.1f} TFLOP/s")
