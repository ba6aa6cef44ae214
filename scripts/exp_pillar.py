"""The pillar stage alone (12 sweeps = 4 samples per launch group), per kernel, alternating two sample sets as the bench does.
A/B harness for kernel variants: HIMO_AMD_LIB=build/variants/<name>/libhimo_amd.so python scripts/exp_pillar.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
from himo_amd import _lib
from himo_amd.seflow import spec
from himo_amd.seflow.model import SeFlowNet

dev = torch.device("cuda", 0)
net = SeFlowNet(spec.init_params(0), device=dev, max_points=120_000, precision="f16x2", max_batch=4)
sets, _ = bench.synthetic_sample_sets(2, 4, 120_000, dev, seed=0, cloud=sys.argv[1] if len(sys.argv) > 1 else "uniform")


def jobs(i):
    out = []
    for k, s in enumerate(sets[i % 2]):
        inv1 = np.linalg.inv(np.asarray(s.pose1, np.float64))
        out.append((k, (s.pch1, s.pc0, s.pc1), (inv1 @ np.asarray(s.pose_h1, np.float64), inv1 @ np.asarray(s.pose0, np.float64), np.eye(4))))
    return out


for i in range(4):
    net.pillarize_many(jobs(i))
torch.cuda.synchronize()
_lib.prof_start()
for i in range(10):
    net.pillarize_many(jobs(i))
torch.cuda.synchronize()
p = _lib.prof_stop()
print({k: round(v["avg_ms"] * 1e3, 1) for k, v in p.items()}, "us per launch group of 12 sweeps; sum", round(sum(v["total_ms"] for v in p.values()) / 10 * 1e3, 1), "us")
