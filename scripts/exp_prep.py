"""GPU experiment: where does frame_prep_kernel's time go? (run via gpurun)"""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib, utils
from himo_amd.compdis import CompDisEngine
sys.argv = ["bench"]
import bench

dev = torch.device("cuda", 0)
res = {}
for name, P in (("120000", 120000), ("122880_aligned", 122880)):
    b = bench.synthetic_batch(256, P, dev, 0)
    eng = CompDisEngine(device=dev, max_frames=256)
    out = {}
    for _ in range(3): eng.run(b, out=out)
    torch.cuda.synchronize()
    _lib.prof_start()
    for _ in range(10): eng.run(b, out=out)
    torch.cuda.synchronize()
    res[name] = _lib.prof_stop()
    del b, out
# single huge frame through himo_dt0 (no ego, no straddle)
dt = torch.rand(256 * 120000, device=dev)
for _ in range(3): utils.dt0_from_lidar_dt(dt)
torch.cuda.synchronize()
_lib.prof_start()
for _ in range(10): utils.dt0_from_lidar_dt(dt)
torch.cuda.synchronize()
res["dt0_one_frame"] = _lib.prof_stop()
print(json.dumps(res, indent=1))
