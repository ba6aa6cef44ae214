"""GPU experiment: fused head kernel time vs number of GRU iterations (fixed cost vs per-iteration cost)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from himo_amd import _lib
from himo_amd.seflow import spec
from himo_amd.seflow.model import SeFlowNet
from himo_amd.synthetic import make_frame

dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
net = SeFlowNet(spec.init_params(0), device=dev, max_points=120_000, precision=prec)
f = [make_frame(i, n_points=120_000) for i in range(3)]
up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
pch, pc0, pc1 = up(f[0]["pc0"]), up(f[1]["pc0"]), up(f[2]["pc0"])
net.forward_device(pch, pc0, pc1, f[0]["pose0"], f[1]["pose0"], f[1]["pose1"])
torch.cuda.synchronize()
for iters in (0, 1, 2, 4):
    spec.GRU_ITERS = iters
    import himo_amd.seflow.model as M
    M.spec.GRU_ITERS = iters
    for _ in range(3): net.head(pc0)
    torch.cuda.synchronize()
    _lib.prof_start(only="gru_head")
    for _ in range(10): net.head(pc0)
    torch.cuda.synchronize()
    p = _lib.prof_stop()
    print(iters, {k: round(v["avg_ms"] * 1e3, 1) for k, v in p.items()})
