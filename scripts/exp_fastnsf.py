"""GPU experiment: FastNSF fit time at 120k points (BASELINE config 3)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from himo_amd.fastnsf import FastNSF
from himo_amd.synthetic import make_frame

dev = torch.device("cuda", 0)
f = make_frame(5, n_points=120_000)
pc0 = torch.from_numpy(f["pc0"][:, :3].copy()).to(dev)
pc1 = torch.from_numpy((f["pc0"][:, :3] + f["flow"]).astype(np.float32)).to(dev)
for iters in (100,):
    m = FastNSF(device=dev, iters=iters)
    m.fit(pc0, pc1, f["pose0"], f["pose1"]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    flow = m.fit(pc0, pc1, f["pose0"], f["pose1"]); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{iters} iterations at 120k points: {dt:.3f} s ({dt / iters * 1e3:.2f} ms / iteration), loss {m.loss_history[-1]}")
