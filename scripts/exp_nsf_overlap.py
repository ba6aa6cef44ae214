"""FastNSF: K fits one after another on one engine vs two engines on two streams (OverlappedFastNSF): frames/s, and the flows must
be the same bits.  python scripts/exp_nsf_overlap.py [K] [engines]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from himo_amd.fastnsf import FastNSF, OverlappedFastNSF
from himo_amd.synthetic import make_frame

dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pairs = []
for i in range(4):
    f = make_frame(300 + i, n_points=120_000 - 1013 * i)
    p0 = torch.from_numpy(np.ascontiguousarray(f["pc0"][:, :3])).to(dev)
    p1 = torch.from_numpy((f["pc0"][:, :3] + f["flow"]).astype(np.float32)).to(dev)
    pairs.append((p0, p1, f["pose0"], f["pose1"]))
one = FastNSF(device=dev, iters=100)
two = OverlappedFastNSF(device=dev, engines=E, iters=100)
ref = [one.fit(*pairs[i % 4]).clone() for i in range(4)]
got = list(two.fits(pairs[i % 4] for i in range(4)))
torch.cuda.synchronize()
print("same bits:", all(torch.equal(a, b) for a, b in zip(ref, got)))
for name, run in (("one engine", lambda: [one.fit(*pairs[i % 4]) for i in range(K)]),
                  (f"{E} engines in flight", lambda: list(two.fits(pairs[i % 4] for i in range(K)))),
                  ("one engine", lambda: [one.fit(*pairs[i % 4]) for i in range(K)]),
                  (f"{E} engines in flight", lambda: list(two.fits(pairs[i % 4] for i in range(K))))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:22s}: {K / dt:6.2f} frames/s ({dt / K * 1e3:.1f} ms per fit)")
