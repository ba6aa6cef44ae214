"""FastNSF's forward kernel with one ingredient removed at a time (scripts/exp_nsf_forward_ablation.sh builds the copies): average launch
time of the three kernels of an optimiser iteration over one 100-iteration fit of a 120k-point pair, each library in its own interpreter.
usage (GPU box): python scripts/exp_nsf_forward_ablation.py"""
import os, subprocess, sys
from pathlib import Path
R = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1:
    sys.path.insert(0, str(R))
    import numpy as np, torch
    from himo_amd import _lib
    from himo_amd.fastnsf import FastNSF
    from himo_amd.synthetic import make_frame
    dev = torch.device("cuda", 0)
    f = make_frame(300, n_points=120_000)
    p0 = torch.from_numpy(np.ascontiguousarray(f["pc0"][:, :3])).to(dev)
    p1 = torch.from_numpy((f["pc0"][:, :3] + f["flow"]).astype(np.float32)).to(dev)
    eng = FastNSF(device=dev, iters=100)
    eng.fit(p0, p1, f["pose0"], f["pose1"])
    torch.cuda.synchronize()
    _lib.prof_start("nsf_")
    eng.fit(p0, p1, f["pose0"], f["pose1"])
    torch.cuda.synchronize()
    got = _lib.prof_stop()
    print(sys.argv[1].ljust(28), "  ".join(f"{k} {1e3 * v['avg_ms']:7.1f} us x {v['count']}" for k, v in sorted(got.items())))
    sys.exit(0)
libs = [("shipped", R / "himo_amd" / "libhimo_amd.so")] + [(v, R / "build" / "variants" / f"nsf_fwd_{v}" / "libhimo_amd.so") for v in ("nospill", "nolds", "nomfma", "nomask")]
for name, lib in libs:
    if lib.exists():
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, HIMO_AMD_LIB=str(lib)))
