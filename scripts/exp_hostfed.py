"""GPU experiment: HOST-FED pipeline rate (PCIe-inclusive; not bench.py's `value`, which starts from HBM-resident inputs).
Frames live as host numpy arrays; feeder.SampleFeeder stages and copies them two batches ahead, the pipeline runs flow +
comp_dis, feeder.ResultDrain brings comp_dis back to pinned host memory.  Compared with the same frames resident in HBM."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from himo_amd import _lib
from himo_amd.feeder import ResultDrain, SampleFeeder
from himo_amd.pipeline import HiMoPipeline, Sample
from himo_amd.seflow import spec
from himo_amd.seflow.model import SeFlowNet
from himo_amd.synthetic import make_frame

dev = torch.device("cuda", 0)
B, P, STEPS = 16, 120_000, int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = [make_frame(i, n_points=P) for i in range(B + 2)]
for f in frames:
    f["pc0"] = np.ascontiguousarray(f["pc0"], dtype=np.float32); f["lidar_dt"] = np.ascontiguousarray(f["lidar_dt"], dtype=np.float32)
pipe = HiMoPipeline(SeFlowNet(spec.init_params(0), device=dev, max_points=P, precision="f16x2", max_batch=B), device=dev)

def source(n_batches):
    for b in range(n_batches):
        for k in range(B):
            yield b * B + k, frames[k], frames[k + 1], frames[k + 2]

# resident baseline
samples = [Sample.from_frames(frames[k], frames[k + 1], frames[k + 2], device=dev) for k in range(B)]
for _ in range(3): pipe.run(samples)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS): pipe.run(samples)
pipe.sync_check(); torch.cuda.synchronize()
res = B * STEPS / (time.perf_counter() - t0)

def hostfed(n_batches):
    landed = [0]
    drain = ResultDrain(lambda key, arr: landed.__setitem__(0, landed[0] + 1), device=dev)
    for batch in SampleFeeder(source(n_batches), device=dev, batch=B, depth=2):
        out = pipe.run([s for _, _, s in batch])
        o = out["batch"].offsets_host
        for k, (i, _, _) in enumerate(batch):
            drain.put(i, out["comp_dis"][int(o[k]):int(o[k + 1])])
    drain.close()
    torch.cuda.synchronize()
    return landed[0]

hostfed(3)
t0 = time.perf_counter()
n = hostfed(STEPS)
fed = n / (time.perf_counter() - t0)
mb = (3 * P * frames[0]["pc0"].shape[1] * 4 + P * 4 + P * 12) / 1e6
print(f"resident {res:.1f} frames/s   host-fed {fed:.1f} frames/s ({n} frames, {mb:.2f} MB over PCIe per frame = {fed * mb / 1e3:.2f} GB/s)")
