"""GPU experiment: evaluator throughput (rows a6-a8): InstanceMetrics.step_frames on 120k-point synthetic sweeps (host
frame dicts in, per-instance records out) next to the pinned numpy/cKDTree oracle on the same frames."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "oracle"))
import torch
import himo_oracle as oracle
from himo_amd.eval import InstanceMetrics
from himo_amd.synthetic import make_frame

frames = [make_frame(500 + i, n_points=120_000, n_instances=30) for i in range(16)]
m = InstanceMetrics("av2")
m.step_frames(frames[:4], res_name="seflowpp_best")
torch.cuda.synchronize()
t0 = time.perf_counter()
for lo in range(0, 16, 8):
    m.step_frames(frames[lo:lo + 8], res_name="seflowpp_best")
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / 16
ref = oracle.InstanceMetrics("av2")
t0 = time.perf_counter()
for f in frames[:4]:
    oracle.eval_frame(ref, f, "seflowpp_best")
cpu = (time.perf_counter() - t0) / 4
print(f"evaluator: HIP path {gpu * 1e3:.2f} ms/frame ({1 / gpu:.0f} frames/s, incl. host->device of the frame dicts); "
      f"numpy + cKDTree oracle {cpu * 1e3:.1f} ms/frame ({1 / cpu:.1f} frames/s, 1 thread)")
