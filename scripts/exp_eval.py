"""GPU experiment: evaluator throughput (rows a6-a8) on 120k-point synthetic sweeps with 30 instances:
  resident   device-resident EvalBatch objects -> InstanceMetrics.step_batch (kernels + pipelined record readback + host
             bucket bookkeeping): the number comparable with bench.py's "inputs already in HBM" convention,
  host-fed   host frame dicts -> feeder.EvalFeeder (pinned staging, side-stream copies) -> step_batch,
  serial     InstanceMetrics.step_frames (pack, upload, run, digest -- one batch at a time),
next to the pinned numpy + cKDTree oracle on the same frames."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "oracle"))
import torch
import himo_oracle as oracle
from himo_amd.eval import EvalBatch, InstanceMetrics
from himo_amd.feeder import EvalFeeder
from himo_amd.synthetic import make_frame

B, NB = 16, 12
frames = [make_frame(500 + i, n_points=120_000, n_instances=30) for i in range(B * 2)]
dev = torch.device("cuda", 0)
m = InstanceMetrics("av2")
m.step_frames(frames[:4], res_name="seflowpp_best")                 # warm-up (workspace growth)
torch.cuda.synchronize()

ebs = [EvalBatch.from_frames(frames[k * B:(k + 1) * B], "seflowpp_best", device=dev) for k in range(2)]
for eb in ebs:
    m.step_batch(eb)
m.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(NB):
    m.step_batch(ebs[k % 2])
m.flush(); torch.cuda.synchronize()
resident = (time.perf_counter() - t0) / (NB * B)

for eb in EvalFeeder((frames[(k % 2) * B:(k % 2 + 1) * B] for k in range(4)), res_name="seflowpp_best", device=dev):
    m.step_batch(eb)                                               # warm-up: the pinned staging arenas are allocated once
m.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for eb in EvalFeeder((frames[(k % 2) * B:(k % 2 + 1) * B] for k in range(NB)), res_name="seflowpp_best", device=dev):
    m.step_batch(eb)
m.flush(); torch.cuda.synchronize()
fed = (time.perf_counter() - t0) / (NB * B)

m.step_frames(frames[:B], res_name="seflowpp_best")                  # warm-up at this batch size (pinned record buffers, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(4):
    m.step_frames(frames[(k % 2) * B:(k % 2 + 1) * B], res_name="seflowpp_best")
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / (4 * B)

ref = oracle.InstanceMetrics("av2")
t0 = time.perf_counter()
for f in frames[:4]:
    oracle.eval_frame(ref, f, "seflowpp_best")
cpu = (time.perf_counter() - t0) / 4
print(f"evaluator, {B} sweeps of 120k points per batch: resident {resident * 1e3:.3f} ms/sweep ({1 / resident:.0f} sweeps/s); "
      f"host-fed through EvalFeeder {fed * 1e3:.3f} ms/sweep ({1 / fed:.0f} sweeps/s); serial step_frames {serial * 1e3:.2f} ms/sweep "
      f"({1 / serial:.0f} sweeps/s); numpy + cKDTree oracle {cpu * 1e3:.1f} ms/sweep ({1 / cpu:.1f} sweeps/s, 1 thread)")
