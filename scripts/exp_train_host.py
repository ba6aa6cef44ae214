"""Is the training step host-bound?  Per step: the wall time of a synchronised step, and the host time spent ENQUEUEING a step (no
synchronisation until the end of 20 steps).  If the enqueue time is close to the step time the Python / HIP launch path is the limit.
usage (GPU box): python scripts/exp_train_host.py [samples per pass = 1]"""
import sys, time, types
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

TB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = types.SimpleNamespace(frames_per_step=TB, points=120_000, train_precision="mixed", train_batchnorm="batch", cloud="uniform", train_batch=TB,
                             sample_sets=3)
dev = torch.device("cuda", 0)
res = {}
step, trainer = bench.make_train_step(args, 0, dev, res)
for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
    torch.cuda.synchronize()
t_sync = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20):
    step()
t_enq = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
print(f"synchronised step {1e3 * t_sync:.2f} ms; enqueue only {1e3 * t_enq:.2f} ms per step; 20 steps back to back {1e3 * t_all:.2f} ms per step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(24)
pstats.Stats(pr).sort_stats("cumulative").print_stats("train.py|ssl_loss", 24)
