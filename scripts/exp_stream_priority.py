"""Does a HIGH-priority main stream help the training step?  The data-gradient chain is the critical path and the weight gradients on
the side streams compete with it for CUs: the step is timed with the caller's stream at the default priority and at the highest one
(side streams at the default), interleaved.  usage (GPU box): python scripts/exp_stream_priority.py"""
import sys, time, types
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
args = types.SimpleNamespace(frames_per_step=1, points=120_000, train_precision="mixed", train_batchnorm="batch", cloud="uniform")
dev = torch.device("cuda", 0)
step, tr = bench.make_train_step(args, 0, dev, {})
streams = {"default stream": None, "own stream, default priority": torch.cuda.Stream(device=dev), "own stream, highest priority": torch.cuda.Stream(device=dev, priority=-1)}
for _ in range(4):
    step()
torch.cuda.synchronize()
for r in range(3):
    for name, st in streams.items():
        ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.default_stream(dev))
        with ctx:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                step()
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        print(f"{name:32s}: {ms:.3f} ms per step = {1e3 / ms:.2f} frames/s", flush=True)
