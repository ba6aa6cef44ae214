"""Same-process A/B of a SeFlowTrainer attribute: the training step timed with the attribute off / on, interleaved.
usage (GPU box): python scripts/exp_train_toggle.py <attribute> [rounds]      e.g. stuffed_dgrad"""
import sys, time, types
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

attr = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args = types.SimpleNamespace(frames_per_step=1, points=120_000, train_precision="mixed", train_batchnorm="batch", cloud="uniform")
step, trainer = bench.make_train_step(args, 0, torch.device("cuda", 0), {})
for on in (False, True):
    setattr(trainer, attr, on)
    for _ in range(4):
        step()
torch.cuda.synchronize()
for r in range(rounds):
    for on in (False, True):
        setattr(trainer, attr, on)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        print(f"{attr} = {on!s:5s}: {ms:.3f} ms per step = {1e3 / ms:.2f} frames/s", flush=True)
