import sys, time, types
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, bench
dev = torch.device("cuda", 0)
for tb in (1, 8):
    args = types.SimpleNamespace(frames_per_step=1, points=120_000, train_precision="mixed", train_batchnorm="batch", cloud="uniform", train_batch=tb)
    res = {}
    step, tr = bench.make_train_step(args, 0, dev, res)
    for _ in range(4): step()
    torch.cuda.synchronize()
    for room in (True, False, True, False):
        tr.wgrad_leave_room = room
        for _ in range(2): step()
        torch.cuda.synchronize()
        n = 40 if tb == 1 else 8
        t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        print(f"train-batch {tb} wgrad_leave_room={room}: {1e3 * el:.2f} ms per step = {tb / el:.1f} frames/s", flush=True)
    del step, tr; torch.cuda.empty_cache()
