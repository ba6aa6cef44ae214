"""What do the weight gradients on the side streams cost the training step?  The step timed with all of them, without the linear / 1x1
ones, without the 3x3 ones and without any (the entry points replaced by no-ops: the parameters stop learning, the timing is what is
asked).  If the side streams were hidden under the main stream's chain the four numbers would agree.
usage (GPU box): python scripts/exp_train_skip_wgrad.py"""
import sys, time, types
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

args = types.SimpleNamespace(frames_per_step=1, points=120_000, train_precision="mixed", train_batchnorm="batch", cloud="uniform")
step, tr = bench.make_train_step(args, 0, torch.device("cuda", 0), {})
for _ in range(6):
    step()
torch.cuda.synchronize()
lib = tr.lib
orig = {n: getattr(lib, n) for n in ("himo_linear_wgrad_ex", "himo_conv3x3_wgrad_batch", "himo_conv3x3_wgrad_batch_bias")}
skip = lambda *a: 0


def timeit(tag):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print(f"{tag:42s}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step", flush=True)


def use(linear: bool, conv: bool):
    for n, f in orig.items():
        on = linear if n == "himo_linear_wgrad_ex" else conv
        setattr(lib, n, f if on else skip)
        setattr(tr.head.lib, n, f if on else skip)


for r in range(2):
    use(True, True); timeit("all weight gradients")
    use(False, True); timeit("without the linear / 1x1 weight gradients")
    use(True, False); timeit("without the 3x3 weight gradients")
    use(False, False); timeit("without any weight gradient")
use(True, True)
