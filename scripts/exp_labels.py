"""Where do a label pair's 2 ms go, and why 8 ms beside a training step?  auto_labels on resident 120k-point pairs: (a) alone on the
main thread, (b) on a second thread while the main thread idles, (c) ... while the main thread spins in Python (holds the GIL, no GPU
work), (d) ... while the main thread runs training steps.  usage: python scripts/exp_labels.py [switch_interval_s]
(under rocprofv3 --kernel-trace --stats with HIMO_EXP_LABELS_ONLY=1: the kernels of (a) only)"""
import os, sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from himo_amd.seflow import spec
from himo_amd.seflow.ssl_label import auto_labels
from himo_amd.seflow.train import SeFlowTrainer
from himo_amd.synthetic import make_scene

if len(sys.argv) > 1:
    sys.setswitchinterval(float(sys.argv[1]))
dev = torch.device("cuda", 0)
P = 120_000
fr = make_scene(3, 6, n_points=P)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
pairs = [(up(fr[k]["pc0"]), up(fr[k + 1]["pc0"]), up(fr[k]["gm0"]), up(fr[k + 1]["gm0"]), fr[k]["pose0"], fr[k]["pose1"]) for k in range(5)]


def label_loop(n, stream=None, out=None):
    torch.cuda.set_device(dev)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        t0 = time.perf_counter()
        for k in range(n):
            l0, l1 = auto_labels(*pairs[k % len(pairs)])
            nl = int(torch.maximum(l0.max(), l1.max()).item())
        el = (time.perf_counter() - t0) / n
    if out is not None:
        out.append(el)
    return el


label_loop(5)
torch.cuda.synchronize()
print(f"(a) alone, main thread: {1e3 * label_loop(40):.3f} ms per pair")
if os.environ.get("HIMO_EXP_LABELS_ONLY"):
    sys.exit(0)
for prio in (0, -1):
    st = torch.cuda.Stream(device=dev, priority=prio)
    out = []
    th = threading.Thread(target=label_loop, args=(40, st, out)); th.start(); th.join()
    print(f"(b) second thread, main idle, stream priority {prio}: {1e3 * out[0]:.3f} ms per pair")
    out, stop = [], [False]
    th = threading.Thread(target=label_loop, args=(40, st, out)); th.start()
    spins = 0
    while th.is_alive():
        spins += 1
    print(f"(c) second thread, main spinning in Python: {1e3 * out[0]:.3f} ms per pair")

tr = SeFlowTrainer(spec.init_params(0), device=dev, max_points=P, precision="mixed")
g = torch.Generator(device=dev); g.manual_seed(1)
lab = (torch.randint(1, 31, (P,), generator=g, device=dev, dtype=torch.int32) * (torch.rand(P, generator=g, device=dev) < 0.1).to(torch.int32))
smp = (pairs[0][0], pairs[1][0], pairs[2][0], fr[0]["pose0"], fr[1]["pose0"], fr[1]["pose1"], lab, lab.clone(), 31)
for _ in range(3):
    tr.train_step(*smp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.train_step(*smp)
torch.cuda.synchronize()
print(f"train step alone: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")
for prio in (0, -1):
    st = torch.cuda.Stream(device=dev, priority=prio)
    out = []
    th = threading.Thread(target=label_loop, args=(30, st, out)); th.start()
    n, t0 = 0, time.perf_counter()
    while th.is_alive():
        tr.train_step(*smp); n += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"(d) second thread beside training steps, stream priority {prio}: {1e3 * out[0]:.3f} ms per pair; step {1e3 * el / n:.3f} ms "
          f"({n} steps, {30} pairs in {el:.3f} s)")
