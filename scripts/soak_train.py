"""GPU soak of the training step's side streams: N optimiser steps on rotating 120k-point samples, (a) twice with the side streams
on, (b) once with every weight gradient back on the main stream -- the parameter bits after N steps must be the same in all three
runs (the streams reorder work, not arithmetic).  A race between the data-gradient chain and a weight gradient reading its operands
would show up as a difference.  usage: python scripts/soak_train.py [steps]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd.dataset import ListDataset
from himo_amd.seflow import spec
from himo_amd.seflow.fit import make_sample, triplets
from himo_amd.seflow.train import SeFlowTrainer
from himo_amd.synthetic import make_frame

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
ds = ListDataset([make_frame(1200 + i, n_points=120_000 - 2111 * i, scene_id="soak") for i in range(5)])
trips = triplets(ds)
samples = [make_sample(ds, t, dev) for t in trips]


def run(side: bool, precision: str = "mixed", batchnorm: str = "batch", steps: int = N, leave_room: bool = False):
    tr = SeFlowTrainer(spec.init_params(9), device=dev, max_points=121_000, batchnorm=batchnorm, precision=precision)
    tr.set_side_streams(side)
    # the product default gives the side-stream weight gradients ONE block per CU (flags bit 2: another split of the pixel chunks, so
    # other bits than the main-stream launch); the race check compares like with like, the default is checked run against run below
    tr.wgrad_leave_room = leave_room
    losses = [float(tr.train_batch([samples[k % len(samples)]], lr=2e-4).item()) for k in range(steps)]
    torch.cuda.synchronize()
    return tr.flat_p.clone(), losses


t0 = time.perf_counter()
a, la = run(True)
b, lb = run(True)
c, lc = run(False)
print(f"{N} steps x 3 runs in {time.perf_counter() - t0:.1f} s; loss first / last {la[0]:.6f} / {la[-1]:.6f}")
print("side streams on vs on :", "same bits" if torch.equal(a, b) and la == lb else f"{int((a != b).sum())} parameters differ")
print("side streams on vs off:", "same bits" if torch.equal(a, c) and la == lc else f"{int((a != c).sum())} parameters differ")
ok = torch.equal(a, b) and torch.equal(a, c)
d, ld = run(True, leave_room=True)
e, le = run(True, leave_room=True)
print("side streams on, one weight-gradient block per CU (the default), run vs run:", "same bits" if torch.equal(d, e) and ld == le else f"{int((d != e).sum())} parameters differ",
      f"; loss last {ld[-1]:.6f} (two blocks per CU: {la[-1]:.6f})")
ok = ok and torch.equal(d, e)
# the other arithmetics and the frozen-BatchNorm mode take other branches of the backward pass (float32: flipped weights through ONE
# scratch buffer; frozen: bias column sums on the side stream): 8 steps each, on vs off
for prec, bn in (("bf16x3", "batch"), ("f32", "batch"), ("mixed", "frozen")):
    x, lx = run(True, prec, bn, 8)
    y, ly = run(False, prec, bn, 8)
    same = torch.equal(x, y) and lx == ly
    ok = ok and same
    print(f"{prec:6s} / BatchNorm {bn:6s}: side streams on vs off:", "same bits" if same else f"{int((x != y).sum())} parameters differ")
sys.exit(0 if ok else 1)
