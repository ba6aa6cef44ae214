"""Do a `mixed` training run (fp16-split forward, two-term bf16 gradients) and an fp32-class one (`bf16x3`, and `f32`) follow the same
trajectory?  200 optimiser steps from one initialisation on 5 rotating samples of one synthetic drive, BatchNorm in training mode;
per step the loss of each run, at the end the flow of a held-out sample under each final model.
usage: python scripts/exp_trajectory.py [points] [steps] [lr] > gpurun_out/r06_train_trajectory.txt"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from himo_amd.dataset import ListDataset
from himo_amd.seflow import spec
from himo_amd.seflow.fit import make_sample, triplets
from himo_amd.seflow.train import SeFlowTrainer
from himo_amd.synthetic import make_scene

P = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
LR = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
dev = torch.device("cuda", 0)
ds, held = ListDataset(make_scene(7, 7, n_points=P)), ListDataset(make_scene(8, 3, n_points=P))
samples = [make_sample(ds, t, dev, "flow_instance_id") for t in triplets(ds)[:5]]
held_s = make_sample(held, (0, 1, 2), dev, "flow_instance_id")
runs = {}
for prec in ("mixed", "bf16x3", "f32", "mixed"):
    tr = SeFlowTrainer(spec.init_params(3, fresh_bn=True), device=dev, max_points=P, precision=prec, batchnorm="batch")
    t0 = time.perf_counter()
    losses = [tr.train_batch([samples[k % len(samples)]], lr=LR) for k in range(STEPS)]
    curve = torch.stack(losses).cpu().numpy().astype(np.float64)
    el = time.perf_counter() - t0
    flow = tr.forward(*held_s[:6], training=False)[:, :3].cpu().numpy()
    val = float(tr.loss_only(*held_s).item())
    name = prec if prec not in runs else prec + "_again"
    runs[name] = {"curve": curve, "flow": flow, "val": val, "ms_per_step": 1e3 * el / STEPS}
    del tr
    torch.cuda.empty_cache()
ref = runs["bf16x3"]
print(f"{STEPS} steps, {P} points, lr {LR}; loss at steps 0 / 49 / 99 / 199 and held-out loss, per arithmetic:")
for name, r in runs.items():
    c = r["curve"]
    print(f"  {name:12s} {c[0]:.6f} {c[min(49, STEPS - 1)]:.6f} {c[min(99, STEPS - 1)]:.6f} {c[-1]:.6f}   held-out {r['val']:.6f}   {r['ms_per_step']:.2f} ms per step")
print("against the bf16x3 run: worst relative loss difference over the steps (all / first 50), held-out flow max abs and mean EPE difference (m):")
out = {}
for name, r in runs.items():
    if name == "bf16x3":
        continue
    rel = np.abs(r["curve"] - ref["curve"]) / np.maximum(np.abs(ref["curve"]), 1e-12)
    d = np.linalg.norm(r["flow"] - ref["flow"], axis=1)
    out[name] = {"loss_rel_max": float(rel.max()), "loss_rel_max_first50": float(rel[:50].max()), "loss_rel_mean": float(rel.mean()),
                 "flow_max_abs": float(np.abs(r["flow"] - ref["flow"]).max()), "flow_mean_epe": float(d.mean()),
                 "held_out_loss_rel": abs(r["val"] - ref["val"]) / abs(ref["val"])}
    print(f"  {name:12s} {out[name]}")
print("flow magnitude of the bf16x3 model on the held-out sample: mean", float(np.linalg.norm(ref["flow"], axis=1).mean()), "max", float(np.abs(ref["flow"]).max()))
print(json.dumps(out))
