"""The stages of the training PROGRAM (``seflow.fit.fit`` over .h5 scenes, ssl_label=seflow_auto), each ALONE and then together:
read (h5 -> pinned), upload (pinned -> HBM), labels (auto_labels on a resident pair), step (train_batch on resident, labelled
samples), and the loop fed by ``feeder.TrainFeeder`` against the loop that builds every sample inside the step.
usage (GPU box): python scripts/prof_fit_stages.py [points] [scenes] [sweeps] > gpurun_out/r06_fit_stages.txt"""
import shutil
import sys
import tempfile
import time
import warnings
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from himo_amd.dataset import HDF5Dataset
from himo_amd.feeder import TrainFeeder
from himo_amd.seflow import spec
from himo_amd.seflow.fit import fit, host_sample, make_sample, train_fields, triplets
from himo_amd.seflow.ssl_label import auto_labels
from himo_amd.seflow.train import SeFlowTrainer
from himo_amd.synthetic import make_scene, write_h5_scenes

P = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
N_SCENES = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SWEEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 21
dev = torch.device("cuda", 0)
root = Path(tempfile.mkdtemp(prefix="himo_fit_stages_"))
try:
    with ThreadPoolExecutor(max_workers=N_SCENES) as pool:
        scenes = list(pool.map(lambda sc: make_scene(500 + sc, SWEEPS, n_points=P, scene_id=f"drive{sc:02d}"), range(N_SCENES)))
    write_h5_scenes(root, scenes)
    del scenes
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = HDF5Dataset(root, fields=train_fields("seflow_auto"), zero_copy=True)
    trips = triplets(ds)
    print(f"{N_SCENES} scenes x {SWEEPS} sweeps of {P} points -> {len(trips)} samples; scene file {(root / 'drive00.h5').stat().st_size / 1e6:.0f} MB")

    # ---- read alone: host_sample + ONE copy into pinned memory, 1 / 4 / 8 threads -------------------------------------
    def read_one(t, pins):
        h = host_sample(ds, t, "seflow_auto")
        for k, pin in zip(("pch1", "pc0", "pc1"), pins[:3]):
            np.copyto(pin.numpy()[:len(h[k])], h[k])
        for k, pin in zip(("gm0", "gm1"), pins[3:]):
            np.copyto(pin.numpy()[:len(h[k])], h[k], casting="unsafe")
    warm = [torch.empty((P, 4), dtype=torch.float32).pin_memory() for _ in range(3)] + [torch.empty(P, dtype=torch.uint8).pin_memory() for _ in range(2)]
    t0 = time.perf_counter()
    for t in trips:                                                          # first pass: page cache, file headers
        read_one(t, warm)
    print(f"read alone  (first pass, 1 thread): {1e3 * (time.perf_counter() - t0) / len(trips):7.3f} ms per sample")
    for workers in (1, 4, 8):
        pins = [[torch.empty((P, 4), dtype=torch.float32).pin_memory() for _ in range(3)] + [torch.empty(P, dtype=torch.uint8).pin_memory() for _ in range(2)]
                for _ in range(workers)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(lambda kt: read_one(kt[1], pins[kt[0] % workers]), enumerate(trips)))
        el = time.perf_counter() - t0
        print(f"read alone  ({workers} threads): {1e3 * el / len(trips):7.3f} ms per sample = {len(trips) / el:8.1f} samples/s")

    # ---- upload alone -----------------------------------------------------------------------------------------------
    pin = pins[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        got = [p.to(dev, non_blocking=True) for p in pin]
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 50
    mb = sum(p.numel() * p.element_size() for p in pin) / 1e6
    print(f"upload alone: {1e3 * el:7.3f} ms per sample as five copies ({mb:.1f} MB -> {mb / el / 1e3:.1f} GB/s)")
    for size_mb in (1, 6, 32, 128):
        block = torch.empty(size_mb << 20, dtype=torch.uint8).pin_memory()
        block.to(dev, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            got = block.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        el1 = (time.perf_counter() - t0) / 20
        print(f"upload alone: one pinned block of {size_mb:3d} MB: {1e3 * el1:7.3f} ms -> {size_mb * 1.048576 / el1 / 1e3:.1f} GB/s" +
              ("   (what the feeder issues per sample)" if size_mb == 6 else ""))

    # ---- labels alone -----------------------------------------------------------------------------------------------
    smp = [make_sample(ds, t, dev, "seflow_auto") for t in trips[:8]]
    hs = [host_sample(ds, t, "seflow_auto") for t in trips[:8]]
    gms = [(torch.from_numpy(np.ascontiguousarray(h["gm0"])).to(dev), torch.from_numpy(np.ascontiguousarray(h["gm1"])).to(dev)) for h in hs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(3):
        for s, (g0, g1), h in zip(smp, gms, hs):
            l0, l1 = auto_labels(s[1], s[2], g0, g1, h["pose0"], h["pose1"])
            n_labels = int(torch.maximum(l0.max(), l1.max()).item()) + 1
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / (3 * len(smp))
    dyn = float((smp[0][6] > 0).float().mean().item())
    print(f"labels alone: {1e3 * el:7.3f} ms per pair (resident sweeps; label count read back); sample 0: {smp[0][8] - 1} clusters, {100 * dyn:.1f} % of pc0 dynamic")

    # ---- step alone -------------------------------------------------------------------------------------------------
    tr = SeFlowTrainer(spec.init_params(0), device=dev, max_points=int(P * 1.02), precision="mixed", batch=8)      # a step = ONE pass over its 8 samples
    for k in range(3):
        tr.train_batch(smp, lr=6e-5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(6):
        tr.train_batch(smp, lr=6e-5)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / (6 * len(smp))
    print(f"step alone  : {1e3 * el:7.3f} ms per sample (train_batch of 8 resident, labelled samples in ONE pass: one Adam step per 8) = {1 / el:6.1f} samples/s")
    step_alone = el
    del smp

    # ---- the loop ---------------------------------------------------------------------------------------------------
    for workers, prefetch, lanes, cache in ((0, 0, 1, False), (1, 2, 1, False), (1, 2, 2, False), (1, 2, 3, False), (2, 2, 2, False), (1, 4, 2, False), (2, 2, 2, True), (1, 2, 1, True)):
        fit(ds, trainer=tr, epochs=1, batch_size=8, max_steps=2, log=None, num_workers=workers, prefetch=max(prefetch, 1), label_lanes=lanes, cache_labels=cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fit(ds, trainer=tr, epochs=2, batch_size=8, log=None, num_workers=workers, prefetch=max(prefetch, 1), label_lanes=lanes, cache_labels=cache)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        n = sum(h["samples"] for h in out["history"])
        what = "samples built inside the step loop (num_workers=0)" if workers == 0 else f"TrainFeeder, {workers} reader threads, {lanes} label lanes, {prefetch} samples ahead"
        fd = out["history"][-1].get("feeder")
        extra = "" if not fd else (f"   feeder host ms per sample: read {1e3 * fd['read'] / fd['samples']:.2f} (summed over threads), "
                                   f"upload {1e3 * fd['upload'] / fd['samples']:.2f}, labels {1e3 * fd['labels'] / fd['samples']:.2f}")
        what += f", labels {'generated once, uploaded afterwards' if cache else 'generated every epoch'}"
        per_epoch = ", ".join(f"{1e3 * h['train_seconds'] / h['samples']:.3f}" for h in out["history"])
        extra += f"   ms per sample by epoch: {per_epoch}"
        print(f"fit, 2 epochs ({n} samples, batch_size 8): {1e3 * el / n:7.3f} ms per sample = {n / el:6.1f} samples/s = {step_alone / (el / n):.3f} of the step alone; {what}{extra}")
    ds.close()
finally:
    shutil.rmtree(root, ignore_errors=True)
