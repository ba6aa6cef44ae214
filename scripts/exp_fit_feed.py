"""What does feeding cost the training loop?  fit() over .h5 scenes, per-sample time, for: stored labels vs generated labels,
interpreter switch interval, reader / lane counts.  usage: python scripts/exp_fit_feed.py"""
import shutil, sys, tempfile, time, warnings
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd.dataset import HDF5Dataset
from himo_amd.seflow import spec
from himo_amd.seflow.fit import fit, make_sample, train_fields, triplets
from himo_amd.seflow.train import SeFlowTrainer
from himo_amd.synthetic import make_scene, write_h5_scenes

P, N_SCENES, SWEEPS = 120_000, 4, 21
dev = torch.device("cuda", 0)
root = Path(tempfile.mkdtemp(prefix="himo_fit_feed_"))
try:
    with ThreadPoolExecutor(max_workers=N_SCENES) as pool:
        scenes = list(pool.map(lambda sc: make_scene(500 + sc, SWEEPS, n_points=P, scene_id=f"drive{sc:02d}"), range(N_SCENES)))
    write_h5_scenes(root, scenes)
    del scenes
    tr = SeFlowTrainer(spec.init_params(0), device=dev, max_points=int(P * 1.02), precision="mixed")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dss = {k: HDF5Dataset(root, fields=train_fields(k), zero_copy=True) for k in ("seflow_auto", "flow_instance_id")}
    smp = [make_sample(dss["seflow_auto"], t, dev, "seflow_auto") for t in triplets(dss["seflow_auto"])[:8]]
    for _ in range(3):
        tr.train_batch(smp, lr=6e-5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        tr.train_batch(smp, lr=6e-5)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / 64
    print(f"step alone: {1e3 * alone:.3f} ms per sample")
    del smp
    for label, workers, lanes, si in (("flow_instance_id", 1, 1, 0.005), ("flow_instance_id", 2, 1, 0.005), ("seflow_auto", 1, 1, 0.005),
                                      ("seflow_auto", 1, 1, 0.0005), ("seflow_auto", 1, 1, 0.00005), ("seflow_auto", 2, 2, 0.0005),
                                      ("seflow_auto", 1, 1, 0.02)):
        sys.setswitchinterval(si)
        ds = dss[label]
        fit(ds, trainer=tr, epochs=1, batch_size=8, max_steps=2, log=None, num_workers=workers, label_lanes=lanes, ssl_label=label)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fit(ds, trainer=tr, epochs=2, batch_size=8, log=None, num_workers=workers, label_lanes=lanes, ssl_label=label)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        n = sum(h["samples"] for h in out["history"])
        fd = out["history"][-1]["feeder"]
        print(f"{label:17s} readers {workers} lanes {lanes} switch interval {si * 1e3:5.2f} ms: {1e3 * el / n:7.3f} ms per sample = {alone / (el / n):.3f} of the step alone; "
              f"feeder host ms: read {1e3 * fd['read'] / fd['samples']:.2f} upload {1e3 * fd['upload'] / fd['samples']:.2f} labels {1e3 * fd['labels'] / fd['samples']:.2f}")
finally:
    shutil.rmtree(root, ignore_errors=True)
