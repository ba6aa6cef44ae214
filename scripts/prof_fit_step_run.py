"""The fed training program for scripts/prof_fit_step.sh: 4 scenes x 21 sweeps of 120k points, batch_size 8, 3 epochs (76 samples = 10 steps each)."""
import shutil, sys, tempfile, time, warnings
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd.dataset import HDF5Dataset
from himo_amd.seflow import spec
from himo_amd.seflow.fit import fit, train_fields
from himo_amd.seflow.train import SeFlowTrainer
from himo_amd.synthetic import make_scene, write_h5_scenes

P = 120_000
dev = torch.device("cuda", 0)
root = Path(tempfile.mkdtemp(prefix="himo_fit_step_"))
try:
    with ThreadPoolExecutor(max_workers=4) as pool:
        scenes = list(pool.map(lambda sc: make_scene(500 + sc, 21, n_points=P, scene_id=f"drive{sc:02d}"), range(4)))
    write_h5_scenes(root, scenes)
    del scenes
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = HDF5Dataset(root, fields=train_fields("seflow_auto"), zero_copy=True)
    tr = SeFlowTrainer(spec.init_params(0), device=dev, max_points=int(P * 1.02), precision="mixed", batch=8)
    t0 = time.perf_counter()
    out = fit(ds, trainer=tr, epochs=3, batch_size=8, log=None)
    torch.cuda.synchronize()
    for h in out["history"]:
        print(f"epoch: {h['samples']} samples, {1e3 * h['train_seconds'] / h['samples']:.3f} ms per sample; feeder {h.get('feeder')}")
    ds.close()
finally:
    shutil.rmtree(root, ignore_errors=True)
