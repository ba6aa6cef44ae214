"""save_zip.run_dataset: serial loop vs the overlapped feeder -> kernel -> drain form, sweeps/s on 120k-point sweeps
held in host memory (the dataset read is a dict lookup here, so this isolates packing + copies + Feather encoding)."""
import sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from himo_amd import save_zip
from himo_amd.dataset import ListDataset
from himo_amd.synthetic import make_frame

frames = [make_frame(i, n_points=120_000) for i in range(96)]
ds = ListDataset(frames)
for overlap in (False, True, False, True):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        n = save_zip.run_dataset(ds, "seflowpp_best", Path(d), batch_frames=16, overlap=overlap)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"overlap={overlap}: {n / dt:.1f} sweeps/s ({dt * 1e3 / n:.2f} ms per sweep)", flush=True)
