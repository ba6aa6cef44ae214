"""save_zip.run_dataset: serial loop vs the overlapped feeder -> kernel -> drain form, sweeps/s on 120k-point sweeps
held in host memory (the dataset read is a dict lookup here, so this isolates packing + copies + Feather encoding), and the
stages of the overlapped form on their own: the feeder (pack into pinned memory + copy to the device) and the drain (copy back +
encode + write one Feather file per sweep)."""
import sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from himo_amd import save_zip
from himo_amd.compdis import FrameBatch
from himo_amd.dataset import ListDataset
from himo_amd.feeder import BatchFeeder, ResultDrain
from himo_amd.synthetic import make_frame

N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
base = [make_frame(i, n_points=120_000) for i in range(48)]
frames = [dict(base[i % 48], timestamp=base[i % 48]["timestamp"] + 1000 * i) for i in range(N)]
ds = ListDataset(frames)
for overlap in (False, True, False, True, True):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        n = save_zip.run_dataset(ds, "seflowpp_best", Path(d), batch_frames=16, overlap=overlap)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"overlap={overlap}: {n / dt:.1f} sweeps/s ({dt * 1e3 / n:.2f} ms per sweep)", flush=True)

dev = torch.device("cuda:0")


def build(fr, upload):
    b = FrameBatch.from_frames(fr, "seflowpp_best", device=dev, upload=upload)
    return (fr, b), [b.offsets, b.pose0, b.pose1, b.pc0, b.lidar_dt, b.flow]


for rep in range(2):
    t0 = time.perf_counter()
    n = 0
    for fr, b in BatchFeeder((frames[lo:lo + 16] for lo in range(0, N, 16)), build, device=dev):
        n += len(fr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"feeder alone (pack + pinned + copy to the device): {n / dt:.1f} sweeps/s", flush=True)

cd = torch.randn(120_000, 3, device=dev)
for threads in (1, 4, 8):
    with tempfile.TemporaryDirectory() as d:
        drain = ResultDrain(lambda key, arr: save_zip.write_output_file(arr, key, Path(d)), device=dev, threads=threads, copy=False)
        t0 = time.perf_counter()
        for i in range(N):
            drain.put((f"scene{i % 4}", str(i)), cd)
        drain.close()
        dt = time.perf_counter() - t0
    print(f"drain alone, {threads} writer thread(s) (copy back + Feather + file): {N / dt:.1f} sweeps/s", flush=True)
