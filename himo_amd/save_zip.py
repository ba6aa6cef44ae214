"""Drop-in for the reference's ``save_zip.py``: flow results -> per-point compensation distances
-> one Feather file per sweep -> a *stored* zip for the leaderboard.

Same public names and wire format as the reference:
    read_output_zip(zip_path, (scene_id, ts)) -> (N,3) float32         save_zip.py:30-54
    write_output_file(comp_dis, (scene_id, ts), output_dir)            save_zip.py:56-81
    zip_res(res_folder, output_file)                                   save_zip.py:84-100
    main(data_dir, res_name)                                           save_zip.py:102-125

What differs is where the arithmetic runs: ``main`` packs sweeps into ragged batches in HBM and
runs the fused HIP path (himo_amd/compdis.py) instead of six numpy passes per sweep, and with
``torch.distributed`` initialised it shards sweeps across ranks (frame i -> rank i % world) so
each rank writes its own Feather files; rank 0 zips.
"""
from __future__ import annotations

import os
import shutil
import time
from pathlib import Path
from typing import Tuple
from zipfile import ZipFile

import numpy as np

from . import feather

COLUMNS = ("comp_dis_x_m", "comp_dis_y_m", "comp_dis_z_m")


def read_output_zip(zip_path: str, sweep_uuid: Tuple[str, int]) -> np.ndarray:
    """(N,3) float32 compensation distances of one sweep; a missing member raises ``KeyError``
    like ``ZipFile.open`` does in the reference."""
    with ZipFile(zip_path, "r") as myzip:
        with myzip.open(f"{sweep_uuid[0]}/{sweep_uuid[1]}.feather") as f:
            table = feather.read_table(f.read())
    return np.stack([table[c].astype(np.float32) for c in COLUMNS], axis=1)


def _frame_table(compensation_dis):
    """Feather V2 file image (bytes-like) with the three float32 columns of save_zip.py:74-80 (own writer: himo_amd/feather.py;
    the reference goes through pandas -> pyarrow, which the GPU box image does not have).  The framing of a sweep depends on its
    row count alone (cached); the columns are gathered from the (N,3) rows straight into the file image."""
    cd = np.asarray(compensation_dis)
    if cd.dtype != np.float32:
        cd = cd.astype(np.float32)
    return feather.write_matrix(cd, COLUMNS)


def write_output_file(compensation_dis, sweep_uuid: Tuple[str, int], output_dir: Path) -> None:
    """``<output_dir>/<scene_id>/<timestamp>.feather`` with three float32 columns."""
    output_log_dir = Path(output_dir) / sweep_uuid[0]
    output_log_dir.mkdir(exist_ok=True, parents=True)
    with open(output_log_dir / f"{sweep_uuid[1]}.feather", "wb") as fh:
        fh.write(memoryview(_frame_table(compensation_dis)))


def zip_res(res_folder, output_file="submit.zip"):
    """Zip every ``<scene>/<ts>.feather`` under ``res_folder`` (stored, not deflated -- the default
    ``ZipFile`` mode the reference uses) and remove the scene folders afterwards."""
    res_folder = str(res_folder)
    all_scenes = [f for f in os.listdir(res_folder) if os.path.isdir(os.path.join(res_folder, f))]
    with ZipFile(output_file, "w") as myzip:
        for scene in all_scenes:
            scene_folder = os.path.join(res_folder, scene)
            for log in os.listdir(scene_folder):
                if log.endswith(".feather") and os.path.isfile(os.path.join(scene_folder, log)):
                    myzip.write(os.path.join(scene_folder, log), arcname=os.path.join(scene, log))
    for scene in all_scenes:
        shutil.rmtree(os.path.join(res_folder, scene), ignore_errors=True)
    print(f"Zipped results to {res_folder} into {output_file}. Submit your result by uploading this zip file.")
    return output_file


class ZipSink:
    """Streams sweeps straight into a stored zip (no temporary Feather files).  Same member
    names and bytes-level format as ``write_output_file`` + ``zip_res``."""

    def __init__(self, output_file):
        self.path = str(output_file)
        self._zip = ZipFile(self.path, "w")

    def add(self, compensation_dis, sweep_uuid: Tuple[str, int]) -> None:
        self._zip.writestr(f"{sweep_uuid[0]}/{sweep_uuid[1]}.feather", memoryview(_frame_table(compensation_dis)))

    def close(self):
        self._zip.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist
    return 0, 1, None


OVERLAP = True          # read / stage / copy ahead and write behind on background threads (False: the reference's plain serial loop)


def run_dataset(dataset, res_name: str, output_dir: Path, batch_frames: int = 32, sensor_dt: float = 0.1, overlap: bool | None = None) -> int:
    """Shared body of ``main``: iterate ``dataset`` (frame i on rank i % world), batch sweeps into HBM, run the fused path,
    write one Feather per sweep.  Returns the sweeps written by this rank.

    With ``overlap`` (default) the reference's serial read -> compute -> write loop (save_zip.py:112-123) becomes three
    overlapped stages: a feeder thread reads the frames, packs them into pinned memory and copies them to the device two
    batches ahead (``feeder.BatchFeeder``), the launch thread only enqueues the fused kernel, and the compensation distances
    leave through pinned buffers to a writer thread that encodes and writes the Feather files (``feeder.ResultDrain``)."""
    from .compdis import CompDisEngine, FrameBatch

    overlap = OVERLAP if overlap is None else overlap
    rank, world, _ = _dist()
    eng = CompDisEngine(max_frames=batch_frames)
    mine = list(range(rank, len(dataset), world))

    def batches():
        for lo in range(0, len(mine), batch_frames):
            frames = [dataset[i] for i in mine[lo:lo + batch_frames]]
            for f in frames:
                if len(f["lidar_dt"]) == 0:
                    raise ValueError("max() arg is an empty sequence")       # save_zip.py:120
            yield frames

    written = 0
    if not overlap:
        for frames in batches():
            batch = FrameBatch.from_frames(frames, res_name)
            host = eng.run(batch, sensor_dt=sensor_dt)["comp_dis"].cpu().numpy()      # one D2H copy per batch
            o = batch.offsets_host
            for k, f in enumerate(frames):
                write_output_file(host[o[k]:o[k + 1]], (f["scene_id"], str(f["timestamp"])), output_dir)
                written += 1
        return written

    from .feeder import BatchFeeder, ResultDrain
    dev = eng.device

    def build(frames, upload):
        b = FrameBatch.from_frames(frames, res_name, device=dev, upload=upload)
        return (frames, b), [b.offsets, b.pose0, b.pose1, b.pc0, b.lidar_dt, b.flow]
    made = set()
    made_lock = __import__("threading").Lock()

    def write_sweep(key, arr):                  # (a writer thread; arr is a view of the drain's pinned buffer, gone when this returns)
        scene_dir = Path(output_dir) / key[0]
        if key[0] not in made:
            with made_lock:
                scene_dir.mkdir(exist_ok=True, parents=True)
                made.add(key[0])
        with open(scene_dir / f"{key[1]}.feather", "wb") as fh:
            fh.write(memoryview(_frame_table(arr)))
    # one Feather file per sweep, independent of each other: four writer threads (the column gather and the file write release the
    # GIL), fed views of the pinned drain buffers -- a single writer thread was the bound of this loop (1.25 k sweeps/s in round 4)
    drain = ResultDrain(write_sweep, device=dev, threads=4, copy=False)
    feed = BatchFeeder(batches(), build, device=dev)
    try:
        for frames, batch in feed:
            cd = eng.run(batch, sensor_dt=sensor_dt)["comp_dis"]
            o = batch.offsets_host
            for k, f in enumerate(frames):
                drain.put((f["scene_id"], str(f["timestamp"])), cd[int(o[k]):int(o[k + 1])])
                written += 1
    except BaseException:
        feed.close()
        try:
            drain.close()                       # the sweeps already computed still reach the disk, as in the serial loop
        except BaseException:
            pass
        raise
    drain.close()
    return written


def main(data_dir: str = "/home/kin/data/av2/h5py/sensor/himo/demo", res_name: str = "seflowpp_best",
         batch_frames: int = 32, allow_dropped_eval: bool | None = None):
    """save_zip.py:102-125.  Under ``torchrun`` (one rank per GPU) the sweeps are sharded i % world, every rank writes its
    own Feather files, and rank 0 zips once all of them are on disk (distenv.process_group joins / leaves the job's group;
    a rank that fails still reaches the rendezvous, so nobody zips a partial result or waits for a dead process)."""
    from . import distenv
    from .dataset import open_dataset

    data_dir = Path(data_dir)
    output_dir = data_dir / "results"
    output_dir.mkdir(exist_ok=True, parents=True)
    with distenv.process_group() as (rank, world):
        err = None
        try:
            dataset = open_dataset(data_dir, vis_name=res_name, eval=True, allow_dropped_eval=allow_dropped_eval)
            run_dataset(dataset, res_name, output_dir, batch_frames=batch_frames)
        except Exception as e:                                   # (an interrupt leaves at once; the launcher ends the job)
            err = e
        distenv.rendezvous(err, "its Feather files, but no submit zip was written")   # every rank's files are on disk -- or somebody failed
        if rank == 0:
            zip_res(output_dir, output_file=f"{output_dir}/{res_name}-submit.zip")
        if world > 1:
            distenv.all_ranks_ok(True)                           # nobody leaves before the zip exists


def _cli(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="flow -> comp_dis -> leaderboard zip (MI355X path)")
    ap.add_argument("--data_dir", default="/home/kin/data/av2/h5py/sensor/himo/demo")
    ap.add_argument("--res_name", default="seflowpp_best")
    ap.add_argument("--batch_frames", type=int, default=32)
    ap.add_argument("--allow_dropped_eval", action="store_true", default=None,
                    help="skip index_eval.pkl sweeps that have no successor sweep in their h5 scene instead of failing")
    a = ap.parse_args(argv)
    main(a.data_dir, a.res_name, a.batch_frames, a.allow_dropped_eval)


if __name__ == "__main__":
    start_time = time.time()
    _cli()
    print(f"Time used: {time.time() - start_time:.2f} s")
