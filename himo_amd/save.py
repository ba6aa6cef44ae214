"""Counterpart of the reference's ``OpenSceneFlow/save.py`` call site (README.md:46-54:
``python save.py checkpoint=... dataset_path=...``): run the scene-flow network over a dataset and attach the
``(N,3) float32`` flow -- INCLUDING ego motion, row-aligned with ``pc0`` -- to every frame under ``<res_name>``,
which is exactly what ``save_zip.py:117`` / ``eval.py:302`` read back.

The reference's ``save.py`` itself is in the absent submodule, so only the data contract is reproduced: the key
name defaults to the checkpoint's stem (``seflowpp_best`` for ``seflowpp_best.ckpt``, README.md:50), frames are
walked in dataset order, and under ``torchrun`` frame i goes to rank i % world.  Results are written through a
``sink(frame_index, frame, flow)`` callable: ``NpzResultSink`` rewrites the frame's npz, ``dict_sink`` keeps them
in memory; an h5 sink needs ``h5py`` (not installed here).
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

from .pipeline import HiMoPipeline, Sample


def history_of(dataset, i: int):
    """The history sweep of frame i: the previous frame of the same scene, else the frame itself."""
    f = dataset[i]
    if i > 0:
        p = dataset[i - 1]
        if p.get("scene_id") == f.get("scene_id"):
            return p
    return f


class NpzResultSink:
    def __init__(self, directory, res_name: str):
        self.directory, self.res_name = Path(directory), res_name

    def __call__(self, index: int, frame: dict, flow: np.ndarray):
        path = self.directory / frame["scene_id"] / f"{frame['timestamp']}.npz"
        with np.load(path) as z:
            arrays = {k: z[k] for k in z.files}
        arrays[self.res_name] = flow.astype(np.float32)
        tmp = path.with_name(path.stem + ".writing.npz")       # written aside, then renamed: the feeder thread may be reading
        np.savez(tmp, **arrays)                                 # this very file as a later frame's history sweep
        os.replace(tmp, path)


def frame_source(dataset, rank: int = 0, world: int = 1):
    """(index, history frame, frame, next frame | None) for every frame of this rank that has a next sweep to flow into."""
    for i in range(rank, len(dataset), world):
        f0 = dataset[i]
        if "pc1" not in f0:
            if i + 1 >= len(dataset) or dataset[i + 1].get("scene_id") != f0.get("scene_id"):
                continue                                       # last sweep of a scene: no pc1 to flow into
            f1 = dataset[i + 1]
        else:
            f1 = None
        yield i, history_of(dataset, i), f0, f1


def run(dataset, res_name: str = "seflowpp_best", params: dict | None = None, sink=None, pipeline: HiMoPipeline | None = None,
        batch_frames: int = 4) -> int:
    """Flow for every frame of ``dataset`` that has a ``pc1`` / next sweep.  Returns the frames this rank processed.
    Frames are read, staged in pinned memory and copied to the device by a background thread two batches ahead of the
    network (``feeder.SampleFeeder``); results leave through pinned buffers and a writer thread (``feeder.ResultDrain``),
    so neither the dataset reads nor the sink's file writes stall the launch thread."""
    import torch.distributed as dist
    from .feeder import ResultDrain, SampleFeeder
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    pipe = pipeline if pipeline is not None else HiMoPipeline(params=params, max_batch=max(1, batch_frames))
    results = {} if sink is None else None

    def deliver(key, flow):
        i, f0 = key
        if sink is None:
            results[i] = flow
        else:
            sink(i, f0, flow)

    drain = ResultDrain(deliver, device=pipe.device)
    done = 0
    try:
        for batch in SampleFeeder(frame_source(dataset, rank, world), device=pipe.device, batch=max(1, batch_frames)):
            for (i, f0, _), flow in zip(batch, pipe.flows([s for _, _, s in batch])):
                drain.put((i, f0), flow)
                done += 1
    finally:
        drain.close()
    return results if sink is None else done


def main(checkpoint: str = "", dataset_path: str = "", res_name: str = ""):
    from .dataset import HDF5Dataset, NpzDataset
    name = res_name or (Path(checkpoint).stem if checkpoint else "seflowpp_best")
    params = None
    if checkpoint:
        with np.load(checkpoint) as z:                        # an .npz of the arrays named in seflow/spec.py
            params = {k: z[k] for k in z.files}
    root = Path(dataset_path)
    if (root / "index_total.pkl").exists() and any(root.glob("*/*.npz")):
        ds = NpzDataset(root)
        return run(ds, name, params, sink=NpzResultSink(root, name))
    ds = HDF5Dataset(root, vis_name=name, eval=False)         # raises a clear ImportError without h5py
    raise NotImplementedError("writing results back into .h5 needs h5py, which is not installed in this image")


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--dataset_path", required=True)
    ap.add_argument("--res_name", default="")
    a = ap.parse_args()
    main(a.checkpoint, a.dataset_path, a.res_name)
