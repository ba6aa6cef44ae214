"""Counterpart of the reference's ``OpenSceneFlow/save.py`` call site (README.md:46-54:
``python save.py checkpoint=... dataset_path=...``): run the scene-flow network over a dataset and attach the
``(N,3) float32`` flow -- INCLUDING ego motion, row-aligned with ``pc0`` -- to every frame under ``<res_name>``,
which is exactly what ``save_zip.py:117`` / ``eval.py:302`` read back.

The reference's ``save.py`` itself is in the absent submodule, so only the data contract is reproduced: the key
name defaults to the checkpoint's stem (``seflowpp_best`` for ``seflowpp_best.ckpt``, README.md:50), frames are
walked in dataset order, and under ``torchrun`` frame i goes to rank i % world (h5 scene files: scene k goes to rank
k % world, so every file has exactly one writer).  Results are written through a ``sink(frame_index, frame, flow)``
callable: ``NpzResultSink`` rewrites the frame's npz, ``H5ResultSink`` creates / replaces the ``<res_name>`` dataset in
group ``<timestamp>`` of ``<scene_id>.h5`` (the dataset ``tools/test/repack_h5_scania.py:50`` deletes by name) through
``h5py`` or, without it, the HDF5 C library by ``ctypes`` (``h5c``); with neither it writes a result file beside the scene
file and says so (``dataset.result_file``; the loader reads it back), no sink keeps them in memory.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

from .pipeline import HiMoPipeline, Sample


class _FrameCache:
    """``dataset[i]`` with the last few frames kept: the walk below touches frame i as "next" of i-1, as itself, and as
    "history" of i+1 -- one file read instead of three (h5 / npz reads are the slow part of ``save``).  With ``workers`` > 0
    and a planned walk (``plan``) the frames of the next ``ahead`` steps are read by a small thread pool before they are asked
    for: page faults, decompression and array copies of several frames overlap (they release the GIL); the frames still come
    back in order."""

    def __init__(self, dataset, keep: int = 4, workers: int = 0, ahead: int = 8):
        self.dataset, self.keep, self._frames = dataset, keep, {}
        self._pool, self._order, self._pos, self._ahead = None, None, 0, ahead
        if workers > 0:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="himo-reader")

    def __len__(self):
        return len(self.dataset)

    def scene(self, j):
        """scene id of item j WITHOUT reading it (the dataset's ``index``: ``[scene_id, timestamp]`` pairs), or None when the
        dataset has no index to ask"""
        index = getattr(self.dataset, "index", None)
        return index[j][0] if index is not None and 0 <= j < len(index) else None

    def plan(self, order):
        """the ascending indices the walk will ask for (each also as history of the next and next of the previous)"""
        self._order, self._pos = list(order), 0

    def _prefetch(self, i):
        order, n = self._order, len(self.dataset)
        while self._pos < len(order) and order[self._pos] <= i + self._ahead:
            o = order[self._pos]
            self._pos += 1
            # the neighbours only where the walk will ask for them: the history sweep o - 1 and (for frames that do not carry their
            # successor) the next sweep o + 1, and only inside o's OWN scene -- with whole scenes per rank (h5 sinks) a neighbour
            # across the boundary belongs to another rank, which may be rewriting that very file
            here = self.scene(o)
            for j in ((o - 1, o) if getattr(self.dataset, "carries_next", False) else (o - 1, o, o + 1)):
                if 0 <= j < n and j >= i - 1 and j not in self._frames and (j == o or here is None or self.scene(j) == here):
                    self._frames[j] = self._pool.submit(self.dataset.__getitem__, j)

    def __getitem__(self, i):
        if self._pool is not None and self._order is not None:
            self._prefetch(i)
        f = self._frames.get(i)
        if f is None:
            f = self._frames[i] = self.dataset[i]
        elif hasattr(f, "result"):
            f = self._frames[i] = f.result()
        if self._pool is not None and self._order is not None:
            for j in [j for j in self._frames if j < i - 2]:      # the walk is ascending: what lies two behind is stale
                del self._frames[j]
        else:
            while len(self._frames) > self.keep:
                self._frames.pop(min(self._frames))               # the walk is ascending: the lowest index is the stale one
        return f

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=False, cancel_futures=True)
            self._pool = None


def _scene_of(dataset, j):
    """scene id of item j from the dataset's index (no read), else None"""
    scene = getattr(dataset, "scene", None)
    if scene is not None:
        return scene(j)
    index = getattr(dataset, "index", None)
    return index[j][0] if index is not None and 0 <= j < len(index) else None


def history_of(dataset, i: int):
    """The history sweep of frame i: the previous frame of the same scene, else the frame itself.  Where the dataset has an index
    the scene of i - 1 is looked up there: a frame of another scene is never read (it may be another rank's file)."""
    f = dataset[i]
    if i > 0:
        here, there = _scene_of(dataset, i), _scene_of(dataset, i - 1)
        if here is not None and there is not None:
            return dataset[i - 1] if there == here else f
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


def h5_writer():
    """(module, how) that can open a scene file for modification: ``h5py``, else libhdf5 through ``ctypes``, else None."""
    try:
        import h5py
        return h5py, "h5py"
    except ImportError:
        pass
    from . import h5c
    if h5c.available():
        return h5c, f"libhdf5 {'.'.join(map(str, h5c.load().version))} via ctypes"
    return None, "no HDF5 library"


class H5ResultSink:
    """``(N,3) float32`` under ``<scene_id>.h5 : <timestamp>/<res_name>`` (consumers: save_zip.py:117 via the
    loader's ``vis_name``, eval.py:302).  A scene's results are held back until the walk has moved on to the next scene
    (frames arrive in dataset order and the reader runs ahead of the results, never behind), so the file is never open for
    reading by the loader and for writing here at the same time; ``close()`` writes what is left.

    Modifying an existing HDF5 file needs an HDF5 library (``h5_writer``).  Where none can be loaded the results go to
    ``dataset.result_file(directory, res_name, scene_id)`` instead -- a new HDF5 file with the same ``<timestamp>/<res_name>``
    layout, written by ``h5lite`` -- with a warning naming the file; ``HDF5Dataset`` falls back to it when reading."""

    def __init__(self, directory, res_name: str, opener=None, before_write=None):
        """``before_write(scene_id)``: called before a scene file is opened for modification (``HDF5Dataset.forget``: a reader
        that keeps scene files open must let go of its handle and of what it parsed from the file)"""
        self.directory, self.res_name = Path(directory), res_name
        self._before_write = before_write
        self.how = "opener"
        if opener is None:
            mod, self.how = h5_writer()
            opener = (lambda path: mod.File(path, "a")) if mod is not None else None
        self._open = opener
        self._scene, self._pending = None, []
        self.side_files = []

    def __call__(self, index: int, frame: dict, flow: np.ndarray):
        if frame["scene_id"] != self._scene:
            self.flush()
            self._scene = frame["scene_id"]
        if flow.shape != (len(frame["pc0"]), 3):
            raise ValueError(f"flow {flow.shape} is not row-aligned with pc0 ({len(frame['pc0'])} points)")   # score.py:583
        self._pending.append((str(frame["timestamp"]), np.ascontiguousarray(flow, dtype=np.float32)))

    def flush(self):
        if not self._pending:
            return
        if self._before_write is not None:
            self._before_write(self._scene)
        if self._open is not None:
            with self._open(self.directory / f"{self._scene}.h5") as f:
                for ts, flow in self._pending:
                    g = f[ts]
                    if self.res_name in g:
                        del g[self.res_name]                       # re-running a checkpoint replaces its result
                    g.create_dataset(self.res_name, data=flow)
            self._supersede_beside()
        else:
            self._flush_beside()
        self._pending = []

    def _supersede_beside(self):
        """After an in-place write: drop the same sweeps from a result file an earlier library-less run left beside the scene
        (the loader prefers that file while it names a sweep)."""
        from . import h5lite
        from .dataset import result_file
        path = result_file(self.directory, self.res_name, self._scene)
        if not path.exists():
            return
        with h5lite.File(path) as old:
            tree = {ts: {k: old[ts][k][:] for k in old[ts].keys()} for ts in old.keys()}
        for ts, _ in self._pending:
            tree.pop(ts, None)
        if tree:
            h5lite.write_file(path, tree)
        else:
            path.unlink()

    def _flush_beside(self):
        import warnings
        from . import h5lite
        from .dataset import result_file
        path = result_file(self.directory, self.res_name, self._scene)
        tree = {}
        if path.exists():                                          # an earlier run / an earlier part of this scene: keep, then replace
            with h5lite.File(path) as old:
                tree = {ts: {k: old[ts][k][:] for k in old[ts].keys()} for ts in old.keys()}
        # provenance, per sweep: what the SCENE file held under <res_name> when this result was written (present?, CRC-32 of its bytes).
        # The loader uses it when the scene file is later found modified: an unchanged in-scene array is older than this result (the
        # scene was touched, copied, or given another <res_name>), a changed one was written by another tool afterwards and wins.
        from .dataset import scene_stamp
        scene_path = Path(self.directory) / f"{self._scene}.h5"
        held = {}
        if scene_path.exists():
            with h5lite.File(scene_path) as sc:
                for ts, _ in self._pending:
                    if ts in sc and self.res_name in sc[ts]:
                        held[ts] = sc[ts][self.res_name][:]
        for ts, flow in self._pending:
            entry = tree.setdefault(ts, {})
            entry[self.res_name] = flow
            entry[self.res_name + "@scene"] = scene_stamp(held.get(ts))
        path.parent.mkdir(parents=True, exist_ok=True)
        h5lite.write_file(path, tree)
        if path not in self.side_files:
            self.side_files.append(path)
            warnings.warn(f"no HDF5 library (h5py / libhdf5) to modify {self._scene}.h5 with: '{self.res_name}' for "
                          f"{len(tree)} sweep(s) written to {path} instead (himo_amd's loader reads it from there; "
                          f"merge with h5copy or h5py to hand the scene file to other tools)", stacklevel=3)

    def close(self):
        self.flush()


def frame_source(dataset, rank: int = 0, world: int = 1, by_scene: bool = False, readers: int = 2):
    """(index, history frame, frame, next frame | None) for every frame of this rank that has a next sweep to flow into.
    ``by_scene``: shard whole scenes (scene k of the walk -> rank k % world) instead of frames.  ``readers``: threads that read
    the coming frames ahead of the walk (0 = read on the calling thread)."""
    index = getattr(dataset, "index", None)
    dataset = _FrameCache(dataset, workers=readers)
    if by_scene and index is not None:
        scenes = {}
        for s, _ in index:
            scenes.setdefault(s, len(scenes))
        mine = [i for i, (s, _) in enumerate(index) if scenes[s] % world == rank]
    else:
        mine = range(rank, len(dataset), world)
    dataset.plan(mine)
    try:
        for i in mine:
            f0 = dataset[i]
            if "pc1" not in f0:
                if i + 1 >= len(dataset):
                    continue
                here, there = _scene_of(dataset, i), _scene_of(dataset, i + 1)
                if (there != here) if (here is not None and there is not None) else (dataset[i + 1].get("scene_id") != f0.get("scene_id")):
                    continue                                       # last sweep of a scene: no pc1 to flow into
                f1 = dataset[i + 1]
            else:
                f1 = None
            yield i, history_of(dataset, i), f0, f1
    finally:
        dataset.close()


def run(dataset, res_name: str = "seflowpp_best", params: dict | None = None, sink=None, pipeline: HiMoPipeline | None = None,
        batch_frames: int = 4, by_scene: bool = False) -> int:
    """Flow for every frame of ``dataset`` that has a ``pc1`` / next sweep.  Returns the frames this rank processed.
    Frames are read, staged in pinned memory and copied to the device by a background thread two batches ahead of the
    network (``feeder.SampleFeeder``); results leave through pinned buffers and a writer thread (``feeder.ResultDrain``),
    so neither the dataset reads nor the sink's file writes stall the launch thread.  The network runs several batches in flight
    (``pipeline.OverlappedPipeline``; pass a ``HiMoPipeline`` as ``pipeline`` for the single-stream path); every batch is
    checked for fp16-range overflow before it is handed to the sink (auto: redone in the bf16 split)."""
    import torch.distributed as dist
    from .feeder import ResultDrain, SampleFeeder
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    from .pipeline import OverlappedPipeline
    pipe = pipeline if pipeline is not None else OverlappedPipeline(params=params, max_batch=max(1, batch_frames))
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
        feeder = SampleFeeder(frame_source(dataset, rank, world, by_scene=by_scene), device=pipe.device, batch=max(1, batch_frames),
                              depth=max(2, len(getattr(pipe, "pipes", [0, 0]))))       # as many batches ahead as the pipeline keeps in flight
        if isinstance(pipe, OverlappedPipeline):               # several batches in flight: batch k's finite-flow check under the following batches
            queued = []

            def sample_lists():
                for batch in feeder:
                    queued.append(batch)
                    yield [s for _, _, s in batch]
            batches = ((queued.pop(0), flows) for _, flows in pipe.flows_stream(sample_lists()))
        else:
            batches = ((batch, pipe.flows([s for _, _, s in batch])) for batch in feeder)
        for batch, flows in batches:
            for (i, f0, _), flow in zip(batch, flows):
                drain.put((i, f0), flow)
                done += 1
    finally:
        drain.close()
        if hasattr(sink, "close"):
            sink.close()
    return results if sink is None else done


def run_fastnsf(dataset, res_name: str = "fastnsf", sink=None, by_scene: bool = False, iters: int = 100, **fit_options):
    """``python save.py model=fastnsf dataset_path=...`` (README.md:53): the optimisation-based baseline instead of the network --
    one coordinate MLP fitted per sweep pair (``himo_amd/fastnsf.py``; PARITY UNPINNED like the network), the flow of every pc0 row
    including ego motion stored under ``res_name`` exactly like the network's.  Two fits in flight on two HIP streams
    (``fastnsf.OverlappedFastNSF``); results leave through the same pinned-buffer writer thread as ``run``'s."""
    import torch.distributed as dist
    from .fastnsf import OverlappedFastNSF
    from .feeder import ResultDrain
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    nsf = OverlappedFastNSF(iters=iters, **fit_options)
    results = {} if sink is None else None

    def deliver(key, flow):
        i, f0 = key
        if sink is None:
            results[i] = flow
        else:
            sink(i, f0, flow)

    drain = ResultDrain(deliver, device=nsf.device)
    keys, done = [], 0

    def pairs():
        for i, _, f0, f1 in frame_source(dataset, rank, world, by_scene=by_scene):
            pc1 = f0["pc1"] if "pc1" in f0 else f1["pc0"]
            keys.append((i, f0))
            yield np.asarray(f0["pc0"])[:, :3], np.asarray(pc1)[:, :3], f0["pose0"], f0["pose1"]
    try:
        for flow in nsf.fits(pairs()):
            drain.put(keys.pop(0), flow)
            done += 1
    finally:
        drain.close()
        if hasattr(sink, "close"):
            sink.close()
    return results if sink is None else done


def main(checkpoint: str = "", dataset_path: str = "", res_name: str = "", model: str = "", iters: int = 100):
    """``python -m himo_amd.save --checkpoint <weights.npz> --dataset_path <dir>`` (the feed-forward network), or
    ``--model fastnsf --dataset_path <dir>`` (the optimisation-based baseline, README.md:50-53); under ``torchrun`` one rank per GPU."""
    from . import distenv
    from .dataset import SAVE_FIELDS, NpzDataset, open_dataset
    if model not in ("", "seflowpp", "deflowpp", "fastnsf"):
        raise ValueError(f"model={model!r}: this build runs the SeFlow++-style network (default) and 'fastnsf'")
    fastnsf = model == "fastnsf"
    name = res_name or ("fastnsf" if fastnsf else (Path(checkpoint).stem if checkpoint else "seflowpp_best"))
    params = None
    if checkpoint and not fastnsf:
        from .seflow.checkpoint import load_params
        params = load_params(checkpoint)
    root = Path(dataset_path)
    with distenv.process_group():
        # inference reads the sweeps, poses and time stamps only (no labels, masks, ground-truth flow, earlier results), as views
        # of the file mappings where the files allow: the feeder stages them straight into its pinned buffers
        ds = open_dataset(root, vis_name=name, eval=False, fields=SAVE_FIELDS, zero_copy=True)
        npz = isinstance(ds, NpzDataset)
        sink = NpzResultSink(root, name) if npz else H5ResultSink(root, name, before_write=ds.forget)
        done, err = 0, None
        try:
            done = (run_fastnsf(ds, name, sink=sink, by_scene=not npz, iters=iters) if fastnsf else
                    run(ds, name, params, sink=sink, by_scene=not npz))
        except Exception as e:                                  # arrive at the rendezvous anyway, then re-raise
            err = e
        distenv.rendezvous(err, "writing its share of the flow results")
        return done


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--dataset_path", required=True)
    ap.add_argument("--res_name", default="")
    ap.add_argument("--model", default="", help="'fastnsf': fit the optimisation-based baseline per sweep pair instead of running the network")
    ap.add_argument("--iters", type=int, default=100, help="optimiser iterations per sweep pair (--model fastnsf)")
    import sys
    # the reference's program takes hydra-style overrides (`save.py checkpoint=... dataset_path=...`, `model=fastnsf`: README.md:46-53)
    argv = [("--" + x) if (not x.startswith("-") and "=" in x and x.split("=", 1)[0] in ("checkpoint", "dataset_path", "res_name", "model", "iters"))
            else x for x in sys.argv[1:]]
    a = ap.parse_args(argv)
    main(a.checkpoint, a.dataset_path, a.res_name, a.model, a.iters)
