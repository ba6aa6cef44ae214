"""Frame sources with the ``len`` / ``[i]`` protocol of the reference's ``HDF5Dataset``
(constructed at save_zip.py:111 and eval.py:279 as ``HDF5Dataset(dir, vis_name=<res>, eval=True)``).

The reference's loader lives in the EMPTY ``OpenSceneFlow`` submodule, so the h5 layout below is
taken from the in-tree writers and consumers only:
    per-scene file ``<scene_id>.h5`` with one group per timestamp holding ``lidar`` (N,4) f32,
    ``lidar_dt`` (N,) f32, ``lidar_id`` u8, ``pose`` (4,4) f64, ``ground_mask`` bool, ``flow`` (N,3) f32,
    ``flow_is_valid``, ``flow_category_indices`` u8, ``flow_instance_id`` u32 and result datasets
    named after the checkpoint (dataprocess/extract_sca.py:76-93, tools/test/repack_h5_scania.py:23-36,
    downstream/eval_seg.py:205-225); frame order from ``index_total.pkl`` / ``index_eval.pkl``
    (lists of ``[scene_id, timestamp]``, tools/pkl_extract.py:5-19).
``pose1`` is the pose of the next timestamp in the same scene (the flow is pc0 -> pc1).

``h5py`` is not installed in the build image; ``HDF5Dataset`` raises ImportError with that message
rather than guessing.  ``NpzDataset`` is this package's own container for the same dicts.
"""
from __future__ import annotations

import pickle
from pathlib import Path

import numpy as np

FRAME_KEYS = ("pc0", "pose0", "pose1", "lidar_dt", "gm0", "flow", "flow_is_valid",
              "flow_category_indices", "flow_instance_id", "lidar_id")


class ListDataset:
    def __init__(self, frames):
        self.frames = list(frames)

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i]


class NpzDataset:
    """``<dir>/<scene_id>/<timestamp>.npz`` + ``index_total.pkl`` (same index format as the reference)."""

    def __init__(self, directory, vis_name: str = "", eval: bool = False):  # noqa: A002
        self.directory = Path(directory)
        idx = self.directory / ("index_eval.pkl" if eval and (self.directory / "index_eval.pkl").exists() else "index_total.pkl")
        with open(idx, "rb") as f:
            self.index = pickle.load(f)
        self.vis_name = vis_name

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        scene_id, ts = self.index[i]
        with np.load(self.directory / scene_id / f"{ts}.npz") as z:
            d = {k: z[k] for k in z.files}
        d["scene_id"], d["timestamp"] = scene_id, int(ts)
        return d

    @staticmethod
    def write(directory, frames, eval_subset=None):
        directory = Path(directory)
        index = []
        for f in frames:
            (directory / f["scene_id"]).mkdir(parents=True, exist_ok=True)
            arrays = {k: np.asarray(v) for k, v in f.items() if k not in ("scene_id", "timestamp")}
            np.savez(directory / f["scene_id"] / f"{f['timestamp']}.npz", **arrays)
            index.append([f["scene_id"], str(f["timestamp"])])
        with open(directory / "index_total.pkl", "wb") as fh:
            pickle.dump(index, fh)
        if eval_subset is not None:
            with open(directory / "index_eval.pkl", "wb") as fh:
                pickle.dump([index[i] for i in eval_subset], fh)


class HDF5Dataset:
    """h5 scene files -> frame dicts (see module docstring for the provenance of the layout)."""

    def __init__(self, directory, vis_name: str = "", eval: bool = False, n_frames: int = 2):  # noqa: A002
        try:
            import h5py  # noqa: F401
        except ImportError as e:
            raise ImportError("HDF5Dataset needs h5py, which is not installed in this image; "
                              "use NpzDataset / SyntheticDataset, or install h5py on the target box") from e
        self._h5py = h5py
        self.directory = Path(directory)
        self.vis_name = vis_name if isinstance(vis_name, (list, tuple)) else [vis_name]
        name = "index_eval.pkl" if eval and (self.directory / "index_eval.pkl").exists() else "index_total.pkl"
        with open(self.directory / name, "rb") as f:
            self.index = pickle.load(f)
        with open(self.directory / "index_total.pkl", "rb") as f:
            total = pickle.load(f)
        self._next = {}
        for (s0, t0), (s1, t1) in zip(total[:-1], total[1:]):
            if s0 == s1:
                self._next[(s0, str(t0))] = str(t1)

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        scene_id, ts = self.index[i]
        ts = str(ts)
        with self._h5py.File(self.directory / f"{scene_id}.h5", "r") as f:
            g = f[ts]
            d = {"scene_id": scene_id, "timestamp": int(ts), "pc0": g["lidar"][:], "pose0": g["pose"][:],
                 "lidar_dt": g["lidar_dt"][:] if "lidar_dt" in g else np.zeros(g["lidar"].shape[0], np.float32)}
            if "ground_mask" in g:
                d["gm0"] = g["ground_mask"][:].astype(bool)
            for k in ("flow", "flow_is_valid", "flow_category_indices", "flow_instance_id", "lidar_id", "ego_motion"):
                if k in g:
                    d[k] = g[k][:]
            for name in self.vis_name:
                if name and name not in ("raw",) and name in g:
                    d[name] = g[name][:]
            nxt = self._next.get((scene_id, ts))
            if nxt is not None and nxt in f:
                d["pose1"] = f[nxt]["pose"][:]
                d["pc1"] = f[nxt]["lidar"][:]
        return d
