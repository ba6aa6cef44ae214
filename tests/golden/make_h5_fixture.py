"""Writes ``tests/golden/h5/`` -- REAL HDF5 scene files for the h5 boundary -- through the
HDF5 C library itself (``ctypes`` on libhdf5 1.10.6, the library present in the build container; ``h5py`` is not installed
anywhere).  Run in the build container:  ``python tests/golden/make_h5_fixture.py``

What is written follows the reference's h5py writers call for call:
  * one file per scene ``<scene_id>.h5``, one group per timestamp, ``group.create_dataset(name, data=array)`` with default
    properties (contiguous, no filter), names and dtypes of dataprocess/extract_sca.py:76-93: ``lidar`` f32 (N,4),
    ``lidar_id`` u8, ``lidar_dt`` f32, ``SensorsCenter`` f32 (6,3), ``pose`` f64 (4,4), ``timestamp`` (an int scalar),
    ``flow`` f32 (N,3), ``flow_is_valid`` bool, ``flow_category_indices`` u8, ``flow_instance_id`` u32, ``ego_motion``;
  * ``ground_mask`` bool (tools/test/repack_h5_scania.py:29; the loader renames it ``gm0``, eval.py:293);
  * an existing result dataset ``seflowpp_best`` f32 (N,3) (the one repack_h5_scania.py:50 skips by name);
  * numpy ``bool`` goes to disk the way h5py maps it: an 8-bit enum {FALSE=0, TRUE=1};
  * ``index_total.pkl`` / ``index_eval.pkl``: lists of ``[scene_id, timestamp]`` (tools/pkl_extract.py:5-19).
The arrays come from ``himo_amd.synthetic.make_frame`` seeds, so the tests re-make them and compare bit for bit; poses are
chained so that ``inv(pose[t+1]) @ pose[t]`` is the ego motion each frame's ``flow`` was built with.

Also written: ``h5dump_H.txt`` (``h5dump -H`` of both files: the library's own listing of what is inside).
The directory NAME is what selects a dataset's evaluation rules (utils/__init__.py:9-24 ``check_valid`` looks for "av2" /
"scania" in the path), so the tests copy this one fixture under ``.../av2/...`` or ``.../scania/...`` as they need.
"""
from __future__ import annotations

import pickle
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))

from himo_amd.synthetic import make_frame  # noqa: E402

OUT = HERE / "h5"
N_SCENES, N_SWEEPS, N_POINTS = 2, 4, 640


def frames():
    """2 scenes x 4 sweeps; ``pose0`` chained through the scene, ``pose1`` = the next sweep's pose (last sweep: its own
    make_frame pose1 applied once more, unused by the loader)."""
    out = []
    for s in range(N_SCENES):
        pose = np.eye(4)
        for k in range(N_SWEEPS):
            i = s * N_SWEEPS + k
            f = make_frame(500 + i, n_points=N_POINTS + 17 * i, scene_id=f"scene-{s:02d}")
            step = f["pose1"]
            f["pose0"] = pose.copy()
            pose = pose @ step
            f["pose1"] = pose.copy()
            out.append(f)
    return out


def group_arrays(f: dict) -> dict:
    ego = np.linalg.inv(f["pose1"]) @ f["pose0"]
    rng = np.random.default_rng(f["timestamp"] % 1000)
    return {"lidar": f["pc0"].astype(np.float32), "lidar_id": f["lidar_id"].astype(np.uint8),
            "lidar_dt": f["lidar_dt"].astype(np.float32), "SensorsCenter": rng.normal(size=(6, 3)).astype(np.float32),
            "pose": f["pose0"].astype(np.float64), "timestamp": np.int64(f["timestamp"]),
            "flow": f["flow"].astype(np.float32), "flow_is_valid": f["flow_is_valid"].astype(bool),
            "flow_category_indices": f["flow_category_indices"].astype(np.uint8),
            "flow_instance_id": f["flow_instance_id"].astype(np.uint32), "ego_motion": ego.astype(np.float32),
            "ground_mask": f["gm0"].astype(bool), "seflowpp_best": f["seflowpp_best"].astype(np.float32)}


def main():
    from himo_amd import h5c
    lib = h5c.load()
    OUT.mkdir(exist_ok=True)
    fr = frames()
    scenes = {}
    for f in fr:
        scenes.setdefault(f["scene_id"], []).append(f)
    for scene, fs in scenes.items():
        with h5c.File(OUT / f"{scene}.h5", "w") as h:
            for f in fs:
                g = h.create_group(str(f["timestamp"]))
                for name, a in group_arrays(f).items():
                    g.create_dataset(name, data=a)
    index = [[f["scene_id"], str(f["timestamp"])] for f in fr]
    with open(OUT / "index_total.pkl", "wb") as fh:
        pickle.dump(index, fh)
    with open(OUT / "index_eval.pkl", "wb") as fh:                    # every sweep that has a successor except one per scene
        pickle.dump([index[i] for i in (0, 2, 4, 5)], fh)
    listing = [f"# libhdf5 {'.'.join(map(str, lib.version))} ({lib.path}); h5dump -H of each file\n"]
    for scene in scenes:
        r = subprocess.run(["/opt/conda/bin/h5dump", "-H", f"{scene}.h5"], capture_output=True, text=True, check=True, cwd=OUT)
        listing.append(r.stdout)
    (OUT / "h5dump_H.txt").write_text("".join(listing))
    print("wrote", sorted(p.name for p in OUT.iterdir()), sum(p.stat().st_size for p in OUT.iterdir()), "bytes")


if __name__ == "__main__":
    main()
