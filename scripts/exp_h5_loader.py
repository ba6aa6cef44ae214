"""The h5 frame loader alone (no GPU): items per second of ``HDF5Dataset[i]`` over 120k-point scenes written by ``h5lite.write_file``
(page cache warm), all datasets vs the fields ``save`` needs, one reader vs the prefetching reader pool.
usage: python scripts/exp_h5_loader.py [scenes] [sweeps per scene]"""
import pickle, sys, tempfile, time
from pathlib import Path

import numpy as np

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
from himo_amd import h5lite
from himo_amd.dataset import HDF5Dataset
from himo_amd.synthetic import make_frame


def write_scenes(root: Path, n_scenes: int, n_sweeps: int, n_points: int = 120_000):
    index = []
    for s in range(n_scenes):
        scene = f"scene{s:03d}"
        tree = {}
        for k in range(n_sweeps):
            f = make_frame(1000 * s + k, n_points=n_points, scene_id=scene)
            ts = str(f["timestamp"])
            tree[ts] = {"lidar": f["pc0"], "lidar_dt": f["lidar_dt"], "lidar_id": f["lidar_id"], "pose": f["pose0"],
                        "ground_mask": f["gm0"], "flow": f["flow"], "flow_is_valid": f["flow_is_valid"],
                        "flow_category_indices": f["flow_category_indices"], "flow_instance_id": f["flow_instance_id"]}
            index.append([scene, ts])
        h5lite.write_file(root / f"{scene}.h5", tree)
    with open(root / "index_total.pkl", "wb") as fh:
        pickle.dump(index, fh)


def rate(fn, n):
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    return n / (time.perf_counter() - t0)


if __name__ == "__main__":
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp)
        write_scenes(root, n_scenes, n_sweeps)
        ds = HDF5Dataset(root)
        n = len(ds)
        for _ in range(2):
            for i in range(n):
                ds[i]                                  # warm the page cache
        print(f"{n} items of {n_scenes} scenes x {n_sweeps} sweeps x 120000 points")
        print(f"  every dataset of the group, one thread:          {rate(lambda i: ds[i], n):8.0f} items/s")
        if "fields" in HDF5Dataset.__init__.__code__.co_varnames:
            from himo_amd.dataset import SAVE_FIELDS
            dsf = HDF5Dataset(root, fields=SAVE_FIELDS, zero_copy=True)
            for rep in range(3):
                print(f"  fields={SAVE_FIELDS}, views into the mapping: {rate(lambda i: dsf[i], n):8.0f} items/s")
            touch = lambda i: sum(float(np.asarray(v).ravel()[0]) for v in dsf[i].values() if isinstance(v, np.ndarray))
            print(f"  the same, first element of every array touched:  {rate(touch, n):8.0f} items/s")
            stage = np.empty(4 * 120_000 * 4 + 64, np.float32)

            def copy_out(i):
                d = dsf[i]
                for k in ("pc0", "pc1", "lidar_dt"):
                    a = np.asarray(d[k], np.float32).ravel()
                    np.copyto(stage[:a.size], a)
            print(f"  the same + pc0 / pc1 / lidar_dt copied out:      {rate(copy_out, n):8.0f} items/s")
