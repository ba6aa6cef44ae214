"""Seeded synthetic LiDAR sweeps shaped like the reference's per-frame dict.

There is no dataset (and no network) in this environment, so every test and
the benchmark run on frames generated here.  The layout follows the dict that
the reference reads from ``HDF5Dataset(dir, vis_name=<res>, eval=True)[i]``
(keys consumed at /root/reference/save_zip.py:113-123 and eval.py:282-310; on-disk
dtypes at dataprocess/extract_sca.py:78-93 and tools/test/repack_h5_scania.py:23-36)
and the distributions of SURVEY.md section 8(d).

Pure numpy; no GPU work happens in this module.
"""
from __future__ import annotations

import numpy as np

# Category indices: "NONE" = 0, then 1-based position in the Argoverse-2 annotation
# list (reference copy at tools/test/score.py:29-65).
REGULAR_VEHICLE = 19
BOX_TRUCK = 6
BUS = 7
TRUCK = 25
PEDESTRIAN = 17
_VEHICLE_CLASSES = (REGULAR_VEHICLE, REGULAR_VEHICLE, REGULAR_VEHICLE, TRUCK, BUS, BOX_TRUCK)

# network range used by the SeFlow++ launchers (assets/slurm/ssl-train-av2.sh:32)
POINT_CLOUD_RANGE = (-51.2, -51.2, -3.0, 51.2, 51.2, 3.0)


def _yaw_pose(yaw: float, tx: float, ty: float) -> np.ndarray:
    pose = np.eye(4, dtype=np.float64)
    c, s = np.cos(yaw), np.sin(yaw)
    pose[:3, :3] = [[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]
    pose[0, 3], pose[1, 3] = tx, ty
    return pose


GROUND_Z = -1.8          # ground plane of the "rings" cloud (sensor at the origin, 1.8 m above it)


def lidar_rings(rng, n: int, n_beams: int = 64) -> np.ndarray:
    """(n,3) float64 returns of a spinning multi-beam LiDAR at the origin: ``n_beams`` elevation angles between -25 and
    +15 degrees, uniform azimuth; a downward ray ends on the ground plane or on the first vertical surface of its azimuth
    sector (walls 12-70 m out), an upward ray on that surface.  The result has what real sweeps have and a uniform cloud
    lacks: concentric ground rings whose spacing grows with range, hundreds of points per 0.2 m cell near the sensor,
    and most of the 512 x 512 grid empty; returns beyond +-51.2 m / above 3 m exist and fall outside the network range."""
    elev = np.deg2rad(np.linspace(-25.0, 15.0, n_beams))[rng.integers(0, n_beams, n)] + rng.normal(0.0, 2e-4, n)
    az = rng.uniform(-np.pi, np.pi, n)
    n_sectors = 72
    wall = rng.uniform(12.0, 70.0, n_sectors)[((az + np.pi) / (2 * np.pi) * n_sectors).astype(np.int64) % n_sectors]
    with np.errstate(divide="ignore"):
        ground = np.where(elev < 0, -GROUND_Z / np.tan(-elev), np.inf)
    r = np.minimum(ground, wall) + rng.normal(0.0, 0.02, n)
    return np.stack([r * np.cos(az), r * np.sin(az), r * np.tan(elev)], axis=1)


def make_frame(
    frame_idx: int,
    n_points: int = 120_000,
    n_instances: int = 30,
    res_name: str = "seflowpp_best",
    scene_id: str | None = None,
    est_noise: float = 0.05,
    data_name: str = "av2",
    cloud: str = "uniform",
) -> dict:
    """One synthetic sweep as a reference-style frame dict.

    ``flow`` includes ego-motion (the reference subtracts ``pose_flow`` from it,
    save_zip.py:117); ``<res_name>`` is an "estimated" flow = GT + N(0, est_noise).
    ``cloud``: "uniform" = SURVEY.md 8(d) (xyz uniform in the network range); "rings" = a spinning-LiDAR-like
    background (``lidar_rings``): range-dependent density, crowded cells near the sensor, empty space far out.
    """
    rng = np.random.default_rng(frame_idx)
    n = int(n_points)
    lo = np.array(POINT_CLOUD_RANGE[:3], dtype=np.float64)
    hi = np.array(POINT_CLOUD_RANGE[3:], dtype=np.float64)

    if cloud == "uniform":
        xyz = rng.uniform(lo, hi, size=(n, 3))
    elif cloud == "rings":
        xyz = lidar_rings(rng, n)
    else:
        raise ValueError(f"cloud={cloud!r}")
    category = np.zeros(n, dtype=np.uint8)
    instance = np.zeros(n, dtype=np.uint32)
    obj_flow = np.zeros((n, 3), dtype=np.float64)

    # carve a share of the points into box-shaped moving instances
    n_instances = int(min(n_instances, max(n // 40, 0)))
    if n_instances > 0:
        per_inst = rng.integers(12, max(13, min(5000, n // (2 * n_instances))), size=n_instances)
        start = 0
        for k in range(n_instances):
            cnt = int(per_inst[k])
            if start + cnt > n:
                break
            cls = _VEHICLE_CLASSES[k % len(_VEHICLE_CLASSES)]
            dims = np.array([4.5, 1.9, 1.6]) if cls == REGULAR_VEHICLE else np.array([10.0, 2.5, 3.2])
            rad = rng.uniform(4.0, 48.0)
            ang = rng.uniform(-np.pi, np.pi)
            centre = np.array([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(-1.6, -0.4)])
            heading = rng.uniform(-np.pi, np.pi)
            c, s = np.cos(heading), np.sin(heading)
            rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
            local = rng.uniform(-0.5, 0.5, size=(cnt, 3)) * dims
            xyz[start:start + cnt] = local @ rot.T + centre
            speed = rng.uniform(0.0, 35.0)  # m/s, spans the buckets of eval.py:101-108
            obj_flow[start:start + cnt] = speed * 0.1 * np.array([c, s, 0.0])
            category[start:start + cnt] = cls
            instance[start:start + cnt] = k + 1
            start += cnt

    pc0 = np.empty((n, 4), dtype=np.float32)
    pc0[:, :3] = xyz.astype(np.float32)
    pc0[:, 3] = rng.uniform(0.0, 1.0, size=n).astype(np.float32)

    lidar_dt = rng.uniform(0.0, 0.1, size=n).astype(np.float32)
    lidar_id = rng.integers(1, 7, size=n).astype(np.uint8)

    pose0 = np.eye(4, dtype=np.float64)
    pose1 = _yaw_pose(np.deg2rad(rng.uniform(-2.0, 2.0)), rng.uniform(-3.0, 3.0), rng.uniform(-0.5, 0.5))

    ego = np.linalg.inv(pose1) @ pose0
    pose_flow = pc0[:, :3].astype(np.float64) @ ego[:3, :3].T + ego[:3, 3] - pc0[:, :3]
    flow = (pose_flow + obj_flow + rng.normal(0.0, 0.02, size=(n, 3)) * (instance[:, None] > 0)).astype(np.float32)
    est = (flow + rng.normal(0.0, est_noise, size=(n, 3)) * (instance[:, None] > 0)).astype(np.float32)

    gm0 = (pc0[:, 2] < (-2.6 if cloud == "uniform" else GROUND_Z + 0.12)) & (instance == 0)
    flow_is_valid = rng.uniform(size=n) > 0.01

    return {
        "scene_id": scene_id if scene_id is not None else f"synthetic-{data_name}-{frame_idx // 8:04d}",
        "timestamp": 315_965_785_000_000_000 + int(frame_idx) * 100_000_000,
        "pc0": pc0,
        "pose0": pose0,
        "pose1": pose1,
        "lidar_dt": lidar_dt,
        "lidar_id": lidar_id,
        "gm0": gm0,
        "flow": flow,
        "flow_is_valid": flow_is_valid,
        "flow_category_indices": category,
        "flow_instance_id": instance,
        res_name: est,
    }


def make_frames(n_frames: int, n_points: int = 120_000, first: int = 0, **kw) -> list[dict]:
    return [make_frame(first + i, n_points=n_points, **kw) for i in range(n_frames)]


class SyntheticDataset:
    """Sequence of frame dicts with the ``len`` / ``[i]`` protocol of the reference's
    ``HDF5Dataset`` (constructed at save_zip.py:111, eval.py:279)."""

    def __init__(self, n_frames: int, n_points: int = 120_000, ragged: bool = False, **kw):
        self.n_frames = int(n_frames)
        self.n_points = int(n_points)
        self.ragged = ragged
        self.kw = kw

    def __len__(self) -> int:
        return self.n_frames

    def __getitem__(self, i: int) -> dict:
        if not 0 <= i < self.n_frames:
            raise IndexError(i)
        n = self.n_points
        if self.ragged:  # sweeps differ in point count, as real sweeps do
            n = max(1, int(n * (0.6 + 0.4 * ((i * 2654435761) % 1000) / 999.0)))
        return make_frame(i, n_points=n, **self.kw)


def make_scene(seed: int, n_sweeps: int, n_points: int = 120_000, n_instances: int = 30, scene_id: str | None = None,
               noise: float = 0.02, cloud: str = "uniform") -> list[dict]:
    """``n_sweeps`` CONSECUTIVE sweeps of one drive as reference-style frame dicts: a static world fixed in world coordinates
    (re-sampled with ``noise`` m of range noise per sweep), ``n_instances`` box-shaped objects at constant world velocities
    (0-35 m/s, the speed buckets of eval.py:101-108) and an ego vehicle driving ~10 m/s with a slow yaw -- so that the sweep
    AFTER a sweep is the same world 0.1 s later, which is what the training loop's label generator and self-supervised loss
    assume (``make_frame``'s sweeps are unrelated draws: fine for inference timing, meaningless as a (pc0, pc1) pair).
    ``pose1`` of sweep k is ``pose0`` of sweep k + 1; ``flow`` is the ground-truth motion of every pc0 point into sweep k + 1's
    sensor frame (ego motion included, as save_zip.py:117 expects)."""
    rng = np.random.default_rng(1_000_003 * seed + 17)
    n, dt = int(n_points), 0.1
    lo, hi = np.array(POINT_CLOUD_RANGE[:3]), np.array(POINT_CLOUD_RANGE[3:])
    speed_ego, yaw_rate = rng.uniform(6.0, 12.0), np.deg2rad(rng.uniform(-4.0, 4.0))
    poses, x, y, yaw = [], 0.0, 0.0, 0.0
    for _ in range(n_sweeps + 1):
        poses.append(_yaw_pose(yaw, x, y))
        x, y, yaw = x + speed_ego * dt * np.cos(yaw), y + speed_ego * dt * np.sin(yaw), yaw + yaw_rate * dt
    # the world: static points around the whole path (so that every sweep's range is populated), objects on top
    travel = speed_ego * dt * n_sweeps
    if cloud == "rings":
        world = lidar_rings(rng, n)
    else:
        world = rng.uniform(lo, hi, size=(n, 3))
    world[:, 0] += rng.uniform(0.0, 1.0, n) * travel * (cloud == "uniform")
    category, instance = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
    vel = np.zeros((n, 3))
    n_instances = int(min(n_instances, max(n // 40, 0)))
    start = 0
    if n_instances > 0:
        per_inst = rng.integers(12, max(13, min(5000, n // (2 * n_instances))), size=n_instances)
        for k in range(n_instances):
            cnt = int(per_inst[k])
            if start + cnt > n:
                break
            cls = _VEHICLE_CLASSES[k % len(_VEHICLE_CLASSES)]
            dims = np.array([4.5, 1.9, 1.6]) if cls == REGULAR_VEHICLE else np.array([10.0, 2.5, 3.2])
            rad, ang, heading = rng.uniform(4.0, 48.0), rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi)
            c, s = np.cos(heading), np.sin(heading)
            rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
            centre = np.array([rad * np.cos(ang) + travel / 2, rad * np.sin(ang), rng.uniform(-1.6, -0.4)])
            world[start:start + cnt] = (rng.uniform(-0.5, 0.5, size=(cnt, 3)) * dims) @ rot.T + centre
            vel[start:start + cnt] = rng.uniform(0.0, 35.0) * np.array([c, s, 0.0])
            category[start:start + cnt], instance[start:start + cnt] = cls, k + 1
            start += cnt
    ground = (world[:, 2] < (-2.6 if cloud == "uniform" else GROUND_Z + 0.12)) & (instance == 0)
    scene = scene_id if scene_id is not None else f"synthetic-drive-{seed:04d}"
    frames = []
    for k in range(n_sweeps):
        at0, at1 = world + vel * (dt * k), world + vel * (dt * (k + 1))
        inv0, inv1 = np.linalg.inv(poses[k]), np.linalg.inv(poses[k + 1])
        p0 = at0 @ inv0[:3, :3].T + inv0[:3, 3] + rng.normal(0.0, noise, (n, 3))
        p1 = at1 @ inv1[:3, :3].T + inv1[:3, 3]
        pc0 = np.empty((n, 4), np.float32)
        pc0[:, :3], pc0[:, 3] = p0, rng.uniform(0.0, 1.0, n)
        order = rng.permutation(n)                              # rows of successive sweeps do not correspond
        frames.append({"scene_id": scene, "timestamp": 315_965_785_000_000_000 + (1000 * seed + k) * 100_000_000,
                       "pc0": pc0[order], "pose0": poses[k], "pose1": poses[k + 1],
                       "lidar_dt": rng.uniform(0.0, 0.1, n).astype(np.float32), "lidar_id": rng.integers(1, 7, n).astype(np.uint8),
                       "gm0": ground[order], "flow": (p1 - p0).astype(np.float32)[order], "flow_is_valid": rng.uniform(size=n) > 0.01,
                       "flow_category_indices": category[order], "flow_instance_id": instance[order]})
    return frames


def write_h5_scenes(root, scenes: list[list[dict]]) -> list:
    """``<root>/<scene_id>.h5`` + ``index_total.pkl`` for lists of frame dicts, with every dataset the reference's extractors
    write (dataprocess/extract_sca.py:76-93; ``h5lite.write_file``).  Returns the index."""
    import pickle
    from pathlib import Path
    from . import h5lite
    root, index = Path(root), []
    for frames in scenes:
        tree = {}
        for f in frames:
            tree[str(f["timestamp"])] = {"lidar": f["pc0"], "lidar_dt": f["lidar_dt"], "lidar_id": f["lidar_id"], "pose": f["pose0"],
                                         "ground_mask": f["gm0"], "flow": f["flow"], "flow_is_valid": f["flow_is_valid"],
                                         "flow_category_indices": f["flow_category_indices"], "flow_instance_id": f["flow_instance_id"]}
            index.append([f["scene_id"], str(f["timestamp"])])
        h5lite.write_file(root / f"{frames[0]['scene_id']}.h5", tree)
    with open(root / "index_total.pkl", "wb") as fh:
        pickle.dump(index, fh)
    return index
