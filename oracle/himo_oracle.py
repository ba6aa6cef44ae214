"""CPU ORACLE -- test infrastructure, NOT product code.

A numpy/scipy restatement of the reference's motion-compensation hot path
(/root/reference, KTH-RPL/HiMo snapshot 2026-01-30).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module, and only as the checker / the timed CPU baseline.  Nothing under
``himo_amd/`` imports it.

Pinning: every function below is checked against the reference's own code imported
from /root/reference (see tests/golden/make_golden.py, which runs the reference's
``save_zip.main``, ``eval.main``, ``tools/test/save_zip_gt.main`` and
``tools/test/score.score`` on seeded frames and commits inputs + outputs as
fixtures under tests/golden/).  Stages a1-a9 of SURVEY.md section 8 are therefore
PINNED.  The scene-flow producer (a10-a12) lives in an empty submodule and has no
oracle here; see oracle/seflow_oracle.py, whose header says "parity unpinned".

Every function cites the reference lines it follows.  Arithmetic keeps numpy's
dtype promotion and operation order so results are bit-comparable.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

# --------------------------------------------------------------------------
# constants
# --------------------------------------------------------------------------
# `CLOSE_DISTANCE_THRESHOLD` is imported by the reference from the absent
# OpenSceneFlow submodule (eval.py:21, save_zip.py:26); the value is not in the
# tree.  35.0 m is the Argoverse-2 scene-flow convention.
CLOSE_DISTANCE_THRESHOLD = 35.0

# tools/test/score.py:29-94 (in-tree copy of the av2_eval tables)
ANNOTATION_CATEGORIES = [
    "ANIMAL", "ARTICULATED_BUS", "BICYCLE", "BICYCLIST", "BOLLARD", "BOX_TRUCK", "BUS",
    "CONSTRUCTION_BARREL", "CONSTRUCTION_CONE", "DOG", "LARGE_VEHICLE", "MESSAGE_BOARD_TRAILER",
    "MOBILE_PEDESTRIAN_CROSSING_SIGN", "MOTORCYCLE", "MOTORCYCLIST", "OFFICIAL_SIGNALER", "PEDESTRIAN",
    "RAILED_VEHICLE", "REGULAR_VEHICLE", "SCHOOL_BUS", "SIGN", "STOP_SIGN", "STROLLER",
    "TRAFFIC_LIGHT_TRAILER", "TRUCK", "TRUCK_CAB", "VEHICULAR_TRAILER", "WHEELCHAIR", "WHEELED_DEVICE",
    "WHEELED_RIDER",
]
CATEGORY_TO_INDEX = {"NONE": 0}
CATEGORY_TO_INDEX.update({cat: i + 1 for i, cat in enumerate(ANNOTATION_CATEGORIES)})
BUCKETED_METACATAGORIES = {
    "BACKGROUND": ["NONE"],
    "CAR": ["REGULAR_VEHICLE"],
    "PEDESTRIAN": ["PEDESTRIAN", "STROLLER", "WHEELCHAIR", "OFFICIAL_SIGNALER"],
    "WHEELED_VRU": ["BICYCLE", "BICYCLIST", "MOTORCYCLE", "MOTORCYCLIST", "WHEELED_DEVICE", "WHEELED_RIDER"],
    "OTHER_VEHICLES": ["BOX_TRUCK", "LARGE_VEHICLE", "RAILED_VEHICLE", "TRUCK", "TRUCK_CAB",
                       "VEHICULAR_TRAILER", "ARTICULATED_BUS", "BUS", "SCHOOL_BUS"],
}
RANGES = ["0-10", "10-20", "20-30", "30+"]


# --------------------------------------------------------------------------
# a1-a5: utils/__init__.py and the inline math of save_zip.py / eval.py
# --------------------------------------------------------------------------
def ego_pts_mask(pts, min_bound=(-9.5, -3 / 2, 0), max_bound=(5, 2.760004 / 2, 5)):
    """utils/__init__.py:26-34 -- True for points OUTSIDE the ego box (strict compares)."""
    mask = ((pts[:, 0] > min_bound[0]) & (pts[:, 0] < max_bound[0])
            & (pts[:, 1] > min_bound[1]) & (pts[:, 1] < max_bound[1])
            & (pts[:, 2] > min_bound[2]) & (pts[:, 2] < max_bound[2]))
    return ~mask


def flow2compDis(flow, dt0, sensor_dt=10):
    """utils/__init__.py:36-43 -- divide THEN multiply (kept for bit parity)."""
    return flow / sensor_dt * dt0[:, None]


def refine_pts(pc, ds):
    """utils/__init__.py:45-47."""
    return pc[:, :3] + ds


def ego_pose(pose0, pose1):
    """save_zip.py:115 / eval.py:284."""
    return np.linalg.inv(pose1) @ pose0


def pose_flow(pc0, pose0, pose1):
    """save_zip.py:115-116 / eval.py:284-285 (f32 points @ f64 pose -> f64)."""
    ego = ego_pose(pose0, pose1)
    return pc0[:, :3] @ ego[:3, :3].T + ego[:3, 3] - pc0[:, :3]


def remove_ego_motion(pc0, pose0, pose1, flow, raw=False):
    """save_zip.py:117 / eval.py:302 -- ``est_flow``; ``raw`` = the ``res_name == "raw"`` branch."""
    pf = pose_flow(pc0, pose0, pose1)
    return np.zeros_like(pf) if raw else (flow - pf)


def dt0_from_lidar_dt(lidar_dt):
    """save_zip.py:120 / eval.py:299 -- time remaining to the latest point of the sweep."""
    return max(lidar_dt) - lidar_dt


def comp_dis_frame(frame: dict, res_name: str, sensor_dt: float = 0.1):
    """The body of the per-frame loop, save_zip.py:113-121.  Returns what the reference
    hands to ``write_output_file`` (dtype follows numpy promotion: f64 for f64 poses)."""
    est_flow = remove_ego_motion(frame["pc0"], frame["pose0"], frame["pose1"],
                                 None if res_name == "raw" else frame[res_name], raw=(res_name == "raw"))
    dt0 = dt0_from_lidar_dt(frame["lidar_dt"])
    return flow2compDis(est_flow, dt0, sensor_dt=sensor_dt)


def comp_dis_frame_f32(frame: dict, res_name: str, sensor_dt: float = 0.1):
    """What lands in the Feather file: the f32 cast of save_zip.py:70-72."""
    return comp_dis_frame(frame, res_name, sensor_dt).astype(np.float32)


# --------------------------------------------------------------------------
# a6: evaluation mask, eval.py:288-296
# --------------------------------------------------------------------------
def eval_mask(frame: dict, data_name: str):
    pc0 = frame["pc0"]
    pc_dis = np.linalg.norm(pc0[:, :2], axis=1)           # eval.py:288
    dis_mask = pc_dis <= CLOSE_DISTANCE_THRESHOLD          # eval.py:289
    notgm_mask = ~frame["gm0"]                             # eval.py:290
    if data_name == "scania":                              # eval.py:293-294
        return dis_mask & frame["flow_is_valid"] & notgm_mask & ego_pts_mask(pc0)
    return dis_mask & notgm_mask & ego_pts_mask(pc0, min_bound=[-1.5, -1.5, -2.0], max_bound=[1.5, 1.5, 2.0])


# --------------------------------------------------------------------------
# a8: Chamfer / MPE, eval.py:50-62, :95; tools/test/score.py:180-197
# --------------------------------------------------------------------------
def cal_chamfer(pc1, pc2) -> float:
    if len(pc1) == 0 or len(pc2) == 0:
        return float("nan")
    d12, _ = cKDTree(pc2).query(pc1, k=1)
    d21, _ = cKDTree(pc1).query(pc2, k=1)
    return float((np.nanmean(d12) + np.nanmean(d21)) / 2.0)


def cal_mpe(pc1, pc2) -> float:
    return np.linalg.norm(pc1 - pc2, axis=1).mean()


def nearest_neighbor(query, ref):
    """k=1 Euclidean NN distances and indices of ``query`` in ``ref`` (the two halves of
    cal_chamfer; also the correspondence search of the self-supervised loss, a11)."""
    d, i = cKDTree(ref).query(query, k=1)
    return d, i


def _bucket(values):
    """eval.py:99-110."""
    if 0 < values < 10:
        return "0-10"
    if 10 <= values < 20:
        return "10-20"
    if 20 <= values < 30:
        return "20-30"
    if values >= 30:
        return "30+"
    return None


# --------------------------------------------------------------------------
# a7: InstanceMetrics, eval.py:24-149 (print/JSON side: eval.py:151-268)
# --------------------------------------------------------------------------
class InstanceMetrics:
    def __init__(self, data_name, sensor_hz=10.0):
        self.frame_cnt = 0
        self.sensor_dt = 1.0 / sensor_hz
        self.data_name = data_name
        self.min_vel = 1.5 if data_name in ["scania"] else 3.0   # eval.py:33-36
        self.evaluate_data = self.init_evaluate_data()

    @staticmethod
    def init_evaluate_data():
        init = lambda: {"num_pts": [], "mpe": [], "cham": [], "std_mpe": [], "std_cham": []}
        d = {"CAR": {}, "OTHER_VEHICLES": {}}
        for c in d:
            d[c]["vel"] = {r: init() for r in RANGES}
            d[c]["dis"] = {r: init() for r in RANGES}
            d[c]["mean"] = init()
        return d

    def step_eval(self, pc, gt_flow, pc_dt0, gt_category, gt_instance, est_flow=None, est_dis=None):
        frame_score = self.init_evaluate_data()
        if est_flow is not None:                                           # eval.py:67-70
            refine_pc = refine_pts(pc, flow2compDis(est_flow, pc_dt0, sensor_dt=self.sensor_dt))
        elif est_dis is not None:
            refine_pc = refine_pts(pc, est_dis)
        gt_refine_pc = refine_pts(pc, flow2compDis(gt_flow, pc_dt0, sensor_dt=self.sensor_dt))  # :72

        for cats_name in ["CAR", "OTHER_VEHICLES"]:                          # eval.py:75
            ids = [CATEGORY_TO_INDEX[c] for c in BUCKETED_METACATAGORIES[cats_name]]
            mask_class = np.isin(gt_category, np.array(ids))
            if np.sum(mask_class) == 0:
                continue
            ins = gt_instance[mask_class]
            gt_flow_c = gt_flow[mask_class]
            ref_c = refine_pc[mask_class]
            gt_ref_c = gt_refine_pc[mask_class]
            pc_c = pc[mask_class]
            for instance_id in np.unique(ins):                               # eval.py:88
                m = ins == instance_id
                num_pts = np.sum(m)
                vel_ins = np.linalg.norm(gt_flow_c[m], axis=1).mean() / self.sensor_dt
                if num_pts < 10 or vel_ins < self.min_vel:
                    continue
                dis_ins = np.linalg.norm(pc_c[m], axis=1).mean()             # eval.py:94 (all columns of pc)
                mpe = np.linalg.norm(gt_ref_c[m] - ref_c[m], axis=1).mean()  # eval.py:95
                cham = cal_chamfer(gt_ref_c[m], ref_c[m])                    # eval.py:96
                for metric, values in [("vel", vel_ins), ("dis", dis_ins)]:
                    r = _bucket(values)
                    if r is None:
                        continue
                    frame_score[cats_name][metric][r]["num_pts"].append(num_pts)
                    frame_score[cats_name][metric][r]["mpe"].append(mpe)
                    frame_score[cats_name][metric][r]["cham"].append(cham)

        for cats_name in frame_score:                                        # eval.py:116-147
            total_num_list, mpe_list, cham_list = [], [], []
            for metric in ["vel", "dis"]:
                for r in frame_score[cats_name][metric]:
                    fs = frame_score[cats_name][metric][r]
                    if len(fs["num_pts"]) > 0:
                        ed = self.evaluate_data[cats_name][metric][r]
                        ed["num_pts"] += fs["num_pts"]
                        ed["mpe"] += fs["mpe"]
                        ed["cham"] += fs["cham"]
                        if metric == "vel":
                            mpe_list.append(np.average(fs["mpe"], weights=fs["num_pts"]))
                            cham_list.append(np.average(fs["cham"], weights=fs["num_pts"]))
                            total_num_list.append(sum(fs["num_pts"]))
            num_pts = sum(total_num_list)
            if num_pts == 0:
                continue
            mean = self.evaluate_data[cats_name]["mean"]
            mean["num_pts"].append(num_pts)
            mean["mpe"].append(np.nanmean(mpe_list))
            mean["cham"].append(np.nanmean(cham_list))
            mean["std_mpe"].append(np.nanstd(mpe_list))
            mean["std_cham"].append(np.nanstd(cham_list))
        self.frame_cnt += 1

    def summary(self) -> dict:
        """The numbers eval.py:151-268 prints and writes to ``res-<data>.json`` (the
        ``entry`` dict of ``savejson``), keyed by category, plus the Total row."""
        def safe_average(v, w):
            return float(np.average(v, weights=w)) if len(v) > 0 and np.sum(w) > 0 else 0.0

        def safe_std(v):
            return float(np.std(v)) if len(v) > 0 else 0.0

        out, tot = {}, {"mpe": [], "cham": [], "num_pts": []}
        for cat in ["CAR", "OTHER_VEHICLES"]:
            raw = self.evaluate_data[cat]
            mean = raw["mean"]
            if len(mean["num_pts"]) == 0:
                continue
            entry = {
                "overall": {
                    "mpe": safe_average(mean["mpe"], mean["num_pts"]),
                    "cd": safe_average(mean["cham"], mean["num_pts"]),
                    "std_mpe": safe_std(mean["std_mpe"]), "std_cd": safe_std(mean["std_cham"]),
                    "num_pts": int(np.sum(mean["num_pts"])), "num_obj": int(len(mean["num_pts"])),
                },
                "velocity": {}, "distance": {},
            }
            for r in RANGES:
                for key, metric in (("velocity", "vel"), ("distance", "dis")):
                    v = raw[metric][r]
                    entry[key][r] = {"mpe": safe_average(v["mpe"], v["num_pts"]),
                                     "cd": safe_average(v["cham"], v["num_pts"]),
                                     "num_pts": int(np.sum(v["num_pts"])), "num_obj": int(len(v["num_pts"]))}
            out[cat] = entry
            tot["mpe"].extend(mean["mpe"]); tot["cham"].extend(mean["cham"]); tot["num_pts"].extend(mean["num_pts"])
        if len(tot["num_pts"]) > 0:
            out["Total"] = {"mpe": safe_average(tot["mpe"], tot["num_pts"]),
                            "cd": safe_average(tot["cham"], tot["num_pts"]),
                            "num_pts": int(np.sum(tot["num_pts"])), "num_obj": int(len(tot["num_pts"]))}
        return out


def eval_frame(metrics: InstanceMetrics, frame: dict, res_name: str = "", comp_dis=None):
    """One iteration of the loop at eval.py:281-310 (EVAL_FLAG 2 when ``comp_dis`` is None,
    EVAL_FLAG 1 when a comp_dis array read from a zip is given)."""
    pc0 = frame["pc0"]
    pf = pose_flow(pc0, frame["pose0"], frame["pose1"])
    gt_flow = frame["flow"] - pf
    m = eval_mask(frame, metrics.data_name)
    dt0 = dt0_from_lidar_dt(frame["lidar_dt"])
    cat, ins = frame["flow_category_indices"][m], frame["flow_instance_id"][m]
    if comp_dis is None:
        est_flow = np.zeros_like(pf) if res_name == "raw" else (frame[res_name] - pf)
        metrics.step_eval(pc0[m, :], gt_flow[m, :], dt0[m], cat, ins, est_flow=est_flow[m, :])
    else:
        metrics.step_eval(pc0[m, :], gt_flow[m, :], dt0[m], cat, ins, est_dis=comp_dis[m, :])


# --------------------------------------------------------------------------
# GT side of the leaderboard: tools/test/save_zip_gt.py:140-172
# --------------------------------------------------------------------------
def gt_frame(frame: dict, data_name: str, sensor_dt: float = 0.1) -> dict:
    pc0 = frame["pc0"]
    pf = pose_flow(pc0, frame["pose0"], frame["pose1"])
    gt_flow = frame["flow"] - pf
    dt0 = dt0_from_lidar_dt(frame["lidar_dt"])
    return {
        "comp_dis": flow2compDis(gt_flow, dt0, sensor_dt=sensor_dt).astype(np.float32),
        "eval_mask": eval_mask(frame, data_name),
        "gt_flow_norm": np.linalg.norm(gt_flow, axis=1).astype(np.float32),
        "pc0": pc0[:, :3],
    }


# --------------------------------------------------------------------------
# Leaderboard scorer: tools/test/score.py:200-456
# --------------------------------------------------------------------------
class ScoreMetrics:
    def __init__(self):
        self.frame_cnt = 0
        init = lambda: {"num_pts": [], "mpe": [], "cham": []}
        self.evaluate_data = {c: {"vel": {r: init() for r in RANGES},
                                  "mean": {"num_pts": [], "mpe": [], "cham": [], "std_mpe": [], "std_cham": []}}
                              for c in ("CAR", "OTHER_VEHICLES")}

    def step(self, gt_dis, est_dis, eval_mask, gt_category=None, gt_instance=None, gt_flow_norm=None,
             pc0=None, sensor_dt=0.1, data_name="av2"):
        self.frame_cnt += 1                                                 # score.py:240
        mask = eval_mask.astype(bool)
        gt_dis, est_dis = gt_dis[mask], est_dis[mask]
        if gt_category is None or gt_instance is None:
            return
        gt_category, gt_instance = gt_category[mask], gt_instance[mask]
        gt_flow_norm = gt_flow_norm[mask] if gt_flow_norm is not None else None
        pc0 = pc0[mask] if pc0 is not None else None
        min_vel = 1.5 if data_name == "scania" else 3.0
        frame_score = {c: {r: {"num_pts": [], "mpe": [], "cham": []} for r in RANGES} for c in ("CAR", "OTHER_VEHICLES")}
        for cats_name in ["CAR", "OTHER_VEHICLES"]:
            ids = [CATEGORY_TO_INDEX[c] for c in BUCKETED_METACATAGORIES[cats_name]]
            mc = np.isin(gt_category, np.array(ids))
            if not np.any(mc):
                continue
            ins_c, gt_c, est_c = gt_instance[mc], gt_dis[mc], est_dis[mc]
            fn_c = gt_flow_norm[mc] if gt_flow_norm is not None else None
            pc_c = pc0[mc] if pc0 is not None else None
            for instance_id in np.unique(ins_c):
                m = ins_c == instance_id
                num_pts = np.sum(m)
                if num_pts < 10:
                    continue
                vel_ins = np.mean(fn_c[m]) / sensor_dt if fn_c is not None else min_vel + 1
                if vel_ins < min_vel:
                    continue
                mpe = cal_mpe(gt_c[m], est_c[m])                              # score.py:300
                if pc_c is not None:
                    cham = cal_chamfer(pc_c[m] + gt_c[m], pc_c[m] + est_c[m])  # score.py:303-306
                else:
                    cham = cal_chamfer(gt_c[m], est_c[m])
                r = _bucket(vel_ins)
                if r is None:
                    continue
                frame_score[cats_name][r]["num_pts"].append(num_pts)
                frame_score[cats_name][r]["mpe"].append(mpe)
                frame_score[cats_name][r]["cham"].append(cham)
        for cats_name in frame_score:
            tot, mpes, chams = [], [], []
            for r in RANGES:
                fs = frame_score[cats_name][r]
                if len(fs["num_pts"]) > 0:
                    ed = self.evaluate_data[cats_name]["vel"][r]
                    ed["num_pts"] += fs["num_pts"]; ed["mpe"] += fs["mpe"]; ed["cham"] += fs["cham"]
                    mpes.append(np.average(fs["mpe"], weights=fs["num_pts"]))
                    chams.append(np.average(fs["cham"], weights=fs["num_pts"]))
                    tot.append(sum(fs["num_pts"]))
            if sum(tot) == 0:
                continue
            mean = self.evaluate_data[cats_name]["mean"]
            mean["num_pts"].append(sum(tot))
            mean["mpe"].append(np.nanmean(mpes)); mean["cham"].append(np.nanmean(chams))
            mean["std_mpe"].append(np.nanstd(mpes)); mean["std_cham"].append(np.nanstd(chams))

    def compute_scores(self) -> dict:
        """score.py:362-456 (flat leaderboard keys only)."""
        def safe_average(v, w):
            return float(np.average(v, weights=w)) if len(v) > 0 and np.sum(w) > 0 else 0.0
        out, mp, ch, pts = {}, [], [], []
        for cat, key in (("CAR", "car"), ("OTHER_VEHICLES", "others")):
            mean = self.evaluate_data[cat]["mean"]
            out[f"{key}_cde"] = safe_average(mean["cham"], mean["num_pts"])
            out[f"{key}_mpe"] = safe_average(mean["mpe"], mean["num_pts"])
            out[f"{key}_num_objs"] = len(mean["num_pts"])
            out[f"{key}_num_pts"] = int(np.sum(mean["num_pts"])) if len(mean["num_pts"]) else 0
            mp.extend(mean["mpe"]); ch.extend(mean["cham"]); pts.extend(mean["num_pts"])
        out["mpe"] = safe_average(mp, pts)
        out["chamfer"] = safe_average(ch, pts)
        out["num_frames"] = self.frame_cnt
        out["num_instances"] = len(pts)
        out["total_points"] = int(np.sum(pts)) if len(pts) else 0
        return out
