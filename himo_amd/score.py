"""Drop-in for the reference's leaderboard scorer ``tools/test/score.py``: GT zip vs prediction zip
-> MPE / Chamfer per CAR / OTHER_VEHICLES, speed buckets, ``scores.json``.

    read_data_file(path, (scene, ts))     score.py:96-144
    list_sweep_uuids(path)                score.py:147-177
    cal_chamfer / cal_mpe                 score.py:180-197
    ScoreMetrics.step / compute_scores / save_detailed_json      score.py:200-542
    score(gt_zip, pred_zip, output_dir, flow_mode)               score.py:545-667

The per-instance work (float32 point error, Chamfer on pc0 + comp_dis by exact 1-NN) runs on the GPU
through himo_eval_instances in HIMO_EVAL_SCORE mode, for a batch of sweeps at a time; the bucket
bookkeeping stays on the host in the reference's order.
"""
from __future__ import annotations

import json
from pathlib import Path
from zipfile import ZipFile

import numpy as np
import torch

from . import feather

from .eval import (BUCKETED_METACATAGORIES, CATEGORY_TO_INDEX, EVAL_GROUPS, MODE_SCORE, RANGES,  # noqa: F401
                   InstanceEvaluator, chamfer_distance, range_name_of)


def read_data_file(data_path: str, sweep_uuid: tuple) -> tuple:
    """(comp_dis, eval_mask, flow_category, flow_instance, gt_flow_norm, pc0) of one sweep from a zip or an
    extracted directory; columns that are absent come back as ``None`` (``eval_mask``: all True)."""
    member = f"{sweep_uuid[0]}/{sweep_uuid[1]}.feather"
    data_path = Path(data_path)
    if data_path.is_dir():
        df = feather.read_table((data_path / member).read_bytes())
    else:
        with ZipFile(data_path, "r") as z:
            df = feather.read_table(z.read(member))
    col = lambda name, dt: df[name].astype(dt) if name in df else None
    comp_dis = np.stack([df[f"comp_dis_{a}_m"].astype(np.float32) for a in "xyz"], axis=1)
    eval_mask = col("eval_mask", bool)
    if eval_mask is None:
        eval_mask = np.ones(len(comp_dis), dtype=bool)
    pc0 = None
    if all(f"pc0_{a}" in df for a in "xyz"):
        pc0 = np.stack([df[f"pc0_{a}"].astype(np.float32) for a in "xyz"], axis=1)
    return (comp_dis, eval_mask, col("flow_category_indices", np.uint8), col("flow_instance_id", np.uint32),
            col("gt_flow_norm", np.float32), pc0)


def list_sweep_uuids(data_path: str) -> list:
    data_path = Path(data_path)
    if data_path.is_dir():
        parts = [p.relative_to(data_path).parts for p in data_path.rglob("*.feather")]
    else:
        with ZipFile(data_path, "r") as z:
            parts = [tuple(n.split("/")) for n in z.namelist() if n.endswith(".feather")]
    return [(p[0], p[1].replace(".feather", "")) for p in parts if len(p) == 2]


def cal_chamfer(pc1: np.ndarray, pc2: np.ndarray) -> float:
    if len(pc1) == 0 or len(pc2) == 0:
        return float("nan")
    return chamfer_distance(pc1, pc2)


def cal_mpe(pc1: np.ndarray, pc2: np.ndarray) -> float:
    return np.linalg.norm(np.asarray(pc1) - np.asarray(pc2), axis=1).mean()


class ScoreMetrics:
    def __init__(self):
        self.frame_cnt = 0
        self.evaluate_data = self._init_evaluate_data()
        self._evaluator = None

    def _init_evaluate_data(self):
        new = lambda: {"num_pts": [], "mpe": [], "cham": []}
        return {c: {"vel": {r: new() for r in RANGES},
                    "mean": {"num_pts": [], "mpe": [], "cham": [], "std_mpe": [], "std_cham": []}} for c in EVAL_GROUPS}

    def step(self, gt_dis, est_dis, eval_mask, gt_category=None, gt_instance=None, gt_flow_norm=None, pc0=None,
             sensor_dt: float = 0.1, data_name: str = "av2"):
        self.step_many([(gt_dis, est_dis, eval_mask, gt_category, gt_instance, gt_flow_norm, pc0)],
                       sensor_dt=sensor_dt, data_name=data_name)

    def step_many(self, sweeps, sensor_dt: float = 0.1, data_name: str = "av2"):
        """``sweeps``: list of (gt_dis, est_dis, eval_mask, gt_category, gt_instance, gt_flow_norm, pc0)."""
        self.frame_cnt += len(sweeps)                                  # score.py:240 counts every sweep
        sweeps = [s for s in sweeps if s[3] is not None and s[4] is not None]    # score.py:251-252
        if not sweeps:
            return
        if self._evaluator is None:
            self._evaluator = InstanceEvaluator()
        min_vel = 1.5 if data_name == "scania" else 3.0
        # The reference decides PER SWEEP whether the velocity filter and the pc0 offset apply (score.py:291-296, :270-283:
        # ``if gt_flow_norm is not None`` / ``if pc0 is not None``), so sweeps only share a launch with sweeps of the same
        # kind; the per-sweep results are then accumulated in the original sweep order.
        kinds = {}
        for k, sw in enumerate(sweeps):
            kinds.setdefault((sw[5] is not None, sw[6] is not None), []).append(k)
        per_sweep = [None] * len(sweeps)
        for (have_norm, have_pc0), members in kinds.items():
            recs = self._run_group([sweeps[k] for k in members], have_norm, have_pc0, sensor_dt)
            bounds = np.searchsorted(recs["frame"], np.arange(len(members) + 1))
            for j, k in enumerate(members):
                per_sweep[k] = (recs[bounds[j]:bounds[j + 1]], have_norm)
        for recs, have_norm in per_sweep:
            self._accumulate(recs, min_vel, have_norm)

    def _run_group(self, sweeps, have_norm: bool, have_pc0: bool, sensor_dt: float):
        ev, dev = self._evaluator, self._evaluator.device
        counts = [len(s[0]) for s in sweeps]
        offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(dev)
        cat = lambda k, dt: torch.from_numpy(np.ascontiguousarray(np.concatenate([np.asarray(s[k]).astype(dt) for s in sweeps]))).to(dev)
        return ev.run(len(sweeps), offsets, cat(6, np.float32) if have_pc0 else None, cat(0, np.float32), cat(1, np.float32),
                      cat(5, np.float32) if have_norm else None, cat(3, np.uint8), cat(4, np.int64), cat(2, np.uint8),
                      MODE_SCORE, sensor_dt=sensor_dt)

    def _accumulate(self, recs, min_vel, have_norm):
        frame_score = {c: {r: {"num_pts": [], "mpe": [], "cham": []} for r in RANGES} for c in EVAL_GROUPS}
        for gid, cats_name in enumerate(EVAL_GROUPS, start=1):
            for r in recs[recs["group"] == gid]:
                num_pts = int(r["num_pts"])
                if num_pts < 10:
                    continue
                vel_ins = float(r["vel"]) if have_norm else min_vel + 1       # score.py:291-296
                if vel_ins < min_vel:
                    continue
                name = range_name_of(vel_ins)
                if name is None:
                    continue
                slot = frame_score[cats_name][name]
                slot["num_pts"].append(num_pts)
                slot["mpe"].append(float(r["mpe"]))
                slot["cham"].append(float(r["cham"]))
        for cats_name in EVAL_GROUPS:
            totals, mpes, chams = [], [], []
            for name in RANGES:
                got = frame_score[cats_name][name]
                if not got["num_pts"]:
                    continue
                keep = self.evaluate_data[cats_name]["vel"][name]
                for k in ("num_pts", "mpe", "cham"):
                    keep[k] += got[k]
                mpes.append(np.average(got["mpe"], weights=got["num_pts"]))
                chams.append(np.average(got["cham"], weights=got["num_pts"]))
                totals.append(sum(got["num_pts"]))
            if sum(totals) == 0:
                continue
            mean = self.evaluate_data[cats_name]["mean"]
            mean["num_pts"].append(sum(totals))
            mean["mpe"].append(np.nanmean(mpes)); mean["cham"].append(np.nanmean(chams))
            mean["std_mpe"].append(np.nanstd(mpes)); mean["std_cham"].append(np.nanstd(chams))

    @staticmethod
    def _wavg(v, w):
        return float(np.average(v, weights=w)) if len(v) > 0 and np.sum(w) > 0 else 0.0

    def _category_summary(self, cat):
        mean, vel = self.evaluate_data[cat]["mean"], self.evaluate_data[cat]["vel"]
        velocity = {r: {"mpe": self._wavg(vel[r]["mpe"], vel[r]["num_pts"]), "cd": self._wavg(vel[r]["cham"], vel[r]["num_pts"]),
                        "num_pts": int(np.sum(vel[r]["num_pts"])) if vel[r]["num_pts"] else 0, "num_obj": len(vel[r]["num_pts"])}
                    for r in RANGES}
        if not mean["num_pts"]:
            return {"mpe_mean": 0.0, "mpe_std": 0.0, "cham_mean": 0.0, "cham_std": 0.0, "num_pts": 0, "num_objs": 0,
                    "velocity": velocity}
        return {"mpe_mean": self._wavg(mean["mpe"], mean["num_pts"]), "mpe_std": float(np.std(mean["std_mpe"])),
                "cham_mean": self._wavg(mean["cham"], mean["num_pts"]), "cham_std": float(np.std(mean["std_cham"])),
                "num_pts": int(np.sum(mean["num_pts"])), "num_objs": len(mean["num_pts"]), "velocity": velocity}

    def compute_scores(self) -> dict:
        per_cat = {c: self._category_summary(c) for c in EVAL_GROUPS}
        mp, ch, pts = [], [], []
        for c in EVAL_GROUPS:
            mean = self.evaluate_data[c]["mean"]
            mp += mean["mpe"]; ch += mean["cham"]; pts += mean["num_pts"]
        out = {"mpe": self._wavg(mp, pts), "chamfer": self._wavg(ch, pts), "num_frames": self.frame_cnt,
               "num_instances": len(pts), "total_points": int(np.sum(pts)) if pts else 0}
        for c, key in (("CAR", "car"), ("OTHER_VEHICLES", "others")):
            out[f"{key}_cde"] = float(per_cat[c]["cham_mean"]); out[f"{key}_mpe"] = float(per_cat[c]["mpe_mean"])
            out[f"{key}_num_objs"] = int(per_cat[c]["num_objs"]); out[f"{key}_num_pts"] = int(per_cat[c]["num_pts"])
        out["per_category"] = per_cat
        return out

    def save_detailed_json(self, data_name: str, flow_mode: str, file_path: str):
        file_path = Path(file_path)
        data = {}
        if file_path.exists():
            try:
                data = json.loads(file_path.read_text())
            except json.JSONDecodeError:
                data = {}
        slot = data.setdefault(data_name, {}).setdefault(flow_mode, {})
        zero = {"mpe": 0.0, "cd": 0.0, "num_pts": 0, "num_obj": 0}
        for c in EVAL_GROUPS:
            if not self.evaluate_data[c]["mean"]["num_pts"]:
                continue
            s = self._category_summary(c)
            slot[c] = {"overall": {"mpe": s["mpe_mean"], "cd": s["cham_mean"], "std_mpe": s["mpe_std"], "std_cd": s["cham_std"],
                                   "num_pts": s["num_pts"], "num_obj": s["num_objs"]},
                       "velocity": s["velocity"], "distance": {r: dict(zero) for r in RANGES}}   # distance not tracked (score.py:529)
        file_path.write_text(json.dumps(data, indent=4))
        return file_path


def score(gt_zip_path: str, pred_zip_path: str, output_dir: str = None, flow_mode: str = "submission",
          batch_sweeps: int = 32) -> dict:
    from tabulate import tabulate
    low = (gt_zip_path.lower(), pred_zip_path.lower())
    data_name = "scania" if any("scania" in s for s in low) else ("av2" if any("av2" in s for s in low) else "scania")
    pred_sweeps = set(list_sweep_uuids(pred_zip_path))
    metrics = ScoreMetrics()
    missing, mismatch, pending = [], [], []

    def flush():
        if pending:
            metrics.step_many(pending, data_name=data_name)
            pending.clear()

    for uuid in list_sweep_uuids(gt_zip_path):
        if uuid not in pred_sweeps:
            missing.append(uuid)
            print(f"Warning: Missing prediction for {uuid}")
            continue
        gt_dis, eval_mask, cat, ins, norm, pc0 = read_data_file(gt_zip_path, uuid)
        est_dis = read_data_file(pred_zip_path, uuid)[0]
        if len(gt_dis) != len(est_dis):
            mismatch.append((uuid, len(gt_dis), len(est_dis)))
            print(f"Warning: Point count mismatch for {uuid}: GT={len(gt_dis)}, Pred={len(est_dis)}")
            continue
        pending.append((gt_dis, est_dis, eval_mask, cat, ins, norm, pc0))
        if len(pending) >= batch_sweeps:
            flush()
    flush()
    scores = metrics.compute_scores()

    rows, pts, objs, wc, wm = [], 0, 0, 0.0, 0.0
    for c, shown in (("CAR", "CAR"), ("OTHER_VEHICLES", "OTHERS")):
        s = scores["per_category"][c]
        rows.append([shown, f"{s['cham_mean']:.3f} ± {s['cham_std']:.2f}", f"{s['mpe_mean']:.3f} ± {s['mpe_std']:.2f}",
                     s["num_pts"], s["num_objs"]])
        pts += s["num_pts"]; objs += s["num_objs"]
        wc += s["cham_mean"] * s["num_pts"]; wm += s["mpe_mean"] * s["num_pts"]
    rows.insert(0, ["Total", f"{wc / max(1, pts):.3f}", f"{wm / max(1, pts):.3f}", pts, objs])     # score.py:622-628
    print(f"\n{'=' * 50}\nHiMo refinement metrics in {data_name}:")
    print(tabulate(rows, headers=["Class", "CDE (Chamfer) ↓", "MPE (Point Err) ↓", "# Points", "# Objs"],
                   tablefmt="fancy_grid", stralign="center"))
    print(f"Total frames processed: {scores['num_frames']}\n{'=' * 50}\n")
    if missing:
        print(f"Missing predictions for {len(missing)} sweeps. Examples:\n{missing[:5]}")
    if mismatch:
        print(f"Point-count mismatches for {len(mismatch)} sweeps. Examples (sweep, GT_count, Pred_count):\n{mismatch[:5]}")
    if output_dir is not None:
        out = Path(output_dir)
        out.mkdir(exist_ok=True, parents=True)
        (out / "scores.json").write_text(json.dumps(scores, indent=2))
        metrics.save_detailed_json(data_name, flow_mode, str(out / f"res-{data_name}.json"))
        print(f"Scores saved to {out / 'scores.json'}")
    return scores


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="HiMo Benchmark Scoring Program (MI355X path)")
    ap.add_argument("--gt_zip", required=True, help="ground-truth zip file or extracted directory")
    ap.add_argument("--pred_zip", required=True, help="prediction zip file or extracted directory")
    ap.add_argument("--output_dir", default=None)
    ap.add_argument("--flow_mode", default="submission")
    a = ap.parse_args(argv)
    score(a.gt_zip, a.pred_zip, a.output_dir, a.flow_mode)


if __name__ == "__main__":
    main()
