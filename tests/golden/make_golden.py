"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE'S OWN CODE.

Run once in the build container (needs /root/reference; never runs on the GPU box):

    python tests/golden/make_golden.py

What it does
  * puts /root/reference on sys.path and imports the reference's ``utils``, ``save_zip``,
    ``eval``, ``tools/test/save_zip_gt`` and ``tools/test/score`` unmodified;
  * the three imports those files make from the EMPTY ``OpenSceneFlow`` submodule are stubbed
    (and only those): ``fire.Fire`` (a CLI launcher), ``src.dataset.HDF5Dataset`` (replaced by an
    in-memory list of seeded frames with the same ``len``/``[i]`` protocol) and
    ``src.utils.av2_eval`` (``CATEGORY_TO_INDEX`` / ``BUCKETED_METACATAGORIES`` taken from the
    in-tree copy at tools/test/score.py:29-94; ``CLOSE_DISTANCE_THRESHOLD`` = 35.0, a value that
    is NOT in the reference tree -- SURVEY.md section 0.1);
  * runs ``save_zip.main``, ``eval.main`` (flow mode, zip mode and "raw"), ``save_zip_gt.main``
    and ``score.score`` end to end on small seeded frames and stores inputs + the reference's
    outputs as fixtures (data only -- no reference source is copied);
  * asserts that oracle/himo_oracle.py reproduces every one of those outputs, which is what
    pins the oracle.

Fixtures written: himo_golden.npz, eval_golden.json, *_pred.zip, *_gt.zip.
"""
from __future__ import annotations

import importlib.util
import io
import json
import os
import shutil
import sys
import tempfile
import types
import contextlib
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))

from himo_amd.synthetic import make_frame  # noqa: E402  (input generator only)

RES = "seflowpp_best"
_REGISTRY: dict[str, list] = {}


class _FakeHDF5Dataset:
    """Stand-in for the absent ``src.dataset.HDF5Dataset`` (constructor use: save_zip.py:111)."""

    def __init__(self, directory, vis_name="", eval=False, **kw):  # noqa: A002
        self.frames = _REGISTRY[str(directory)]

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i]


def _install_stubs():
    fire = types.ModuleType("fire")
    fire.Fire = lambda fn=None, *a, **k: None
    sys.modules["fire"] = fire

    spec = importlib.util.spec_from_file_location("ref_score", REF / "tools/test/score.py")
    score = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(score)

    src = types.ModuleType("src"); src.__path__ = []
    ds = types.ModuleType("src.dataset"); ds.HDF5Dataset = _FakeHDF5Dataset
    ut = types.ModuleType("src.utils"); ut.__path__ = []
    av2 = types.ModuleType("src.utils.av2_eval")
    av2.CLOSE_DISTANCE_THRESHOLD = 35.0
    av2.CATEGORY_TO_INDEX = score.CATEGORY_TO_INDEX
    av2.BUCKETED_METACATAGORIES = score.BUCKETED_METACATAGORIES
    sys.modules.update({"src": src, "src.dataset": ds, "src.utils": ut, "src.utils.av2_eval": av2})
    return score


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _frames(data_name: str):
    if data_name == "av2":
        sizes = [(0, 3000), (1, 2500), (2, 3500)]
    else:
        sizes = [(10, 3000), (11, 2047)]
    out = []
    for idx, n in sizes:
        f = make_frame(idx, n_points=n, n_instances=12, res_name=RES, data_name=data_name,
                       scene_id=f"{data_name}-scene-{idx % 2}")
        if data_name == "scania":   # put a handful of points inside the Scania ego box (utils/__init__.py:26)
            f["pc0"][:40, :3] = np.random.default_rng(idx).uniform([-9, -1.4, 0.1], [4.9, 1.3, 4.9], (40, 3)).astype(np.float32)
        else:
            f["pc0"][:40, :3] = np.random.default_rng(idx).uniform(-1.4, 1.4, (40, 3)).astype(np.float32)
        out.append(f)
    return out


def main():
    score_mod = _install_stubs()
    sys.path.insert(0, str(REF))
    ref_utils = importlib.import_module("utils")
    ref_save_zip = importlib.import_module("save_zip")
    ref_eval = _load("ref_eval", REF / "eval.py")
    ref_gt = _load("ref_save_zip_gt", REF / "tools/test/save_zip_gt.py")

    sys.path.insert(0, str(REPO / "oracle"))
    import himo_oracle as oracle

    gold: dict[str, np.ndarray] = {}
    evalj: dict = {}
    tmp = Path(tempfile.mkdtemp(prefix="himo_golden_"))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        for data_name in ("av2", "scania"):
            frames = _frames(data_name)
            data_dir = tmp / "data" / data_name / "val"
            data_dir.mkdir(parents=True)
            _REGISTRY[str(data_dir)] = frames
            gold[f"{data_name}/n_frames"] = np.array(len(frames))
            for i, f in enumerate(frames):
                for k, v in f.items():
                    gold[f"{data_name}/{i}/{k}"] = np.asarray(v)

            # ---- a1-a4 + a9: the reference's save_zip.main, end to end ----------------------
            with contextlib.redirect_stdout(io.StringIO()):
                ref_save_zip.main(data_dir=str(data_dir), res_name=RES)
            pred_zip = data_dir / "results" / f"{RES}-submit.zip"
            shutil.copy(pred_zip, HERE / f"{data_name}_pred.zip")
            for i, f in enumerate(frames):
                cd = ref_save_zip.read_output_zip(str(pred_zip), (f["scene_id"], str(f["timestamp"])))
                gold[f"{data_name}/{i}/ref_comp_dis"] = cd
                assert cd.dtype == np.float32
                mine = oracle.comp_dis_frame_f32(f, RES)
                assert np.array_equal(mine, cd), "oracle comp_dis != reference"
                # direct calls of utils (f64 and f32 chains)
                ego = np.linalg.inv(f["pose1"]) @ f["pose0"]
                pf = f["pc0"][:, :3] @ ego[:3, :3].T + ego[:3, 3] - f["pc0"][:, :3]
                est = f[RES] - pf
                dt0 = max(f["lidar_dt"]) - f["lidar_dt"]
                cd64 = ref_utils.flow2compDis(est, dt0, sensor_dt=0.1)
                gold[f"{data_name}/{i}/ref_comp_dis_f64"] = cd64
                gold[f"{data_name}/{i}/ref_refined_f64"] = ref_utils.refine_pts(f["pc0"], cd64)
                cd32 = ref_utils.flow2compDis(est.astype(np.float32), dt0, sensor_dt=0.1)
                assert cd32.dtype == np.float32
                gold[f"{data_name}/{i}/ref_comp_dis_f32chain"] = cd32
                gold[f"{data_name}/{i}/ref_refined_f32chain"] = ref_utils.refine_pts(f["pc0"], cd32)
                gold[f"{data_name}/{i}/ref_ego_mask_default"] = ref_utils.ego_pts_mask(f["pc0"])
                gold[f"{data_name}/{i}/ref_ego_mask_av2"] = ref_utils.ego_pts_mask(
                    f["pc0"], min_bound=[-1.5, -1.5, -2.0], max_bound=[1.5, 1.5, 2.0])
                assert np.array_equal(oracle.ego_pts_mask(f["pc0"]), gold[f"{data_name}/{i}/ref_ego_mask_default"])
                assert np.array_equal(oracle.comp_dis_frame(f, RES), cd64)

            # ---- a5-a8: the reference's eval.main in its three modes ----------------------------
            captured = {}
            orig_print = ref_eval.InstanceMetrics.print

            def _capture(self, res_name="flow", file_name="x.json"):
                captured["evaluate_data"] = json.loads(json.dumps(self.evaluate_data, default=float))
                captured["frame_cnt"] = self.frame_cnt
                return orig_print(self, res_name=res_name, file_name=file_name)

            ref_eval.InstanceMetrics.print = _capture
            for mode, kwargs in (("flow", dict(res_name=RES)), ("raw", dict(res_name="raw")),
                                 ("zip", dict(res_name=RES, comp_dis_zip=str(pred_zip)))):
                resfile = tmp / f"res-{data_name}.json"
                if resfile.exists():
                    resfile.unlink()
                with contextlib.redirect_stdout(io.StringIO()):
                    ref_eval.main(data_dir=str(data_dir), **kwargs)
                ref_json = json.loads(resfile.read_text())
                evalj[f"{data_name}/{mode}"] = {
                    "res_json": ref_json[data_name][kwargs["res_name"]],
                    "evaluate_data": captured["evaluate_data"],
                    "frame_cnt": captured["frame_cnt"],
                }
                # pin the oracle's InstanceMetrics against it
                om = oracle.InstanceMetrics(data_name)
                for f in frames:
                    cd = None
                    if mode == "zip":
                        cd = ref_save_zip.read_output_zip(str(pred_zip), (f["scene_id"], str(f["timestamp"])))
                    oracle.eval_frame(om, f, res_name=kwargs["res_name"], comp_dis=cd)
                got = json.loads(json.dumps(om.evaluate_data, default=float))
                assert got == captured["evaluate_data"], f"oracle InstanceMetrics != reference ({data_name}/{mode})"
                summ = om.summary()
                for cat, entry in ref_json[data_name][kwargs["res_name"]].items():
                    assert json.loads(json.dumps(summ[cat], default=float)) == entry, (cat, mode)
            ref_eval.InstanceMetrics.print = orig_print

            # eval mask as the reference builds it (eval.py:288-296) -- recorded via save_zip_gt below
            # ---- GT zip + leaderboard scorer ----------------------------------------------------
            gt_out = tmp / "gt" / data_name
            with contextlib.redirect_stdout(io.StringIO()):
                ref_gt.main(data_dir=str(data_dir), output_dir=str(gt_out), res_name="flow")
            gt_zip = gt_out / "flow-submit.zip"
            shutil.copy(gt_zip, HERE / f"{data_name}_gt.zip")
            for i, f in enumerate(frames):
                cd, em, cat, ins, fn, p0 = score_mod.read_data_file(str(gt_zip), (f["scene_id"], str(f["timestamp"])))
                gold[f"{data_name}/{i}/ref_gt_comp_dis"] = cd
                gold[f"{data_name}/{i}/ref_eval_mask"] = em
                gold[f"{data_name}/{i}/ref_gt_flow_norm"] = fn
                g = oracle.gt_frame(f, data_name)
                assert np.array_equal(g["comp_dis"], cd) and np.array_equal(g["eval_mask"], em)
                assert np.array_equal(g["gt_flow_norm"], fn)
            # scorer wants the dataset name in the path (score.py:556-561)
            with contextlib.redirect_stdout(io.StringIO()):
                scores = score_mod.score(str(gt_zip), str(HERE / f"{data_name}_pred.zip"), output_dir=str(tmp / "score" / data_name))
            scores.pop("per_category")
            evalj[f"{data_name}/scores"] = scores
            sm = oracle.ScoreMetrics()
            for uuid in score_mod.list_sweep_uuids(str(gt_zip)):   # the scorer walks the zip's own order (score.py:564)
                gd, em, cat, ins, fn, p0 = score_mod.read_data_file(str(gt_zip), uuid)
                ed, *_ = score_mod.read_data_file(str(HERE / f"{data_name}_pred.zip"), uuid)
                sm.step(gd, ed, em, gt_category=cat, gt_instance=ins, gt_flow_norm=fn, pc0=p0, data_name=data_name)
            mine = sm.compute_scores()
            for k, v in mine.items():
                assert scores[k] == v, (k, scores[k], v)

        # ---- a8 stand-alone: cal_chamfer / cal_mpe on random clouds -----------------------------
        rng = np.random.default_rng(2024)
        for j, (na, nb) in enumerate([(10, 10), (257, 300), (1500, 1200), (1, 5)]):
            a = rng.normal(size=(na, 3)).astype(np.float32) * 3
            b = (a[rng.integers(0, na, nb)] + rng.normal(size=(nb, 3)).astype(np.float32) * 0.2).astype(np.float32)
            gold[f"chamfer/{j}/a"], gold[f"chamfer/{j}/b"] = a, b
            gold[f"chamfer/{j}/ref"] = np.array(score_mod.cal_chamfer(a, b))
            assert oracle.cal_chamfer(a, b) == score_mod.cal_chamfer(a, b)
            assert ref_eval.InstanceMetrics("av2").cal_chamfer(a, b) == score_mod.cal_chamfer(a, b)
        gold["chamfer/n"] = np.array(4)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)

    np.savez_compressed(HERE / "himo_golden.npz", **gold)
    (HERE / "eval_golden.json").write_text(json.dumps(evalj, indent=1, sort_keys=True))
    print(f"wrote {len(gold)} arrays; oracle matches the reference on every fixture")


if __name__ == "__main__":
    main()
