"""Leaderboard scorer (tools/test/score.py) on the GPU vs the reference's own scores.json numbers.
Inputs are rebuilt from the golden arrays (the GPU box has no pyarrow to read the fixture zips)."""
import numpy as np
import pytest

from conftest import golden_frames

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("data_name", ["av2", "scania"])
def test_scores_match_reference(gpu, gold, eval_gold, data_name):
    from himo_amd.score import ScoreMetrics
    ref = eval_gold[f"{data_name}/scores"]
    frames = golden_frames(gold, data_name)
    sweeps = []
    for i, f in enumerate(frames):
        sweeps.append((gold[f"{data_name}/{i}/ref_gt_comp_dis"], gold[f"{data_name}/{i}/ref_comp_dis"],
                       gold[f"{data_name}/{i}/ref_eval_mask"], f["flow_category_indices"],
                       f["flow_instance_id"].astype(np.uint32), gold[f"{data_name}/{i}/ref_gt_flow_norm"],
                       np.ascontiguousarray(f["pc0"][:, :3])))
    batched, single = ScoreMetrics(), ScoreMetrics()
    batched.step_many(sweeps, data_name=data_name)
    for s in sweeps:
        single.step(s[0], s[1], s[2], gt_category=s[3], gt_instance=s[4], gt_flow_norm=s[5], pc0=s[6], data_name=data_name)
    for m in (batched, single):
        got = m.compute_scores()
        for k, v in ref.items():
            if isinstance(v, float):
                # float32 means in the reference (score.py:197): agree to float32 resolution
                assert got[k] == pytest.approx(v, rel=2e-6), k
            else:
                assert got[k] == v, k
    assert batched.compute_scores()["num_instances"] > 0


def test_missing_labels_only_counts_the_frame(gpu, gold):
    from himo_amd.score import ScoreMetrics
    m = ScoreMetrics()
    cd = gold["av2/0/ref_comp_dis"]
    m.step(cd, cd, np.ones(len(cd), bool))                 # no category / instance columns (a prediction zip)
    assert m.frame_cnt == 1 and m.compute_scores()["num_instances"] == 0


def test_without_pc0_and_flow_norm(gpu, gold, oracle):
    from himo_amd.score import ScoreMetrics
    f = golden_frames(gold, "av2")[0]
    args = (gold["av2/0/ref_gt_comp_dis"], gold["av2/0/ref_comp_dis"], gold["av2/0/ref_eval_mask"])
    kw = dict(gt_category=f["flow_category_indices"], gt_instance=f["flow_instance_id"].astype(np.uint32))
    mine, ref = ScoreMetrics(), oracle.ScoreMetrics()
    mine.step(*args, **kw)
    ref.step(*args, **kw)
    a, b = mine.compute_scores(), ref.compute_scores()
    for k, v in b.items():
        assert a[k] == pytest.approx(v, rel=2e-6), k


@pytest.mark.parametrize("data_name", ["av2", "scania"])
def test_score_program_on_the_reference_written_zips(gpu, eval_gold, data_name, tmp_path, capsys):
    """tools/test/score.py's whole program: GT zip + prediction zip (both written by the reference, LZ4 Feather)
    -> scores.json.  Runs on the GPU box without pyarrow thanks to himo_amd/feather.py."""
    import json
    from conftest import GOLDEN
    from himo_amd.score import score
    ref = eval_gold[f"{data_name}/scores"]
    got = score(str(GOLDEN / f"{data_name}_gt.zip"), str(GOLDEN / f"{data_name}_pred.zip"), output_dir=str(tmp_path))
    for k, v in ref.items():
        if isinstance(v, float):
            assert got[k] == pytest.approx(v, rel=2e-6), k
        else:
            assert got[k] == v, k
    saved = json.loads((tmp_path / "scores.json").read_text())
    assert saved["num_frames"] == ref["num_frames"] and (tmp_path / f"res-{data_name}.json").exists()
    assert "HiMo refinement metrics" in capsys.readouterr().out


def test_save_zip_program_round_trip_on_gpu_box(gpu, gold, tmp_path):
    """save_zip.run_dataset -> zip_res -> read_output_zip with the package's own Feather writer/reader."""
    from conftest import golden_frames
    from himo_amd import save_zip
    from himo_amd.dataset import ListDataset
    frames = golden_frames(gold, "av2")
    out = tmp_path / "results"
    out.mkdir()
    assert save_zip.run_dataset(ListDataset(frames), "seflowpp_best", out, batch_frames=2) == len(frames)
    z = save_zip.zip_res(out, output_file=str(out / "seflowpp_best-submit.zip"))
    for i, f in enumerate(frames):
        cd = save_zip.read_output_zip(z, (f["scene_id"], str(f["timestamp"])))
        ref = gold[f"av2/{i}/ref_comp_dis"]
        assert cd.dtype == np.float32 and np.abs(cd.astype(np.float64) - ref).max() <= 1e-9


def test_overlapped_save_zip_writes_the_serial_loops_files(gpu, tmp_path):
    """run_dataset's feeder -> kernel -> drain form writes byte-identical Feather files to the plain serial loop, over ragged
    sweeps and a last short batch; a sweep without points / without the result key fails as save_zip.py:117,120 would, with
    the files of the earlier batches on disk."""
    from himo_amd import save_zip
    from himo_amd.synthetic import SyntheticDataset
    ds = SyntheticDataset(11, n_points=20_000, ragged=True)
    outs = []
    for overlap in (False, True):
        out = tmp_path / f"o{int(overlap)}"
        out.mkdir()
        assert save_zip.run_dataset(ds, "seflowpp_best", out, batch_frames=4, overlap=overlap) == 11
        outs.append(out)
    names = sorted(p.relative_to(outs[0]) for p in outs[0].rglob("*.feather"))
    assert len(names) == 11 and names == sorted(p.relative_to(outs[1]) for p in outs[1].rglob("*.feather"))
    for n in names:
        assert (outs[0] / n).read_bytes() == (outs[1] / n).read_bytes(), n

    class Broken:
        def __init__(self, bad, how):
            self.bad, self.how = bad, how

        def __len__(self):
            return len(ds)

        def __getitem__(self, i):
            f = dict(ds[i])
            if i == self.bad and self.how == "empty":
                f["lidar_dt"] = f["lidar_dt"][:0]
            if i == self.bad and self.how == "key":
                del f["seflowpp_best"]
            return f
    for how, exc in (("empty", ValueError), ("key", KeyError)):
        out = tmp_path / how
        out.mkdir()
        with pytest.raises(exc):
            save_zip.run_dataset(Broken(9, how), "seflowpp_best", out, batch_frames=4)
        assert len(list(out.rglob("*.feather"))) == 8                # batches 0 and 1 were complete


def test_mixed_sweep_kinds_in_one_batch_are_decided_per_sweep(gpu, gold, oracle):
    """score.py:270-296 tests ``gt_flow_norm is not None`` / ``pc0 is not None`` for EACH sweep: a batch that mixes sweeps
    with and without them must score every sweep as it would be scored alone (one sweep without gt_flow_norm must not
    switch the velocity filter off for its neighbours, one without pc0 must not change their Chamfer inputs)."""
    from himo_amd.score import ScoreMetrics
    frames = golden_frames(gold, "av2")
    sweeps = []
    for i, f in enumerate(frames):
        full = (gold[f"av2/{i}/ref_gt_comp_dis"], gold[f"av2/{i}/ref_comp_dis"], gold[f"av2/{i}/ref_eval_mask"],
                f["flow_category_indices"], f["flow_instance_id"].astype(np.uint32), gold[f"av2/{i}/ref_gt_flow_norm"],
                np.ascontiguousarray(f["pc0"][:, :3]))
        kind = i % 4
        sweeps.append(full if kind == 0 else full[:5] + (None, full[6]) if kind == 1 else full[:6] + (None,) if kind == 2
                      else full[:5] + (None, None))
    assert len({(s[5] is None, s[6] is None) for s in sweeps}) >= 3
    batched, ref = ScoreMetrics(), oracle.ScoreMetrics()
    batched.step_many(sweeps, data_name="av2")
    for s in sweeps:
        ref.step(s[0], s[1], s[2], gt_category=s[3], gt_instance=s[4], gt_flow_norm=s[5], pc0=s[6], data_name="av2")
    a, b = batched.compute_scores(), ref.compute_scores()
    assert b["num_instances"] > 0
    for k, v in b.items():
        if isinstance(v, float):
            assert a[k] == pytest.approx(v, rel=2e-6), k
        elif not isinstance(v, dict):
            assert a[k] == v, k
