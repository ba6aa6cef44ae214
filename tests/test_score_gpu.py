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
