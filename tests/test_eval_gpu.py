"""GPU parity of a7/a8: exact 1-NN / Chamfer and the per-instance metrics of eval.py.

Bars: nearest-neighbour distances in float64 are bit-identical to cKDTree's; per-instance MPE /
Chamfer and every number of res-<data>.json agree with the reference's own output to 1e-9
(different summation trees: numpy's pairwise sums vs fixed GPU trees)."""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN, RES, golden_frames

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _approx_tree(got, ref, path=""):
    if isinstance(ref, dict):
        assert set(got) == set(ref), path
        for k in ref:
            _approx_tree(got[k], ref[k], f"{path}/{k}")
    elif isinstance(ref, list):
        assert len(got) == len(ref), path
        for i, (g, r) in enumerate(zip(got, ref)):
            _approx_tree(g, r, f"{path}[{i}]")
    elif isinstance(ref, float):
        assert got == pytest.approx(ref, rel=TOL, abs=TOL), path
    else:
        assert got == ref, path


@pytest.mark.parametrize("nq,nr", [(1, 1), (10, 7), (257, 300), (3000, 1), (5000, 4999), (20000, 30000)])
def test_nn_float64_is_bitwise_ckdtree(gpu, oracle, nq, nr):
    from himo_amd.eval import nearest_neighbor
    rng = np.random.default_rng(nq * 31 + nr)
    q = rng.normal(size=(nq, 3)) * 5
    r = rng.normal(size=(nr, 3)) * 5
    d, i = nearest_neighbor(q, r)
    rd, ri = oracle.nearest_neighbor(q, r)
    assert d.dtype == np.float64 and np.array_equal(d, rd)
    assert np.array_equal(np.linalg.norm(q - r[i], axis=1), np.linalg.norm(q - r[ri], axis=1))


def test_nn_float32_and_ties(gpu, oracle):
    from himo_amd.eval import nearest_neighbor
    rng = np.random.default_rng(5)
    q = rng.uniform(-50, 50, (40000, 3)).astype(np.float32)
    r = rng.uniform(-50, 50, (35000, 3)).astype(np.float32)
    d, i = nearest_neighbor(q, r)
    rd, _ = oracle.nearest_neighbor(q, r)
    assert d.dtype == np.float32 and np.abs(d - rd).max() <= 1e-5
    # duplicates: the lowest reference index wins
    r2 = np.concatenate([r[:100], r[:100]])
    d2, i2 = nearest_neighbor(r[:100], r2)
    assert not d2.any() and np.array_equal(i2, np.arange(100))


def test_nn_segments_and_empty_ranges(gpu, oracle):
    from himo_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(9)
    qn, rn = [5, 0, 700, 33], [9, 4, 0, 1200]
    q = rng.normal(size=(sum(qn), 3))
    r = rng.normal(size=(sum(rn), 3))
    qo = torch.tensor(np.concatenate([[0], np.cumsum(qn)]), dtype=torch.int64, device=gpu)
    ro = torch.tensor(np.concatenate([[0], np.cumsum(rn)]), dtype=torch.int64, device=gpu)
    qd, rd = torch.from_numpy(q).to(gpu), torch.from_numpy(r).to(gpu)
    d2 = torch.empty(len(q), dtype=torch.float64, device=gpu)
    idx = torch.empty(len(q), dtype=torch.int32, device=gpu)
    _lib.check(lib.himo_nn_search(4, qo.data_ptr(), ro.data_ptr(), len(q), len(r), qd.data_ptr(), rd.data_ptr(), 1,
                                  d2.data_ptr(), idx.data_ptr(), _lib.stream_handle()))
    d2, idx = d2.cpu().numpy(), idx.cpu().numpy()
    qs, rs = np.cumsum([0] + qn), np.cumsum([0] + rn)
    for s in range(4):
        sl = slice(qs[s], qs[s + 1])
        if rn[s] == 0:
            assert np.isinf(d2[sl]).all() and (idx[sl] == -1).all()
            continue
        ref_d, ref_i = oracle.nearest_neighbor(q[sl], r[rs[s]:rs[s + 1]])
        assert np.array_equal(np.sqrt(d2[sl]), ref_d)
        assert ((idx[sl] >= rs[s]) & (idx[sl] < rs[s + 1])).all()


def test_chamfer_known_answers(gpu, gold):
    from himo_amd.eval import InstanceMetrics
    m = InstanceMetrics("av2")
    for j in range(int(gold["chamfer/n"])):
        got = m.cal_chamfer(gold[f"chamfer/{j}/a"], gold[f"chamfer/{j}/b"])
        assert got == pytest.approx(float(gold[f"chamfer/{j}/ref"]), rel=1e-15, abs=0)
    assert np.isnan(m.cal_chamfer(np.zeros((0, 3)), np.zeros((3, 3))))


@pytest.mark.parametrize("data_name", ["av2", "scania"])
@pytest.mark.parametrize("mode", ["flow", "raw", "zip"])
def test_instance_metrics_against_reference_output(gpu, gold, eval_gold, data_name, mode):
    from himo_amd.eval import InstanceMetrics
    frames = golden_frames(gold, data_name)
    ref = eval_gold[f"{data_name}/{mode}"]
    m = InstanceMetrics(data_name)
    cds = None
    if mode == "zip":   # the payload of <data>_pred.zip as the reference's read_output_zip returned it (the GPU box
        cds = [gold[f"{data_name}/{i}/ref_comp_dis"] for i in range(len(frames))]   # has no pyarrow to re-read the zip)
    m.step_frames(frames, res_name="raw" if mode == "raw" else RES, comp_dis=cds)
    assert m.frame_cnt == ref["frame_cnt"]
    _approx_tree(json.loads(json.dumps(m.evaluate_data, default=float)), ref["evaluate_data"])
    summ = json.loads(json.dumps(m.summary(), default=float))
    for cat, entry in ref["res_json"].items():
        _approx_tree(summ[cat], entry, cat)
    # one sweep at a time == the whole batch (sweeps are independent)
    m1 = InstanceMetrics(data_name)
    for k, f in enumerate(frames):
        m1.step_frames([f], res_name="raw" if mode == "raw" else RES, comp_dis=None if cds is None else [cds[k]])
    assert json.dumps(m1.evaluate_data, default=float) == json.dumps(m.evaluate_data, default=float)


def test_step_eval_signature_matches_reference_semantics(gpu, gold, oracle):
    """The reference's own call (eval.py:303-310): masked, ego-motion-free arrays of one sweep."""
    from himo_amd.eval import InstanceMetrics
    for data_name in ("av2", "scania"):
        mine, ref = InstanceMetrics(data_name), oracle.InstanceMetrics(data_name)
        for f in golden_frames(gold, data_name):
            pf = oracle.pose_flow(f["pc0"], f["pose0"], f["pose1"])
            gt_flow, est_flow = f["flow"] - pf, f[RES] - pf
            msk = oracle.eval_mask(f, data_name)
            dt0 = oracle.dt0_from_lidar_dt(f["lidar_dt"])
            args = (f["pc0"][msk, :], gt_flow[msk, :], dt0[msk], f["flow_category_indices"][msk], f["flow_instance_id"][msk])
            mine.step_eval(*args, est_flow=est_flow[msk, :])
            ref.step_eval(*args, est_flow=est_flow[msk, :])
            cd = oracle.comp_dis_frame_f32(f, RES)
            mine.step_eval(*args, est_dis=cd[msk, :])
            ref.step_eval(*args, est_dis=cd[msk, :])
        _approx_tree(json.loads(json.dumps(mine.evaluate_data, default=float)),
                     json.loads(json.dumps(ref.evaluate_data, default=float)))
        assert mine.frame_cnt == ref.frame_cnt


def test_full_size_sweeps_against_oracle(gpu, oracle):
    from himo_amd.eval import InstanceMetrics
    from himo_amd.synthetic import make_frame
    frames = [make_frame(40 + i, n_points=120_000) for i in range(2)]
    mine, ref = InstanceMetrics("av2"), oracle.InstanceMetrics("av2")
    mine.step_frames(frames, res_name=RES)
    for f in frames:
        oracle.eval_frame(ref, f, res_name=RES)
    _approx_tree(json.loads(json.dumps(mine.evaluate_data, default=float)),
                 json.loads(json.dumps(ref.evaluate_data, default=float)))
    assert sum(len(v["num_pts"]) for v in mine.evaluate_data["CAR"]["vel"].values()) > 5


def test_many_instances_in_any_point_order_against_oracle(gpu, oracle):
    """The counting sort by (group, instance) (evalmetrics.hip rank_frames_kernel): points of an instance scattered through the sweep
    (the fixtures keep them in runs), a few hundred instances per sweep (the hash-table path, several tiles per frame) and more
    distinct instances than the table admits (the all-pairs path of that frame) -- in one batch, against the numpy + cKDTree oracle."""
    from himo_amd.eval import InstanceMetrics
    from himo_amd.synthetic import make_frame
    frames = []
    for seed, n_inst in ((60, 400), (61, 3000), (62, 30)):
        f = make_frame(seed, n_points=120_000, n_instances=n_inst)
        order = np.random.default_rng(seed).permutation(len(f["pc0"]))
        for k in ("pc0", "lidar_dt", "lidar_id", "gm0", "flow", "flow_is_valid", "flow_category_indices", "flow_instance_id", RES):
            f[k] = np.ascontiguousarray(f[k][order])
        frames.append(f)
    in_range = np.linalg.norm(frames[1]["pc0"][:, :2], axis=1) < 30.0
    assert len(np.unique(frames[1]["flow_instance_id"][in_range])) > 1600          # (the table admits 1536 distinct labels)
    mine, ref = InstanceMetrics("av2"), oracle.InstanceMetrics("av2")
    mine.step_frames(frames, res_name=RES)
    for f in frames:
        oracle.eval_frame(ref, f, res_name=RES)
    _approx_tree(json.loads(json.dumps(mine.evaluate_data, default=float)),
                 json.loads(json.dumps(ref.evaluate_data, default=float)))
    assert sum(len(v["num_pts"]) for v in mine.evaluate_data["CAR"]["vel"].values()) > 500


def test_no_evaluated_points_is_a_noop(gpu):
    from himo_amd.eval import InstanceMetrics
    from himo_amd.synthetic import make_frame
    f = make_frame(7, n_points=2000, n_instances=0)
    m = InstanceMetrics("av2")
    m.step_frames([f], res_name=RES)
    assert m.frame_cnt == 1 and m.summary() == {}


def test_step_batch_pipeline_and_feeder_equal_the_serial_path(gpu, frames_av2, frames_scania):
    """Device-resident batches with the records digested ``depth`` batches late (step_batch / flush), fed by
    feeder.EvalFeeder (pinned staging, side-stream copies), must reproduce step_frames element for element -- including a
    record buffer that is too small on the first try (the batch is run again with room for all)."""
    import json
    from himo_amd.eval import EvalBatch, InstanceMetrics
    from himo_amd.feeder import EvalFeeder
    for data_name, frames in (("av2", frames_av2), ("scania", frames_scania)):
        want = InstanceMetrics(data_name)
        for f in frames:
            want.step_frames([f], res_name="seflowpp_best")
        groups = [frames[i:i + 3] for i in range(0, len(frames), 3)]
        got = InstanceMetrics(data_name)
        for eb in EvalFeeder(iter(groups), res_name="seflowpp_best", device=gpu, depth=2):
            got.step_batch(eb, depth=2)
        assert got.frame_cnt < len(frames)                       # still in flight: nothing was waited for
        got.flush()
        tiny = InstanceMetrics(data_name)                         # record buffer of ONE entry: every batch overflows and reruns
        ev = tiny.evaluator
        launch = ev.launch
        ev.launch = lambda *a, **k: launch(*a, **{**k, "max_records": k.get("max_records") or 1})
        for g in groups:
            tiny.step_batch(EvalBatch.from_frames(g, "seflowpp_best", device=gpu))
        tiny.flush()
        dump = lambda m: json.dumps(m.evaluate_data, default=float, sort_keys=True)
        assert got.frame_cnt == tiny.frame_cnt == want.frame_cnt == len(frames)
        assert dump(got) == dump(want) and dump(tiny) == dump(want)
        assert [k for k, _ in got._log] == list(range(len(frames)))
