"""End-to-end per-frame pipeline on the GPU (network flow -> comp_dis) against the CPU restatements."""
import numpy as np
from pathlib import Path
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipeline_matches_cpu_restatement(gpu, oracle):
    import seflow_oracle as so
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.seflow import spec
    from himo_amd.seflow.model import SeFlowNet
    from himo_amd.synthetic import make_frame
    params = spec.init_params(1)
    pipe = HiMoPipeline(SeFlowNet(params, device=gpu, max_points=30_000), device=gpu)
    frames = [make_frame(20 + i, n_points=n) for i, n in enumerate([20_000, 25_000, 18_000, 22_000])]
    samples = [Sample.from_frames(frames[0], frames[1], frames[2], device=gpu),
               Sample.from_frames(frames[1], frames[2], frames[3], device=gpu)]
    out = pipe.run(samples, refined=True)
    torch.cuda.synchronize()
    flows = out["batch"].split(out["flow"])
    cds = out["batch"].split(out["comp_dis"])
    for k, (fh, f0, f1) in enumerate([(frames[0], frames[1], frames[2]), (frames[1], frames[2], frames[3])]):
        ref_flow = so.forward(params, fh["pc0"], f0["pc0"], f1["pc0"], fh["pose0"], f0["pose0"], f0["pose1"])
        got_flow = flows[k].cpu().numpy()
        assert np.abs(got_flow - ref_flow).max() <= 1e-4
        # comp_dis from the GPU's own flow must equal the reference arithmetic applied to that flow (pinned stage)
        frame = dict(f0, seflowpp_best=got_flow)
        ref_cd = oracle.comp_dis_frame_f32(frame, "seflowpp_best")
        assert np.abs(cds[k].cpu().numpy().astype(np.float64) - ref_cd).max() <= 1e-9
        # and end to end against the all-CPU path
        ref_cd_cpu = oracle.comp_dis_frame_f32(dict(f0, seflowpp_best=ref_flow), "seflowpp_best")
        assert np.abs(cds[k].cpu().numpy() - ref_cd_cpu).max() <= 1e-4
    # running the same batch again reuses buffers and reproduces the result bit for bit
    first = out["comp_dis"].clone()
    again = pipe.run(samples, refined=True)
    assert torch.equal(first, again["comp_dis"])


def test_save_then_save_zip_then_eval_round_trip(gpu, oracle, tmp_path):
    """The reference's three-program flow (save.py -> save_zip.py / eval.py) on an on-disk dataset, GPU end to end.
    Feather I/O needs pyarrow, absent on the GPU box: the zip step is exercised only where pyarrow exists."""
    from himo_amd import save
    from himo_amd.dataset import NpzDataset
    from himo_amd.eval import InstanceMetrics
    from himo_amd.synthetic import make_frame
    frames = [make_frame(70 + i, n_points=12_000, scene_id="sceneA", res_name="unused") for i in range(3)]
    for f in frames:
        f.pop("unused")
    NpzDataset.write(tmp_path, frames)
    ds = NpzDataset(tmp_path)
    done = save.run(ds, "seflowpp_best", sink=save.NpzResultSink(tmp_path, "seflowpp_best"))
    assert done == 2                                          # the last sweep of the scene has no successor
    ds = NpzDataset(tmp_path)
    f0 = ds[0]
    assert f0["seflowpp_best"].shape == (12_000, 3) and f0["seflowpp_best"].dtype == np.float32
    # evaluate the stored flow exactly like eval.py does (frames that carry a result)
    m, ref = InstanceMetrics("av2"), oracle.InstanceMetrics("av2")
    scored = [dict(ds[i], pose1=frames[i + 1]["pose0"] if False else ds[i]["pose1"]) for i in range(2)]
    m.step_frames(scored, res_name="seflowpp_best")
    for f in scored:
        oracle.eval_frame(ref, f, res_name="seflowpp_best")
    assert m.frame_cnt == ref.frame_cnt == 2
    import json
    a, b = json.loads(json.dumps(m.summary(), default=float)), json.loads(json.dumps(ref.summary(), default=float))
    assert a.keys() == b.keys()
    for cat in b:
        for k in ("mpe", "cd"):
            va = a[cat]["overall"][k] if "overall" in a[cat] else a[cat][k]
            vb = b[cat]["overall"][k] if "overall" in b[cat] else b[cat][k]
            assert va == pytest.approx(vb, rel=1e-9, abs=1e-12)


def test_fp16_split_pipeline_matches_and_flags_overflow(gpu):
    """precision='f16x2' (two-term fp16 split): same 1e-4 flow parity as the bf16 split, and weights that push an
    activation past fp16's range make the pipeline raise instead of returning NaN flows."""
    import seflow_oracle as so
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.seflow import spec
    from himo_amd.seflow.model import SeFlowNet
    from himo_amd.synthetic import make_frame
    params = spec.init_params(3)
    frames = [make_frame(40 + i, n_points=16_000) for i in range(3)]
    samples = [Sample.from_frames(frames[0], frames[1], frames[2], device=gpu)]
    pipe = HiMoPipeline(SeFlowNet(params, device=gpu, max_points=20_000, precision="f16x2"), device=gpu)
    out = pipe.run(samples)
    pipe.sync_check()
    ref = so.forward(params, frames[0]["pc0"], frames[1]["pc0"], frames[2]["pc0"], frames[0]["pose0"], frames[1]["pose0"], frames[1]["pose1"])
    assert np.abs(out["flow"].cpu().numpy() - ref).max() <= 1e-4
    big = dict(params)
    big["dec3.u5.weight"] = params["dec3.u5.weight"] * 1e7          # dec4 then sees inputs ~1e7 > 65504
    pipe = HiMoPipeline(SeFlowNet(big, device=gpu, max_points=20_000, precision="f16x2"), device=gpu)
    pipe.run(samples)
    with pytest.raises(FloatingPointError):
        pipe.sync_check()
    # the bf16 split has float32's range: same weights, finite result
    pipe = HiMoPipeline(SeFlowNet(big, device=gpu, max_points=20_000, precision="bf16x3"), device=gpu)
    want = pipe.run(samples)["flow"].clone()
    assert torch.isfinite(want).all()
    # precision="auto": starts in the fp16 split, notices the overflow, rebuilds in the bf16 split and redoes the batch
    auto = HiMoPipeline(device=gpu, max_points=20_000, max_batch=2, params=big)
    assert auto.net.precision == "f16x2"
    got = auto.run(samples)["flow"]
    assert auto.net.precision == "bf16x3" and torch.equal(got, want)
    ok = HiMoPipeline(device=gpu, max_points=20_000, max_batch=2, params=params)
    ok.run(samples)
    assert ok.net.precision == "f16x2"                           # well-scaled weights stay on the fast path


def test_batched_backbone_equals_one_sample_at_a_time(gpu):
    """max_batch > 1 (every backbone layer one launch over several samples: two-level image addressing, batched upsample,
    up to 12 sweeps per pillar launch) must reproduce the per-sample path bit for bit, for full and partial groups."""
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.seflow import spec
    from himo_amd.seflow.model import SeFlowNet
    from himo_amd.synthetic import make_frame
    params = spec.init_params(2)
    frames = [make_frame(60 + i, n_points=9_000 + 700 * i) for i in range(8)]
    samples = [Sample.from_frames(frames[i], frames[i + 1], frames[i + 2], device=gpu) for i in range(6)]     # ragged sizes
    for prec in ("f16x2", "bf16x3"):
        one = HiMoPipeline(SeFlowNet(params, device=gpu, max_points=16_000, precision=prec, max_batch=1, autotune=False), device=gpu)
        ref = one.run(samples)["flow"].clone()
        many = HiMoPipeline(SeFlowNet(params, device=gpu, max_points=16_000, precision=prec, max_batch=4, autotune=False), device=gpu)
        got = many.run(samples)["flow"].clone()                 # groups of 4 + 2
        assert torch.equal(got, ref), prec
        assert torch.equal(many.run(samples[:5])["flow"], ref[: sum(s.pc0.shape[0] for s in samples[:5])]), prec      # 4 + 1


def test_feeder_and_drain_deliver_the_same_bits_as_the_serial_path(gpu):
    """feeder.SampleFeeder (pinned staging + side-stream copies, batches ahead) and feeder.ResultDrain (pinned D2H + writer
    thread) against the serial Sample.from_frames / .cpu() path: same samples, same flows, in order; a failing source or
    sink surfaces on the caller's thread."""
    from himo_amd.feeder import ResultDrain, SampleFeeder
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.synthetic import make_frame
    frames = [make_frame(90 + i, n_points=5_000 + 700 * i) for i in range(7)]
    source = [(i, frames[max(i - 1, 0)], frames[i], frames[i + 1]) for i in range(6)]
    pipe = HiMoPipeline(device=gpu, max_points=12_000, max_batch=4, precision="f16x2")
    want = {}
    for i, fh, f0, f1 in source:
        s = Sample.from_frames(fh, f0, f1, device=gpu)
        want[i] = pipe.flows([s])[0].cpu().numpy()
    got, order = {}, []
    drain = ResultDrain(lambda key, arr: (got.__setitem__(key, arr), order.append(key)), device=gpu)
    n_batches = 0
    for batch in SampleFeeder(iter(source), device=gpu, batch=4, depth=2):
        n_batches += 1
        for (i, f0, s) in batch:
            ref = Sample.from_frames(source[i][1], f0, source[i][3], device=gpu)
            assert torch.equal(s.pc0, ref.pc0) and torch.equal(s.pch1, ref.pch1) and torch.equal(s.pc1, ref.pc1)
            assert torch.equal(s.lidar_dt, ref.lidar_dt) and np.array_equal(s.pose1, ref.pose1)
        for (i, _, _), flow in zip(batch, pipe.flows([s for _, _, s in batch])):
            drain.put(i, flow)
    drain.close()
    assert n_batches == 2 and order == list(range(6))
    for i in want:
        assert np.array_equal(got[i], want[i]), i

    def bad_source():
        yield source[0]
        raise OSError("disk gone")
    with pytest.raises(OSError):
        for _ in SampleFeeder(bad_source(), device=gpu, batch=1):
            pass
    drain = ResultDrain(lambda key, arr: (_ for _ in ()).throw(ValueError("sink failed")), device=gpu)
    drain.put(0, torch.zeros(4, device=gpu))
    with pytest.raises(ValueError):
        drain.close()


def test_degenerate_sweeps_in_a_batch(gpu):
    """Sweeps with no points at all (sensor drop-out) and with a single point travel through a batch next to ordinary
    ones: every sample's result equals what it gets on its own, and empty pc0 sweeps give empty outputs."""
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.synthetic import make_frame
    pipe = HiMoPipeline(device=gpu, max_points=9_000, max_batch=4, precision="f16x2")
    full = [make_frame(120 + i, n_points=6_000 + 500 * i) for i in range(3)]
    def variant(f, n):
        g = dict(f)
        g["pc0"], g["lidar_dt"] = f["pc0"][:n].copy(), f["lidar_dt"][:n].copy()
        return g
    cases = [(full[0], full[1], full[2]),                       # ordinary
             (variant(full[0], 0), full[1], full[2]),            # empty history sweep
             (full[0], variant(full[1], 0), full[2]),            # empty pc0: nothing to compensate
             (full[0], full[1], variant(full[2], 0)),            # empty pc1
             (full[0], variant(full[1], 1), full[2])]            # a single point
    samples = [Sample.from_frames(fh, f0, f1, device=gpu) for fh, f0, f1 in cases]
    alone = []
    for s in samples:
        out = pipe.run([s])
        alone.append((out["flow"].clone(), out["comp_dis"].clone()))
    for lo in (0, 1):
        grp = samples[lo:lo + 4]
        out = pipe.run(grp)
        o = out["batch"].offsets_host
        for k in range(len(grp)):
            a, b = int(o[k]), int(o[k + 1])
            assert b - a == grp[k].pc0.shape[0]
            assert torch.equal(out["flow"][a:b], alone[lo + k][0]) and torch.equal(out["comp_dis"][a:b], alone[lo + k][1]), (lo, k)
    pipe.sync_check()
    assert alone[2][0].shape == (0, 3) and alone[4][0].shape == (1, 3) and torch.isfinite(alone[4][0]).all()


def test_streaming_temporaries_never_see_a_previous_batch(gpu, oracle):
    """A streaming caller builds its Samples on the fly and drops them after the call, so CPython hands the next Sample
    the same ``id()``: every call must still use ITS points, poses and lidar_dt (same-sized sweeps, so a stale batch
    could not be noticed by a shape check)."""
    import gc
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.synthetic import make_frame
    pipe = HiMoPipeline(device=gpu, max_points=9_000, max_batch=2, precision="f16x2")
    frames = [make_frame(300 + i, n_points=7_000) for i in range(8)]
    ids = set()
    for i in range(6):
        fh, f0, f1 = frames[i], frames[i + 1], frames[i + 2]
        out = pipe.run([Sample.from_frames(fh, f0, f1, device=gpu)], refined=True)     # the Sample dies with this statement
        ids.add(id(out["batch"]))
        gc.collect()
        flow = out["flow"].cpu().numpy()
        alone = HiMoPipeline(net=pipe.net, device=gpu).flow(Sample.from_frames(fh, f0, f1, device=gpu)).cpu().numpy()
        assert np.array_equal(flow, alone), i                          # this sample's sweeps went through the network
        ref_cd = oracle.comp_dis_frame_f32(dict(f0, seflowpp_best=flow), "seflowpp_best")
        assert np.abs(out["comp_dis"].cpu().numpy().astype(np.float64) - ref_cd).max() <= 1e-9, i    # ITS pc0 / pose / dt
        ref_rf = f0["pc0"][:, :3].astype(np.float64) + ref_cd
        assert np.abs(out["refined"].cpu().numpy() - ref_rf).max() <= 1e-5, i
    # results of batch k survive batch k+1 (two alternating buffer sets)
    a = pipe.run([Sample.from_frames(frames[0], frames[1], frames[2], device=gpu)])
    keep = a["comp_dis"].clone()
    pipe.run([Sample.from_frames(frames[3], frames[4], frames[5], device=gpu)])
    assert torch.equal(a["comp_dis"], keep)
    # refined is only handed out when asked for
    assert a["refined"] is None
    # ... and are overwritten by batch k+2 unless the caller asked for copies (the lifetime is part of run()'s contract)
    kept = [pipe.run([Sample.from_frames(frames[i], frames[i + 1], frames[i + 2], device=gpu)], copy=True) for i in range(4)]
    for i, r in enumerate(kept):
        alone = HiMoPipeline(net=pipe.net, device=gpu).flow(Sample.from_frames(frames[i], frames[i + 1], frames[i + 2], device=gpu))
        assert torch.equal(r["flow"], alone), i


def test_flows_never_returns_an_overflowed_batch(gpu):
    """``flows`` (what ``save.run`` writes under <res_name>) checks before it returns: auto falls back to the bf16 split and
    redoes the batch, an explicit f16x2 raises."""
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    params = spec.init_params(3)
    big = dict(params)
    big["dec3.u5.weight"] = params["dec3.u5.weight"] * 1e7
    frames = [make_frame(40 + i, n_points=8_000) for i in range(3)]
    samples = [Sample.from_frames(frames[0], frames[1], frames[2], device=gpu)]
    want = HiMoPipeline(device=gpu, max_points=9_000, max_batch=1, params=big, precision="bf16x3").flows(samples)[0].clone()
    assert torch.isfinite(want).all()
    auto = HiMoPipeline(device=gpu, max_points=9_000, max_batch=1, params=big)
    got = auto.flows(samples)[0]
    assert auto.net.precision == "bf16x3" and torch.equal(got, want)
    with pytest.raises(FloatingPointError):
        HiMoPipeline(device=gpu, max_points=9_000, max_batch=1, params=big, precision="f16x2").flows(samples)


def test_an_empty_batch_is_not_an_underflow(gpu):
    """once the low-side guard words of the fp16 split are registered, a call with NO samples (nothing ran between the
    clear and the read-back: every word still 0) must not read as "activations on the split's floor": explicit f16x2 does not
    raise, auto does not leave the fp16 split -- and the next real batch is still checked"""
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    frames = [make_frame(60 + i, n_points=6_000) for i in range(3)]
    samples = [Sample.from_frames(frames[0], frames[1], frames[2], device=gpu)]
    for precision in ("f16x2", "auto"):
        pipe = HiMoPipeline(device=gpu, max_points=7_000, max_batch=1, params=spec.init_params(3), precision=precision)
        first = pipe.flows(samples)[0].clone()
        assert pipe.net._range_slots                                  # the words exist now
        assert pipe.flows([]) == []
        assert pipe.net.precision == "f16x2"
        assert torch.equal(pipe.flows(samples)[0], first)
    tiny = dict(spec.init_params(3))
    tiny["enc1.0.weight"] = tiny["enc1.0.weight"] * 1e-6             # ... and a layer that really sits on the floor is still caught
    tiny["enc1.0.bias"] = tiny["enc1.0.bias"] * 1e-6
    tiny["enc1.0.bn.gamma"] = tiny["enc1.0.bn.gamma"] * 1e-6
    tiny["enc1.0.bn.beta"] = tiny["enc1.0.bn.beta"] * 1e-6
    with pytest.raises(FloatingPointError):
        HiMoPipeline(device=gpu, max_points=7_000, max_batch=1, params=tiny, precision="f16x2").flows(samples)


@pytest.mark.parametrize("n_points", [2_000, 120_000])
def test_config2_run_over_the_reference_frame_list(gpu, oracle, tmp_path, monkeypatch, n_points):
    """BASELINE config 2 (the data itself is absent): the reference's own frame lists -- 70 eval frames of 13 scenes
    (index_eval.pkl) inside their index_total.pkl neighbourhood -- filled with synthetic sweeps (small ones, and BASELINE-size
    120k-point ones), then the three programs in a row through their command-line mains: save (network flow under <res_name>
    for every frame with a successor) -> save_zip (the eval list only -> one Feather member per distinct sweep, stored zip)
    -> eval, once from the stored flow and once from the zip.  Both tables are checked against the pinned oracle's
    ``InstanceMetrics`` (eval.py:64-149 restated) fed the SAME stored flow / the SAME zip payload, sweep by sweep in the
    list's order -- not only against each other."""
    import json
    import pickle
    from pathlib import Path
    from zipfile import ZIP_STORED, ZipFile
    from himo_amd import eval as ev, save, save_zip
    from himo_amd.dataset import NpzDataset
    from himo_amd.synthetic import make_frame
    idx = json.loads((Path(__file__).resolve().parent / "golden" / "av2_index.json").read_text())
    total, evl = idx["index_total"], idx["index_eval"]
    pos = {tuple(k): i for i, k in enumerate(total)}
    keep = set()
    for s, t in evl:                                             # each eval frame with its history sweep and its successor
        i = pos[(s, t)]
        keep.update(j for j in (i - 1, i, i + 1) if 0 <= j < len(total) and total[j][0] == s)
    sub = [total[i] for i in sorted(keep)]
    root = tmp_path / "av2" / "himo"
    frames = []
    for k, (s, t) in enumerate(sub):                               # ragged: 120k, 119k, ... (2000, 2100, ... in the small run)
        f = make_frame(9000 + k, n_points=n_points - (k % 7) * (n_points // 120) if n_points > 10_000 else n_points + (k % 7) * 100,
                       scene_id=s)
        f["timestamp"] = int(t)
        f.pop("seflowpp_best")
        frames.append(f)
    NpzDataset.write(root, frames)
    del frames
    with open(root / "index_eval.pkl", "wb") as fh:
        pickle.dump(evl, fh)
    assert len(NpzDataset(root, eval=True)) == 70 and len({s for s, _ in evl}) == 13
    done = save.main(dataset_path=str(root), res_name="seflowpp_best")
    has_next = sum(1 for a, b in zip(sub[:-1], sub[1:]) if a[0] == b[0])
    assert done == has_next
    ds = NpzDataset(root, eval=True)
    assert all("seflowpp_best" in ds[i] and ds[i]["seflowpp_best"].shape == (len(ds[i]["pc0"]), 3) for i in (0, 33, 69))
    save_zip.main(str(root), "seflowpp_best", batch_frames=16)
    z = root / "results" / "seflowpp_best-submit.zip"
    with ZipFile(z) as zf:
        names = zf.namelist()
        # the reference's list names two sweeps twice (70 entries, 68 distinct): as with the reference's own save_zip.py the
        # second write replaces the first file, so the zip has 68 members -- and eval.py still walks all 70 entries
        assert len(names) == 68 and all(i.compress_type == ZIP_STORED for i in zf.infolist())
        assert set(names) == {f"{s}/{t}.feather" for s, t in evl}
    monkeypatch.chdir(tmp_path)
    direct = ev.main(str(root), res_name="seflowpp_best", batch_frames=16, file_name=str(tmp_path / "res-direct.json"))
    via_zip = ev.main(str(root), res_name="seflowpp_best", comp_dis_zip=str(z), batch_frames=16, file_name=str(tmp_path / "res-zip.json"))
    assert direct.frame_cnt == via_zip.frame_cnt == 70
    a, b = (json.loads(json.dumps(m.summary(), default=float)) for m in (direct, via_zip))
    assert "Total" in a and a["Total"]["num_obj"] > 0

    def close(x, y, rel, path=""):
        if isinstance(x, dict):
            assert x.keys() == y.keys(), (path, sorted(x), sorted(y))
            return all(close(x[k], y[k], rel, f"{path}/{k}") for k in x)
        assert x == pytest.approx(y, rel=rel, abs=1e-9, nan_ok=True), (path, x, y)
        return True
    # the zip carries comp_dis rounded to float32 (save_zip.py:70-72), the direct mode keeps the float64 chain: same table to ~1e-7
    assert close(a, b, 1e-6)
    # ... and against the pinned oracle: eval.py's loop over the same 70 entries, on the flow `save` stored / the zip `save_zip` wrote
    from himo_amd.save_zip import read_output_zip
    ref_direct, ref_zip = oracle.InstanceMetrics("av2"), oracle.InstanceMetrics("av2")
    for i in range(len(ds)):
        f = ds[i]
        oracle.eval_frame(ref_direct, f, res_name="seflowpp_best")
        oracle.eval_frame(ref_zip, f, comp_dis=read_output_zip(z, (f["scene_id"], f["timestamp"])))
    assert ref_direct.frame_cnt == ref_zip.frame_cnt == 70
    for mine, ref in ((direct, ref_direct), (via_zip, ref_zip)):
        close(json.loads(json.dumps(mine.evaluate_data, default=float)), json.loads(json.dumps(ref.evaluate_data, default=float)), 1e-9)
        close(json.loads(json.dumps(mine.summary(), default=float)), json.loads(json.dumps(ref.summary(), default=float)), 1e-9)


def test_cli_entry_points_on_a_one_rank_rccl_group(gpu, tmp_path, monkeypatch):
    """save_zip.main / eval.main with HIMO_DIST_FORCE=1: the entry points join a one-rank "nccl" (RCCL) group on cuda:0 and
    run their rendezvous (int32 min all-reduce) and the metric gather (all_gather_object) through it -- the collectives of
    the sharded runs, executed on the GPU box's RCCL."""
    import json
    import socket
    import torch.distributed as dist
    from himo_amd import eval as ev, save_zip
    from himo_amd.dataset import NpzDataset
    from himo_amd.synthetic import make_frame
    root = tmp_path / "av2" / "demo"
    frames = [make_frame(1200 + i, n_points=3_000, scene_id=f"scene{i // 3}") for i in range(6)]
    NpzDataset.write(root, frames)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for k, v in {"HIMO_DIST_FORCE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1",
                 "LOCAL_RANK": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}.items():
        monkeypatch.setenv(k, v)
    monkeypatch.chdir(tmp_path)
    seen = []
    real = dist.init_process_group
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: (seen.append(backend), real(backend, **kw))[1])
    save_zip.main(str(root), "seflowpp_best", batch_frames=4)
    assert (root / "results" / "seflowpp_best-submit.zip").exists() and not dist.is_initialized()
    m = ev.main(str(root), res_name="seflowpp_best", batch_frames=4, file_name=str(tmp_path / "res.json"))
    assert m.frame_cnt == 6 and not dist.is_initialized()
    assert seen == ["nccl", "nccl"]
    single = ev.InstanceMetrics("av2")
    single.step_frames(frames, res_name="seflowpp_best")
    assert json.dumps(m.evaluate_data, default=float, sort_keys=True) == json.dumps(single.evaluate_data, default=float, sort_keys=True)


@pytest.mark.parametrize("workload", ["pipeline", "train"])
def test_bench_two_ranks_on_this_box(gpu, workload):
    """`python bench.py --gpus 2 --share-gpu`: the driver's command line for N = 2 on a one-GPU box -- bench.py re-launches itself
    as two ranks under torch.distributed.run, both do REAL device work (on the one device; collectives over gloo because RCCL
    refuses two ranks per device), barrier, max-over-ranks time, per-rank frame counts gathered, one JSON line from rank 0.
    Everything the 8-GPU run does except RCCL itself (which the one-rank `--force-process-group` tests exercise)."""
    import json, os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--share-gpu", "--workload", workload, "--steps", "2", "--warmup", "1",
           "--points", "20000", "--no-cpu-baseline", "--no-extra-precisions"] + (["--frames-per-step", "2"] if workload == "pipeline" else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 alone prints
    line = json.loads(lines[0])
    per_step = 2 if workload == "pipeline" else 1
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["collectives"] == "gloo x2"
    assert line["config"]["frames_per_rank"] == [2 * per_step, 2 * per_step] and "shared_gpu" in line["config"]
    assert line["value"] > 0 and np.isfinite(line["value"])


def test_overlapped_gradient_exchange_ends_in_the_flat_exchange_bits_two_ranks(gpu):
    """the gradient all-reduce bucket by bucket UNDER the backward pass (seflow.train.BucketedAllReduce: head + decoder,
    encoder stages 3 + 2, stage 1 + pillar net; the sample count and loss ride with the first) against ONE flat all-reduce after it --
    two ranks with real device work on this box (gloo: RCCL refuses two ranks per device), 2 samples per rank and pass, 10+ optimiser
    steps: the same parameter bits on both paths."""
    import json, os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    bits = {}
    for overlap in ("1", "0"):
        env = dict(os.environ, HIMO_TRAIN_OVERLAP_ALLREDUCE=overlap)
        env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
        cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "train", "--train-batch", "2", "--steps", "4",
               "--warmup", "2", "--points", "20000", "--no-cpu-baseline", "--no-extra-precisions"]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert line["n_gpus"] == 2 and line["config"]["samples_per_pass_and_optimiser_step"] == 2
        assert ("bucket by bucket" in line["gradient_exchange"]) == (overlap == "1")
        bits[overlap] = line["param_bits_sum_at_end"]
    assert bits["1"] == bits["0"], bits


def test_cli_programs_as_two_ranks_on_this_box(gpu, tmp_path):
    """The reference's job shape (assets/slurm/ssl-train-av2.sh:3: one process per GPU under a launcher) for the four programs,
    with REAL device work on the one-GPU box: `python -m torch.distributed.run --nproc-per-node 2 -m himo_amd.{save, save_zip,
    eval, seflow.fit}` with HIMO_DIST_BACKEND=gloo (both ranks drive cuda:0; RCCL refuses two ranks per device).  The sharded
    runs must leave exactly what the single-process programs leave."""
    import json, os, pickle, socket, subprocess, sys
    from zipfile import ZipFile
    from himo_amd import eval as ev, save, save_zip
    from himo_amd.dataset import NpzDataset
    from himo_amd.synthetic import make_frame
    repo = Path(__file__).resolve().parents[1]

    def dataset(root):
        frames = [make_frame(4100 + i, n_points=3_000 + 100 * (i % 3), scene_id=f"scene{i // 4}") for i in range(8)]
        for i, f in enumerate(frames):
            f["timestamp"] = 1000 + i
            f.pop("seflowpp_best")
        NpzDataset.write(root, frames)
        with open(root / "index_eval.pkl", "wb") as fh:             # save_zip / eval walk the frames that have a successor
            pickle.dump([[f["scene_id"], f["timestamp"]] for i, f in enumerate(frames) if i % 4 != 3], fh)

    def torchrun(module, *args, cwd):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HIMO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=str(repo))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HIMO_DIST_FORCE"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), "-m", module, *args], env=env, cwd=cwd, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (module, out.stderr[-3000:])
        return out

    two, one = tmp_path / "two" / "av2" / "demo", tmp_path / "one" / "av2" / "demo"
    dataset(two); dataset(one)
    # single-process programs
    save.main(dataset_path=str(one), res_name="seflowpp_best")
    save_zip.main(str(one), "seflowpp_best", batch_frames=4)
    ref = ev.main(str(one), res_name="seflowpp_best", batch_frames=4, file_name=str(tmp_path / "one" / "res.json"))
    # the same three as two ranks each
    torchrun("himo_amd.save", "--dataset_path", str(two), "--res_name", "seflowpp_best", cwd=tmp_path / "two")
    a, b = NpzDataset(one), NpzDataset(two)
    assert len(a) == len(b)
    for i in range(len(a)):
        assert ("seflowpp_best" in a[i]) == ("seflowpp_best" in b[i])
        if "seflowpp_best" in a[i]:
            assert np.array_equal(a[i]["seflowpp_best"], b[i]["seflowpp_best"]), i      # frame i came from rank i % 2: same bits
    torchrun("himo_amd.save_zip", "--data_dir", str(two), "--res_name", "seflowpp_best", "--batch_frames", "4", cwd=tmp_path / "two")
    with ZipFile(one / "results" / "seflowpp_best-submit.zip") as z1, ZipFile(two / "results" / "seflowpp_best-submit.zip") as z2:
        assert sorted(z1.namelist()) == sorted(z2.namelist()) and len(z1.namelist()) > 0
        for n in z1.namelist():
            assert z1.read(n) == z2.read(n), n
    torchrun("himo_amd.eval", "--data_dir", str(two), "--res_name", "seflowpp_best", cwd=tmp_path / "two")
    got = json.loads((tmp_path / "two" / "res-av2.json").read_text())
    want = json.loads((tmp_path / "one" / "res.json").read_text())
    assert ref.frame_cnt > 0 and got == want
    # the training loop: 2 ranks x 1 sample per optimiser step, one epoch; rank 0 leaves the checkpoints
    torchrun("himo_amd.seflow.fit", "--dataset_path", str(two), "--out_dir", str(tmp_path / "ckpt"), "--epochs", "1", "--batch_size", "2",
             cwd=tmp_path / "two")
    assert list((tmp_path / "ckpt").glob("*.npz"))
    # ... and an UNEVEN split: 6 samples in steps of 5 -> shares 3 + 2, then 1 + 0: the rank without a sample sends zeros through the same
    # bucketed collectives as the rank that runs a pass (a rank choosing another exchange than its peer would hang here)
    torchrun("himo_amd.seflow.fit", "--dataset_path", str(two), "--out_dir", str(tmp_path / "ckpt5"), "--epochs", "1", "--batch_size", "5",
             cwd=tmp_path / "two")
    assert list((tmp_path / "ckpt5").glob("*.npz"))


@pytest.mark.parametrize("in_flight", [2, 3])
def test_two_batches_in_flight_deliver_the_single_stream_bits(gpu, in_flight):
    """pipeline.OverlappedPipeline (two networks' buffers, two HIP streams, batches alternate) against HiMoPipeline on the same
    stream of ragged batches: flow and comp_dis of every batch bit-identical, whatever co-runs; and ``flows_stream`` (the
    ``save`` program's path, finite-flow check of batch k under batch k + 1) against ``HiMoPipeline.flows``."""
    from himo_amd.pipeline import HiMoPipeline, OverlappedPipeline, Sample
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    params = spec.init_params(2)
    frames = [make_frame(300 + i, n_points=9_000 + 137 * (i % 5)) for i in range(12)]
    batches = [[Sample.from_frames(frames[j], frames[j + 1], frames[j + 2], device=gpu) for j in range(lo, lo + n)]
               for lo, n in ((0, 3), (3, 2), (5, 3), (8, 1), (2, 3), (6, 2))]
    single = HiMoPipeline(device=gpu, max_points=10_000, max_batch=3, params=params, precision="f16x2")
    want = []
    for b in batches:
        r = single.run(b, copy=True)
        want.append((r["flow"].clone(), r["comp_dis"].clone()))
    single.sync_check()
    two = OverlappedPipeline(params=params, device=gpu, max_points=10_000, max_batch=3, precision="f16x2", in_flight=in_flight)
    side = torch.cuda.Stream(device=gpu)
    got = []
    for b in batches:                                             # results consumed on a side stream, as feeder.ResultDrain does
        r = two.run(b)
        with torch.cuda.stream(side):
            side.wait_event(r["ready"])
            got.append((r["flow"].clone(), r["comp_dis"].clone()))
    two.sync_check()
    torch.cuda.synchronize()
    for (f0, c0), (f1, c1) in zip(want, got):
        assert torch.equal(f0, f1) and torch.equal(c0, c1)
    ref_flows = [[o.clone() for o in single.flows(b)] for b in batches]
    seen = 0
    for (samples, flows), ref in zip(two.flows_stream(iter(batches)), ref_flows):
        assert len(flows) == len(ref) == len(samples)
        for a, b in zip(flows, ref):
            assert torch.equal(a, b)
        seen += 1
    assert seen == len(batches)
    r = two.wait(two.run(batches[0]))                             # the convenience path: the current stream waits
    assert torch.equal(r["flow"], want[0][0])
