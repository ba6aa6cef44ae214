"""The h5 boundary end to end on the GPU box: the reference's three command lines
(``save.py`` -> ``save_zip.py`` -> ``eval.py``, README.md:46-70) over REAL HDF5 scene files, with the tables checked against
the pinned oracle's ``InstanceMetrics`` (eval.py:64-149 restated) fed the same stored flow / the same zip payload.

* the committed fixture (tests/golden/h5: written by libhdf5 1.10.6 with the datasets / dtypes of extract_sca.py:76-93),
  once under an ``av2`` path and once under a ``scania`` path (the path selects the evaluation rules, utils/__init__.py:9-24);
* BASELINE config 3's shape: 10 Scania scenes of 120k-point sweeps as h5 scene files, the three programs as TWO ranks under
  ``torch.distributed.run`` (whole scenes per rank: one writer per file), default ego box, ``flow_is_valid`` holes.
"""
import json
import os
import pickle
import shutil
import socket
import subprocess
import sys
from pathlib import Path
from zipfile import ZIP_STORED, ZipFile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parents[1]
H5 = REPO / "tests" / "golden" / "h5"


def _close(x, y, rel, path=""):
    if isinstance(x, dict):
        assert x.keys() == y.keys(), (path, sorted(x), sorted(y))
        return all(_close(x[k], y[k], rel, f"{path}/{k}") for k in x)
    assert x == pytest.approx(y, rel=rel, abs=1e-9, nan_ok=True), (path, x, y)
    return True


def _plain(obj):
    return json.loads(json.dumps(obj, default=float))


def _check_against_oracle(oracle, data_name, root, res_name, direct, via_zip, z, n_frames):
    from himo_amd.dataset import open_dataset
    from himo_amd.save_zip import read_output_zip
    ds = open_dataset(root, vis_name=res_name, eval=True)
    assert len(ds) == n_frames
    ref_direct, ref_zip = oracle.InstanceMetrics(data_name), oracle.InstanceMetrics(data_name)
    for i in range(len(ds)):
        f = ds[i]
        assert f[res_name].shape == (len(f["pc0"]), 3) and f[res_name].dtype == np.float32
        oracle.eval_frame(ref_direct, f, res_name=res_name)
        oracle.eval_frame(ref_zip, f, comp_dis=read_output_zip(z, (f["scene_id"], f["timestamp"])))
    assert ref_direct.frame_cnt == n_frames
    for mine, ref in ((direct, ref_direct), (via_zip, ref_zip)):
        _close(_plain(mine["evaluate_data"]), _plain(ref.evaluate_data), 1e-9)
        _close(_plain(mine["summary"]), _plain(ref.summary()), 1e-9)
    return ref_direct


@pytest.mark.parametrize("data_name,res_name", [("av2", "seflowpp_best"), ("scania", "seflowpp_r4")])
def test_h5_fixture_through_the_three_programs(gpu, oracle, tmp_path, monkeypatch, data_name, res_name):
    """save.main -> save_zip.main -> eval.main on the libhdf5-written fixture directory.  ``seflowpp_best`` already exists in
    the files (the av2 run REPLACES it, as re-running a checkpoint does); ``seflowpp_r4`` is a new dataset name."""
    from himo_amd import eval as ev, h5lite, save, save_zip
    from himo_amd.dataset import HDF5Dataset, open_dataset, result_file
    root = tmp_path / data_name / "himo"
    shutil.copytree(H5, root)
    before = open_dataset(root, vis_name="seflowpp_best")[0]["seflowpp_best"].copy()
    done = save.main(dataset_path=str(root), res_name=res_name)
    assert done == 6                                                  # 2 scenes x 4 sweeps, the last of each has no successor
    how = save.h5_writer()[1]
    ds = open_dataset(root, vis_name=res_name)
    assert isinstance(ds, HDF5Dataset) and len(ds) == 6
    flow0 = ds[0][res_name]
    assert flow0.shape == before.shape and np.isfinite(flow0).all() and not np.array_equal(flow0, before)
    with h5lite.File(root / "scene-00.h5") as f:                      # where the result went: into the scene file when a library exists
        inside = res_name in f[sorted(f.keys())[0]]
    assert inside == (how != "no HDF5 library") or res_name == "seflowpp_best"
    if how == "no HDF5 library":
        assert result_file(root, res_name, "scene-00").exists()
    save_zip.main(str(root), res_name, batch_frames=4)
    z = root / "results" / f"{res_name}-submit.zip"
    with ZipFile(z) as zf:
        assert len(zf.namelist()) == 4 and all(i.compress_type == ZIP_STORED for i in zf.infolist())
    monkeypatch.chdir(tmp_path)
    direct = ev.main(str(root), res_name=res_name, batch_frames=4, file_name=str(tmp_path / "d.json"))
    via_zip = ev.main(str(root), res_name=res_name, comp_dis_zip=str(z), batch_frames=4, file_name=str(tmp_path / "z.json"))
    assert direct.frame_cnt == via_zip.frame_cnt == 4 and direct.data_name == data_name
    # the programs above were fed by reader threads (short evaluations, and this process holds device memory); forked reader
    # PROCESSES -- what a long evaluation started from a shell gets -- give the same lists
    assert direct.loop["reader_processes"] == 0 and direct.loop["reader_threads"] == 4
    procs = ev.main(str(root), res_name=res_name, batch_frames=4, file_name=str(tmp_path / "p.json"), num_workers=2)
    procs_zip = ev.main(str(root), res_name=res_name, comp_dis_zip=str(z), batch_frames=4, file_name=str(tmp_path / "pz.json"), num_workers=2)
    assert procs.loop["reader_processes"] == 2 and procs.feed_stats["batches"] == 1 and procs.feed_stats["restarts"] == 0
    assert procs.evaluate_data == direct.evaluate_data and procs_zip.evaluate_data == via_zip.evaluate_data
    assert [k for k, _ in procs._log] == [k for k, _ in direct._log]
    ref = _check_against_oracle(oracle, data_name, root, res_name,
                                {"evaluate_data": direct.evaluate_data, "summary": direct.summary()},
                                {"evaluate_data": via_zip.evaluate_data, "summary": via_zip.summary()}, z, 4)
    assert ref.summary()["Total"]["num_obj"] > 0


def test_reader_processes_on_the_device_errors_and_slots(gpu, tmp_path):
    """feeder.ProcessBatchFeeder on the device, forked from a process that already holds device state: the same result lists as the
    reader threads; a batch larger than a slot is an error that names the size to ask for (no second generation of workers is
    forked from the feeder thread); a reader's error arrives where the serial loop would meet it."""
    import warnings
    from himo_amd import eval as ev
    from himo_amd.dataset import EVAL_FIELDS, open_dataset
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = open_dataset(H5, vis_name="seflowpp_best", eval=True, fields=EVAL_FIELDS + ("seflowpp_best",), zero_copy=True, allow_dropped_eval=True)
    assert ds.fork_safe and len(ds) >= 4
    key_lists = [list(range(lo, min(lo + 2, len(ds)))) for lo in range(0, len(ds), 2)]
    read = lambda k: ([ds[i] for i in key_lists[k]], None)
    want = ev.InstanceMetrics("av2")
    ev.stream_batches(want, ((key_lists[k],) + read(k) for k in range(len(key_lists))), "seflowpp_best")
    got = ev.InstanceMetrics("av2")
    ev.stream_batches_from_processes(got, key_lists, read, "seflowpp_best", workers=2, slot_bytes=1 << 20)
    assert got.feed_stats["restarts"] == 0 and got.feed_stats["batches"] == len(key_lists)
    assert got.evaluate_data == want.evaluate_data and got.frame_cnt == want.frame_cnt == len(ds)
    assert [k for k, _ in got._log] == [k for k, _ in want._log]
    small = ev.InstanceMetrics("av2")
    with pytest.raises(RuntimeError, match="slot_bytes >= "):
        ev.stream_batches_from_processes(small, key_lists, read, "seflowpp_best", workers=2, slot_bytes=4096)
    assert small.frame_cnt == 0 and small.feed_stats["restarts"] == 0

    def failing(k):
        if k == 1:
            raise KeyError("seflowpp_missing")
        return read(k)
    partial = ev.InstanceMetrics("av2")
    with pytest.raises(KeyError, match="seflowpp_missing"):
        ev.stream_batches_from_processes(partial, key_lists, failing, "seflowpp_best", workers=2)
    assert partial.frame_cnt + sum(t[1] for t in partial._pending) == len(key_lists[0])        # the batch before the failing one was scored


def _torchrun(module, *args, cwd, nproc=2):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HIMO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=str(REPO))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HIMO_DIST_FORCE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), "-m", module, *args], env=env, cwd=cwd, capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0, (module, out.stderr[-3000:])
    return out


def test_config3_scania_h5_programs_as_two_ranks(gpu, oracle, tmp_path):
    """BASELINE config 3 ("Scania val (10 scenes), SeFlow++ inference sharded over the GPUs") with the data synthetic: 10 scenes
    x 3 sweeps of 120k points as h5 scene files on a ``.../scania/val`` path, the three programs as two ranks sharing this
    box's GPU (gloo collectives: RCCL refuses two ranks per device).  What is Scania-specific and now exercised at full size
    through the programs: the default ego box (utils/__init__.py:26), ``min_vel`` 1.5 (eval.py:32-35), ``flow_is_valid`` in
    the mask (eval.py:293-294).  The table the two ranks leave must equal the oracle's over the same 20 sweeps."""
    from himo_amd import h5c, h5lite
    from himo_amd.dataset import open_dataset
    from himo_amd.synthetic import make_frame
    root = tmp_path / "scania" / "val"
    root.mkdir(parents=True)
    index, in_box, holes = [], 0, 0
    for s in range(10):
        pose, groups = np.eye(4), {}
        for k in range(3):
            f = make_frame(7000 + 3 * s + k, n_points=120_000 - 1_000 * ((s + k) % 5), scene_id=f"scania-{s:02d}", data_name="scania")
            groups[str(f["timestamp"])] = {
                "lidar": f["pc0"], "lidar_id": f["lidar_id"], "lidar_dt": f["lidar_dt"], "pose": pose.copy(), "flow": f["flow"],
                "flow_is_valid": f["flow_is_valid"], "flow_category_indices": f["flow_category_indices"],
                "flow_instance_id": f["flow_instance_id"], "ground_mask": f["gm0"], "timestamp": np.int64(f["timestamp"])}
            pose = pose @ f["pose1"]
            index.append([f["scene_id"], str(f["timestamp"])])
            p = f["pc0"]
            in_box += int(((p[:, 0] > -9.5) & (p[:, 0] < 5) & (p[:, 1] > -1.5) & (p[:, 1] < 1.380002) & (p[:, 2] > 0) & (p[:, 2] < 5)).sum())
            holes += int((~f["flow_is_valid"]).sum())
        if h5c.available() and s % 2 == 0:                            # half the scenes by the real library, half by h5lite
            with h5c.File(root / f"scania-{s:02d}.h5", "w") as h:
                for ts, arrays in groups.items():
                    g = h.create_group(ts)
                    for name, a in arrays.items():
                        g.create_dataset(name, data=a)
        else:
            h5lite.write_file(root / f"scania-{s:02d}.h5", groups)
    assert in_box > 1000 and holes > 10_000                           # the Scania-specific mask terms have something to remove
    with open(root / "index_total.pkl", "wb") as fh:
        pickle.dump(index, fh)
    with open(root / "index_eval.pkl", "wb") as fh:
        pickle.dump([e for i, e in enumerate(index) if i % 3 != 2], fh)
    res = "seflowpp_best"
    _torchrun("himo_amd.save", "--dataset_path", str(root), "--res_name", res, cwd=tmp_path)
    ds = open_dataset(root, vis_name=res, eval=True)
    assert len(ds) == 20 and all(res in ds[i] for i in (0, 7, 19))
    _torchrun("himo_amd.save_zip", "--data_dir", str(root), "--res_name", res, "--batch_frames", "8", cwd=tmp_path)
    z = root / "results" / f"{res}-submit.zip"
    with ZipFile(z) as zf:
        assert len(zf.namelist()) == 20 and all(i.compress_type == ZIP_STORED for i in zf.infolist())
    _torchrun("himo_amd.eval", "--data_dir", str(root), "--res_name", res, cwd=tmp_path)
    direct = json.loads((tmp_path / "res-scania.json").read_text())
    (tmp_path / "res-scania.json").rename(tmp_path / "res-scania-direct.json")
    # (the second evaluation is fed by two forked reader processes per rank -- forked inside the process group, before the ranks start
    # the HIP runtime; the workers read the scene files AND the zip members)
    out = _torchrun("himo_amd.eval", "--data_dir", str(root), "--res_name", res, "--comp_dis_zip", str(z), "--num_workers", "2", cwd=tmp_path)
    assert "2 reader processes" in out.stdout
    via_zip = json.loads((tmp_path / "res-scania.json").read_text())
    from himo_amd.save_zip import read_output_zip
    ref_direct, ref_zip = oracle.InstanceMetrics("scania"), oracle.InstanceMetrics("scania")
    for i in range(len(ds)):
        f = ds[i]
        oracle.eval_frame(ref_direct, f, res_name=res)
        oracle.eval_frame(ref_zip, f, comp_dis=read_output_zip(z, (f["scene_id"], f["timestamp"])))
    assert ref_direct.frame_cnt == 20 and ref_direct.summary()["Total"]["num_obj"] > 0
    for mine, ref in ((direct, ref_direct), (via_zip, ref_zip)):       # the file holds {data: {res: {CAR: ..., OTHER_VEHICLES: ...}}}
        want = {k: v for k, v in _plain(ref.summary()).items() if k in ("CAR", "OTHER_VEHICLES")}
        assert want and _close(mine["scania"][res], want, 1e-9)
