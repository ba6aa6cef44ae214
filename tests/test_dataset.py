"""Frame sources and result sinks at the h5 boundary: index handling on the reference's own frame
lists, the loader's key renaming and successor logic, the ``<res_name>`` result dataset.  The index / successor logic runs
through the ``opener`` hook on an in-memory scene mapping; the file format itself is exercised on REAL HDF5 files: the
committed fixtures the HDF5 library wrote (tests/golden/h5), files written here through libhdf5 (``h5c``) where it loads,
and the result files ``H5ResultSink`` writes with ``h5lite`` where nothing can modify a scene file in place."""
import json
import pickle
from contextlib import contextmanager
from pathlib import Path

import numpy as np
import pytest

from himo_amd import save
from himo_amd.dataset import HDF5Dataset, NpzDataset, load_index, open_dataset
from himo_amd.synthetic import make_frame

GOLDEN = Path(__file__).resolve().parent / "golden"


def _scene_groups(frames):
    """frames -> {scene: {timestamp: {h5 dataset name: array}}} with the on-disk names / dtypes of
    dataprocess/extract_sca.py:76-93 and tools/test/repack_h5_scania.py:23-36"""
    scenes = {}
    for f in frames:
        g = {"lidar": f["pc0"], "lidar_dt": f["lidar_dt"], "lidar_id": f["lidar_id"], "pose": f["pose0"],
             "ground_mask": f["gm0"], "flow": f["flow"], "flow_is_valid": f["flow_is_valid"],
             "flow_category_indices": f["flow_category_indices"], "flow_instance_id": f["flow_instance_id"]}
        scenes.setdefault(f["scene_id"], {})[str(f["timestamp"])] = g
    return scenes


def _memory_opener(scenes, log=None):
    @contextmanager
    def opener(path):
        if log is not None:
            log.append(Path(path).name)
        yield scenes[Path(path).stem]
    return opener


def _dataset_dir(tmp_path, frames, eval_subset=None):
    index = [[f["scene_id"], str(f["timestamp"])] for f in frames]
    with open(tmp_path / "index_total.pkl", "wb") as fh:
        pickle.dump(index, fh)
    if eval_subset is not None:
        with open(tmp_path / "index_eval.pkl", "wb") as fh:
            pickle.dump([index[i] for i in eval_subset], fh)
    return index


def test_reference_frame_lists_load_and_every_eval_frame_has_a_successor(tmp_path):
    """The reference's own index files (BASELINE config 2's frame list): 70 eval frames of 13 scenes, all of which have a
    next sweep in index_total -- so dropping successor-less frames loses none of them."""
    idx = json.loads((GOLDEN / "av2_index.json").read_text())
    for name in ("index_eval", "index_total"):
        with open(tmp_path / f"{name}.pkl", "wb") as fh:
            pickle.dump(idx[name], fh)
    ev, total = load_index(tmp_path, eval=True), load_index(tmp_path)
    assert len(ev) == 70 and len({s for s, _ in ev}) == 13 and len(total) == 2040
    assert all(isinstance(s, str) and isinstance(t, str) for s, t in ev)
    ds = HDF5Dataset(tmp_path, vis_name="seflowpp_best", eval=True, opener=_memory_opener({}))
    assert len(ds) == 70 and ds.index == ev and ds.dropped == []     # every entry of the shipped eval list survives the loader
    full = HDF5Dataset(tmp_path, opener=_memory_opener({}))
    assert len(full) == 2040 - 13                                  # the last sweep of each scene cannot be compensated
    nxt = {(s, t): t1 for (s, t), (s1, t1) in zip(total[:-1], total[1:]) if s == s1}
    assert all(int(nxt[(s, t)]) > int(t) for s, t in ev)


def test_hdf5_dataset_logic_through_the_opener_hook(tmp_path):
    frames = [make_frame(i, n_points=50 + i, scene_id=f"scene{i // 3}") for i in range(6)]     # two scenes of three sweeps
    flow_name = "seflowpp_best"
    scenes = _scene_groups(frames)
    for f in frames:
        scenes[f["scene_id"]][str(f["timestamp"])][flow_name] = f[flow_name]
    _dataset_dir(tmp_path, frames, eval_subset=[0, 2, 4])
    log = []
    ds = HDF5Dataset(tmp_path, vis_name=flow_name, opener=_memory_opener(scenes, log))
    assert len(ds) == 4                                            # frames 2 and 5 end their scenes
    d = ds[1]
    assert log == ["scene0.h5"]
    assert d["scene_id"] == "scene0" and d["timestamp"] == frames[1]["timestamp"] and isinstance(d["timestamp"], int)
    assert np.array_equal(d["pc0"], frames[1]["pc0"]) and np.array_equal(d["pose0"], frames[1]["pose0"])
    assert np.array_equal(d["pose1"], frames[2]["pose0"]) and np.array_equal(d["pc1"], frames[2]["pc0"])   # successor's pose / points
    assert d["gm0"].dtype == bool and np.array_equal(d["gm0"], frames[1]["gm0"]) and "ground_mask" not in d   # the loader's rename
    assert np.array_equal(d[flow_name], frames[1][flow_name]) and d["lidar_dt"].dtype == np.float32
    for k in ("flow", "flow_is_valid", "flow_category_indices", "flow_instance_id", "lidar_id"):
        assert np.array_equal(d[k], frames[1][k]), k
    # frame 2 of the eval list has no successor: a sweep the leaderboard expects would silently go missing -> loud by default
    with pytest.raises(KeyError, match="pose1"):
        HDF5Dataset(tmp_path, vis_name=flow_name, eval=True, opener=_memory_opener(scenes))
    with pytest.warns(UserWarning, match="no successor"):
        ev = HDF5Dataset(tmp_path, vis_name=flow_name, eval=True, opener=_memory_opener(scenes), allow_dropped_eval=True)
    assert [t for _, t in ev.index] == [str(frames[0]["timestamp"]), str(frames[4]["timestamp"])]
    assert ev.dropped == [("scene0", str(frames[2]["timestamp"]))]
    assert len(ds.dropped) == 2                                     # the scene ends of index_total.pkl: noted, not an error
    raw = HDF5Dataset(tmp_path, vis_name="raw", opener=_memory_opener(scenes))[0]
    assert flow_name not in raw
    # the frames it yields are exactly what the comp_dis path consumes
    from himo_amd.compdis import FrameBatch
    import torch
    b = FrameBatch.from_frames([ds[0], ds[1]], flow_name, device=torch.device("cpu"), with_masks=True)
    assert b.total_points == 50 + 51


def test_training_triplets_use_the_successor_a_frame_carries(tmp_path):
    """the last usable sweep of every h5 scene was lost to the training loop, because its successor is not in
    the index although the frame itself carries pc1 (and now the successor's cluster labels)."""
    from himo_amd.seflow.fit import triplets
    frames = [make_frame(i, n_points=40 + i, scene_id=f"scene{i // 3}") for i in range(6)]
    _dataset_dir(tmp_path, frames)
    ds = HDF5Dataset(tmp_path, opener=_memory_opener(_scene_groups(frames)))
    assert triplets(ds) == [(0, 0, None), (0, 1, None), (2, 2, None), (2, 3, None)]      # 4 usable sweeps, all of them
    d = ds[1]
    assert np.array_equal(d["flow_instance_id_next"], frames[2]["flow_instance_id"]) and len(d["pc1"]) == len(frames[2]["pc0"])
    from himo_amd.dataset import ListDataset
    assert triplets(ListDataset(frames)) == [(0, 0, 1), (0, 1, 2), (3, 3, 4), (3, 4, 5)]


def test_h5_result_sink_writes_res_name_per_timestamp_and_only_after_the_scene(tmp_path):
    frames = [make_frame(i, n_points=40, scene_id=f"scene{i // 3}") for i in range(6)]
    scenes = _scene_groups(frames)

    class Group(dict):                                            # the three h5py.Group calls the sink makes
        def create_dataset(self, name, data):
            self[name] = np.array(data)

    store = {s: {ts: Group(g) for ts, g in gs.items()} for s, gs in scenes.items()}
    opened = []

    @contextmanager
    def opener(path):
        opened.append(Path(path).name)
        yield store[Path(path).stem]

    sink = save.H5ResultSink(tmp_path, "seflowpp_best", opener=opener)
    flows = [np.full((40, 3), i, np.float64) for i in range(6)]
    for i in (0, 1):
        sink(i, frames[i], flows[i])
    assert opened == []                                            # scene0 may still be open for reading
    sink(3, frames[3], flows[3])
    assert opened == ["scene0.h5"]
    sink.close()
    assert opened == ["scene0.h5", "scene1.h5"]
    for i in (0, 1, 3):
        got = store[frames[i]["scene_id"]][str(frames[i]["timestamp"])]["seflowpp_best"]
        assert got.dtype == np.float32 and got.shape == (40, 3) and (got == i).all()
    assert "seflowpp_best" not in store["scene0"][str(frames[2]["timestamp"])]
    sink(0, frames[0], flows[5])                                   # re-running a checkpoint replaces its dataset
    sink.close()
    assert (store["scene0"][str(frames[0]["timestamp"])]["seflowpp_best"] == 5).all()
    with pytest.raises(ValueError):
        sink(1, frames[1], np.zeros((39, 3)))                      # not row-aligned with pc0 (score.py:583)


def test_frame_source_shards_whole_scenes_for_h5(tmp_path):
    frames = [make_frame(i, n_points=30, scene_id=f"scene{i // 3}") for i in range(9)]
    _dataset_dir(tmp_path, frames)
    ds = HDF5Dataset(tmp_path, opener=_memory_opener(_scene_groups(frames)))
    seen = []
    for rank in range(2):
        mine = [(f0["scene_id"], i) for i, _, f0, _ in save.frame_source(ds, rank, 2, by_scene=True)]
        assert {s for s, _ in mine} == ({"scene0", "scene2"} if rank == 0 else {"scene1"})     # one writer per scene file
        seen += [i for _, i in mine]
    assert sorted(seen) == list(range(len(ds)))
    i, fh, f0, f1 = next(iter(save.frame_source(ds, 0, 1)))
    assert f1 is None and "pc1" in f0 and fh is f0 or fh["scene_id"] == f0["scene_id"]


def test_open_dataset_picks_the_container(tmp_path):
    frames = [make_frame(i, n_points=30) for i in range(3)]
    NpzDataset.write(tmp_path, frames)
    assert isinstance(open_dataset(tmp_path, vis_name="x", eval=True), NpzDataset)


def _write_scene_files(directory, frames, writer):
    for scene, groups in _scene_groups(frames).items():
        if writer == "h5lite":
            from himo_amd import h5lite
            h5lite.write_file(directory / f"{scene}.h5", groups)
        else:
            with writer.File(directory / f"{scene}.h5", "w") as f:
                for ts, arrays in groups.items():
                    g = f.create_group(ts)
                    for name, a in arrays.items():
                        g.create_dataset(name, data=a)


def _round_trip(tmp_path, writer, force_beside=False, monkeypatch=None):
    """the extract_sca.py:76-93 schema on disk -> HDF5Dataset -> H5ResultSink -> read back as <res_name>"""
    frames = [make_frame(i, n_points=64 + i, scene_id=f"scene{i // 3}") for i in range(6)]
    _write_scene_files(tmp_path, frames, writer)
    _dataset_dir(tmp_path, frames)
    ds = HDF5Dataset(tmp_path, vis_name="seflowpp_best")
    assert len(ds) == 4 and np.array_equal(ds[0]["pc0"], frames[0]["pc0"]) and np.array_equal(ds[0]["pose1"], frames[1]["pose0"])
    assert ds[0]["gm0"].dtype == bool and np.array_equal(ds[0]["gm0"], frames[0]["gm0"])
    assert ds[0]["flow_is_valid"].dtype == bool and ds[0]["flow_instance_id"].dtype == np.uint32 and ds[0]["pose0"].dtype == np.float64
    assert "seflowpp_best" not in ds[0]
    if force_beside:
        monkeypatch.setattr(save, "h5_writer", lambda: (None, "no HDF5 library"))
    sink = save.H5ResultSink(tmp_path, "seflowpp_best")
    for i in range(len(ds)):
        f0 = ds[i]
        sink(i, f0, np.full((len(f0["pc0"]), 3), i, np.float32))
    if force_beside:
        with pytest.warns(UserWarning, match="results_h5"):
            sink.close()
        assert [p.name for p in sink.side_files] == ["scene0.h5", "scene1.h5"]
    else:
        sink.close()
        assert sink.side_files == []
    again = HDF5Dataset(tmp_path, vis_name="seflowpp_best")
    for i in range(len(again)):
        assert (again[i]["seflowpp_best"] == i).all() and again[i]["seflowpp_best"].dtype == np.float32
        assert again[i]["seflowpp_best"].shape == (len(again[i]["pc0"]), 3)
    sink = save.H5ResultSink(tmp_path, "seflowpp_best")               # a re-run replaces, never duplicates
    sink(0, ds[0], np.full((len(ds[0]["pc0"]), 3), 7, np.float32))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sink.close()
    again = HDF5Dataset(tmp_path, vis_name="seflowpp_best")
    assert (again[0]["seflowpp_best"] == 7).all() and (again[1]["seflowpp_best"] == 1).all()
    return frames


def test_real_h5_files_round_trip_through_libhdf5(tmp_path):
    """Scene files written AND modified by the real HDF5 library (ctypes), read by h5lite (or h5py where it exists)."""
    from himo_amd import h5c
    if not h5c.available():
        pytest.skip("no HDF5 C library on this box")
    _round_trip(tmp_path, h5c)
    from himo_amd import h5lite
    with h5lite.File(tmp_path / "scene0.h5") as f:                     # the result really is inside the scene file
        assert all("seflowpp_best" in f[ts] for ts in sorted(f.keys())[:2])


def test_real_h5_files_round_trip_without_any_hdf5_library(tmp_path, monkeypatch):
    """No h5py, no libhdf5: results go to <dir>/results_h5/<res_name>/<scene>.h5 (real HDF5, written by h5lite), the sink says
    so, and the loader reads them from there."""
    _round_trip(tmp_path, "h5lite", force_beside=True, monkeypatch=monkeypatch)
    from himo_amd.dataset import result_file
    assert result_file(tmp_path, "seflowpp_best", "scene0").exists()
    assert not any(p.name.endswith(".writing") for p in (tmp_path / "results_h5" / "seflowpp_best").iterdir())


def test_result_beside_the_scene_file_is_the_newer_one_until_superseded(tmp_path, monkeypatch):
    """A scene file that already holds ``seflowpp_best`` + a library-less re-run: the loader must hand out the re-run's result,
    and a later in-place write must retire the file beside the scene."""
    import shutil
    import warnings
    from himo_amd import h5c
    from himo_amd.dataset import result_file
    shutil.copytree(GOLDEN / "h5", tmp_path / "d")
    root = tmp_path / "d"
    ds = HDF5Dataset(root, vis_name="seflowpp_best")
    old = ds[0]["seflowpp_best"].copy()
    real_writer = save.h5_writer
    monkeypatch.setattr(save, "h5_writer", lambda: (None, "no HDF5 library"))
    sink = save.H5ResultSink(root, "seflowpp_best")
    sink(0, ds[0], np.full_like(old, 3.0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sink.close()
    assert (HDF5Dataset(root, vis_name="seflowpp_best")[0]["seflowpp_best"] == 3).all()
    assert np.array_equal(HDF5Dataset(root, vis_name="seflowpp_best")[1]["seflowpp_best"], ds[1]["seflowpp_best"])   # untouched sweep: in-file
    if not h5c.available():
        return
    monkeypatch.setattr(save, "h5_writer", real_writer)
    sink = save.H5ResultSink(root, "seflowpp_best")
    sink(0, ds[0], np.full_like(old, 4.0))
    sink.close()
    assert (HDF5Dataset(root, vis_name="seflowpp_best")[0]["seflowpp_best"] == 4).all()
    assert not result_file(root, "seflowpp_best", ds[0]["scene_id"]).exists()


def test_fixture_directory_opens_as_the_reference_call_does(tmp_path):
    """``HDF5Dataset(data_dir, vis_name=res_name, eval=True)[i]`` (save_zip.py:111-113, eval.py:279-282) over the committed
    libhdf5-written fixture: every key the two consumers read, with the dtypes extract_sca.py:76-93 put on disk."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_h5_fixture", GOLDEN / "make_h5_fixture.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    frames = mod.frames()
    ds = open_dataset(GOLDEN / "h5", vis_name="seflowpp_best", eval=True)
    assert isinstance(ds, HDF5Dataset) and len(ds) == 4
    for d, i in zip((ds[k] for k in range(4)), (0, 2, 4, 5)):
        f = frames[i]
        assert d["scene_id"] == f["scene_id"] and d["timestamp"] == f["timestamp"]
        for k in ("pc0", "pose0", "pose1", "lidar_dt", "gm0", "flow", "flow_is_valid", "flow_category_indices",
                  "flow_instance_id", "seflowpp_best", "lidar_id"):
            assert np.array_equal(d[k], f[k]) and d[k].dtype == f[k].dtype, k
        assert np.array_equal(d["pc1"], frames[i + 1]["pc0"])
    total = open_dataset(GOLDEN / "h5", vis_name="raw")
    assert len(total) == 6 and "seflowpp_best" not in total[0]


def test_fields_and_views_read_only_what_inference_needs_and_equal_the_full_read():
    """``fields=SAVE_FIELDS, zero_copy=True`` (what ``save.main`` opens): the five keys, as views of the file mapping where the
    libhdf5-written fixture stores plain runs, equal to the copying read; nothing else is touched"""
    from himo_amd.dataset import SAVE_FIELDS
    full = HDF5Dataset(GOLDEN / "h5", vis_name="seflowpp_best")
    lean = HDF5Dataset(GOLDEN / "h5", vis_name="seflowpp_best", fields=SAVE_FIELDS, zero_copy=True)
    for i in range(len(full)):
        a, b = full[i], lean[i]
        assert set(b) == set(SAVE_FIELDS) | {"scene_id", "timestamp"}
        for k in SAVE_FIELDS:
            assert np.array_equal(a[k], b[k]) and a[k].dtype == b[k].dtype, k
    assert not lean[0]["pc0"].flags.writeable and lean[0]["pc0"].base is not None      # a view of the mapping, not a copy
    view = lean[0]["pc0"]
    lean.close()                                                        # the view outlives the dataset's handle (the mapping goes with it)
    assert np.array_equal(view, full[0]["pc0"])
    with_result = HDF5Dataset(GOLDEN / "h5", vis_name="seflowpp_best", fields=SAVE_FIELDS + ("seflowpp_best",))
    assert np.array_equal(with_result[0]["seflowpp_best"], full[0]["seflowpp_best"])


def test_scene_files_stay_open_between_items(tmp_path):
    frames = [make_frame(i, n_points=40, scene_id=f"scene{i // 4}") for i in range(12)]     # three scenes of four sweeps
    _dataset_dir(tmp_path, frames)
    log = []
    ds = HDF5Dataset(tmp_path, opener=_memory_opener(_scene_groups(frames), log), keep_open=2)
    for i in range(len(ds)):
        ds[i]
    assert log == ["scene0.h5", "scene1.h5", "scene2.h5"]               # one open per scene, not one per item
    ds[0]                                                               # scene0 was evicted (two stay open): opened again
    assert log[-1] == "scene0.h5" and len(log) == 4
    ds.forget("scene0")
    ds[0]
    assert len(log) == 5


def test_newer_scene_file_wins_over_a_stale_result_file_beside_it(tmp_path, monkeypatch):
    """a result file left beside a scene by a library-less run must not shadow a ``<res_name>`` that
    another tool wrote INTO the scene file later.  ADVICE r05 (low): ... and a scene file that was merely touched (or given another
    result name) afterwards must not bring its OLD ``<res_name>`` back: the side writer stamps what the scene held, per sweep."""
    import os
    import shutil
    import warnings
    from himo_amd import h5lite
    from himo_amd.dataset import result_file
    shutil.copytree(GOLDEN / "h5", tmp_path / "d")
    root = tmp_path / "d"
    ds = HDF5Dataset(root, vis_name="seflowpp_best")
    in_file = ds[0]["seflowpp_best"].copy()
    scene = root / f"{ds[0]['scene_id']}.h5"
    monkeypatch.setattr(save, "h5_writer", lambda: (None, "no HDF5 library"))
    sink = save.H5ResultSink(root, "seflowpp_best")
    sink(0, ds[0], np.full_like(in_file, 3.0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sink.close()
    ds.close()
    side = result_file(root, "seflowpp_best", ds[0]["scene_id"])
    assert (HDF5Dataset(root, vis_name="seflowpp_best")[0]["seflowpp_best"] == 3).all()         # the side file is the newer one
    later = side.stat().st_mtime_ns + 5_000_000_000
    os.utime(scene, ns=(later, later))                                  # touched: the in-scene array is still the one the result replaced
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        touched = HDF5Dataset(root, vis_name="seflowpp_best")
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                  # (no "modified after the result file" warning)
        assert (touched[0]["seflowpp_best"] == 3).all()
    with h5lite.File(scene) as f:                                       # "another tool wrote <res_name> into the scene file afterwards"
        tree = {ts: {k: f[ts][k].read() for k in f[ts].keys()} for ts in f.keys()}
    ts0 = str(ds[0]["timestamp"])
    tree[ts0]["seflowpp_best"] = np.full_like(in_file, 7.0)
    h5lite.write_file(scene, tree)
    later += 5_000_000_000
    os.utime(scene, ns=(later, later))
    with pytest.warns(UserWarning, match="modified after the result file"):
        got = HDF5Dataset(root, vis_name="seflowpp_best")[0]["seflowpp_best"]
    assert (got == 7).all()
    # a side file from before the stamps existed: the scene file's copy wins, as it did
    with h5lite.File(side) as f:
        old = {ts: {k: f[ts][k][:] for k in f[ts].keys() if not k.endswith("@scene")} for ts in f.keys()}
    h5lite.write_file(side, old)
    os.utime(side, ns=(later - 1_000_000_000, later - 1_000_000_000))
    with pytest.warns(UserWarning, match="modified after the result file"):
        got = HDF5Dataset(root, vis_name="seflowpp_best")[0]["seflowpp_best"]
    assert (got == 7).all()


def test_prefetching_frame_cache_returns_the_same_frames_in_order(tmp_path):
    frames = [make_frame(i, n_points=30 + i, scene_id=f"scene{i // 5}") for i in range(15)]
    _dataset_dir(tmp_path, frames)
    reads = []

    class Counting(HDF5Dataset):
        def __getitem__(self, i):
            reads.append(i)
            return super().__getitem__(i)
    ds = Counting(tmp_path, opener=_memory_opener(_scene_groups(frames)))
    plain = [(i, fh["timestamp"], f0["timestamp"]) for i, fh, f0, f1 in save.frame_source(ds, readers=0)]
    n_plain, reads[:] = len(reads), []
    ahead = [(i, fh["timestamp"], f0["timestamp"]) for i, fh, f0, f1 in save.frame_source(ds, readers=3)]
    assert ahead == plain and len(plain) == len(ds)
    assert sorted(set(reads)) == list(range(len(ds))) and len(reads) == len(ds) == n_plain     # every frame read exactly once either way
    for readers in (0, 2):
        reads[:] = []
        mine = [i for i, *_ in save.frame_source(ds, rank=1, world=3, by_scene=True, readers=readers)]
        assert mine == [i for i, (s, _) in enumerate(ds.index) if s == "scene1"]
        # a rank never reads a frame of a scene it does not own -- not as the history of its scene's first sweep, not as
        # the successor of its last, not ahead of the walk (the owner may be rewriting that file in place)
        assert sorted(set(reads)) == mine and len(reads) == len(mine), (readers, reads)


def test_training_host_sample_reads_only_its_fields_and_names_what_is_missing(tmp_path):
    """the host half of a training sample (``seflow.fit.host_sample``, shared by ``make_sample`` and ``feeder.TrainFeeder``): the
    history sweep is read with ``fields=("pc0", "pose0")`` only, generated labels need both ground masks, stored labels both label
    arrays -- a missing one is a KeyError that names it"""
    from himo_amd.seflow.fit import host_sample, train_fields, triplets
    frames = [make_frame(i, n_points=40 + i, scene_id="scene0") for i in range(4)]
    _dataset_dir(tmp_path, frames)
    groups = _scene_groups(frames)
    ds = HDF5Dataset(tmp_path, opener=_memory_opener(groups), fields=train_fields("seflow_auto"))
    assert set(ds[1]) == {"pc0", "pose0", "pose1", "pc1", "gm0", "gm1", "scene_id", "timestamp"}
    assert set(ds.read(0, ("pc0", "pose0"))) == {"pc0", "pose0", "scene_id", "timestamp"}
    h = host_sample(ds, triplets(ds)[1], "seflow_auto")
    assert np.array_equal(h["pch1"], frames[0]["pc0"]) and np.array_equal(h["pc0"], frames[1]["pc0"]) and np.array_equal(h["pc1"], frames[2]["pc0"])
    assert np.array_equal(h["gm0"], frames[1]["gm0"]) and np.array_equal(h["gm1"], frames[2]["gm0"]) and "lab0" not in h
    assert np.array_equal(h["pose_h1"], frames[0]["pose0"]) and np.array_equal(h["pose1"], frames[2]["pose0"])
    with pytest.raises(KeyError, match="flow_instance_id"):
        host_sample(ds, triplets(ds)[1], "flow_instance_id")                    # the dataset was opened without the label fields
    lab = HDF5Dataset(tmp_path, opener=_memory_opener(groups), fields=train_fields("flow_instance_id"))
    h = host_sample(lab, triplets(lab)[0], "flow_instance_id")
    assert np.array_equal(h["lab0"], frames[0]["flow_instance_id"]) and np.array_equal(h["lab1"], frames[1]["flow_instance_id"])
    assert h["pch1"] is h["pc0"]                                                # a scene's first sweep is its own history
    nogm = HDF5Dataset(tmp_path, opener=_memory_opener(groups), fields=("pc0", "pose0", "pose1", "pc1"))
    with pytest.raises(KeyError, match="gm0 / gm1"):
        host_sample(nogm, triplets(nogm)[0], "seflow_auto")
