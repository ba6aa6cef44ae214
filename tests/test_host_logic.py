"""Host-side logic that needs no GPU: dataset sniffing, batch packing, synthetic frames, datasets."""
import numpy as np
import pytest
import torch

from himo_amd import utils
from himo_amd.compdis import FrameBatch
from himo_amd.dataset import ListDataset, NpzDataset
from himo_amd.synthetic import SyntheticDataset, make_frame


def test_check_valid_quirks(tmp_path, capsys):
    assert utils.check_valid("/data/scania/val", "x") == ("scania", 2)
    assert utils.check_valid("/data/Scania/val", "x") == ("scania", 2)
    assert utils.check_valid("/data/av2/h5", "x")[0] == "av2"
    assert utils.check_valid("/data/AV2/h5", "x")[0] == "av2"
    with pytest.raises(ValueError):
        utils.check_valid("/data/nuscenes", "x")
    with pytest.raises(ValueError):                     # name at position 0 does not count (find() > 0)
        utils.check_valid("av2/h5", "x")
    z = tmp_path / "a.zip"
    z.write_bytes(b"")
    assert utils.check_valid("/data/av2", "x", str(z)) == ("av2", 1)
    assert utils.check_valid("/data/av2", "x", str(tmp_path / "missing.zip")) == ("av2", 2)
    assert "comp_dis_zip" in capsys.readouterr().out


def test_synthetic_frame_contract():
    f = make_frame(3, n_points=5000)
    assert f["pc0"].shape == (5000, 4) and f["pc0"].dtype == np.float32
    assert f["pose0"].dtype == np.float64 and f["pose1"].shape == (4, 4)
    assert f["lidar_dt"].dtype == np.float32 and 0 <= f["lidar_dt"].min() and f["lidar_dt"].max() <= 0.1
    assert f["flow"].dtype == np.float32 and f["seflowpp_best"].shape == (5000, 3)
    assert f["flow_category_indices"].dtype == np.uint8 and f["flow_instance_id"].dtype == np.uint32
    assert f["gm0"].dtype == bool and f["flow_is_valid"].dtype == bool
    g = make_frame(3, n_points=5000)
    assert all(np.array_equal(f[k], g[k]) for k in f if isinstance(f[k], np.ndarray))   # seeded
    assert len(np.unique(f["flow_instance_id"])) > 5


def test_frame_batch_packing_ragged():
    frames = [make_frame(i, n_points=n) for i, n in enumerate([100, 1, 257, 64])]
    b = FrameBatch.from_frames(frames, "seflowpp_best", device=torch.device("cpu"), with_masks=True)
    assert b.n_frames == 4 and b.total_points == 422
    assert b.offsets_host.tolist() == [0, 100, 101, 358, 422]
    assert b.pc0.shape == (422, 4) and b.flow.shape == (422, 3) and b.gm0.dtype == torch.uint8
    assert not b.f32_chain
    parts = b.split(b.lidar_dt)
    assert [len(p) for p in parts] == [100, 1, 257, 64]
    assert np.array_equal(parts[2].numpy(), frames[2]["lidar_dt"])
    raw = FrameBatch.from_frames(frames, "raw", device=torch.device("cpu"))
    assert raw.flow is None
    with pytest.raises(KeyError):
        FrameBatch.from_frames(frames, "no_such_result", device=torch.device("cpu"))
    f32 = [dict(f, pose0=f["pose0"].astype(np.float32), pose1=f["pose1"].astype(np.float32)) for f in frames]
    assert FrameBatch.from_frames(f32, "raw", device=torch.device("cpu")).f32_chain


def test_datasets_roundtrip(tmp_path):
    frames = [make_frame(i, n_points=300) for i in range(5)]
    NpzDataset.write(tmp_path, frames, eval_subset=[1, 3])
    full = NpzDataset(tmp_path)
    ev = NpzDataset(tmp_path, eval=True)
    assert len(full) == 5 and len(ev) == 2
    assert ev[1]["timestamp"] == frames[3]["timestamp"] and ev[1]["scene_id"] == frames[3]["scene_id"]
    assert np.array_equal(full[2]["pc0"], frames[2]["pc0"])
    assert len(ListDataset(frames)) == 5
    ds = SyntheticDataset(4, n_points=1000, ragged=True)
    assert len(ds) == 4 and len({len(ds[i]["pc0"]) for i in range(4)}) > 1
    with pytest.raises(IndexError):
        ds[4]


def test_hdf5_dataset_says_what_is_missing(tmp_path):
    """No h5py needed any more (h5lite reads the scene files): what can be missing is the index, or a scene file -- both by name."""
    import pickle
    from himo_amd.dataset import HDF5Dataset
    with pytest.raises(FileNotFoundError, match="index_total.pkl"):
        HDF5Dataset(tmp_path)
    with open(tmp_path / "index_total.pkl", "wb") as fh:
        pickle.dump([["sceneA", "100"], ["sceneA", "200"]], fh)
    ds = HDF5Dataset(tmp_path)
    assert len(ds) == 1
    with pytest.raises((FileNotFoundError, OSError), match="sceneA.h5"):
        ds[0]


def test_bench_cli_contract(monkeypatch):
    """bench.py's command line: --gpus/--steps/--warmup exist, no flags means N=1 and minutes-scale defaults per workload."""
    import sys
    import bench
    for argv, steps, warmup, frames in (([], 100, 3, 16), (["--workload", "compdis"], 50, 5, 256), (["--workload", "train"], 10, 2, 1),
                                        (["--gpus", "8", "--steps", "7", "--warmup", "1"], 7, 1, 16)):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        a = bench.parse_args()
        assert (a.steps, a.warmup, a.frames_per_step) == (steps, warmup, frames), argv
        assert a.points == 120_000 and a.precision == "f16x2"
    assert a.gpus == 8


def test_save_program_takes_the_reference_style_overrides():
    """`python save.py model=fastnsf dataset_path=...` / `checkpoint=... dataset_path=...` (README.md:46-53, hydra overrides) and the
    --flag form reach the same ``save.main``; an optimisation-based model this build does not have is refused by name before any
    device work."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    for argv in (["model=nsfp", "dataset_path=/nonexistent"], ["--model", "nsfp", "--dataset_path", "/nonexistent"]):
        r = subprocess.run([sys.executable, "-m", "himo_amd.save", *argv], capture_output=True, text=True, cwd=root)
        assert r.returncode != 0 and "model='nsfp'" in r.stderr and "fastnsf" in r.stderr, r.stderr[-400:]


def test_evaluator_bucket_statistics_have_numpys_bits():
    """the evaluator's per-sweep bucket means (eval.py:113-141: np.average / np.nanmean / np.nanstd on lists of a handful of floats)
    are formed without numpy's per-call dispatch for short NaN-free lists: the same bits as numpy, over 60k random lists of every
    length that takes the short path and the lengths / NaNs that fall back to numpy"""
    from himo_amd.eval import InstanceMetrics as M
    rng = np.random.default_rng(0)
    for trial in range(60_000):
        n = int(rng.integers(1, 11))
        vals = [float(x) for x in rng.uniform(0, 3, n) * 10.0 ** rng.integers(-3, 2)]
        if trial % 50 == 0:
            vals[int(rng.integers(0, n))] = float("nan")
        w = [int(x) for x in rng.integers(10, 5000, n)]
        a, b = M._average(vals, w), np.average(vals, weights=w)
        assert a == b or (a != a and b != b), (vals, w, a, b)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got, want = M._nanmean_nanstd([np.float64(v) for v in vals]), (float(np.nanmean(vals)), float(np.nanstd(vals)))
        assert got == want or all(g != g and x != x or g == x for g, x in zip(got, want)), (vals, got, want)


def test_the_host_knows_a_steps_sample_count_without_a_process_group():
    """seflow.train.global_count_known: one process -> the local count decides (no device read-back at the end of a step); a step
    without any sample is an error either way"""
    from himo_amd.seflow.train import global_count_known
    assert global_count_known(8, None) is True and global_count_known(3, 99) is True      # (no group: the caller's figure is ignored)
    with pytest.raises(ValueError, match="at least one sample"):
        global_count_known(0, None)
    with pytest.raises(ValueError, match="at least one sample"):
        global_count_known(0, 8)


def test_when_the_evaluator_forks_reader_processes(monkeypatch):
    """eval.reader_processes_default: 4 processes for a long evaluation of a dataset that survives a fork in a process without device
    memory; reader threads otherwise; the environment overrides"""
    from himo_amd import eval as ev

    class Ds:
        fork_safe = True
    monkeypatch.delenv("HIMO_EVAL_WORKERS", raising=False)
    assert ev.reader_processes_default(5000, Ds()) == 4
    assert ev.reader_processes_default(1023, Ds()) == 0                      # too short to repay the start
    assert ev.reader_processes_default(5000, object()) == 0                  # (an h5py-backed or in-memory dataset)
    monkeypatch.setenv("HIMO_EVAL_WORKERS", "2")
    assert ev.reader_processes_default(10, object()) == 2                    # main() still checks fork_safe before it forks
    monkeypatch.setenv("HIMO_EVAL_WORKERS", "0")
    assert ev.reader_processes_default(5000, Ds()) == 0
    monkeypatch.delenv("HIMO_EVAL_WORKERS")
    monkeypatch.setattr(ev.torch.cuda, "is_initialized", lambda: True)
    monkeypatch.setattr(ev.torch.cuda, "memory_reserved", lambda *a: 1 << 20)
    assert ev.reader_processes_default(5000, Ds()) == 0                      # a process that already holds device memory: a fork is paid for
    monkeypatch.setattr(ev.torch.cuda, "memory_reserved", lambda *a: 0)
    assert ev.reader_processes_default(5000, Ds()) == 4
