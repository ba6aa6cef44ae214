"""The N>1 path on CPU: two processes over gloo (127.0.0.1).  Covers what is distributed in this design --
frame sharding i % world, per-rank Feather files + rank-0 zip, the final gather of results -- without a GPU
(device arithmetic is replaced by the oracle INSIDE THESE TESTS ONLY, to exercise the host logic)."""
import json
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "oracle"))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _fake_records(frame_idx):
    """Deterministic per-instance records of one sweep (what himo_eval_instances returns)."""
    from himo_amd.eval import RECORD_DTYPE
    rng = np.random.default_rng(1000 + frame_idx)
    n = int(rng.integers(3, 9))
    r = np.zeros(n, RECORD_DTYPE)
    r["frame"] = 0
    r["group"] = rng.integers(1, 3, n)
    r["instance"] = np.arange(n) + 10
    r["num_pts"] = rng.integers(5, 400, n)
    r["vel"] = rng.uniform(0, 40, n)
    r["dis"] = rng.uniform(1, 50, n)
    r["mpe"] = rng.uniform(0, 1, n)
    r["cham"] = rng.uniform(0, 1, n)
    return np.sort(r, order=["frame", "group", "instance"])


def _worker_metrics(rank, world, port, n_frames, out_dir):
    _init(rank, world, port)
    from himo_amd.eval import InstanceMetrics
    m = InstanceMetrics("av2")
    for i in range(rank, n_frames, world):                    # frame i -> rank i % world
        m._accumulate_frame(_fake_records(i), key=i)
    m.gather()
    Path(out_dir, f"rank{rank}.json").write_text(json.dumps({"data": m.evaluate_data, "cnt": m.frame_cnt}, default=float))
    dist.destroy_process_group()


def test_metrics_gather_reproduces_single_process_exactly(tmp_path):
    sys.path.insert(0, str(REPO))
    from himo_amd.eval import InstanceMetrics
    n_frames, world = 11, 2
    mp.spawn(_worker_metrics, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    single = InstanceMetrics("av2")
    for i in range(n_frames):
        single._accumulate_frame(_fake_records(i), key=i)
    want = json.loads(json.dumps({"data": single.evaluate_data, "cnt": single.frame_cnt}, default=float))
    for rank in range(world):
        got = json.loads(Path(tmp_path, f"rank{rank}.json").read_text())
        assert got == want                                     # element for element, same order
    assert want["cnt"] == n_frames and len(want["data"]["CAR"]["mean"]["num_pts"]) > 0


def _worker_save_zip(rank, world, port, data_dir):
    _init(rank, world, port)
    import himo_oracle as oracle
    from himo_amd import compdis, save_zip
    from himo_amd.synthetic import SyntheticDataset

    class CpuEngine:                                           # test double for the device arithmetic
        def __init__(self, *a, **k):
            pass

        def run(self, batch, sensor_dt=0.1, **k):
            cds = [oracle.comp_dis_frame_f32(f, "seflowpp_best", sensor_dt) for f in batch._frames]
            return {"comp_dis": torch.from_numpy(np.concatenate(cds))}

    real_from_frames = compdis.FrameBatch.from_frames.__func__

    def from_frames(cls, frames, res_name="seflowpp_best", device=None, with_masks=False):
        b = real_from_frames(cls, frames, res_name, device=torch.device("cpu"), with_masks=with_masks)
        b._frames = list(frames)
        return b

    compdis.CompDisEngine = CpuEngine
    compdis.FrameBatch.from_frames = classmethod(from_frames)
    ds = SyntheticDataset(7, n_points=500, ragged=True)
    out = Path(data_dir) / "results"
    out.mkdir(exist_ok=True, parents=True)
    written = save_zip.run_dataset(ds, "seflowpp_best", out, batch_frames=2)
    assert written == len(range(rank, 7, world))
    dist.barrier()
    if rank == 0:
        save_zip.zip_res(out, output_file=str(out / "seflowpp_best-submit.zip"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_save_zip_writes_every_sweep_once(tmp_path):
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "oracle"))
    import himo_oracle as oracle
    from himo_amd.save_zip import read_output_zip
    from himo_amd.synthetic import SyntheticDataset
    mp.spawn(_worker_save_zip, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    z = tmp_path / "results" / "seflowpp_best-submit.zip"
    ds = SyntheticDataset(7, n_points=500, ragged=True)
    from zipfile import ZipFile
    with ZipFile(z) as zf:
        assert len(zf.namelist()) == 7
    for i in range(7):
        f = ds[i]
        cd = read_output_zip(str(z), (f["scene_id"], str(f["timestamp"])))
        assert np.array_equal(cd, oracle.comp_dis_frame_f32(f, "seflowpp_best"))


def _worker_bench(rank, world, port, out_dir):
    _init(rank, world, port)
    sys.argv = ["bench.py"]
    import bench
    elapsed, total = bench.reduce_job(0.5 + rank, 80 * (rank + 1), torch.device("cpu"), world, rank)
    Path(out_dir, f"b{rank}.json").write_text(json.dumps([elapsed, total]))
    dist.destroy_process_group()


def test_bench_reduction_is_max_time_and_total_frames(tmp_path):
    mp.spawn(_worker_bench, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    e0, t0 = json.loads(Path(tmp_path, "b0.json").read_text())
    e1, t1 = json.loads(Path(tmp_path, "b1.json").read_text())
    assert e0 == e1 == 1.5            # max over ranks
    assert t0 == 240 and t1 == 0      # whole-job frame count lands on rank 0


def _worker_grad_allreduce(rank, world, port, out_dir):
    _init(rank, world, port)
    from himo_amd.seflow.train import allreduce_mean_
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank r holds (r+1) * [0..999]
    allreduce_mean_(g)
    Path(out_dir, f"g{rank}.npy").write_bytes(g.numpy().tobytes())
    dist.destroy_process_group()


def test_training_gradient_exchange_is_one_mean_allreduce(tmp_path):
    """The data-parallel training step (config 5) exchanges exactly one flat buffer: its mean over ranks."""
    mp.spawn(_worker_grad_allreduce, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = np.arange(1000, dtype=np.float32) * 1.5
    for r in range(2):
        got = np.frombuffer(Path(tmp_path, f"g{r}.npy").read_bytes(), np.float32)
        assert np.array_equal(got, want)
