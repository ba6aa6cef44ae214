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
    save_zip.OVERLAP = False                                   # the feeder / drain threads need HIP streams and pinned memory
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
    elapsed, total, per_rank = bench.reduce_job(0.5 + rank, 80 * (rank + 1), torch.device("cpu"), world, rank)
    Path(out_dir, f"b{rank}.json").write_text(json.dumps([elapsed, total, per_rank]))
    dist.destroy_process_group()


def test_bench_reduction_is_max_time_and_total_frames(tmp_path):
    mp.spawn(_worker_bench, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    e0, t0, p0 = json.loads(Path(tmp_path, "b0.json").read_text())
    e1, t1, p1 = json.loads(Path(tmp_path, "b1.json").read_text())
    assert e0 == e1 == 1.5            # max over ranks
    assert t0 == t1 == 240 and p0 == p1 == [80, 160]      # whole-job frame count + what every rank did


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


# ---- bench.py --gpus N: the self-launch path -------------------------------------------------------------------------
def _run_bench(argv, env_extra=None, timeout=240):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("workload", ["pipeline", "train"])
def test_bench_gpus_n_self_launches_n_ranks(workload):
    """``python bench.py --gpus 2`` outside torchrun must start TWO ranks (the driver's SCALE command line is exactly this),
    take the max-over-ranks time and report n_gpus: 2 with both ranks' frame counts.  --dry-run-cpu swaps only the device
    work (host stand-in; gloo for RCCL): launch, barrier, timing, gather and the JSON line are the real code.  For the
    train workload the stand-in step runs the real gradient exchange (allreduce_mean_) and checks its result."""
    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", workload, "--dry-run-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    line = json.loads(lines[0])
    per_step = 16 if workload == "pipeline" else 1
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["frames_per_rank"] == [4 * per_step, 4 * per_step]
    assert line["value"] == pytest.approx(2 * 4 * per_step / (line["ms_per_step"] * 4e-3), rel=1e-6)
    assert "INVALID" in line["data"] and "cpu_baseline" not in line       # a dry run can never pass for a measurement


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    r = _run_bench(["--gpus", "2", "--dry-run-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr
    r = _run_bench(["--gpus", "2"])                              # no HIP devices here: fails loudly instead of measuring N=1
    assert r.returncode != 0 and "HIP device" in r.stderr


# ---- the command-line entry points under torchrun-style environments ---------------------------------------------
def _cli_env(rank, world, port):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "oracle"))


def _cpu_compdis_double():
    """test double for the device arithmetic of save_zip.run_dataset (the oracle computes; host logic is the real one)"""
    import himo_oracle as oracle
    from himo_amd import compdis

    class CpuEngine:
        def __init__(self, *a, **k):
            pass

        def run(self, batch, sensor_dt=0.1, **k):
            cds = [oracle.comp_dis_frame_f32(f, "seflowpp_best", sensor_dt) for f in batch._frames]
            return {"comp_dis": torch.from_numpy(np.concatenate(cds))}

    real = compdis.FrameBatch.from_frames.__func__

    def from_frames(cls, frames, res_name="seflowpp_best", device=None, with_masks=False):
        b = real(cls, frames, res_name, device=torch.device("cpu"), with_masks=with_masks)
        b._frames = list(frames)
        return b

    compdis.CompDisEngine = CpuEngine
    compdis.FrameBatch.from_frames = classmethod(from_frames)
    from himo_amd import save_zip
    save_zip.OVERLAP = False                                   # the feeder / drain threads need HIP streams and pinned memory


def _worker_save_zip_cli(rank, world, port, data_dir, fail_rank):
    _cli_env(rank, world, port)
    _cpu_compdis_double()
    from himo_amd import save_zip
    if rank == fail_rank:
        real = save_zip.write_output_file

        def broken(*a, **k):
            raise OSError("disk full on this rank")
        save_zip.write_output_file = broken
    try:
        save_zip._cli(["--data_dir", data_dir, "--res_name", "seflowpp_best", "--batch_frames", "2"])
        Path(data_dir, f"ok{rank}").write_text("done")
    except BaseException as e:
        Path(data_dir, f"err{rank}").write_text(type(e).__name__)
    assert not dist.is_initialized()                             # the entry point left the group it joined


def _write_npz_dataset(root, n=7):
    from himo_amd.dataset import NpzDataset
    from himo_amd.synthetic import make_frame
    frames = [make_frame(i, n_points=400 + 13 * i, scene_id=f"scene{i // 4}") for i in range(n)]
    NpzDataset.write(root, frames)
    return frames


def test_save_zip_cli_under_torchrun_env_shards_and_zips_once(tmp_path):
    """``torchrun -m himo_amd.save_zip``: the entry point itself joins the process group (nothing else does), every rank
    writes ITS sweeps only, rank 0 zips after the rendezvous."""
    sys.path.insert(0, str(REPO / "oracle"))
    import himo_oracle as oracle
    from himo_amd.save_zip import read_output_zip
    root = tmp_path / "av2" / "demo"
    frames = _write_npz_dataset(root)
    mp.spawn(_worker_save_zip_cli, args=(2, _free_port(), str(root), -1), nprocs=2, join=True)
    assert (root / "ok0").exists() and (root / "ok1").exists()
    z = root / "results" / "seflowpp_best-submit.zip"
    from zipfile import ZipFile
    with ZipFile(z) as zf:
        assert len(zf.namelist()) == len(frames)
    for f in frames:
        cd = read_output_zip(str(z), (f["scene_id"], str(f["timestamp"])))
        assert np.array_equal(cd, oracle.comp_dis_frame_f32(f, "seflowpp_best"))


def test_save_zip_cli_a_failing_rank_stops_everyone_and_no_zip_is_written(tmp_path):
    root = tmp_path / "av2" / "demo"
    _write_npz_dataset(root)
    mp.spawn(_worker_save_zip_cli, args=(2, _free_port(), str(root), 1), nprocs=2, join=True)     # returns: nobody hangs
    assert (root / "err1").read_text() == "OSError" and (root / "err0").read_text() == "RuntimeError"
    assert not (root / "results" / "seflowpp_best-submit.zip").exists()


def _worker_eval_cli(rank, world, port, data_dir, out_dir):
    _cli_env(rank, world, port)
    import himo_oracle as oracle
    from himo_amd import eval as ev

    def stream_batches(metrics, source, res_name=""):                        # device double: the oracle scores the sweep
        for keys, frames, cds in source:
            for f, key in zip(frames, keys):
                ref = oracle.InstanceMetrics(metrics.data_name)
                oracle.eval_frame(ref, f, res_name=res_name)
                metrics._log.append((key, None))
                metrics.frame_cnt += 1
    ev.stream_batches = stream_batches
    ev.InstanceMetrics.flush = lambda self: None
    ev.InstanceMetrics._accumulate_records = lambda self, recs: setattr(self, "frame_cnt", self.frame_cnt + 1)
    os.chdir(out_dir)
    m = ev.main(data_dir, res_name="seflowpp_best", batch_frames=2, file_name=str(Path(out_dir) / f"res-rank{rank}.json"), num_workers=0)
    Path(out_dir, f"cnt{rank}").write_text(f"{m.frame_cnt} {sorted(k for k, _ in m._log)}")
    assert not dist.is_initialized()


def test_eval_cli_under_torchrun_env_shards_the_sweeps(tmp_path):
    root = tmp_path / "av2" / "demo"
    _write_npz_dataset(root)
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_worker_eval_cli, args=(2, _free_port(), str(root), str(out)), nprocs=2, join=True)
    for r in range(2):                                           # after the gather every rank holds all 7 sweeps, once each
        assert (out / f"cnt{r}").read_text() == "7 [0, 1, 2, 3, 4, 5, 6]"


def _worker_combine(rank, world, port, out_dir):
    _init(rank, world, port)
    from himo_amd.seflow.train import combine_batch_, global_count_known
    # under a process group only the caller can know the step's global sample count on the host (fit does); a rank without samples is fine
    assert global_count_known(2, None) is False and global_count_known(0, 3) is True and global_count_known(2, 3) is True
    try:
        global_count_known(0, 0)
        raise AssertionError("a step without any sample must be refused")
    except ValueError:
        pass
    # a partial last batch of 3 samples on 2 ranks: rank 0 holds samples 0 and 2, rank 1 holds sample 1
    grads = [torch.tensor([1.0, 2.0, 3.0]), torch.tensor([10.0, 20.0, 30.0]), torch.tensor([100.0, 200.0, 300.0])]
    losses = [1.0, 2.0, 6.0]
    out = {}
    for name, split in (("2+1", [[0, 2], [1]]), ("3+0", [[0, 1, 2], []])):
        mine = split[rank]
        acc = torch.zeros(5)
        for j in mine:
            acc[:3] += grads[j]; acc[3] += 1.0; acc[4] += losses[j]
        g = torch.zeros(3)
        loss = combine_batch_(acc.clone(), g)
        g2 = torch.zeros(3)
        loss2 = combine_batch_(acc, g2, count_known=True)       # (the host has checked the count: no read-back; the same numbers)
        assert torch.equal(g, g2) and float(loss) == float(loss2)
        out[name] = {"g": g.tolist(), "loss": float(loss)}
    Path(out_dir, f"rank{rank}.json").write_text(json.dumps(out))
    dist.destroy_process_group()


def test_train_batch_exchange_weighs_every_sample_of_the_global_batch_alike(tmp_path):
    """'each rank averages its own samples and the ranks are then averaged' gave the samples of an uneven split
    unequal weight, and an empty rank re-used a sample.  Now: gradient sums + counts all-reduced, divided once."""
    port = _free_port()
    mp.spawn(_worker_combine, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want_g, want_loss = [37.0, 74.0, 111.0], 3.0
    for r in range(2):
        got = json.loads(Path(tmp_path, f"rank{r}.json").read_text())
        for name in ("2+1", "3+0"):
            assert got[name]["g"] == pytest.approx(want_g) and got[name]["loss"] == pytest.approx(want_loss), (r, name)


def _worker_skewed_rendezvous(rank, world, port, out_dir):
    _cli_env(rank, world, port)
    os.environ["HIMO_DIST_TIMEOUT_S"] = "5"                      # the data collectives' clock: far shorter than the skew below
    import time
    from himo_amd import distenv
    with distenv.process_group() as (r, w):
        if r == 1:
            time.sleep(12)                                       # the slow shard (whole scenes per rank, slow h5 reads)
        distenv.rendezvous(None)
        got = [None] * w
        dist.all_gather_object(got, r)                           # the gather that follows the rendezvous still works
    Path(out_dir, f"ok{rank}").write_text(str(got))


def test_rendezvous_waits_for_the_slow_rank_longer_than_the_collective_timeout(tmp_path):
    """the rendezvous completes when the SLOWEST rank is done, so it must not share the short timeout that
    guards the data collectives -- a rank that finishes early waits (HIMO_RENDEZVOUS_TIMEOUT_S, hours by default)."""
    mp.spawn(_worker_skewed_rendezvous, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").read_text() == (tmp_path / "ok1").read_text() == "[0, 1]"


def _worker_buckets(rank, world, port, out_dir):
    _init(rank, world, port)
    from himo_amd.seflow.train import BucketedAllReduce, allreduce_sum_, bucket_ranges
    names = ["pfn.weight", "enc1.0.weight", "enc1.0.bias", "enc2.0.weight", "enc3.0.weight", "dec1.u1.weight", "dec4.weight", "head.offset.weight",
             "head.zr.weight", "pfn.bn.gamma", "enc1.0.bn.gamma", "enc2.0.bn.gamma", "enc3.0.bn.beta"]
    sizes = [288, 18432, 64, 73728, 294912, 196608, 36864, 192, 49152, 32, 64, 128, 256]
    offsets, o = {}, 0
    for k, n in zip(names, sizes):
        offsets[k] = (o, o + n); o += n
    buckets = bucket_ranges(names, offsets)
    ok = buckets == [[(offsets["dec1.u1.weight"][0], offsets["head.zr.weight"][1])],
                     [(offsets["enc2.0.weight"][0], offsets["enc3.0.weight"][1]), (offsets["enc2.0.bn.gamma"][0], offsets["enc3.0.bn.beta"][1])],
                     [(0, offsets["enc1.0.bias"][1]), (offsets["pfn.bn.gamma"][0], offsets["enc1.0.bn.gamma"][1])]]
    g = torch.Generator(); g.manual_seed(100 + rank)
    equal = True
    for step in range(10):
        grad = torch.randn(o, generator=g) * 10.0 ** float(torch.randint(-6, 3, (1,), generator=g))
        words = torch.tensor([float(1 + rank + step % 2), float(torch.rand(1, generator=g))], dtype=torch.float32)
        if rank == 1 and step % 4 == 3:                             # a rank without a sample in a partial last batch: zeros, same collectives
            grad, words = torch.zeros(o), torch.zeros(2)
        flat = torch.cat([grad, words])
        allreduce_sum_(flat)                                        # the single flat all-reduce
        mine, w2 = grad.clone(), words.clone()
        ex = BucketedAllReduce(mine, buckets, words=w2)
        for k in range(3):
            ex.launch(k)
        ex.wait()
        equal = equal and torch.equal(mine, flat[:o]) and torch.equal(w2, flat[o:])
    partial = BucketedAllReduce(grad.clone(), buckets)
    partial.launch(0)
    try:
        partial.wait()
        refused = False
    except RuntimeError:
        refused = True
    for k in (1, 2):                                                # (the other rank entered these collectives too)
        partial.launch(k)
    partial.wait()
    Path(out_dir, f"rank{rank}.json").write_text(json.dumps({"layout": ok, "equal": equal, "refused": refused}))
    dist.destroy_process_group()


def test_bucketed_gradient_exchange_has_the_bits_of_the_flat_all_reduce(tmp_path):
    """three buckets of the flat gradient (head + decoder / encoder stages 3 + 2 / stage 1 + pillar net; the BatchNorm
    gamma / beta at the end of the layout join their stage's bucket), each reduced on its own as the backward pass completes it, with the
    step's [sample count | loss sum] words riding along -- element by element the sums of the ONE flat all-reduce, over 10 steps of
    gradients spanning nine decades; an exchange that skipped a bucket refuses to finish."""
    port = _free_port()
    mp.spawn(_worker_buckets, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = json.loads(Path(tmp_path, f"rank{r}.json").read_text())
        assert got == {"layout": True, "equal": True, "refused": True}, (r, got)
