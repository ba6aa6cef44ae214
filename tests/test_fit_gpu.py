"""Training at BASELINE size and the loop around the step (config 5; assets/slurm/ssl-train-av2.sh:31-34: batch_size=8,
epochs=12, save_top_model=3, lr 6e-5, StepLR(3, 0.5)).  PARITY UNPINNED (the reference's train.py is absent): these tests
pin this build's own behaviour -- finite and reproducible at 120k points, checkpoints that resume bit for bit, the epoch /
batch / schedule bookkeeping."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _labelled(dataset_frames):
    return dataset_frames


def test_one_full_size_train_step_is_finite_and_reproducible(gpu):
    """3 x 120k-point sweeps, labelled clusters: forward with saved activations -> 4-term loss -> full backward -> Adam.
    Two fresh trainers on the same sample give the same loss and the same gradient norm (the backward pass is fixed-order
    reductions; the loss's scatter-add half accumulates fixed point: see test_training_steps_are_bit_reproducible)."""
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import make_sample
    from himo_amd.seflow.train import SeFlowTrainer
    from himo_amd.synthetic import make_frame
    frames = [make_frame(700 + i, n_points=120_000, scene_id="s") for i in range(3)]
    smp = make_sample(frames, (0, 1, 2), gpu)
    runs = []
    for _ in range(2):
        tr = SeFlowTrainer(spec.init_params(8), device=gpu, max_points=120_000, precision="mixed")
        terms, total = tr.loss_and_grad(*smp)
        g = tr.flat_g.double()
        runs.append((float(total.item()), float(g.norm().item()), {k: float(v.item()) for k, v in terms.items()}))
        assert torch.isfinite(tr.flat_g).all()
        p0 = tr.flat_p.clone()
        tr.adam_step(6e-5)
        assert torch.isfinite(tr.flat_p).all() and not torch.equal(tr.flat_p, p0)
        del tr
        torch.cuda.empty_cache()
    (l0, n0, t0), (l1, n1, t1) = runs
    assert np.isfinite(l0) and l0 > 0 and n0 > 0
    assert l0 == pytest.approx(l1, rel=1e-9) and n0 == pytest.approx(n1, rel=1e-5), runs
    assert all(v >= 0 for v in t0.values()) and sum(t0.values()) == pytest.approx(l0, rel=1e-9)


@pytest.mark.parametrize("batch", [1, 3])
def test_training_steps_are_bit_reproducible(gpu, batch):
    """Two fresh trainers, the same four samples (``batch`` = 3: three of them per pass and optimiser step, BatchNorm statistics over the
    pass), BatchNorm in training mode, the encoder's weight gradients on the side stream: the same parameter bits after four optimiser steps.  Every reduction of the step is a fixed-order tree, and the one scatter-add
    (the pc1 -> moved half of the Chamfer gradient, csrc/sslloss.hip) accumulates 64-bit fixed point -- with float atomics the
    last bits of a step depended on the arrival order, and the trained-weights parity case saw different weights on every run."""
    from himo_amd.dataset import ListDataset
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import make_sample, triplets
    from himo_amd.seflow.train import SeFlowTrainer
    from himo_amd.synthetic import make_frame
    ds = ListDataset([make_frame(760 + i, n_points=30_000, scene_id="r") for i in range(6)])
    trips = triplets(ds)
    finals, losses = [], []
    for _ in range(2):
        tr = SeFlowTrainer(spec.init_params(5), device=gpu, max_points=32_000, batchnorm="batch", batch=batch)
        ls = [float(tr.train_batch([make_sample(ds, trips[(k + j) % len(trips)], gpu) for j in range(batch)], lr=3e-4).item()) for k in range(4)]
        torch.cuda.synchronize()
        finals.append(tr.flat_p.clone())
        losses.append(ls)
        del tr
        torch.cuda.empty_cache()
    assert losses[0] == losses[1], losses
    assert torch.equal(finals[0], finals[1]), int((finals[0] != finals[1]).sum())


def test_fit_runs_epochs_batches_schedule_and_keeps_the_top_checkpoints(gpu, tmp_path):
    from himo_amd.dataset import ListDataset
    from himo_amd.seflow import checkpoint as ck
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import fit, triplets
    from himo_amd.synthetic import make_frame
    frames = [make_frame(720 + i, n_points=5_000, scene_id=f"scene{i // 6}") for i in range(12)]     # two scenes of six sweeps
    ds = ListDataset(frames)
    trips = triplets(ds)
    assert len(trips) == 10 and trips[0] == (0, 0, 1) and trips[1] == (0, 1, 2) and trips[5] == (6, 6, 7)      # no pair across scenes
    logs = []
    out = fit(ds, spec.init_params(9), out_dir=tmp_path, epochs=5, batch_size=4, lr=2e-4, step_size=2, gamma=0.5, save_top=2,
              max_points=6_000, device=gpu, log=logs.append, ssl_label="flow_instance_id")
    hist = out["history"]
    assert [h["epoch"] for h in hist] == [0, 1, 2, 3, 4] and all(h["steps"] == 3 for h in hist)            # ceil(10 / 4) steps per epoch
    assert [h["lr"] for h in hist] == [2e-4, 2e-4, 1e-4, 1e-4, 5e-5]                                         # StepLR(2, 0.5)
    assert all(np.isfinite(h["train_loss"]) for h in hist) and hist[-1]["train_loss"] < hist[0]["train_loss"]
    assert out["trainer"].step_count == 15 and len(logs) == 5
    kept = sorted(tmp_path.glob("*.npz"))
    assert len(kept) == 2                                                                                  # save_top_model
    best_val = float(ck.load_params(out["best"], with_extra=True)[1]["val"])
    assert best_val == pytest.approx(min(h["train_loss"] for h in hist), rel=1e-4, abs=1e-4)   # file name carries 4 decimals; the array is exact


def test_fit_trains_from_scratch_with_batchnorm_in_training_mode(gpu, tmp_path):
    """The reference's job shape (ssl-train-av2.sh:31-34: no checkpoint=): random weights, BatchNorm freshly reset (gamma 1,
    beta 0, running statistics 0 / 1) and in TRAINING mode; the loss falls, gamma / beta and the running statistics move,
    validation (running statistics) is finite, and the checkpoint drives the inference network."""
    from himo_amd.dataset import ListDataset
    from himo_amd.seflow import spec
    from himo_amd.seflow.checkpoint import load_params
    from himo_amd.seflow.fit import fit
    from himo_amd.seflow.model import SeFlowNet
    from himo_amd.synthetic import make_frame
    frames = [make_frame(760 + i, n_points=5_000, scene_id=f"scene{i // 5}") for i in range(10)]
    start = spec.init_params(12, fresh_bn=True)
    assert np.all(start["enc2.3.bn.gamma"] == 1) and np.all(start["pfn.bn.var"] == 1) and np.all(start["enc1.0.bn.mean"] == 0)
    out = fit(ListDataset(frames), start, out_dir=tmp_path, epochs=4, batch_size=4, lr=2e-4, save_top=1, max_points=6_000, device=gpu,
              val_dataset=ListDataset(frames[:5]), log=None, ssl_label="flow_instance_id")
    hist = out["history"]
    assert all(np.isfinite(h["train_loss"]) and np.isfinite(h["val_loss"]) for h in hist), hist
    assert hist[-1]["train_loss"] < 0.9 * hist[0]["train_loss"], hist
    trained = load_params(out["best"])
    for k in ("pfn.bn.gamma", "enc1.0.bn.beta", "enc3.5.bn.gamma", "enc2.0.bn.mean", "pfn.bn.var"):
        assert not np.array_equal(trained[k], start[k]), k
    assert np.all(trained["enc1.1.bn.var"] > 0)
    net = SeFlowNet(trained, device=gpu, max_points=6_000, precision="f16x2", autotune=False)
    flow = net.forward(frames[0]["pc0"], frames[1]["pc0"], frames[2]["pc0"], frames[0]["pose0"], frames[1]["pose0"], frames[1]["pose1"])
    assert torch.isfinite(flow).all()


def test_checkpoint_resume_continues_bit_for_bit(gpu, tmp_path):
    """save after 2 steps, load into a FRESH trainer, take 2 more steps on both: identical parameters and moments"""
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import make_sample
    from himo_amd.seflow.model import SeFlowNet
    from himo_amd.seflow.train import SeFlowTrainer
    from himo_amd.synthetic import make_frame
    frames = [make_frame(740 + i, n_points=6_000, scene_id="s") for i in range(4)]
    a, b = make_sample(frames, (0, 1, 2), gpu), make_sample(frames, (1, 2, 3), gpu)
    tr = SeFlowTrainer(spec.init_params(10), device=gpu, max_points=7_000, precision="bf16x3")
    tr.train_step(*a, lr=1e-4); tr.train_step(*b, lr=1e-4)
    path = tr.save_checkpoint(tmp_path / "ck.npz", epoch=0)
    tr.train_step(*a, lr=1e-4); tr.train_step(*b, lr=1e-4)
    other = SeFlowTrainer(spec.init_params(11), device=gpu, max_points=7_000, precision="bf16x3")       # different initial weights
    extra = other.load_checkpoint(path)
    assert int(extra["epoch"]) == 0 and other.step_count == 2
    other.train_step(*a, lr=1e-4); other.train_step(*b, lr=1e-4)
    # (written when the loss's scatter-add half used float atomics; the step is bit-reproducible now, the float32 round-off bar stays)
    for x, y in ((tr.flat_p, other.flat_p), (tr.flat_m, other.flat_m), (tr.flat_v, other.flat_v)):
        assert (x - y).abs().max().item() <= 1e-6 * max(x.abs().max().item(), 1e-30)
    assert other.step_count == tr.step_count == 4
    for k in ("pfn.bn.mean", "pfn.bn.var", "enc1.0.bn.mean", "enc3.5.bn.var"):          # BatchNorm running statistics travel with the file
        x, y = tr.net.p[k], other.net.p[k]
        assert (x - y).abs().max().item() <= 1e-6 * max(x.abs().max().item(), 1e-30), k
        assert not np.array_equal(x.cpu().numpy(), spec.init_params(10)[k]), k                # ... and have moved since initialisation
    # the file is also an inference checkpoint: save.main's loader -> SeFlowNet
    from himo_amd.seflow.checkpoint import load_params
    net = SeFlowNet(load_params(path), device=gpu, max_points=7_000, precision="bf16x3", autotune=False)
    flow = net.forward(frames[0]["pc0"], frames[1]["pc0"], frames[2]["pc0"], frames[0]["pose0"], frames[1]["pose0"], frames[1]["pose1"])
    assert torch.isfinite(flow).all()


@pytest.mark.parametrize("ssl_label", ["seflow_auto", "flow_instance_id"])
def test_fed_loop_ends_in_the_serial_loops_parameter_bits(gpu, tmp_path, ssl_label):
    """``fit`` over ``.h5`` scene files with the samples prepared AHEAD of the step (``feeder.TrainFeeder``: reader threads, pinned
    staging, a copy stream, labels generated on a label stream) against the same run with every sample built inside the step loop
    on the launch thread (``num_workers=0``): the same losses and the same parameter bits -- across an epoch boundary, a partial
    last batch, validation, and with both kinds of labels."""
    import warnings
    from himo_amd.dataset import HDF5Dataset
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import fit, train_fields, triplets
    from himo_amd.synthetic import make_scene, write_h5_scenes
    write_h5_scenes(tmp_path, [make_scene(90 + sc, 5, n_points=6_000, scene_id=f"fit{sc:02d}") for sc in range(2)])
    finals, hists = [], []
    for workers in (0, 3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                   # (the last sweep of a scene has no successor)
            ds = HDF5Dataset(tmp_path, fields=train_fields(ssl_label), zero_copy=True)
        assert len(triplets(ds)) == 8 and triplets(ds)[0] == (0, 0, None) and triplets(ds)[4] == (4, 4, None)
        out = fit(ds, spec.init_params(3), epochs=2, batch_size=3, lr=2e-4, max_points=6_000, device=gpu, log=None, ssl_label=ssl_label,
                  num_workers=workers, prefetch=2, val_dataset=ds)
        torch.cuda.synchronize()
        finals.append(out["trainer"].flat_p.clone())
        hists.append([(h["train_loss"], h["val_loss"], h["steps"]) for h in out["history"]])
        ds.close()
        del out
        torch.cuda.empty_cache()
    assert hists[0] == hists[1] and all(s == 3 for _, _, s in hists[0]), hists
    assert torch.equal(finals[0], finals[1]), int((finals[0] != finals[1]).sum())


def test_train_feeder_surfaces_reader_errors_and_stops_early(gpu):
    """a frame without ground masks under ``ssl_label=seflow_auto`` is the KeyError the serial loop raises, raised on the consumer's
    thread; ``close()`` before the end leaves no thread behind"""
    import threading
    from himo_amd.dataset import ListDataset
    from himo_amd.feeder import TrainFeeder
    from himo_amd.seflow.fit import triplets
    from himo_amd.synthetic import make_frame
    frames = [make_frame(30 + i, n_points=3_000, scene_id="s") for i in range(6)]
    ds = ListDataset(frames)
    feed = TrainFeeder(ds, triplets(ds), device=gpu, depth=1, workers=2)
    first = next(iter(feed))
    assert first[1].shape == (3_000, 4) and first[6].dtype == torch.int32 and isinstance(first[8], int)
    feed.close()
    assert not any(t.name.startswith("himo-train") for t in threading.enumerate() if t.is_alive())
    bad = ListDataset([{k: v for k, v in f.items() if k != "gm0"} for f in frames])
    with pytest.raises(KeyError, match="gm0"):
        list(TrainFeeder(bad, triplets(bad), device=gpu))


def test_fit_program_over_h5_scene_files(gpu, tmp_path):
    """``python -m himo_amd.seflow.fit --dataset_path <dir of .h5 scenes>`` (the reference's ``train.py ... +ssl_label=seflow_auto``,
    assets/slurm/ssl-train-av2.sh:31-34) as a PROGRAM: scenes opened with the training fields as views of the file mapping, samples fed
    ahead of the step, labels generated on the device, a batch in one pass, checkpoints written -- and a second run resumes from one."""
    import subprocess, sys, os
    from pathlib import Path
    from himo_amd.seflow.checkpoint import load_params
    from himo_amd.synthetic import make_scene, write_h5_scenes
    data = tmp_path / "scenes"
    data.mkdir()
    write_h5_scenes(data, [make_scene(70 + sc, 5, n_points=5_000, scene_id=f"h5fit{sc}") for sc in range(2)])
    repo = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=str(repo))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, "-m", "himo_amd.seflow.fit", "--dataset_path", str(data), "--out_dir", str(tmp_path / "ckpt"), "--epochs", "2",
           "--batch_size", "3", "--save_top_model", "2"]
    out = subprocess.run(cmd, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.count("epoch ") == 2 and "train loss" in out.stdout
    kept = sorted((tmp_path / "ckpt").glob("*.npz"))
    assert len(kept) == 2
    params, extra = load_params(kept[0], with_extra=True)
    assert int(extra["step"]) in (3, 6) and np.isfinite(params["enc1.0.weight"]).all()      # 8 samples in steps of 3: 3 steps per epoch
    again = subprocess.run(cmd[:-4] + ["--epochs", "3", "--resume", str(kept[-1])], env=env, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert again.returncode == 0, again.stderr[-3000:]
