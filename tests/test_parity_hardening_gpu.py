"""Harder parity cases for the scene-flow network (a10) -- every one of them a comparison against the CPU restatement
(oracle/seflow_oracle.py; PARITY UNPINNED: the reference's network source is absent), not a self-comparison:

  * one BASELINE-size sample (3 x 120k points) per matrix arithmetic,
  * crowded pillars (100k points in a 2 m square: thousands of points per cell) against ``pillar_image`` -- where the
    "ascending point order" summation contract of the pillar means matters,
  * a LiDAR-like ring cloud (range-dependent density, most of the grid empty),
  * intermediate activations far from O(1): gains {0.3, 1, 3, 10} pushed through the encoder / decoder (compensated one
    layer later so the flow stays O(1) and 1e-4 abs keeps its meaning), and BatchNorm statistics far from unit,

each reporting its margin to north_star's 1e-4 for f32 / bf16x3 / f16x2, and showing that ``precision="auto"`` ends on
an arithmetic that meets the bar.  Margins are written to gpurun_out/parity_margins.json (copied to profiles/)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4
PRECISIONS = ("f32", "bf16x3", "f16x2")
_MARGINS = {}


@pytest.fixture(scope="module")
def so():
    import seflow_oracle
    return seflow_oracle


@pytest.fixture(scope="module", autouse=True)
def _write_margins():
    yield
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parents[1])) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "parity_margins.json").write_text(json.dumps(_MARGINS, indent=1, sort_keys=True))
    except OSError:
        pass


def _compare(case, gpu, so, params, frames, max_points, expect_auto="f16x2"):
    """max abs / EPE of each arithmetic's flow vs the CPU restatement on (history, pc0, pc1) = frames; asserts the bar."""
    from himo_amd.pipeline import HiMoPipeline, Sample
    from himo_amd.seflow.model import SeFlowNet
    fh, f0, f1 = frames
    ref = so.forward(params, fh["pc0"], f0["pc0"], f1["pc0"], fh["pose0"], f0["pose0"], f0["pose1"])
    assert np.isfinite(ref).all() and np.abs(ref).max() < 1e3, "the case itself must keep the flow O(1)"
    sample = Sample.from_frames(fh, f0, f1, device=gpu)
    row = {"flow_abs_max_of_reference": float(np.abs(ref).max())}
    for prec in PRECISIONS:
        net = SeFlowNet(params, device=gpu, max_points=max_points, precision=prec, autotune=False)
        got = net.forward_device(sample.pch1, sample.pc0, sample.pc1, sample.pose_h1, sample.pose0, sample.pose1).cpu().numpy()
        finite = bool(np.isfinite(got).all())
        err = float(np.abs(got - ref).max()) if finite else float("inf")
        row[prec] = {"max_abs": err, "epe": float(np.linalg.norm(got - ref, axis=1).mean()) if finite else float("inf"),
                     "margin_to_1e-4": TOL / err if err > 0 else float("inf"), "finite": finite}
        del net
    auto = HiMoPipeline(device=gpu, max_points=max_points, max_batch=1, params=params)        # precision="auto"
    got = auto.flows([sample])[0].cpu().numpy()
    row["auto"] = {"ends_in": auto.net.precision, "max_abs": float(np.abs(got - ref).max())}
    _MARGINS[case] = row
    torch.cuda.empty_cache()
    assert row["f32"]["max_abs"] <= TOL and row["bf16x3"]["max_abs"] <= TOL, (case, row)
    assert row["auto"]["max_abs"] <= TOL, (case, row)                       # whatever auto chose meets the bar
    if expect_auto is not None:
        assert row["auto"]["ends_in"] == expect_auto, (case, row)
    if row["auto"]["ends_in"] == "f16x2":
        assert row["f16x2"]["max_abs"] <= TOL, (case, row)
    return row


def test_full_size_sample_per_arithmetic_vs_cpu_restatement(gpu, so):
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    frames = [make_frame(500 + i, n_points=120_000) for i in range(3)]
    row = _compare("uniform_120k", gpu, so, spec.init_params(0), frames, 120_000)
    assert all(row[p]["epe"] <= 2e-5 for p in PRECISIONS)


def test_lidar_ring_cloud_vs_cpu_restatement(gpu, so):
    """range-dependent density: ~8 % of the cells occupied, up to ~100 points per cell near the sensor, a fifth of the
    returns outside the network range (they must come back with pose flow only)"""
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    frames = [make_frame(520 + i, n_points=120_000, cloud="rings") for i in range(3)]
    p0 = frames[1]["pc0"]
    assert (np.abs(p0[:, :2]).max(axis=1) > 51.2).mean() > 0.05
    _compare("lidar_rings_120k", gpu, so, spec.init_params(1), frames, 120_000)


def test_crowded_pillars_vs_cpu_restatement(gpu, so):
    """100k points piled into a 2 m square = 100 cells: ~1000 points per pillar.  The pillar means / feature means are
    float32 sums in ASCENDING POINT ORDER (spec step 2; the oracle's index_add_ is sequential) -- compared here against
    the oracle image, then through the whole network."""
    from himo_amd.seflow import spec
    from himo_amd.seflow.model import SeFlowNet
    from himo_amd.synthetic import make_frame
    params = spec.init_params(2)
    rng = np.random.default_rng(77)
    def crowd(seed):
        f = make_frame(seed, n_points=100_000)
        r = np.random.default_rng(seed)
        f["pc0"][:, :2] = r.uniform(-1.0, 1.0, (100_000, 2)).astype(np.float32)
        f["pc0"][:, 2] = r.uniform(-2.5, 2.5, 100_000).astype(np.float32)
        return f
    frames = [crowd(540 + i) for i in range(3)]
    net = SeFlowNet(params, device=gpu, max_points=100_000, precision="bf16x3", autotune=False)     # float32 pillar image
    T = so.ego_transform(frames[1]["pose0"], frames[1]["pose1"])
    net.pillarize_into(1, torch.from_numpy(frames[1]["pc0"]).to(gpu), T)
    torch.cuda.synchronize()
    img, valid, pid, off = so.pillar_image(params, so.transform_points(frames[1]["pc0"], T))
    counts = torch.bincount(pid[valid])
    assert counts.max().item() > 500                                   # really crowded
    got_img = net.B0.view(512, 512, 3, 32)[:, :, 1, :].permute(2, 0, 1).cpu()
    err = (got_img - img).abs().max().item()
    _MARGINS["crowded_pillar_image"] = {"max_abs": err, "max_points_per_cell": int(counts.max()), "image_abs_max": float(img.abs().max())}
    assert err <= 2e-6 * max(1.0, img.abs().max().item()), err         # same summation order: float32 round-off only
    del net
    _compare("crowded_pillars_100k", gpu, so, params, frames, 100_000)


def _with_gain(params, g):
    """activations g x larger from the gained layer to the compensating one; the network function is (nearly) unchanged:
    encoder -- BatchNorm gamma / beta of one block per stage x g (GELU(g x) ~ g GELU(x) once |x| is large; exact for the
    sign pattern), the next block's weights / g; decoder (no activations: exactly linear) -- u4 x g, u5 weights / g."""
    p = {k: v.copy() for k, v in params.items()}
    for up, down in (("enc1.1", "enc1.2"), ("enc2.2", "enc2.3"), ("enc3.3", "enc3.4")):
        p[f"{up}.bn.gamma"] *= g
        p[f"{up}.bn.beta"] *= g
        p[f"{down}.weight"] /= g
    for blk in ("dec1", "dec2", "dec3"):
        p[f"{blk}.u4.weight"] *= g
        p[f"{blk}.u4.bias"] *= g
        p[f"{blk}.u5.weight"] /= g
    return p


@pytest.mark.parametrize("gain", [1e-4, 1e-3, 1e-2, 0.3, 1.0, 3.0, 10.0, 100.0])
def test_intermediate_gain_sweep_vs_cpu_restatement(gpu, so, gain):
    """gains < 1 push the intermediate activations DOWN, towards the two-term fp16 split's absolute
    floor: below |x| = 1/4 the low part is an fp16 subnormal and a value keeps 2^-25 absolute instead of 2^-23 relative."""
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    frames = [make_frame(560 + i, n_points=30_000) for i in range(3)]
    _compare(f"gain_{gain:g}", gpu, so, _with_gain(spec.init_params(3), gain), frames, 30_000, expect_auto=None)


@pytest.mark.parametrize("stage", ["enc1", "enc2", "enc3"])
def test_one_stage_with_tiny_batchnorm_gamma_vs_cpu_restatement(gpu, so, stage):
    """EVERY block of one encoder stage with BatchNorm gamma and beta x 1e-3: the stage's activations are ~1e-3 (GELU is then
    nearly x / 2: the function changes, which is fine -- the CPU restatement runs the same parameters), the next consumers'
    weights x 1e3 bring the signal back to O(1) -- an absolute error floor of the small activations would arrive amplified."""
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    p = {k: v.copy() for k, v in spec.init_params(6).items()}
    blocks = [name for name, _, _, _ in spec.ENCODER if name.startswith(stage + ".")]
    last = blocks[-1]
    p[f"{last}.bn.gamma"] *= 1e-3
    p[f"{last}.bn.beta"] *= 1e-3
    nxt = {"enc1": "enc2.0", "enc2": "enc3.0"}.get(stage)
    if nxt is not None:
        p[f"{nxt}.weight"] *= 1e3
    p[{"enc1": "dec2.u3.weight", "enc2": "dec1.u3.weight", "enc3": "dec1.u1.weight"}[stage]] *= 1e3     # the decoder's 1x1 on this stage's frames
    frames = [make_frame(570 + i, n_points=30_000) for i in range(3)]
    _compare(f"tiny_gamma_{stage}", gpu, so, p, frames, 30_000, expect_auto=None)


def test_batchnorm_statistics_far_from_unit_vs_cpu_restatement(gpu, so):
    """running variances over four decades, means of the size of the activations, per-channel effective scales 0.3 ... 3"""
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    params = spec.init_params(4)
    rng = np.random.default_rng(5)
    for name, _, cout, _ in spec.ENCODER:
        var = np.exp(rng.uniform(np.log(1e-2), np.log(1e2), cout)).astype(np.float32)
        eff = np.exp(rng.uniform(np.log(0.3), np.log(3.0), cout)).astype(np.float32)
        params[f"{name}.bn.var"] = var
        params[f"{name}.bn.gamma"] = (eff * np.sqrt(var + spec.BN_EPS)).astype(np.float32) * 0.75
        params[f"{name}.bn.mean"] = rng.uniform(-1.0, 1.0, cout).astype(np.float32)
        params[f"{name}.bn.beta"] = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    frames = [make_frame(580 + i, n_points=30_000) for i in range(3)]
    _compare("batchnorm_far_from_unit", gpu, so, params, frames, 30_000, expect_auto=None)


def test_overflowing_weights_send_auto_to_the_bf16_split(gpu, so):
    """a gain large enough to leave fp16's range (65504): f16x2 alone is non-finite, auto ends in bf16x3 and meets the bar"""
    from himo_amd.seflow import spec
    from himo_amd.synthetic import make_frame
    frames = [make_frame(590 + i, n_points=20_000) for i in range(3)]
    row = _compare("gain_1e6_overflow", gpu, so, _with_gain(spec.init_params(3), 1e6), frames, 20_000, expect_auto="bf16x3")
    assert not row["f16x2"]["finite"]


def test_trained_looking_weights_per_arithmetic_vs_cpu_restatement(gpu, so):
    """every other case runs ``spec.init_params`` weights (He-uniform, BatchNorm gamma in [0.8, 1.2]).  The
    f16x2 headline's range assumptions (|activation| < 65504, weights packed x 2^6) meet here the only trained-looking weights
    this repo can produce: the parameters after 20 optimiser steps of its own trainer (BatchNorm in training mode: running
    statistics and gamma / beta have moved; Adam has reshaped every weight tensor), on a full-size 3 x 120k sample, all three
    arithmetics against the CPU restatement fed the SAME exported parameters."""
    from himo_amd.dataset import ListDataset
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import make_sample, triplets
    from himo_amd.seflow.train import SeFlowTrainer
    from himo_amd.synthetic import make_frame
    train_frames = [make_frame(900 + i, n_points=30_000, scene_id="t") for i in range(6)]
    ds = ListDataset(train_frames)
    trips = triplets(ds)
    tr = SeFlowTrainer(spec.init_params(3), device=gpu, max_points=32_000, batchnorm="batch")
    losses = []
    for step in range(20):
        losses.append(float(tr.train_batch([make_sample(ds, trips[step % len(trips)], gpu)], lr=3e-4).item()))
    assert np.isfinite(losses).all()          # (different samples per step: the per-step loss is not monotone; what matters is that the weights moved)
    tr.sync_running_stats()
    params = tr.export_params()
    init = spec.init_params(3)
    moved = max(float(np.abs(params[k] - init[k]).max()) for k in init if k.endswith(".weight"))
    bn_moved = max(float(np.abs(params[k] - init[k]).max()) for k in init if k.endswith((".mean", ".var", ".gamma", ".beta")))
    assert moved > 1e-3 and bn_moved > 1e-3, (moved, bn_moved)            # these really are not the initial weights any more
    del tr
    torch.cuda.empty_cache()
    frames = [make_frame(950 + i, n_points=120_000) for i in range(3)]
    row = _compare("trained_20_steps_120k", gpu, so, params, frames, 120_000, expect_auto=None)
    row["weights_moved_max_abs"], row["batchnorm_moved_max_abs"] = moved, bn_moved
    row["train_loss_first_last"] = [losses[0], losses[-1]]
    assert row["f16x2"]["finite"]
