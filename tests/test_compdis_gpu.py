"""GPU parity of the fused flow -> comp_dis path (a1-a6) through the C ABI.

Bars: the float64 chain must equal the reference's Feather payload bit for bit on all but a
vanishing fraction of elements (double rounding at an f32 tie), and never differ by more than
1 float32 ulp / 1e-6 abs (north_star's bar is 1e-4 abs).  Masks are bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import RES, golden_frames

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-9          # north_star allows 1e-4 abs; the float64 chain holds 1e-9
MIN_EXACT = 0.9999      # fraction of significant elements that must be bit-identical to the reference
SIGNIFICANT = 1e-5      # |value| below this is cancellation residue of flow - pose_flow (~1e-8): the
                        # last bits of inv(pose1) decide its float32 rounding, so it is held to ABS_TOL only


def _engine():
    from himo_amd.compdis import CompDisEngine
    return CompDisEngine()


def _close(got, ref, tol=ABS_TOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max() if got.size else 0.0
    assert err <= tol, f"max abs err {err:g} > {tol:g}"


def _exact_fraction(got, ref):
    got, ref = np.asarray(got), np.asarray(ref)
    sig = np.abs(ref) >= SIGNIFICANT
    return float((got[sig] == ref[sig]).mean()) if sig.any() else 1.0


@pytest.mark.parametrize("data_name", ["av2", "scania"])
def test_single_frame_matches_reference_feather_payload(gpu, gold, data_name):
    from himo_amd.compdis import comp_dis_frame
    for i, f in enumerate(golden_frames(gold, data_name)):
        ref = gold[f"{data_name}/{i}/ref_comp_dis"]
        cd = comp_dis_frame(f, RES)                      # ego_pose from numpy, as save_zip.py:115 computes it
        assert cd.dtype == np.float32 and cd.shape == ref.shape
        assert np.array_equal(cd, ref), f"{(cd != ref).sum()} of {ref.size} elements differ from the reference"
        cd2 = comp_dis_frame(f, RES, host_ego=False)     # 4x4 inverse inside the library
        _close(cd2, ref)
        assert _exact_fraction(cd2, ref) >= MIN_EXACT
        sig = np.abs(ref) >= SIGNIFICANT
        ulp = np.abs(cd2.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))[sig].max()
        assert ulp <= 1


@pytest.mark.parametrize("data_name", ["av2", "scania"])
def test_ragged_batch_with_refined_and_mask(gpu, gold, oracle, data_name):
    from himo_amd.compdis import FrameBatch
    frames = golden_frames(gold, data_name)
    b = FrameBatch.from_frames(frames, RES, with_masks=True)
    out = _engine().run(b, refined=True, data_name=data_name)
    torch.cuda.synchronize()
    cds = [t.cpu().numpy() for t in b.split(out["comp_dis"])]
    rfs = [t.cpu().numpy() for t in b.split(out["refined"])]
    masks = [t.cpu().numpy().astype(bool) for t in b.split(out["eval_mask"])]
    for i, f in enumerate(frames):
        ref = gold[f"{data_name}/{i}/ref_comp_dis"]
        _close(cds[i], ref)
        assert _exact_fraction(cds[i], ref) >= MIN_EXACT
        _close(rfs[i], gold[f"{data_name}/{i}/ref_refined_f64"], tol=4e-6)   # f32 rounding of a ~50 m coordinate
        assert np.array_equal(masks[i], gold[f"{data_name}/{i}/ref_eval_mask"])
        assert np.array_equal(masks[i], oracle.eval_mask(f, data_name))


def test_raw_mode_is_all_zero(gpu, frames_av2):
    from himo_amd.compdis import FrameBatch, comp_dis_frame
    assert not comp_dis_frame(frames_av2[0], "raw").any()
    b = FrameBatch.from_frames(frames_av2, "raw")
    out = _engine().run(b, refined=True)
    assert not out["comp_dis"].any().item()
    assert torch.equal(out["refined"], b.pc0[:, :3])


def test_f32_chain_for_f32_poses(gpu, frames_av2, oracle):
    from himo_amd.compdis import comp_dis_frame
    f = dict(frames_av2[1])
    f["pose0"], f["pose1"] = f["pose0"].astype(np.float32), f["pose1"].astype(np.float32)
    ref = oracle.comp_dis_frame(f, RES)
    assert ref.dtype == np.float32                      # numpy stays in float32 for float32 poses
    _close(comp_dis_frame(f, RES), ref, tol=2e-5)       # f32 chain: op order differs inside LAPACK/BLAS


@pytest.mark.parametrize("stride", [3, 5])
def test_other_point_strides(gpu, frames_av2, oracle, stride):
    from himo_amd.compdis import comp_dis_frame
    f = dict(frames_av2[2])
    xyz = f["pc0"][:, :3]
    f["pc0"] = np.ascontiguousarray(xyz) if stride == 3 else np.concatenate([f["pc0"], f["pc0"][:, :1]], 1)
    ref = oracle.comp_dis_frame_f32(f, RES)
    cd = comp_dis_frame(f, RES)
    _close(cd, ref)
    assert _exact_fraction(cd, ref) >= MIN_EXACT


def test_unaligned_views_take_the_scalar_path(gpu, frames_av2, oracle):
    eng = _engine()
    f = frames_av2[0]
    n = len(f["pc0"]) - 1
    dev = eng.device
    pc = torch.from_numpy(f["pc0"]).to(dev)[1:]          # 16-byte aligned rows, still aligned
    fl = torch.from_numpy(f[RES]).to(dev)[1:]            # 12-byte offset: not 16-byte aligned
    dt = torch.from_numpy(f["lidar_dt"]).to(dev)[1:]     # 4-byte offset
    cd = eng.run_frame(pc, fl, dt, f["pose0"], f["pose1"])
    g = {k: (v[1:] if isinstance(v, np.ndarray) and v.shape[:1] == (n + 1,) else v) for k, v in f.items()}
    _close(cd.cpu().numpy(), oracle.comp_dis_frame_f32(g, RES))


@pytest.mark.parametrize("sizes", [[1], [3, 1, 2], [1023, 1025, 4097, 5], [0, 7, 0, 4096, 0]])
def test_ragged_edges_and_empty_frames(gpu, oracle, sizes):
    from himo_amd.compdis import FrameBatch
    from himo_amd.synthetic import make_frame
    frames = []
    for i, n in enumerate(sizes):
        f = make_frame(100 + i, n_points=max(n, 1), n_instances=0)
        if n == 0:
            f = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1 and k not in ("pose0", "pose1") else v)
                 for k, v in f.items()}
        frames.append(f)
    b = FrameBatch.from_frames(frames, RES, with_masks=True)
    out = _engine().run(b, refined=True, data_name="av2")
    torch.cuda.synchronize()
    for f, cd, m in zip(frames, b.split(out["comp_dis"]), b.split(out["eval_mask"])):
        if len(f["pc0"]) == 0:
            assert cd.shape[0] == 0
            continue
        _close(cd.cpu().numpy(), oracle.comp_dis_frame_f32(f, RES))
        assert np.array_equal(m.cpu().numpy().astype(bool), oracle.eval_mask(f, "av2"))


def test_error_behaviour_matches_reference(gpu, frames_av2):
    from himo_amd.compdis import comp_dis_frame
    f = frames_av2[0]
    with pytest.raises(KeyError):
        comp_dis_frame(f, "missing_result")              # data[res_name]
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == len(f["pc0"]) else v) for k, v in f.items()}
    with pytest.raises(ValueError, match="empty sequence"):
        comp_dis_frame(empty, RES)                       # max() of an empty lidar_dt
    sing = dict(f, pose1=np.zeros((4, 4)))
    with pytest.raises(np.linalg.LinAlgError):
        comp_dis_frame(sing, RES)                        # np.linalg.inv(pose1)
    with pytest.raises(np.linalg.LinAlgError):
        comp_dis_frame(sing, RES, host_ego=False)        # the library's own singularity check


def test_full_size_frames_against_oracle_and_invariants(gpu, oracle):
    """BASELINE size (120k points): direct oracle comparison plus size-independent properties."""
    from himo_amd.compdis import FrameBatch
    from himo_amd.synthetic import make_frame
    frames = [make_frame(i, n_points=120_000) for i in range(3)]
    eng = _engine()
    b = FrameBatch.from_frames(frames, RES)
    out = eng.run(b, refined=True)
    cd = out["comp_dis"]
    worst, exact = 0.0, []
    for f, got in zip(frames, b.split(cd)):
        ref = oracle.comp_dis_frame_f32(f, RES)
        g = got.cpu().numpy()
        worst = max(worst, float(np.abs(g.astype(np.float64) - ref).max()))
        exact.append(_exact_fraction(g, ref))
    assert worst <= ABS_TOL and min(exact) >= MIN_EXACT, (worst, exact)
    # (1) the latest point of each sweep is not moved: dt0 == 0 there
    for f, got in zip(frames, b.split(cd)):
        assert not got[int(np.argmax(f["lidar_dt"]))].any().item()
    # (2) linearity in sensor_dt: doubling sensor_dt halves comp_dis exactly (power-of-two scale)
    cd2 = eng.run(b, sensor_dt=0.2)["comp_dis"]
    _close((cd2 * 2).cpu().numpy(), cd.cpu().numpy(), tol=1e-6)
    # (3) refined - pc0 == comp_dis up to f32 rounding of the sum
    assert (out["refined"] - b.pc0[:, :3] - cd).abs().max().item() <= 8e-6
    # (4) batch == frame-by-frame
    one = eng.run(FrameBatch.from_frames(frames[1:2], RES))["comp_dis"]
    assert torch.equal(one, b.split(cd)[1])


@pytest.mark.parametrize("workload", ["compdis", "train"])
def test_bench_nccl_branch_runs_as_a_one_rank_communicator(gpu, workload):
    """A single-GPU box cannot run N > 1, but it CAN execute the RCCL code path the multi-GPU runs use: bench.py
    --force-process-group joins a one-rank "nccl" (= RCCL) communicator bound to cuda:0 and goes through the same barrier,
    max all-reduce of the wall time and all-gather of the frame counts -- and, for the train workload, the all-reduce of the flat
    gradient buffer inside every optimiser step.  (The N = 2 logic itself is covered on CPU over gloo:
    tests/test_distributed_cpu.py.)"""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    extra = ["--frames-per-step", "32", "--steps", "5"] if workload == "compdis" else ["--points", "20000", "--steps", "2"]
    r = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "1", "--force-process-group", "--workload", workload,
                        "--warmup", "1", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collectives"] == "nccl x1" and line["value"] > 0
    if workload == "compdis":
        assert line["config"]["frames_per_rank"] == [5 * 32] and line["parity"]["comp_dis_max_abs_vs_ref"] <= 1e-6
    else:                                   # the step's flat 28 MB gradient went through the RCCL all-reduce
        assert line["config"]["frames_per_rank"] == [2] and np.isfinite(line["parity"]["loss_after_warmup"])
