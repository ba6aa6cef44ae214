"""GPU parity of a12 (per-scene coordinate-MLP fit) against the CPU restatement.  PARITY UNPINNED (own spec)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(seed, n0, n1):
    rng = np.random.default_rng(seed)
    pc1 = rng.uniform([-40, -40, -2], [40, 40, 2], (n1, 3)).astype(np.float32)
    k = min(n0, n1)
    pc0 = np.empty((n0, 3), np.float32)
    pc0[:k] = pc1[:k] - np.array([0.6, 0.2, 0.0], np.float32) + rng.normal(0, 0.02, (k, 3)).astype(np.float32)
    if n0 > k:
        pc0[k:] = rng.uniform([-40, -40, -2], [40, 40, 2], (n0 - k, 3)).astype(np.float32)
    return pc0, pc1


def test_wgrad_and_masked_dgrad_kernels(gpu):
    """The two backward products against torch on the same device-independent data."""
    import ctypes
    from himo_amd import _lib
    from himo_amd.fastnsf import FastNSF
    lib = _lib.load()
    eng = FastNSF(device=gpu)
    rng = np.random.default_rng(0)
    n = 5000
    for cin, cout in [(128, 128), (4, 128), (128, 4)]:
        x = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).to(gpu)
        dz = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).to(gpu)
        dw = torch.empty((cin, cout), dtype=torch.float32, device=gpu)
        db = torch.empty(cout, dtype=torch.float32, device=gpu)
        ws = torch.empty(int(lib.himo_wgrad_workspace_bytes(n)), dtype=torch.uint8, device=gpu)
        _lib.check(lib.himo_linear_wgrad(n, x.data_ptr(), cin, cin, dz.data_ptr(), cout, cout, dw.data_ptr(), db.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _lib.stream_handle()))
        ref_dw = (x.double().T @ dz.double()).float()
        assert (dw - ref_dw).abs().max().item() <= 2e-3 * ref_dw.abs().max().item()
        assert (db - dz.double().sum(0).float()).abs().max().item() <= 1e-3 * n ** 0.5
    w = torch.from_numpy(rng.normal(size=(128, 128)).astype(np.float32) * 0.1).to(gpu)
    act = torch.relu(torch.from_numpy(rng.normal(size=(n, 128)).astype(np.float32)).to(gpu))
    dz = torch.from_numpy(rng.normal(size=(n, 128)).astype(np.float32)).to(gpu)
    wt, out = torch.empty_like(w), torch.empty_like(act)
    _lib.check(lib.himo_transpose(w.data_ptr(), 128, 128, wt.data_ptr(), _lib.stream_handle()))
    assert torch.equal(wt, w.T.contiguous())
    eng._gemm(dz, wt, None, out, n, 128, 128, 6, aux=act)
    ref = (dz @ w.T) * (act > 0)
    assert (out - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("precision,tol", [("f32", 2e-4), ("mixed", 1e-3)])
@pytest.mark.parametrize("n0,n1", [(6000, 5500), (120_000, 119_000)])      # incl. BASELINE size
def test_single_step_gradients_match_autograd(gpu, n0, n1, precision, tol):
    import fastnsf_oracle as fo
    from himo_amd.fastnsf import FastNSF, init_mlp
    pc0, pc1 = _scene(1, n0, n1)
    layers = init_mlp(3)
    eng = FastNSF(device=gpu, iters=1, lr=0.0, seed=3, objective="nn", precision=precision)
    eng.fit(pc0, pc1, layers=layers)
    ref_loss, ref_grads, _ = fo.loss_and_grads(layers, pc0, pc1)
    assert eng.loss_history[0][1] == pytest.approx(ref_loss, rel=1e-5 if precision == "f32" else 1e-4)
    for k, (gw, gb) in enumerate(ref_grads):
        cin, cout = gw.shape
        got_w, got_b = eng.gW[k].cpu().numpy()[:cin, :cout], eng.gb[k].cpu().numpy()[:cout]
        scale = max(np.abs(gw).max(), 1e-8)
        assert np.abs(got_w - gw).max() <= tol * scale, k
        assert np.abs(got_b - gb).max() <= tol * max(np.abs(gb).max(), 1e-8), k


def test_fit_follows_the_cpu_restatement_and_recovers_the_motion(gpu):
    import fastnsf_oracle as fo
    from himo_amd.fastnsf import FastNSF, init_mlp
    pc0, pc1 = _scene(2, 8000, 8000)
    layers = init_mlp(5)
    iters = 25
    eng = FastNSF(device=gpu, iters=iters, lr=1e-3, early_patience=10_000, objective="nn")     # patience on => loss logged every step
    flow = eng.fit(pc0, pc1, layers=layers).cpu().numpy()
    hist, ref_flow = fo.fit(layers, pc0, pc1, iters)
    got_hist = [v for _, v in eng.loss_history]
    assert len(got_hist) == iters
    for a, b in zip(got_hist[:10], hist[:10]):
        assert a == pytest.approx(b, rel=2e-3)                 # trajectories diverge slowly (different rounding), so early steps
    assert got_hist[-1] < 0.5 * got_hist[0] and hist[-1] < 0.5 * hist[0]
    assert got_hist[-1] == pytest.approx(hist[-1], rel=0.1)
    # identity poses: flow == network output; it should already point along the true translation (0.6, 0.2, 0)
    assert np.abs(np.median(flow, axis=0) - np.median(ref_flow, axis=0)).max() < 0.05


def test_flow_includes_ego_motion(gpu):
    from himo_amd.fastnsf import FastNSF
    pc0, pc1 = _scene(4, 3000, 3000)
    pose0, pose1 = np.eye(4), np.eye(4)
    pose1[0, 3] = 1.5                                          # ego moved 1.5 m forward between the sweeps
    eng = FastNSF(device=gpu, iters=0)
    flow = eng.fit(pc0, pc1, pose0, pose1).cpu().numpy()
    # with 0 iterations the MLP output is small but non-zero; flow - f(p') must be the pose flow (-1.5, 0, 0)
    f = eng.OUT[:, :3].cpu().numpy()
    assert np.abs((flow - f) - np.array([-1.5, 0, 0], np.float32)).max() < 1e-5


@pytest.mark.parametrize("objective", ["nn", "dt"])
def test_full_size_fit_is_finite_reproducible_and_reduces_the_objective(gpu, objective):
    """BASELINE config 4 at BASELINE size: one 120k-point sweep pair, 30 iterations.  The objective falls, the flow is
    finite and row-aligned with pc0, and a second fit from the same seed reproduces it (exact NN correspondences and
    fixed-order weight-gradient reductions; the Chamfer gradient's scatter half accumulates fixed point, the
    distance-transform objective has no scatter at all: both reproduce bit for bit)."""
    from himo_amd.fastnsf import FastNSF
    from himo_amd.synthetic import make_frame
    f = make_frame(805, n_points=120_000)
    pc0 = torch.from_numpy(f["pc0"][:, :3].copy()).to(gpu)
    pc1 = torch.from_numpy((f["pc0"][:, :3] + f["flow"]).astype(np.float32)).to(gpu)
    runs = []
    for _ in range(2):
        m = FastNSF(device=gpu, iters=30, seed=3, objective=objective)
        flow = m.fit(pc0, pc1, f["pose0"], f["pose1"])
        assert flow.shape == (120_000, 3) and torch.isfinite(flow).all()
        first, last = m.loss_history[0][1], m.loss_history[-1][1]
        assert np.isfinite(first) and last < 0.95 * first, m.loss_history      # a dense uniform cloud starts near its optimum
        runs.append((flow.clone(), last))
    assert runs[0][1] == pytest.approx(runs[1][1], rel=1e-4)
    assert (runs[0][0] - runs[1][0]).abs().max().item() <= 1e-3
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]


# ---- the distance-transform objective (csrc/dtloss.hip): what the config's `model=fastnsf` names ---------------------------------
DT_BOX = (-12.0, -10.0, -2.0, 12.0, 10.0, 2.0)


def _dt_scene(seed, n0, n1):
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-9, -7, -1.5], np.float32), np.array([9, 7, 1.5], np.float32)
    pc1 = rng.uniform(lo, hi, (n1, 3)).astype(np.float32)
    pc0 = (pc1[rng.integers(0, n1, n0)] - np.array([0.5, 0.15, 0.0], np.float32) + rng.normal(0, 0.03, (n0, 3))).astype(np.float32)
    pc0[:20] += np.array([40.0, 0, 0], np.float32)               # a few points far outside the volume: no loss, no gradient, not counted
    return pc0, pc1


@pytest.mark.parametrize("n1,cell", [(3_000, 0.1), (40, 0.1), (0, 0.2), (20_000, 0.25)])
def test_distance_transform_volume_equals_scipy_edt(gpu, n1, cell):
    """The three windowed min-plus passes against scipy's exact Euclidean distance transform of the same occupancy grid, capped
    at the window: sparse targets (large distances: the cap / INF handling), an empty target, a dense one."""
    import ctypes
    import fastnsf_oracle as fo
    from himo_amd import _lib
    from himo_amd.fastnsf import dt_grid
    lib = _lib.load()
    _, pc1 = _dt_scene(7, 10, max(n1, 1))
    pc1 = pc1[:n1]
    origin, dims, window = dt_grid(2.0, cell, DT_BOX)
    o_c, d_c = (ctypes.c_float * 3)(*origin.tolist()), (ctypes.c_int * 3)(*dims.tolist())
    vol = torch.empty(int(lib.himo_dt_volume_bytes(d_c)), dtype=torch.uint8, device=gpu)
    p1 = torch.from_numpy(pc1.reshape(-1, 3).copy()).to(gpu)
    _lib.check(lib.himo_dt_build(n1, p1.data_ptr() if n1 else None, o_c, cell, d_c, window, vol.data_ptr(), vol.numel(), _lib.stream_handle()))
    nx, ny, nz = (int(d) for d in dims)
    g = vol[: nx * ny * nz * 2].view(torch.int16).cpu().numpy().view(np.uint16).reshape(nz, ny, nx)
    got = np.minimum(np.sqrt(g.astype(np.float32)), np.float32(window)) * np.float32(cell)
    got[g == 0xFFFF] = np.float32(window) * np.float32(cell)
    ref = fo.dt_volume(pc1.reshape(-1, 3), origin, dims, cell, window)
    assert np.abs(got - ref).max() <= 1e-6 * max(ref.max(), 1.0)
    if n1:
        assert (g == 0).sum() > 0 and got.min() == 0.0


def test_fused_mlp_kernels_equal_the_layer_by_layer_products(gpu):
    """csrc/mlpfused.hip against the row-GEMM path it replaces (same split arithmetic, same packed weights): activations, output,
    loss and every gradient of one evaluation -- at a point count that is not a multiple of the 64-row block."""
    from himo_amd.fastnsf import FastNSF, init_mlp
    pc0, pc1 = _dt_scene(21, 10_037, 9_000)
    layers = init_mlp(9)
    runs = {}
    for fused in (False, True):
        eng = FastNSF(device=gpu, iters=1, lr=0.0, seed=3, objective="dt", dt_box=DT_BOX, precision="mixed", fused=fused, three_launch=False)
        eng.fit(pc0, pc1, layers=layers)
        runs[fused] = ([h.clone() for h in eng.H], eng.OUT.clone(), eng.loss_history[0][1], eng.flat_g.clone())
    (H0, out0, loss0, g0), (H1, out1, loss1, g1) = runs[False], runs[True]
    for k, (a, b) in enumerate(zip(H0, H1)):
        assert (a - b).abs().max().item() <= 2e-5 * max(a.abs().max().item(), 1.0), k       # same 22-bit products, different summation grouping
    assert (out0 - out1).abs().max().item() <= 1e-5 * max(out0.abs().max().item(), 1.0)
    assert loss1 == pytest.approx(loss0, rel=1e-5)
    assert (g0 - g1).abs().max().item() <= 5e-4 * g0.abs().max().item()


@pytest.mark.parametrize("precision,tol", [("f32", 2e-4), ("mixed", 1e-3)])
@pytest.mark.parametrize("n0,n1", [(6_000, 5_500), (50_000, 60_000)])
def test_dt_objective_loss_and_gradients_match_autograd(gpu, n0, n1, precision, tol):
    import fastnsf_oracle as fo
    from himo_amd.fastnsf import FastNSF, dt_grid, init_mlp
    pc0, pc1 = _dt_scene(11, n0, n1)
    layers = init_mlp(3)
    eng = FastNSF(device=gpu, iters=1, lr=0.0, seed=3, objective="dt", dt_box=DT_BOX, precision=precision)
    eng.fit(pc0, pc1, layers=layers)
    origin, dims, window = dt_grid(2.0, 0.1, DT_BOX)
    ref_loss, ref_grads, _, ref_gm = fo.dt_loss_and_grads(layers, pc0, pc1, origin, dims, 0.1, window)
    assert eng.loss_history[0][1] == pytest.approx(ref_loss, rel=1e-5 if precision == "f32" else 1e-4)
    gm = eng._gmoved.cpu().numpy()
    assert np.abs(gm - ref_gm).max() <= (1e-4 if precision == "f32" else 2e-3) * np.abs(ref_gm).max()
    assert np.all(gm[:20] == 0)                                   # the points far outside the volume
    for k, (gw, gb) in enumerate(ref_grads):
        cin, cout = gw.shape
        got_w, got_b = eng.gW[k].cpu().numpy()[:cin, :cout], eng.gb[k].cpu().numpy()[:cout]
        assert np.abs(got_w - gw).max() <= tol * max(np.abs(gw).max(), 1e-8), k
        assert np.abs(got_b - gb).max() <= tol * max(np.abs(gb).max(), 1e-8), k


def test_dt_fit_follows_the_cpu_restatement_and_recovers_the_motion(gpu):
    import fastnsf_oracle as fo
    from himo_amd.fastnsf import FastNSF, dt_grid, init_mlp
    pc0, pc1 = _dt_scene(12, 8_000, 8_000)
    layers = init_mlp(5)
    iters = 25
    eng = FastNSF(device=gpu, iters=iters, lr=1e-3, early_patience=10_000, objective="dt", dt_box=DT_BOX)
    flow = eng.fit(pc0, pc1, layers=layers).cpu().numpy()
    origin, dims, window = dt_grid(2.0, 0.1, DT_BOX)
    hist, ref_flow = fo.dt_fit(layers, pc0, pc1, origin, dims, 0.1, window, iters)
    got_hist = [v for _, v in eng.loss_history]
    assert len(got_hist) == iters
    for a, b in zip(got_hist[:10], hist[:10]):
        assert a == pytest.approx(b, rel=2e-3)
    assert got_hist[-1] < 0.7 * got_hist[0] and hist[-1] < 0.7 * hist[0]
    assert got_hist[-1] == pytest.approx(hist[-1], rel=0.1)
    assert np.abs(np.median(flow[20:], axis=0) - np.median(ref_flow[20:], axis=0)).max() < 0.05


@pytest.mark.parametrize("n0,n1", [(6_000, 5_500), (120_000, 119_000)])
def test_three_launch_iteration_equals_the_layer_kernels(gpu, n0, n1):
    """csrc/nsffused.hip (forward + objective | backward + every weight gradient | reduce + Adam + re-pack) against round 3's
    kernels (fused forward / backward + one split-K weight-gradient product per layer + Adam + re-pack): the same arithmetic
    formats, so one iteration's loss, objective gradient and parameter gradients agree to summation order, and a short fit stays
    on the same trajectory.  n0 = 6000 is 94 tiles: the padding to whole 4-tile blocks is exercised."""
    from himo_amd.fastnsf import FastNSF, init_mlp
    pc0, pc1 = _dt_scene(21, n0, n1)
    layers = init_mlp(4)
    a = FastNSF(device=gpu, iters=1, lr=0.0, objective="dt", dt_box=DT_BOX)
    b = FastNSF(device=gpu, iters=1, lr=0.0, objective="dt", dt_box=DT_BOX, three_launch=False)
    fa, fb = a.fit(pc0, pc1, layers=layers), b.fit(pc0, pc1, layers=layers)
    assert a.loss_history[0][1] == pytest.approx(b.loss_history[0][1], rel=1e-6)
    assert torch.equal(fa, fb)                                    # lr = 0: the field is the initial one in both; same forward arithmetic
    ga, gb = a._gmoved.cpu().numpy(), b._gmoved.cpu().numpy()
    assert np.abs(ga - gb).max() <= 1e-6 * np.abs(gb).max()
    for k in range(len(a.gW)):
        wa, wb = a.gW[k].cpu().numpy(), b.gW[k].cpu().numpy()
        ba, bb = a.gb[k].cpu().numpy(), b.gb[k].cpu().numpy()
        assert np.abs(wa - wb).max() <= 2e-4 * max(np.abs(wb).max(), 1e-12), k
        assert np.abs(ba - bb).max() <= 2e-4 * max(np.abs(bb).max(), 1e-12), k
    # the optimiser step + re-pack: ten iterations, same trajectory; and the run is bit-reproducible
    a = FastNSF(device=gpu, iters=10, lr=1e-3, objective="dt", dt_box=DT_BOX, early_patience=10_000)
    b = FastNSF(device=gpu, iters=10, lr=1e-3, objective="dt", dt_box=DT_BOX, early_patience=10_000, three_launch=False)
    fa, fb = a.fit(pc0, pc1, layers=layers), b.fit(pc0, pc1, layers=layers)
    for (_, x), (_, y) in zip(a.loss_history, b.loss_history):
        assert x == pytest.approx(y, rel=2e-3)
    assert a.loss_history[-1][1] < a.loss_history[0][1]
    assert np.abs(fa.cpu().numpy() - fb.cpu().numpy()).max() < 5e-2
    again = FastNSF(device=gpu, iters=10, lr=1e-3, objective="dt", dt_box=DT_BOX, early_patience=10_000).fit(pc0, pc1, layers=layers)
    assert torch.equal(again, fa)


def test_two_fits_in_flight_return_the_single_engine_bits(gpu):
    """fastnsf.OverlappedFastNSF: two engines on two HIP streams, the fit of pair k + 1 queued while pair k's runs (the product's and
    bench.py's default way through a stream of sweep pairs).  Each fit is the launch sequence of FastNSF.fit: the flows, the loss
    trajectories and the point counts are the single engine's, in order, for ragged pair sizes and an odd number of pairs."""
    from himo_amd.fastnsf import FastNSF, OverlappedFastNSF
    from himo_amd.synthetic import make_frame
    pairs = []
    for i in range(5):
        f = make_frame(830 + i, n_points=30_000 - 1777 * i)
        p0 = torch.from_numpy(f["pc0"][:, :3].copy()).to(gpu)
        p1 = torch.from_numpy((f["pc0"][:, :3] + f["flow"]).astype(np.float32)).to(gpu)
        pairs.append((p0, p1, f["pose0"], f["pose1"]))
    one = FastNSF(device=gpu, iters=12, seed=3)
    ref = []
    for p in pairs:
        flow = one.fit(*p)
        ref.append((flow.clone(), list(one.loss_history), one.points_in_volume))
    two = OverlappedFastNSF(device=gpu, engines=2, iters=12, seed=3)
    got = []
    for k, flow in enumerate(two.fits(iter(pairs))):
        eng = two.engines[k % 2]
        got.append((flow.clone(), list(eng.loss_history), eng.points_in_volume))
    assert len(got) == len(ref)
    for (fa, la, ca), (fb, lb, cb) in zip(ref, got):
        assert torch.equal(fa, fb) and la == lb and ca == cb
    # submit / collect by hand, and a drained object can be reused
    k, flow = two.submit(*pairs[0])
    two.drain()
    assert torch.equal(flow, ref[0][0]) and two.engines[k].loss_history == ref[0][1]


def test_save_program_runs_the_fastnsf_baseline_over_a_dataset(gpu, tmp_path):
    """``python save.py model=fastnsf dataset_path=...`` (README.md:50-53) as ``himo_amd.save.main(model="fastnsf")``: every sweep with a
    successor gets a flow under ``fastnsf`` in the dataset, row-aligned with pc0, and it is the flow FastNSF.fit returns for that pair."""
    from himo_amd import save
    from himo_amd.dataset import NpzDataset
    from himo_amd.fastnsf import FastNSF
    from himo_amd.synthetic import make_frame
    frames = [make_frame(860 + i, n_points=9_000 - 500 * i, scene_id="s0" if i < 3 else "s1") for i in range(5)]
    NpzDataset.write(tmp_path, frames)
    done = save.main(dataset_path=str(tmp_path), model="fastnsf", iters=8)
    assert done == 3                                          # two scenes of 3 and 2 sweeps: the last sweep of each has no successor
    ds = NpzDataset(tmp_path, vis_name="fastnsf")
    one = FastNSF(device=gpu, iters=8)
    seen = 0
    for i in range(len(ds)):
        f = ds[i]
        if "fastnsf" not in f:
            continue
        seen += 1
        assert f["fastnsf"].shape == (len(f["pc0"]), 3) and f["fastnsf"].dtype == np.float32
        nxt = ds[i + 1]
        want = one.fit(f["pc0"][:, :3], nxt["pc0"][:, :3], f["pose0"], f["pose1"]).cpu().numpy()
        assert np.array_equal(f["fastnsf"], want)
    assert seen == 3
    with pytest.raises(ValueError):
        save.main(dataset_path=str(tmp_path), model="nsfp")
