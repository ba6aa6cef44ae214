"""Self-supervised cluster labels on the GPU (himo_amd/seflow/ssl_label.py, csrc/dbscan.hip: the launcher's ``+ssl_label=seflow_auto``,
assets/slurm/ssl-train-av2.sh:32) against the CPU oracle (oracle/dbscan_oracle.py: sklearn.cluster.DBSCAN for the core points and
their partition, this build's border rule restated with cKDTree).  PARITY UNPINNED vs the reference (generator absent); the
clustering itself is pinned against sklearn: labels must be EQUAL, not just equal up to renumbering -- both number clusters by
their lowest point index."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def do():
    import dbscan_oracle
    return dbscan_oracle


def _blobs(seed, n, n_blobs, spread=0.25, box=45.0):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(5, 400, n_blobs)
    parts = [rng.normal(rng.uniform(-box, box, 3) * np.array([1, 1, 0.05]), spread * rng.uniform(0.5, 2.0), (int(k), 3)) for k in sizes]
    noise = rng.uniform([-box, -box, -2.5], [box, box, 2.5], (max(n - int(sizes.sum()), 0), 3))
    pts = np.concatenate(parts + [noise]).astype(np.float32)
    return pts[rng.permutation(len(pts))]


@pytest.mark.parametrize("n,n_blobs,eps,min_pts", [(4_000, 12, 0.5, 8), (30_000, 60, 0.4, 5), (120_000, 150, 0.5, 8), (2_000, 3, 1.0, 20)])
def test_dbscan_labels_equal_sklearn(gpu, do, n, n_blobs, eps, min_pts):
    from himo_amd.seflow.ssl_label import dbscan
    pts = _blobs(n, n, n_blobs)
    skip = np.random.default_rng(n + 1).uniform(size=len(pts)) < 0.1
    got, k = dbscan(torch.from_numpy(pts).to(gpu), eps, min_pts, torch.from_numpy(skip).to(gpu))
    got = got.cpu().numpy()
    want = do.dbscan(pts, eps, min_pts, skip)
    assert int(k.item()) == want.max() > 0
    assert np.array_equal(got, want), (np.flatnonzero(got != want)[:10], got[got != want][:10], want[got != want][:10])
    assert (got[skip] == 0).all()
    again, _ = dbscan(torch.from_numpy(pts).to(gpu), eps, min_pts, torch.from_numpy(skip).to(gpu))
    assert torch.equal(again.cpu(), torch.from_numpy(got))        # a pure function of the input (atomics inside, fixed result)


def test_dbscan_edge_cases(gpu, do):
    from himo_amd.seflow.ssl_label import dbscan
    empty, k = dbscan(torch.zeros((0, 3), device=gpu))
    assert empty.shape == (0,) and int(k.item()) == 0
    # all noise; everything skipped; points beyond the cell grid (binned into its border cells); NaN rows; a strided (N,4) input
    rng = np.random.default_rng(3)
    far = np.concatenate([rng.normal([80, -70, 0], 0.2, (50, 3)), rng.normal([-90, 95, 1], 0.2, (40, 3)), rng.uniform(-50, 50, (200, 3))]).astype(np.float32)
    got = dbscan(torch.from_numpy(far).to(gpu), 0.5, 8)[0].cpu().numpy()
    assert np.array_equal(got, do.dbscan(far, 0.5, 8)) and got.max() == 2
    assert dbscan(torch.from_numpy(far).to(gpu), 0.5, 8, torch.ones(len(far), dtype=torch.bool, device=gpu))[0].abs().sum().item() == 0
    sparse = rng.uniform(-50, 50, (500, 3)).astype(np.float32)
    assert dbscan(torch.from_numpy(sparse).to(gpu), 0.3, 5)[0].abs().sum().item() == 0
    bad = far.copy(); bad[::7, 1] = np.nan
    assert np.array_equal(dbscan(torch.from_numpy(bad).to(gpu), 0.5, 8)[0].cpu().numpy(), do.dbscan(bad, 0.5, 8))
    xyzi = np.concatenate([far, rng.uniform(size=(len(far), 1)).astype(np.float32)], 1)
    assert np.array_equal(dbscan(torch.from_numpy(xyzi).to(gpu), 0.5, 8)[0].cpu().numpy(), got)


def test_auto_labels_of_a_sweep_pair(gpu, do):
    """The whole generator on a BASELINE-size pair: ring-cloud background (static: the second sweep re-observes it), moving
    box instances, ground masks, ego motion.  Against the oracle fed the SAME ego-compensated points (the float32 transform
    is the product's own kernel, pinned elsewhere), labels equal; and the labels mean something: points of fast instances
    are labelled dynamic an order of magnitude more often than background points (not all of them: the interior of a solid box
    that moved by less than its length still finds returns of the same box nearby -- the nearest-neighbour residual sees the
    trailing and leading parts of an object, which is what a cluster label needs)."""
    from himo_amd.seflow.ssl_label import DYN_DIST, EPS, MIN_PTS, _moved, auto_labels
    from himo_amd.synthetic import make_frame
    f0 = make_frame(70, n_points=120_000, cloud="rings")
    rng = np.random.default_rng(5)
    # the next sweep: the same static world seen from pose1 (re-sampled with 3 cm noise), instances moved by their flow
    ego = np.linalg.inv(f0["pose1"]) @ f0["pose0"]
    moved = f0["pc0"][:, :3].astype(np.float64) + f0["flow"].astype(np.float64)
    keep = rng.uniform(size=len(moved)) < 0.97
    pc1 = (moved[keep] + rng.normal(0, 0.03, (int(keep.sum()), 3))).astype(np.float32)
    gm1 = f0["gm0"][keep]
    l0, l1 = auto_labels(f0["pc0"], pc1, f0["gm0"], gm1, f0["pose0"], f0["pose1"])
    a = _moved(torch.from_numpy(f0["pc0"]).to(gpu), ego).cpu().numpy()
    (w0, s0), (w1, s1) = do.auto_labels(np.concatenate([a, f0["pc0"][:, 3:]], 1), pc1, f0["gm0"], gm1, np.eye(4), np.eye(4), EPS, MIN_PTS, DYN_DIST)
    g0, g1 = l0.cpu().numpy(), l1.cpu().numpy()
    assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
    inst = f0["flow_instance_id"] > 0
    speed = np.linalg.norm(f0["flow"] - (a - f0["pc0"][:, :3]), axis=1) / 0.1
    fast = inst & (speed > 8.0) & ~f0["gm0"]
    assert fast.sum() > 500 and (g0[fast] > 0).mean() > 0.2
    assert (g0[~inst] > 0).mean() < 0.02 and (g0[f0["gm0"]] == 0).all()


def test_auto_labels_of_a_200_m_sweep_pair(gpu, do):
    """an AV2-like sweep reaches 200 m.  Points beyond the network's +-51.2 m take no part (label 0, not even as
    neighbours); inside, the dynamic-candidate bar grows with range, so the sparse far field is not declared dynamic wholesale."""
    from himo_amd.seflow.ssl_label import DYN_DIST, EPS, MIN_PTS, RANGE_NET, _moved, auto_labels
    from himo_amd.synthetic import make_frame
    f0 = make_frame(71, n_points=120_000, cloud="rings")
    f0["pc0"][:, :3] *= 2.9                                   # walls out to ~200 m, ring spacing x 2.9
    f0["flow"] *= 2.9
    rng = np.random.default_rng(6)
    ego = np.linalg.inv(f0["pose1"]) @ f0["pose0"]
    moved = f0["pc0"][:, :3].astype(np.float64) + f0["flow"].astype(np.float64)
    keep = rng.uniform(size=len(moved)) < 0.97
    pc1 = (moved[keep] + rng.normal(0, 0.03, (int(keep.sum()), 3))).astype(np.float32)
    gm0 = f0["pc0"][:, 2] < -1.8 * 2.9 + 0.3
    gm1 = gm0[keep]
    l0, l1 = auto_labels(f0["pc0"], pc1, gm0, gm1, f0["pose0"], f0["pose1"])
    a = _moved(torch.from_numpy(f0["pc0"]).to(gpu), ego).cpu().numpy()
    (w0, s0), (w1, s1) = do.auto_labels(np.concatenate([a, f0["pc0"][:, 3:]], 1), pc1, gm0, gm1, np.eye(4), np.eye(4), EPS, MIN_PTS, DYN_DIST)
    g0, g1 = l0.cpu().numpy(), l1.cpu().numpy()
    assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
    outside = np.abs(a[:, :2]).max(axis=1) > RANGE_NET
    assert outside.mean() > 0.3 and (g0[outside] == 0).all()
    inst = f0["flow_instance_id"] > 0
    assert (g0[~inst & ~outside] > 0).mean() < 0.08                 # the static world stays static although its sampling is 3x sparser (measured 0.052)


def test_training_loop_with_generated_labels(gpu, tmp_path):
    """``fit(..., ssl_label="seflow_auto")`` (the default, the launcher's option): one short run end to end -- labels generated per
    sample on the device, finite losses, a checkpoint."""
    from himo_amd.dataset import ListDataset
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import fit, make_sample, triplets
    from himo_amd.synthetic import make_frame
    frames = [make_frame(40 + i, n_points=6_000, scene_id="s") for i in range(5)]
    ds = ListDataset(frames)
    smp = make_sample(ds, triplets(ds)[0], gpu, "seflow_auto")
    assert smp[6].dtype == torch.int32 and smp[6].shape == (6_000,) and smp[8] == int(max(smp[6].max(), smp[7].max())) + 1
    out = fit(ds, spec.init_params(1), out_dir=tmp_path, epochs=2, batch_size=2, lr=1e-4, max_points=6_000, device=gpu, log=None)
    assert all(np.isfinite(h["train_loss"]) for h in out["history"]) and list(tmp_path.glob("*.npz"))
