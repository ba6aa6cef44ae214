"""GPU parity of a11 (KNN/Chamfer correspondences + self-supervised loss terms and gradient) against the
CPU restatement (PARITY UNPINNED: own spec, see csrc/sslloss.hip)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(seed, n0, n1, n_clusters=12):
    from himo_amd.synthetic import make_frame
    a, b = make_frame(seed, n_points=n0, n_instances=n_clusters), make_frame(seed + 1, n_points=n1, n_instances=n_clusters)
    rng = np.random.default_rng(seed)
    lab0 = a["flow_instance_id"].astype(np.int32)          # 0 = background, k > 0 = instance k
    lab1 = (b["flow_instance_id"] > 0).astype(np.int32) * 7
    flow = (a["flow"] + rng.normal(0, 0.05, a["flow"].shape)).astype(np.float32)
    # make pc1 contain the moved instances so that dynamic correspondences exist
    pc1 = b["pc0"][:, :3].copy()
    k = min(len(pc1), len(a["pc0"])) // 4
    pc1[:k] = a["pc0"][:k, :3] + a["flow"][:k]
    lab1[:k] = (lab0[:k] > 0) * 7
    return a["pc0"][:, :3].copy(), pc1, flow, lab0, lab1


@pytest.mark.parametrize("nq,nr", [(1, 1), (100, 3), (5000, 7000), (120_000, 120_000)])
def test_nn_grid_is_exact(gpu, nq, nr):
    from himo_amd.ssl_loss import nn_grid
    import sslloss_oracle as so
    rng = np.random.default_rng(nq + nr)
    q = rng.uniform([-60, -60, -3], [60, 60, 3], (nq, 3)).astype(np.float32)     # some points outside the 104 m grid
    r = rng.uniform([-60, -60, -3], [60, 60, 3], (nr, 3)).astype(np.float32)
    d2, idx = nn_grid(torch.from_numpy(q).to(gpu), torch.from_numpy(r).to(gpu))
    ref_d2, ref_i = so.nearest(q, r)
    got_d2 = d2.cpu().numpy()
    assert np.abs(got_d2 - ref_d2).max() <= 1e-4 * max(1.0, ref_d2.max())
    diff = q - r[idx.cpu().numpy().astype(np.int64)]
    assert np.allclose((diff * diff).sum(1), got_d2, rtol=1e-5, atol=1e-6)       # the index really is that neighbour
    exact = (idx.cpu().numpy() == ref_i).mean()
    assert exact >= 0.999


def test_nn_grid_matches_brute_force_kernel_including_ties(gpu):
    from himo_amd.eval import nearest_neighbor
    from himo_amd.ssl_loss import nn_grid
    rng = np.random.default_rng(3)
    r = rng.uniform(-50, 50, (3000, 3)).astype(np.float32)
    r = np.concatenate([r, r])                               # every reference point twice: ties everywhere
    q = rng.uniform(-50, 50, (4000, 3)).astype(np.float32)
    d2, idx = nn_grid(torch.from_numpy(q).to(gpu), torch.from_numpy(r).to(gpu))
    bd, bi = nearest_neighbor(torch.from_numpy(q).to(gpu), torch.from_numpy(r).to(gpu))
    assert torch.equal(idx, bi) and (idx < 3000).all()       # lowest index wins in both kernels


def _density_case(name):
    """query / reference clouds that break a one-lane-per-query grid walk: crowded cells, far walks, empty grids"""
    from himo_amd.synthetic import make_frame
    rng = np.random.default_rng(11)
    if name == "rings":                     # LiDAR-shaped sweeps: ~600 points per cell near the sensor; walls the other sweep lacks
        q, r = (make_frame(s, cloud="rings")["pc0"][:, :3].copy() for s in (0, 1))
    elif name == "crowded":                 # 100k points in 2 m^2 (four cells), every fourth reference point duplicated: ties
        r = rng.uniform([10.0, 10.0, -1.0], [11.4142, 11.4142, 1.0], (100_000, 3)).astype(np.float32)
        r[25_000:50_000] = r[:25_000]
        q = rng.uniform([9.5, 9.5, -1.0], [12.0, 12.0, 1.0], (100_000, 3)).astype(np.float32)
    elif name == "far_apart":               # the two sets in opposite corners, partly outside the grid: every ring but the last is empty
        q = rng.uniform([-60.0, -60.0, -3.0], [-40.0, -40.0, 3.0], (20_000, 3)).astype(np.float32)
        r = rng.uniform([40.0, 40.0, -3.0], [60.0, 60.0, 3.0], (20_000, 3)).astype(np.float32)
    elif name == "one_row":                 # everything on a line along x: one grid row holds all points
        q = np.stack([rng.uniform(-50, 50, 30_000), np.full(30_000, 0.5), rng.uniform(-1, 1, 30_000)], 1).astype(np.float32)
        r = np.stack([rng.uniform(-50, 50, 30_000), np.full(30_000, 0.5), rng.uniform(-1, 1, 30_000)], 1).astype(np.float32)
    elif name == "sparse_queries":          # a handful of queries scattered over the grid against a full sweep
        q = rng.uniform([-60, -60, -3], [60, 60, 3], (37, 3)).astype(np.float32)
        r = make_frame(2, cloud="rings")["pc0"][:, :3].copy()
    else:
        raise ValueError(name)
    return q, r


@pytest.mark.parametrize("case", ["rings", "crowded", "far_apart", "one_row", "sparse_queries"])
def test_nn_grid_is_exact_whatever_the_density(gpu, case):
    """the same ROWS as the exhaustive kernel (csrc/nn.hip, pinned bit-for-bit against cKDTree in test_eval_gpu.py), ties
    included, and cKDTree's distances"""
    from himo_amd.eval import nearest_neighbor
    from himo_amd.ssl_loss import nn_grid
    import sslloss_oracle as so
    q, r = _density_case(case)
    tq, tr = torch.from_numpy(q).to(gpu), torch.from_numpy(r).to(gpu)
    d2, idx = nn_grid(tq, tr)
    bd, bi = nearest_neighbor(tq, tr)
    assert torch.equal(idx, bi.to(idx.dtype))
    diff = q.astype(np.float64) - r[idx.cpu().numpy().astype(np.int64)].astype(np.float64)
    ref_d2, _ = so.nearest(q, r)
    assert np.allclose((diff * diff).sum(1), ref_d2.astype(np.float64), rtol=1e-5, atol=1e-9)   # that row IS a nearest neighbour
    assert np.allclose(d2.cpu().numpy(), ref_d2, rtol=1e-5, atol=1e-9)


def test_nn_grid_does_not_depend_on_the_cell_size(gpu):
    """finer / coarser / non-power-of-two cells return the same rows (the ring bound is conservative for any geometry)"""
    import ctypes
    from himo_amd import _lib
    q, r = _density_case("rings")
    q, r = q[:40_000], r[:40_000]
    tq, tr = torch.from_numpy(q).to(gpu), torch.from_numpy(r).to(gpu)
    lib = _lib.load()
    rows = []
    for cell, w in [(1.0, 104), (0.5, 208), (0.3, 347), (4.0, 26), (104.0, 1)]:
        d2 = torch.empty(len(q), dtype=torch.float32, device=gpu)
        idx = torch.empty(len(q), dtype=torch.int32, device=gpu)
        ws = torch.empty(int(lib.himo_nn_grid_workspace_bytes(max(len(q), len(r)), w, w)), dtype=torch.uint8, device=gpu)
        _lib.check(lib.himo_nn_grid(len(q), _lib.ptr(tq), len(r), _lib.ptr(tr), -52.0, -52.0, cell, w, w, _lib.ptr(d2), _lib.ptr(idx),
                                    _lib.ptr(ws), ws.numel(), _lib.stream_handle()), "himo_nn_grid")
        rows.append((idx.clone(), d2.clone()))
    for idx, d2 in rows[1:]:
        assert torch.equal(idx, rows[0][0]) and torch.equal(d2, rows[0][1])


@pytest.mark.parametrize("n0,n1", [(4000, 3500), (30_000, 32_000), (120_000, 118_000)])      # incl. BASELINE size
def test_loss_terms_and_gradient(gpu, n0, n1):
    import sslloss_oracle as so
    from himo_amd.ssl_loss import SeFlowLoss
    pc0, pc1, flow, lab0, lab1 = _scene(5, n0, n1)
    terms, total, grad = SeFlowLoss()(torch.from_numpy(pc0), torch.from_numpy(pc1), torch.from_numpy(flow),
                                      torch.from_numpy(lab0), torch.from_numpy(lab1))
    ref_terms, ref_total, ref_grad = so.ssl_loss(pc0, pc1, flow, lab0, lab1)
    for k, v in ref_terms.items():
        assert float(terms[k]) == pytest.approx(v, rel=2e-5, abs=1e-7), k
        assert v > 0, k                                      # every term is exercised
    assert float(total) == pytest.approx(ref_total, rel=2e-5)
    g = grad.cpu().numpy()
    assert np.abs(g - ref_grad).max() <= 1e-6 + 1e-4 * np.abs(ref_grad).max()


def test_loss_edge_cases(gpu):
    from himo_amd.ssl_loss import SeFlowLoss
    eng = SeFlowLoss()
    pc0, pc1, flow, lab0, lab1 = _scene(9, 2000, 1800)
    z0, z1 = np.zeros_like(lab0), np.zeros_like(lab1)
    terms, total, grad = eng(torch.from_numpy(pc0), torch.from_numpy(pc1), torch.from_numpy(flow), torch.from_numpy(z0), torch.from_numpy(z1))
    assert float(terms["dynamic_chamfer_dis"]) == 0 and float(terms["cluster_based_pc0pc1"]) == 0   # no dynamic points
    assert float(terms["static_flow_loss"]) > 0 and torch.isfinite(grad).all()
    zero = torch.zeros_like(torch.from_numpy(flow))
    terms, total, grad = eng(torch.from_numpy(pc0), torch.from_numpy(pc1), zero, torch.from_numpy(lab0), torch.from_numpy(lab1))
    assert float(terms["static_flow_loss"]) == 0 and torch.isfinite(grad).all()                       # |0| has zero sub-gradient


def test_autograd_wrapper(gpu):
    from himo_amd.ssl_loss import SeFlowLoss, seflow_loss
    pc0, pc1, flow, lab0, lab1 = (torch.from_numpy(x).to(gpu) for x in _scene(11, 3000, 3000))
    f = flow.clone().requires_grad_(True)
    loss = seflow_loss(f, pc0, pc1, lab0, lab1)
    (loss * 2).backward()
    _, total, grad = SeFlowLoss()(pc0, pc1, flow, lab0, lab1)
    assert torch.allclose(f.grad, 2 * grad, rtol=1e-5, atol=1e-7)


def test_loss_with_the_raw_correspondences_supplied_is_the_same_loss(gpu):
    """himo_ssl_loss_ex: the pc0 -> pc1 correspondences of the cluster term computed ahead (SeFlowLoss.raw_neighbours; the training
    step searches them on a side stream under its forward pass) -- every term and the gradient bit-identical to the all-in-one call;
    a half-given or mis-sized pair is refused."""
    from himo_amd import _lib
    from himo_amd.ssl_loss import SeFlowLoss
    pc0, pc1, flow, lab0, lab1 = _scene(41, 20_000, 18_500)
    t = lambda a: torch.from_numpy(a).to(gpu)
    eng = SeFlowLoss(device=gpu)
    terms, total, grad = eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1))
    d2, idx, _keep = eng.raw_neighbours(t(pc0), t(pc1))
    terms2, total2, grad2 = eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1), raw=(d2, idx))
    assert total.item() == total2.item() and torch.equal(grad, grad2)
    assert all(terms[k].item() == terms2[k].item() for k in terms)
    with pytest.raises(ValueError):
        eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1), raw=(d2[:-1], idx[:-1]))
    lib = _lib.load()
    ws = torch.empty(int(lib.himo_ssl_loss_workspace_bytes(10, 10, 2, 104, 104)), dtype=torch.uint8, device=gpu)
    z = torch.zeros(64, dtype=torch.float32, device=gpu)
    zi = torch.zeros(64, dtype=torch.int32, device=gpu)
    loss = torch.zeros(5, dtype=torch.float64, device=gpu)
    st = lib.himo_ssl_loss_ex(10, 10, z.data_ptr(), z.data_ptr(), z.data_ptr(), zi.data_ptr(), zi.data_ptr(), 2, -52.0, -52.0, 1.0, 104, 104,
                              z.data_ptr(), None, loss.data_ptr(), z.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_handle())
    assert st != 0                                            # distances without indices: invalid argument


@pytest.mark.gpu
def test_loss_with_the_dynamic_subset_sizes_supplied_is_the_same_loss_and_never_blocks(gpu):
    """himo_ssl_loss_presized: the sizes of the two dynamic subsets counted ahead (SeFlowLoss.dyn_sizes; the training step counts them on a
    side stream under its forward pass) -- every term and the gradient bit-identical to the call that copies them back itself, with and
    without the raw correspondences; sizes that do not match the labels give NaN, sizes out of range are refused; edge cases: no
    dynamic point on one side, an empty pc1."""
    from himo_amd import _lib
    from himo_amd.ssl_loss import SeFlowLoss
    pc0, pc1, flow, lab0, lab1 = _scene(43, 20_000, 18_500)
    t = lambda a: torch.from_numpy(a).to(gpu)
    eng = SeFlowLoss(device=gpu)
    terms, total, grad = eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1), n_labels=int(lab0.max()) + 1)
    sizes = eng.dyn_sizes(t(lab0), t(lab1))
    sizes[0].synchronize()
    assert sizes[1].tolist() == [int((lab0 > 0).sum()), int((lab1 > 0).sum())]
    terms2, total2, grad2 = eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1), n_labels=int(lab0.max()) + 1, sizes=sizes[:2])
    assert total.item() == total2.item() and torch.equal(grad, grad2)
    assert all(terms[k].item() == terms2[k].item() for k in terms)
    d2, idx, _keep = eng.raw_neighbours(t(pc0), t(pc1))
    sizes = eng.dyn_sizes(t(lab0), t(lab1))
    terms3, total3, grad3 = eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1), n_labels=int(lab0.max()) + 1, raw=(d2, idx), sizes=sizes[:2])
    assert total.item() == total3.item() and torch.equal(grad, grad3)
    # wrong sizes: loud
    done, host, _ = eng.dyn_sizes(t(lab0), t(lab1))
    done.synchronize()
    wrong = host.clone(); wrong[0] -= 1
    _, total4, _ = eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1), n_labels=int(lab0.max()) + 1, sizes=(done, wrong))
    assert np.isnan(total4.item())
    wrong[0] = pc0.shape[0] + 1
    with pytest.raises(Exception):
        eng(t(pc0), t(pc1), t(flow), t(lab0), t(lab1), n_labels=int(lab0.max()) + 1, sizes=(done, wrong))
    # no dynamic point in pc1; an empty pc1
    for l1, p1 in ((np.zeros_like(lab1), pc1), (lab1[:0], pc1[:0])):
        a = eng(t(pc0), t(p1), t(flow), t(lab0), t(l1), n_labels=int(lab0.max()) + 1)
        sizes = eng.dyn_sizes(t(lab0), t(l1))
        b = eng(t(pc0), t(p1), t(flow), t(lab0), t(l1), n_labels=int(lab0.max()) + 1, sizes=sizes[:2])
        assert a[1].item() == b[1].item() and torch.equal(a[2], b[2])
