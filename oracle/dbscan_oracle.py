"""CPU ORACLE for the self-supervised cluster labels (himo_amd/seflow/ssl_label.py, csrc/dbscan.hip) -- test infrastructure, NOT
product code.  PARITY UNPINNED: the reference's ``ssl_label=seflow_auto`` generator is in the absent OpenSceneFlow submodule
(assets/slurm/ssl-train-av2.sh:32 names the option only).  The clustering itself is pinned against an independent
implementation: sklearn.cluster.DBSCAN (1.7.2 in the build container) for the core points and their partition; the rule this
build adds for border points -- DBSCAN leaves their cluster to the processing order -- is restated here with cKDTree."""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def dbscan(points: np.ndarray, eps: float, min_pts: int, skip: np.ndarray | None = None) -> np.ndarray:
    """int32 labels: 0 noise / skipped, 1 .. K clusters in the order of their lowest point index; border points join the
    neighbouring cluster of lowest such index.  Distances in float32 arithmetic like the product (dx*dx + dy*dy + dz*dz <= eps*eps)."""
    from sklearn.cluster import DBSCAN
    pts = np.ascontiguousarray(points[:, :3], dtype=np.float32)
    n = len(pts)
    take = np.ones(n, bool) if skip is None else ~np.asarray(skip, bool)
    take &= np.isfinite(pts).all(1)
    idx = np.flatnonzero(take)
    labels = np.zeros(n, np.int32)
    if len(idx) == 0:
        return labels
    sub = pts[idx]
    # neighbour lists with the product's float32 distance rule (cKDTree pre-selects generously in float64)
    tree = cKDTree(sub.astype(np.float64))
    cand = tree.query_ball_point(sub.astype(np.float64), r=float(eps) * 1.001 + 1e-6)
    e2 = np.float32(eps) * np.float32(eps)
    neigh = []
    for i, c in enumerate(cand):
        c = np.asarray(c, np.int64)
        d = sub[c] - sub[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        neigh.append(c[d2 <= e2])
    core = np.array([len(v) >= min_pts for v in neigh])
    # sklearn on a precomputed sparse graph of exactly these neighbourhoods: its core set and core partition must be ours
    from scipy.sparse import csr_matrix
    rows = np.concatenate([np.full(len(v), i) for i, v in enumerate(neigh)])
    cols = np.concatenate(neigh)
    graph = csr_matrix((np.ones(len(rows), np.float32) * 0.5, (rows, cols)), shape=(len(sub), len(sub)))
    graph.setdiag(0.0)
    sk = DBSCAN(eps=1.0, min_samples=min_pts, metric="precomputed").fit(graph)
    sk_core = np.zeros(len(sub), bool)
    sk_core[sk.core_sample_indices_] = True
    assert np.array_equal(sk_core, core), "core sets differ from sklearn's"
    # canonical cluster of a core point = lowest (original) index of its sklearn cluster
    root = np.full(len(sub), -1, np.int64)
    for lab in np.unique(sk.labels_[core]):
        members = np.flatnonzero(core & (sk.labels_ == lab))
        root[members] = idx[members].min()
    for i in np.flatnonzero(~core):
        cn = neigh[i][core[neigh[i]]]
        if len(cn):
            root[i] = root[cn].min()
    roots = np.unique(root[root >= 0])
    rank = {r: k + 1 for k, r in enumerate(roots)}
    labels[idx] = [rank[r] if r >= 0 else 0 for r in root]
    return labels


def auto_labels(pc0, pc1, ground0, ground1, pose0, pose1, eps, min_pts, dyn_dist, range_net=51.2, ref_range=30.0):
    """himo_amd/seflow/ssl_label.py steps 0-3: points outside the network's BEV range take no part; a non-ground point is a
    dynamic candidate when its nearest usable point of the other sweep is further than dyn_dist * max(1, r / ref_range)"""
    T = (np.linalg.inv(np.asarray(pose1, np.float64)) @ np.asarray(pose0, np.float64)).astype(np.float32)
    a = (pc0[:, :3].astype(np.float32) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    b = pc1[:, :3].astype(np.float32)
    use_a = ~(np.asarray(ground0, bool) | (np.abs(a[:, :2]).max(axis=1, initial=0.0) > np.float32(range_net)))
    use_b = ~(np.asarray(ground1, bool) | (np.abs(b[:, :2]).max(axis=1, initial=0.0) > np.float32(range_net)))
    far2, inv_ref2 = np.float32(float(dyn_dist) ** 2), np.float32(1.0 / (ref_range * ref_range))
    out = []
    for pts, use, other, ouse in ((a, use_a, b, use_b), (b, use_b, a, use_a)):
        skip = ~use
        mine, oth = pts[use], other[ouse]
        if len(mine) and len(oth):
            _, j = cKDTree(oth.astype(np.float64)).query(mine.astype(np.float64))
            d = mine - oth[j]
            d2 = (d * d).sum(1)
            r2 = mine[:, 0] * mine[:, 0] + mine[:, 1] * mine[:, 1]
            bar2 = far2 * np.maximum(r2 * inv_ref2, np.float32(1.0))
            skip[np.flatnonzero(use)] = d2 <= bar2
        out.append((dbscan(pts, eps, min_pts, skip), skip))
    return out
