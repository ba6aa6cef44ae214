"""CPU ORACLE for the self-supervised loss terms (stage a11) -- test infrastructure, NOT product code.

PARITY UNPINNED: the reference's `seflowppLoss` is in the absent OpenSceneFlow submodule; only
the term names and unit weights are in the tree (assets/slurm/ssl-train-av2.sh:33).  This restates THIS BUILD'S
specification (himo_amd/csrc/sslloss.hip header) with PyTorch CPU ops (autograd for the gradient) and scipy's
cKDTree for the correspondences.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial import cKDTree


def nearest(query: np.ndarray, ref: np.ndarray):
    """float32 squared distances + indices, exact (k=1)."""
    d, i = cKDTree(ref.astype(np.float64)).query(query.astype(np.float64), k=1)
    diff = query.astype(np.float32) - ref.astype(np.float32)[i]
    return (diff * diff).sum(1), i


def ssl_loss(pc0, pc1, flow, label0, label1):
    """-> ({term: float}, total float, grad (N0,3) float32)."""
    p0 = torch.from_numpy(np.ascontiguousarray(pc0[:, :3], dtype=np.float32))
    p1 = torch.from_numpy(np.ascontiguousarray(pc1[:, :3], dtype=np.float32))
    f = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32)).clone().requires_grad_(True)
    l0 = torch.from_numpy(np.asarray(label0).astype(np.int64))
    l1 = torch.from_numpy(np.asarray(label1).astype(np.int64))
    moved = p0 + f
    mv = moved.detach().numpy()

    def chamfer(a_t, a_np, b_t, b_np):
        if len(a_np) == 0 or len(b_np) == 0:
            return torch.zeros((), dtype=torch.float64)
        _, ia = nearest(a_np, b_np)
        _, ib = nearest(b_np, a_np)
        da = ((a_t - b_t[torch.from_numpy(ia)]) ** 2).sum(1)
        db = ((b_t - a_t[torch.from_numpy(ib)]) ** 2).sum(1)
        return da.double().mean() + db.double().mean()

    terms = {}
    terms["chamfer_dis"] = chamfer(moved, mv, p1, p1.numpy())
    st = l0 == 0
    terms["static_flow_loss"] = torch.linalg.vector_norm(f[st], dim=-1).double().mean() if st.any() else torch.zeros((), dtype=torch.float64)
    d0, d1 = l0 > 0, l1 > 0
    terms["dynamic_chamfer_dis"] = chamfer(moved[d0], mv[d0.numpy()], p1[d1], p1[d1].numpy())
    # cluster term
    norms = []
    if len(p0) and len(p1):
        raw_d, raw_i = nearest(p0.numpy(), p1.numpy())
        for lab in torch.unique(l0).tolist():
            if lab <= 0:
                continue
            members = torch.nonzero(l0 == lab).squeeze(1).numpy()
            ok = l1.numpy()[raw_i[members]] > 0
            if not ok.any():
                continue
            cand = members[ok]
            dmax = raw_d[cand].max()
            anchor = cand[raw_d[cand] == dmax].min()                   # ties: lowest index
            target = p1[raw_i[anchor]] - p0[anchor]
            norms.append(torch.linalg.vector_norm(f[torch.from_numpy(members)] - target, dim=-1))
    terms["cluster_based_pc0pc1"] = torch.cat(norms).double().mean() if norms else torch.zeros((), dtype=torch.float64)
    total = sum(terms.values())
    if total.requires_grad:
        total.backward()
        grad = f.grad.numpy()
    else:
        grad = np.zeros_like(flow, dtype=np.float32)
    return {k: float(v) for k, v in terms.items()}, float(total), grad
