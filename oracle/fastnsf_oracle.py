"""CPU ORACLE for the optimisation-based scene flow (stage a12) -- test infrastructure, NOT product code.

PARITY UNPINNED: the reference's `fastnsf` lives in the absent OpenSceneFlow submodule (README.md:53 is the only call
site).  This restates THIS BUILD'S specification (himo_amd/fastnsf.py) with PyTorch CPU autograd + torch.optim.Adam and
cKDTree correspondences.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial import cKDTree


def _nn(q, r):
    return cKDTree(r.astype(np.float64)).query(q.astype(np.float64), k=1)[1]


def mlp(layers, x):
    h = x
    for k, (w, b) in enumerate(layers):
        h = h @ w + b
        if k < len(layers) - 1:
            h = torch.relu(h)
    return h


def loss_and_grads(layers_np, pc0, pc1, trunc=2.0):
    """One evaluation: (loss, [(dW, db)], flow) for the given parameters; pc0 already in pc1's frame."""
    layers = [(torch.from_numpy(w.copy()).requires_grad_(True), torch.from_numpy(b.copy()).requires_grad_(True)) for w, b in layers_np]
    p0, p1 = torch.from_numpy(pc0.astype(np.float32)), torch.from_numpy(pc1.astype(np.float32))
    f = mlp(layers, p0)
    moved = p0 + f
    ia, ib = _nn(moved.detach().numpy(), pc1), _nn(pc1, moved.detach().numpy())
    da = ((moved - p1[torch.from_numpy(ia)]) ** 2).sum(1)
    db = ((p1 - moved[torch.from_numpy(ib)]) ** 2).sum(1)
    t2 = trunc * trunc
    loss = (da * (da.detach() <= t2)).double().sum() / len(p0) + (db * (db.detach() <= t2)).double().sum() / len(p1)
    loss.backward()
    return float(loss), [(w.grad.numpy(), b.grad.numpy()) for w, b in layers], f.detach().numpy()


def fit(layers_np, pc0, pc1, iters, lr=1e-3, trunc=2.0):
    layers = [(torch.from_numpy(w.copy()).requires_grad_(True), torch.from_numpy(b.copy()).requires_grad_(True)) for w, b in layers_np]
    opt = torch.optim.Adam([t for wb in layers for t in wb], lr=lr, betas=(0.9, 0.999), eps=1e-8)
    p0, p1 = torch.from_numpy(pc0.astype(np.float32)), torch.from_numpy(pc1.astype(np.float32))
    hist = []
    t2 = trunc * trunc
    for _ in range(iters):
        opt.zero_grad()
        moved = p0 + mlp(layers, p0)
        ia, ib = _nn(moved.detach().numpy(), pc1), _nn(pc1, moved.detach().numpy())
        da = ((moved - p1[torch.from_numpy(ia)]) ** 2).sum(1)
        db = ((p1 - moved[torch.from_numpy(ib)]) ** 2).sum(1)
        loss = (da * (da.detach() <= t2)).double().sum() / len(p0) + (db * (db.detach() <= t2)).double().sum() / len(p1)
        loss.backward()
        opt.step()
        hist.append(float(loss))
    with torch.no_grad():
        flow = mlp(layers, p0).numpy()
    return hist, flow
