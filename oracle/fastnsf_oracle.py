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


# ---- the distance-transform objective (csrc/dtloss.hip; himo_amd/fastnsf.py objective="dt") -----------------------------------
def dt_volume(pc1, origin, dims, cell, window):
    """(nz, ny, nx) float32 volume D: distance in metres from every cell to the nearest cell that holds a pc1 point, capped at
    ``window`` cells -- scipy's exact Euclidean distance transform of the occupancy grid."""
    from scipy.ndimage import distance_transform_edt
    nx, ny, nz = (int(d) for d in dims)
    c = np.floor((pc1[:, :3].astype(np.float32) - np.asarray(origin, np.float32)) / np.float32(cell))
    ok = (c[:, 0] >= 0) & (c[:, 0] < nx) & (c[:, 1] >= 0) & (c[:, 1] < ny) & (c[:, 2] >= 0) & (c[:, 2] < nz)
    c = c[ok].astype(np.int64)
    occ = np.zeros((nz, ny, nx), bool)
    occ[c[:, 2], c[:, 1], c[:, 0]] = True
    if not occ.any():
        return np.full((nz, ny, nx), np.float32(window) * np.float32(cell), np.float32)
    d = distance_transform_edt(~occ)                                   # in cells
    # the product's passes are windowed: a cell further than `window` cells from every occupied cell ALONG SOME AXIS of the
    # minimising offset is reported as `window`; capping at `window` makes both agree (any distance < window is within the window)
    return (np.minimum(d, float(window)).astype(np.float32) * np.float32(cell)).astype(np.float32)


def dt_lookup(vol, moved, origin, cell):
    """(D, in_volume): trilinear interpolation of ``vol`` (torch float32 (nz, ny, nx)) at ``moved`` (torch (n, 3), may require
    grad) in cell-centre coordinates, and which points are IN THE VOLUME (0 < u < n - 1 on every axis) -- the rule of
    csrc/dtlookup.h, with autograd supplying the gradient.  D of a point outside the volume is meaningless (its indices are
    clamped only to keep the gather legal); callers mask it."""
    nz, ny, nx = vol.shape
    dims = (nx, ny, nz)
    idx, frac = [], []
    inside = torch.ones(len(moved), dtype=torch.bool)
    for k in range(3):
        u = (moved[:, k] - float(np.float32(origin[k]))) / float(np.float32(cell)) - 0.5
        inside = inside & (u.detach() > 0.0) & (u.detach() < float(dims[k] - 1))
        b = u.detach().floor().clamp(min=0.0, max=float(max(dims[k] - 2, 0))).nan_to_num(0.0).long()
        idx.append(b); frac.append(u - b.to(u.dtype))
    def at(dx, dy, dz):
        x = (idx[0] + dx).clamp(max=nx - 1); y = (idx[1] + dy).clamp(max=ny - 1); z = (idx[2] + dz).clamp(max=nz - 1)
        return vol[z, y, x]
    fx, fy, fz = frac
    c00 = at(0, 0, 0) * (1 - fx) + at(1, 0, 0) * fx
    c10 = at(0, 1, 0) * (1 - fx) + at(1, 1, 0) * fx
    c01 = at(0, 0, 1) * (1 - fx) + at(1, 0, 1) * fx
    c11 = at(0, 1, 1) * (1 - fx) + at(1, 1, 1) * fx
    c0 = c00 * (1 - fy) + c10 * fy
    c1 = c01 * (1 - fy) + c11 * fy
    return c0 * (1 - fz) + c1 * fz, inside


def dt_objective(vol, moved, origin, cell, trunc):
    """loss = (1 / m) sum_{i in volume} [D_i <= trunc] D_i, m = points in the volume (0 -> 0): himo_amd/fastnsf.py's "dt" objective"""
    D, inside = dt_lookup(vol, moved, origin, cell)
    keep = inside & (D.detach() <= trunc)
    m = int(inside.sum())
    return torch.where(keep, D, torch.zeros_like(D)).double().sum() / max(m, 1)


def dt_loss_and_grads(layers_np, pc0, pc1, origin, dims, cell, window, trunc=2.0):
    """(loss, [(dW, db)], flow, d loss / d moved) of the distance-transform objective for the given parameters."""
    layers = [(torch.from_numpy(w.copy()).requires_grad_(True), torch.from_numpy(b.copy()).requires_grad_(True)) for w, b in layers_np]
    vol = torch.from_numpy(dt_volume(pc1, origin, dims, cell, window))
    p0 = torch.from_numpy(pc0.astype(np.float32))
    f = mlp(layers, p0)
    moved = p0 + f
    moved.retain_grad()
    loss = dt_objective(vol, moved, origin, cell, trunc)
    loss.backward()
    return float(loss), [(w.grad.numpy(), b.grad.numpy()) for w, b in layers], f.detach().numpy(), moved.grad.numpy()


def dt_fit(layers_np, pc0, pc1, origin, dims, cell, window, iters, lr=1e-3, trunc=2.0):
    layers = [(torch.from_numpy(w.copy()).requires_grad_(True), torch.from_numpy(b.copy()).requires_grad_(True)) for w, b in layers_np]
    opt = torch.optim.Adam([t for wb in layers for t in wb], lr=lr, betas=(0.9, 0.999), eps=1e-8)
    vol = torch.from_numpy(dt_volume(pc1, origin, dims, cell, window))
    p0 = torch.from_numpy(pc0.astype(np.float32))
    hist = []
    for _ in range(iters):
        opt.zero_grad()
        loss = dt_objective(vol, p0 + mlp(layers, p0), origin, cell, trunc)
        loss.backward()
        opt.step()
        hist.append(float(loss))
    with torch.no_grad():
        flow = mlp(layers, p0).numpy()
    return hist, flow
