"""CPU ORACLE for the scene-flow network (stage a10) -- test infrastructure, NOT product code.

PARITY UNPINNED: the reference's network source (OpenSceneFlow submodule) is absent from
/root/reference, there is no checkpoint and no golden flow (SURVEY.md sections 0 and 8c), so this
file restates THIS BUILD'S OWN specification (himo_amd/seflow/spec.py) with plain PyTorch float32
ops on the CPU.  It pins the HIP kernels against an independent implementation of the same spec,
not against the reference.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.

All functions are functional: they take the parameter dict produced by
``himo_amd.seflow.spec.init_params`` (numpy float32 arrays) and numpy / torch inputs.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

VOXEL_SIZE = (0.2, 0.2, 6.0)
PC_RANGE = (-51.2, -51.2, -3.0, 51.2, 51.2, 3.0)
GRID_H, GRID_W = 512, 512
BN_EPS_PFN, BN_EPS = 1e-3, 1e-5
BN_MOMENTUM = 0.1
GRU_ITERS = 4


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def ego_transform(pose_from, pose_to) -> np.ndarray:
    """4x4 float64 taking points of sweep `from` into sweep `to`'s frame: inv(pose_to) @ pose_from."""
    return np.linalg.inv(np.asarray(pose_to, np.float64)) @ np.asarray(pose_from, np.float64)


def transform_points(xyz, T) -> torch.Tensor:
    """float32 p' = R p + t as ((x*R0 + y*R1) + z*R2) + t with every product / sum rounded separately
    (spec.py step 0: the pillar a point lands in is a discrete decision, so the rounding is specified)."""
    p = _t(xyz)[:, :3].float()
    R = np.asarray(T[:3, :3], np.float32)
    t = np.asarray(T[:3, 3], np.float32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    cols = [((x * float(R[r, 0]) + y * float(R[r, 1])) + z * float(R[r, 2])) + float(t[r]) for r in range(3)]
    return torch.stack(cols, dim=1)


def voxelize(xyz: torch.Tensor):
    """valid (N,) bool, iy (N,), ix (N,) int64 -- float32 arithmetic: floor((p - min) / size)."""
    mn = torch.tensor(PC_RANGE[:3], dtype=torch.float32)
    vs = torch.tensor(VOXEL_SIZE, dtype=torch.float32)
    c = torch.floor((xyz - mn) / vs)
    valid = (c[:, 0] >= 0) & (c[:, 0] < GRID_W) & (c[:, 1] >= 0) & (c[:, 1] < GRID_H) & (c[:, 2] >= 0) & (c[:, 2] < 1)
    return valid, c[:, 1].long(), c[:, 0].long()


def _batch_norm(params, prefix, y, eps, training):
    """BatchNorm over dim 0 (and the spatial dims) of y.  Eval: the running statistics folded to scale / shift exactly as the
    product folds them.  Training: torch's own F.batch_norm -- batch statistics normalise, the tensors under
    ``<prefix>.mean`` / ``<prefix>.var`` are updated IN PLACE (momentum 0.1, unbiased variance)."""
    if training:
        return F.batch_norm(y, _t(params[f"{prefix}.mean"]), _t(params[f"{prefix}.var"]), _t(params[f"{prefix}.gamma"]),
                            _t(params[f"{prefix}.beta"]), training=True, momentum=BN_MOMENTUM, eps=eps)
    scale = _t(params[f"{prefix}.gamma"]) / torch.sqrt(_t(params[f"{prefix}.var"]) + eps)
    shift = _t(params[f"{prefix}.beta"]) - _t(params[f"{prefix}.mean"]) * scale
    shape = (1, -1) + (1,) * (y.dim() - 2)
    return y * scale.reshape(shape) + shift.reshape(shape)


def pillar_image(params, xyz: torch.Tensor, training: bool = False):
    """(32, H, W) pseudo-image + per-point (valid, pillar id, offset-to-centre).  ``training``: the pillar net's BatchNorm
    uses the statistics of THIS sweep's in-range points (one call of the embedder per sweep)."""
    valid, iy, ix = voxelize(xyz)
    pid = iy * GRID_W + ix
    pts = xyz[valid]
    pv = pid[valid]
    n_cells = GRID_H * GRID_W
    cnt = torch.zeros(n_cells, dtype=torch.float32).index_add_(0, pv, torch.ones(len(pv)))
    sums = torch.zeros(n_cells, 3, dtype=torch.float32).index_add_(0, pv, pts)       # sequential, in point order
    mean = sums[pv] / cnt[pv, None]
    vs = torch.tensor(VOXEL_SIZE, dtype=torch.float32)
    off = torch.tensor([VOXEL_SIZE[0] / 2 + PC_RANGE[0], VOXEL_SIZE[1] / 2 + PC_RANGE[1], VOXEL_SIZE[2] / 2 + PC_RANGE[2]],
                       dtype=torch.float32)
    cell = torch.stack([ix[valid].float(), iy[valid].float(), torch.zeros(len(pv))], dim=1)
    centre = cell * vs + off
    feats = torch.cat([pts, pts - mean, pts - centre], dim=1)                        # (n, 9)
    y = feats @ _t(params["pfn.weight"])
    y = torch.relu(_batch_norm(params, "pfn.bn", y, BN_EPS_PFN, training and len(pv) > 0))
    acc = torch.zeros(n_cells, y.shape[1], dtype=torch.float32).index_add_(0, pv, y)
    img = acc / cnt.clamp(min=1.0)[:, None]
    img = img.T.reshape(-1, GRID_H, GRID_W).contiguous()
    offsets = torch.zeros_like(xyz)
    offsets[valid] = pts - centre
    return img, valid, pid, offsets


def _conv(params, name, x, stride=1, pad=1):
    w = _t(params[f"{name}.weight"]).permute(3, 2, 0, 1).contiguous()                 # [kh,kw,ci,co] -> [co,ci,kh,kw]
    return F.conv2d(x, w, _t(params[f"{name}.bias"]), stride=stride, padding=pad)


def conv_bn_gelu(params, name, x, stride, training: bool = False):
    y = _conv(params, name, x, stride=stride)
    return F.gelu(_batch_norm(params, f"{name}.bn", y, BN_EPS, training))


ENC_STAGES = (("enc1", 4), ("enc2", 6), ("enc3", 6))


def encoder(params, imgs: torch.Tensor, training: bool = False):
    """imgs (F,32,H,W) -> [(F,64,H/2,..), (F,128,H/4,..), (F,256,H/8,..)].  The F frames of the sample are the batch: in
    training mode a layer's BatchNorm statistics are taken over all of them."""
    outs, x = [], imgs
    for stage, n in ENC_STAGES:
        for i in range(n):
            x = conv_bn_gelu(params, f"{stage}.{i}", x, stride=2 if i == 0 else 1, training=training)
        outs.append(x)
    return outs


def upsample_skip(params, name, coarse, skip):
    a = _conv(params, f"{name}.u1", coarse, pad=0)
    a = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)
    b = _conv(params, f"{name}.u3", skip, pad=0)
    y = _conv(params, f"{name}.u4", torch.cat([a, b], dim=1))
    return _conv(params, f"{name}.u5", y)


def backbone(params, imgs: torch.Tensor, training: bool = False):
    """imgs (F,32,H,W) -> (64,H,W) decoder map."""
    f1, f2, f3 = encoder(params, imgs, training)
    cat = lambda t: t.reshape(1, -1, t.shape[2], t.shape[3])                          # frames stacked on channels
    s = upsample_skip(params, "dec1", cat(f3), cat(f2))
    t = upsample_skip(params, "dec2", s, cat(f1))
    u = upsample_skip(params, "dec3", t, cat(imgs))
    return _conv(params, "dec4", u)[0]


def head(params, img0, img1, dec, pid, offsets):
    """per-point GRU head for the valid pc0 points; pid / offsets already restricted to them."""
    gather = lambda im: im.reshape(im.shape[0], -1)[:, pid].T                          # (n, C)
    h = torch.cat([gather(img0), gather(img1), gather(dec)], dim=1)                    # (n, 128)
    x = offsets @ _t(params["head.offset.weight"]) + _t(params["head.offset.bias"])    # (n, 64)
    W = lambda g: _t(params[f"head.gru.{g}.weight"])
    B = lambda g: _t(params[f"head.gru.{g}.bias"])
    for _ in range(GRU_ITERS):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(hx @ W("z") + B("z"))
        r = torch.sigmoid(hx @ W("r") + B("r"))
        q = torch.tanh(torch.cat([r * h, x], dim=1) @ W("q") + B("q"))
        h = (1 - z) * h + z * q
    y = F.gelu(torch.cat([h, x], dim=1) @ _t(params["head.dec1.weight"]) + _t(params["head.dec1.bias"]))
    return y @ _t(params["head.dec2.weight"]) + _t(params["head.dec2.bias"])


@torch.no_grad()
def forward(params, pch1, pc0, pc1, pose_h1, pose0, pose1, return_intermediates: bool = False):
    """One sample -> (N0,3) float32 flow INCLUDING ego motion, aligned with pc0 rows."""
    p0 = _t(pc0)[:, :3].float()
    T0, Th = ego_transform(pose0, pose1), ego_transform(pose_h1, pose1)
    p0t = transform_points(p0, T0)
    pht = transform_points(pch1, Th)
    p1 = _t(pc1)[:, :3].float()
    pose_flow = p0t - p0
    img_h, *_ = pillar_image(params, pht)
    img0, valid0, pid0, off0 = pillar_image(params, p0t)
    img1, *_ = pillar_image(params, p1)
    imgs = torch.stack([img_h, img0, img1])
    dec = backbone(params, imgs)
    res = head(params, img0, img1, dec, pid0[valid0], off0[valid0])
    flow = pose_flow.clone()
    flow[valid0] = flow[valid0] + res
    out = flow.numpy()
    if return_intermediates:
        return out, {"imgs": imgs.numpy(), "dec": dec.numpy(), "valid0": valid0.numpy(), "pose_flow": pose_flow.numpy(),
                     "res": res.numpy()}
    return out


def forward_train(params, pch1, pc0, pc1, pose_h1, pose0, pose1, training: bool = False):
    """Differentiable restatement for the training tests: ``params`` maps names to torch tensors (leaves that require
    grad for the trainable ones).  ``training`` False: BatchNorm frozen (its tensors are constants).  True: BatchNorm in
    training mode as torch defines it -- batch statistics (per sweep in the pillar net, over the sample's F frames in the
    encoder), gamma / beta trainable leaves, and the ``*.bn.mean`` / ``*.bn.var`` tensors of ``params`` updated in place in
    call order (pillar net: history sweep, pc0, pc1).  Returns (res (n_valid,3) with graph, valid0 (N0,) bool, pc0 in
    pc1's frame (N0,3))."""
    p0 = _t(pc0)[:, :3].float()
    T0, Th = ego_transform(pose0, pose1), ego_transform(pose_h1, pose1)
    p0t = transform_points(p0, T0)
    pht = transform_points(pch1, Th)
    p1 = _t(pc1)[:, :3].float()
    img_h, *_ = pillar_image(params, pht, training)
    img0, valid0, pid0, off0 = pillar_image(params, p0t, training)
    img1, *_ = pillar_image(params, p1, training)
    imgs = torch.stack([img_h, img0, img1])
    dec = backbone(params, imgs, training)
    res = head(params, img0, img1, dec, pid0[valid0], off0[valid0])
    return res, valid0, p0t


def pillar_images_batch(params, sweeps, training: bool = False):
    """The pillar net over a BATCH of sweeps in one call of the embedder (torch semantics of a per-process batch: the pillar net's
    BatchNorm1d sees the in-range points of ALL the sweeps it is handed): per sweep (image (32,H,W), valid, pid, offsets).  With one
    sweep this is ``pillar_image``."""
    n_cells = GRID_H * GRID_W
    vs = torch.tensor(VOXEL_SIZE, dtype=torch.float32)
    off = torch.tensor([VOXEL_SIZE[0] / 2 + PC_RANGE[0], VOXEL_SIZE[1] / 2 + PC_RANGE[1], VOXEL_SIZE[2] / 2 + PC_RANGE[2]], dtype=torch.float32)
    pre, ys = [], []
    for xyz in sweeps:
        valid, iy, ix = voxelize(xyz)
        pid = iy * GRID_W + ix
        pts, pv = xyz[valid], pid[valid]
        cnt = torch.zeros(n_cells, dtype=torch.float32).index_add_(0, pv, torch.ones(len(pv)))
        sums = torch.zeros(n_cells, 3, dtype=torch.float32).index_add_(0, pv, pts)
        mean = sums[pv] / cnt[pv, None]
        centre = torch.stack([ix[valid].float(), iy[valid].float(), torch.zeros(len(pv))], dim=1) * vs + off
        feats = torch.cat([pts, pts - mean, pts - centre], dim=1)
        ys.append(feats @ _t(params["pfn.weight"]))
        pre.append((valid, pid, pv, cnt, pts - centre, xyz))
    total = sum(len(y) for y in ys)
    y_all = torch.relu(_batch_norm(params, "pfn.bn", torch.cat(ys, dim=0), BN_EPS_PFN, training and total > 0))
    out, at = [], 0
    for (valid, pid, pv, cnt, rel, xyz), y0 in zip(pre, ys):
        y = y_all[at:at + len(y0)]
        at += len(y0)
        acc = torch.zeros(n_cells, y.shape[1], dtype=torch.float32).index_add_(0, pv, y)
        img = (acc / cnt.clamp(min=1.0)[:, None]).T.reshape(-1, GRID_H, GRID_W).contiguous()
        offsets = torch.zeros_like(xyz)
        offsets[valid] = rel
        out.append((img, valid, pid, offsets))
    return out


def forward_train_batch(params, samples, training: bool = False):
    """``forward_train`` for a per-process BATCH (the reference launcher's ``batch_size=8`` on one GPU, assets/slurm/ssl-train-av2.sh:32-34):
    ``samples`` = [(pch1, pc0, pc1, pose_h1, pose0, pose1), ...].  Training-mode BatchNorm takes its statistics over the WHOLE batch,
    as torch does: the pillar net per embedder call -- the history sweeps of all samples, then their pc0s, then their pc1s (three
    running-statistics updates, in that order) -- and every encoder layer over the B x F images (image b * F + f).  The decoder and
    the head have no BatchNorm.  Returns [(res (n_valid,3), valid0, pc0 in pc1's frame)] per sample.  With one sample: ``forward_train``."""
    slots = [[], [], []]
    p0ts = []
    for pch1, pc0, pc1, pose_h1, pose0, pose1 in samples:
        p0 = _t(pc0)[:, :3].float()
        T0, Th = ego_transform(pose0, pose1), ego_transform(pose_h1, pose1)
        p0t = transform_points(p0, T0)
        slots[0].append(transform_points(pch1, Th)); slots[1].append(p0t); slots[2].append(_t(pc1)[:, :3].float())
        p0ts.append(p0t)
    per_slot = [pillar_images_batch(params, sw, training) for sw in slots]            # call order: history, pc0, pc1
    B, Fr = len(samples), len(slots)
    imgs = torch.stack([per_slot[f][b][0] for b in range(B) for f in range(Fr)])      # (B * F, 32, H, W), image b * F + f
    f1, f2, f3 = encoder(params, imgs, training)
    cat = lambda t: t.reshape(B, -1, t.shape[2], t.shape[3])                          # a sample's frames stacked on channels
    s = upsample_skip(params, "dec1", cat(f3), cat(f2))
    t = upsample_skip(params, "dec2", s, cat(f1))
    u = upsample_skip(params, "dec3", t, cat(imgs))
    dec = _conv(params, "dec4", u)
    out = []
    for b in range(B):
        img0, valid0, pid0, off0 = per_slot[1][b]
        out.append((head(params, img0, per_slot[2][b][0], dec[b], pid0[valid0], off0[valid0]), valid0, p0ts[b]))
    return out
