"""Self-supervised cluster labels on the GPU -- this build's counterpart of ``+ssl_label=seflow_auto``
(assets/slurm/ssl-train-av2.sh:32, ssl-train-scania.sh:32): the per-point ``0 = static / > 0 = dynamic cluster id`` labels the
training loss consumes (``ssl_loss.SeFlowLoss``), generated from the sweeps themselves instead of read from ground truth.

PARITY UNPINNED.  The reference's generator (DUFOMap dynamic awareness + HDBSCAN clustering, per the SeFlow papers) lives in the
absent OpenSceneFlow submodule together with its dependencies; only the option's name is in the tree.  This build's own
specification, chosen so that every step is exact and order-independent:

  0. only points inside the network's BEV range (+-51.2 m in the common frame, ``RANGE_NET``) take part: the network gives the
     others no flow and the loss no gradient (``mask_rows``), so labelling them would only dilute the loss's per-cluster and
     per-point normalisations -- and a 200 m sweep would pile its far half into the border cells of the search grids.
  1. dynamic candidates of a sweep A against its neighbour sweep B (both in ONE frame: A is moved with ``inv(poseB) @ poseA``):
     non-ground points of A whose nearest non-ground point of B is further than ``dyn_dist`` x max(1, r / 30 m), r = the
     point's BEV range (0.35 m near the sensor: a point on a static surface has a return of the other sweep next to it once
     ego motion is removed, a point on an object faster than 3.5 m/s does not; beyond ~30 m the SAMPLING spacing of a spinning
     LiDAR exceeds 0.35 m, so the bar grows with range instead of declaring most far static returns candidates).
     Exact nearest neighbours through the BEV cell grid (csrc/nngrid.hip).
  2. clusters: DBSCAN(``eps``, ``min_pts``) over the candidates (csrc/dbscan.hip: labels are a pure function of the input --
     clusters numbered by their lowest point index, a border point joins the neighbouring cluster of lowest such index).
  3. everything else -- ground, static, noise -- is label 0.

The oracle is sklearn.cluster.DBSCAN + scipy's cKDTree (oracle/dbscan_oracle.py); tests/test_ssl_label_gpu.py.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from .. import _lib
from ..ssl_loss import nn_grid

_lib.register({
    "himo_dbscan_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "himo_dbscan": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_size_t, ctypes.c_void_p]),
    "himo_rigid_transform": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p]),
})

EPS, MIN_PTS, DYN_DIST = 0.5, 8, 0.35
RANGE_XY = 52.0                     # the BEV cell grid covers +- this (points beyond it are binned into its border cells)
RANGE_NET = 51.2                    # the network's BEV range (assets/slurm/ssl-train-av2.sh:32 point_cloud_range): step 0
DYN_REF_RANGE = 30.0                # BEV range from which the dynamic-candidate bar grows linearly (step 1)


def dbscan(points: torch.Tensor, eps: float = EPS, min_pts: int = MIN_PTS, skip: torch.Tensor | None = None):
    """(labels int32 (n,), number of clusters) of DBSCAN over the xyz of ``points`` (device (n, >= 3) float32).  ``skip``: bool /
    uint8 (n,), True = the point takes no part (label 0)."""
    lib, dev = _lib.load(), _lib.require_gpu()
    p = points.to(device=dev, dtype=torch.float32)
    if p.stride(-1) != 1 or p.stride(0) < 3:
        p = p.contiguous()
    n = p.shape[0]
    gw = gh = int(math.ceil(2 * RANGE_XY / eps))
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.himo_dbscan_workspace_bytes(n, gw, gh)), dtype=torch.uint8, device=dev)
    sk = None if skip is None else skip.to(device=dev, dtype=torch.uint8).contiguous()
    _lib.check(lib.himo_dbscan(n, _lib.ptr(p), int(p.stride(0)) if n else 3, _lib.ptr(sk), float(eps), int(min_pts), -RANGE_XY, -RANGE_XY, float(eps),
                               gw, gh, _lib.ptr(labels), _lib.ptr(count), _lib.ptr(ws), ws.numel(), _lib.stream_handle()), "himo_dbscan")
    return labels, count


def _moved(pc: torch.Tensor, T: np.ndarray) -> torch.Tensor:
    lib, dev = _lib.load(), _lib.require_gpu()
    src = pc[:, :3].contiguous()
    out = torch.empty((src.shape[0], 3), dtype=torch.float32, device=dev)
    T32 = torch.from_numpy(np.ascontiguousarray(T, dtype=np.float32)).to(dev)
    _lib.check(lib.himo_rigid_transform(src.shape[0], _lib.ptr(src), 3, _lib.ptr(T32), _lib.ptr(out), 3, _lib.stream_handle()), "rigid")
    return out


_tls = __import__("threading").local()


def _pinned_counts():
    """this thread's pinned word pair for the sizes of the two compacted sweeps (one outstanding read-back per thread)"""
    buf = getattr(_tls, "counts", None)
    if buf is None:
        buf = _tls.counts = torch.zeros(2, dtype=torch.int32).pin_memory()
    return buf


def _compact(pts: torch.Tensor, use: torch.Tensor):
    """rows of ``pts`` with ``use`` set, in order, WITHOUT asking the host for their number: every row is copied to its rank among
    the kept rows, the others to a dump row at the end.  -> (buffer (n + 1, 3), destination row of every input row (n,) int64, the
    number of kept rows as a 1-element int32 device tensor)"""
    n = pts.shape[0]
    pos = torch.cumsum(use.to(torch.int32), 0, dtype=torch.int32)
    dest = torch.where(use, pos - 1, n).to(torch.int64)
    buf = torch.empty((n + 1, 3), dtype=torch.float32, device=pts.device)
    buf.index_copy_(0, dest, pts)                               # (the dump row takes whichever excluded row lands last: never read)
    return buf, dest, pos[-1:]


def auto_labels(pc0, pc1, ground0, ground1, pose0, pose1, eps: float = EPS, min_pts: int = MIN_PTS, dyn_dist: float = DYN_DIST,
                return_top: bool = False):
    """(label0 (n0,), label1 (n1,)) int32 device tensors for the sweep pair (module docstring).  ``pc*``: (n, >= 3) float32 in their
    own sensor frames, ``ground*``: bool masks, ``pose*``: 4x4 world poses.  ``return_top``: also the highest label of the pair as a
    1-element int32 device tensor (= the larger of the two cluster counts the clustering kernels leave: no reduction over the labels).

    ONE host wait per pair -- the sizes of the two in-range non-ground subsets, which the nearest-neighbour search takes as host
    integers; they come back through pinned memory behind an event on the current stream.  Everything else is enqueued without
    asking the device anything (no boolean-mask indexing: that copies a count back per use -- five waits per pair before round 6,
    each as long as the queue in front of it when a training step shares the device)."""
    dev = _lib.require_gpu()
    up = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(device=dev, dtype=dt)
    p0, p1 = up(pc0, torch.float32), up(pc1, torch.float32)
    g0, g1 = up(ground0, torch.bool), up(ground1, torch.bool)
    T = np.linalg.inv(np.asarray(pose1, np.float64)) @ np.asarray(pose0, np.float64)
    a = _moved(p0, T)                                           # pc0 in pc1's frame
    b = p1[:, :3].contiguous()
    if a.shape[0] == 0 or b.shape[0] == 0:                      # nothing to compare with: every non-ground in-range point is a candidate
        out = []
        for pts, g in ((a, g0), (b, g1)):
            skip = (g | (pts[:, :2].abs().amax(dim=1) > RANGE_NET)) if pts.shape[0] else torch.zeros(0, dtype=torch.bool, device=dev)
            out.append(dbscan(pts, eps, min_pts, skip))
        return (out[0][0], out[1][0], torch.maximum(out[0][1], out[1][1])) if return_top else (out[0][0], out[1][0])
    far2 = float(dyn_dist) ** 2
    inv_ref2 = 1.0 / (DYN_REF_RANGE * DYN_REF_RANGE)
    use_a = ~(g0 | (a[:, :2].abs().amax(dim=1) > RANGE_NET))   # step 0
    use_b = ~(g1 | (b[:, :2].abs().amax(dim=1) > RANGE_NET))
    (buf_a, dest_a, cnt_a), (buf_b, dest_b, cnt_b) = _compact(a, use_a), _compact(b, use_b)
    host = _pinned_counts()
    host.copy_(torch.cat([cnt_a, cnt_b]), non_blocking=True)
    landed = torch.cuda.Event()
    landed.record(torch.cuda.current_stream(dev))
    landed.synchronize()                                        # the pair's one host wait
    na, nb = int(host[0]), int(host[1])
    a_in, b_in = buf_a[:na], buf_b[:nb]
    out = []
    for pts, use, dest, mine, other in ((a, use_a, dest_a, a_in, b_in), (b, use_b, dest_b, b_in, a_in)):
        skip = ~use
        if mine.shape[0] and other.shape[0]:
            n = pts.shape[0]
            d2 = torch.full((n + 1,), float("inf"), dtype=torch.float32, device=dev)
            d2[:mine.shape[0]] = nn_grid(mine, other, return_index=False)
            x, y = pts[:, 0], pts[:, 1]
            bar2 = far2 * torch.clamp((x * x + y * y) * inv_ref2, min=1.0)       # (dyn_dist * max(1, r / 30 m))^2, float32 as the oracle
            skip = skip | (d2[dest] <= bar2)                    # a close return in the other sweep: static (the dump row holds inf)
        out.append(dbscan(pts, eps, min_pts, skip))
    return (out[0][0], out[1][0], torch.maximum(out[0][1], out[1][1])) if return_top else (out[0][0], out[1][0])
