"""Drop-in for the reference's ``utils/__init__.py`` (same names, arguments, results and
errors) with the arithmetic running on the MI355X through libhimo_amd.so.

    check_valid    utils/__init__.py:4-24    (host logic only)
    ego_pts_mask   utils/__init__.py:26-34
    flow2compDis   utils/__init__.py:36-43
    refine_pts     utils/__init__.py:45-47

Inputs may be numpy arrays (as in the reference; they are staged through HBM and the result
comes back as a numpy array of the dtype numpy itself would have produced) or torch tensors
already on the GPU (zero-copy, result stays on the GPU).  There is no CPU implementation
here: without a HIP device the functions raise.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import _lib

__all__ = ["check_valid", "ego_pts_mask", "flow2compDis", "refine_pts"]


def check_valid(data_dir, flow_mode, comp_dis_zip=None):
    """Dataset sniffing + evaluation mode.  Same quirks as the reference: the match is on
    ``str.find(...) > 0`` (a name at position 0 does not count) and an unknown name raises
    ``ValueError`` (utils/__init__.py:6-11)."""
    hit = lambda s: data_dir.find(s) > 0
    if hit("Scania") or hit("scania"):
        data_name = "scania"
    elif hit("av2") or hit("AV2"):
        data_name = "av2"
    else:
        raise ValueError("Unknown dataset name in data_dir.")
    if comp_dis_zip is not None and os.path.exists(comp_dis_zip):
        print(f"Using provided comp_dis_zip: {comp_dis_zip} for evaluation.")
        return data_name, 1
    print(f"No valid comp_dis_zip provided, evaluating based on {flow_mode} directly.")
    return data_name, 2


# ---------------------------------------------------------------------------------------------
def _is_np(x) -> bool:
    return not isinstance(x, torch.Tensor)


def _to_dev(x, dtype, dev):
    """numpy / tensor -> contiguous device tensor of ``dtype`` (no copy when already right)."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=dev, dtype=dtype).contiguous()


def _float_kind(*arrays) -> torch.dtype:
    """numpy's promotion for the float chain: float64 if any operand is float64 (or integer)."""
    for a in arrays:
        dt = a.dtype if isinstance(a, torch.Tensor) else torch.from_numpy(np.empty(0, np.asarray(a).dtype)).dtype
        if dt not in (torch.float32, torch.float16, torch.bfloat16):
            return torch.float64
    return torch.float32


def _rows(pc, dev):
    """(N, S>=3) float32 point rows on the device + row stride in floats."""
    t = _to_dev(pc, torch.float32, dev)
    if t.dim() != 2 or t.shape[1] < 3:
        raise IndexError(f"expected an (N, >=3) point array, got shape {tuple(t.shape)}")
    return t, t.shape[1]


def _back(t, as_numpy: bool):
    return t.cpu().numpy() if as_numpy else t


def ego_pts_mask(pts, min_bound=[-9.5, -3 / 2, 0], max_bound=[5, 2.760004 / 2, 5]):
    """(N,) bool, True for points OUTSIDE the open ego box.  Compares run in float32 with the
    bounds rounded to float32, which is what numpy does for a float32 array against Python
    floats."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    p, stride = _rows(pts, dev)
    n = p.shape[0]
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    bounds = (ctypes.c_float * 6)(*[float(v) for v in list(min_bound)[:3] + list(max_bound)[:3]])
    _lib.check(lib.himo_ego_pts_mask(n, _lib.ptr(p), stride, bounds, _lib.ptr(out), _lib.stream_handle()), "ego_pts_mask")
    return _back(out.bool(), _is_np(pts))


def flow2compDis(flow, dt0, sensor_dt=10):
    """``flow / sensor_dt * dt0[:, None]`` -- divide, then multiply, in the promoted dtype."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    kind = _float_kind(flow, dt0)
    f = _to_dev(flow, kind, dev)
    dt_kind = torch.float64 if (kind == torch.float64 and _float_kind(dt0) == torch.float64) else torch.float32
    d = _to_dev(dt0, dt_kind, dev)
    if f.dim() != 2 or f.shape[1] != 3 or d.dim() != 1 or d.shape[0] != f.shape[0]:
        raise ValueError(f"operands could not be broadcast together with shapes {tuple(f.shape)} {tuple(d.shape)}")
    n = f.shape[0]
    out = torch.empty_like(f)
    flags = (1 if kind == torch.float64 else 0) | (2 if dt_kind == torch.float64 else 0)
    _lib.check(lib.himo_flow2compdis(n, _lib.ptr(f), _lib.ptr(d), float(sensor_dt), flags, _lib.ptr(out),
                                     _lib.stream_handle()), "flow2compDis")
    return _back(out, _is_np(flow))


def refine_pts(pc, ds):
    """``pc[:, :3] + ds``."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    p, stride = _rows(pc, dev)
    kind = _float_kind(ds) if _float_kind(pc) == torch.float32 else torch.float64
    d = _to_dev(ds, kind, dev)
    if d.dim() != 2 or d.shape[1] != 3 or d.shape[0] != p.shape[0]:
        raise ValueError(f"operands could not be broadcast together with shapes {(p.shape[0], 3)} {tuple(d.shape)}")
    out = torch.empty_like(d)
    _lib.check(lib.himo_refine_pts(p.shape[0], _lib.ptr(p), stride, _lib.ptr(d), 1 if kind == torch.float64 else 0,
                                   _lib.ptr(out), _lib.stream_handle()), "refine_pts")
    return _back(out, _is_np(pc))


def dt0_from_lidar_dt(lidar_dt):
    """``max(lidar_dt) - lidar_dt`` (save_zip.py:120, eval.py:299); raises ``ValueError`` on an
    empty sweep like the builtin ``max``."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    d = _to_dev(lidar_dt, torch.float32, dev)
    out = torch.empty_like(d)
    ws = torch.empty(64, dtype=torch.uint8, device=dev)
    _lib.check(lib.himo_dt0(d.shape[0], _lib.ptr(d), _lib.ptr(out), _lib.ptr(ws), _lib.stream_handle()), "dt0")
    return _back(out, _is_np(lidar_dt))
