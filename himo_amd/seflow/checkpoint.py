"""Checkpoint files of the scene-flow network: what the reference passes around as ``checkpoint=...seflowpp_best.ckpt``
(README.md:50) and keeps three of per run (``save_top_model=3``, assets/slurm/ssl-train-av2.sh:32).

The reference's ``.ckpt`` is a PyTorch-Lightning pickle of the absent ``OpenSceneFlow`` model, so its tensor names cannot
be known from this tree (PARITY UNPINNED).  This build's own container is a plain ``.npz``:
    every array of ``spec.param_shapes()`` under its own name (float32),
    optionally ``__adam_m__`` / ``__adam_v__`` (the trainer's flat moment buffers), ``__step__``, ``__epoch__``,
    ``__val__`` (the validation figure ``save_top`` ranks by).
``from_state_dict`` is the adapter hook for the day the reference's weights are readable: it converts torch's layouts
(conv ``[cout, cin, kh, kw]``, linear ``[out, in]``) to this build's (``[kh, kw, cin, cout]``, ``[in, out]``) through a
caller-supplied name map and checks every shape against ``spec.param_shapes()``.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

from . import spec

_EXTRA = ("__adam_m__", "__adam_v__", "__step__", "__epoch__", "__val__")


def check_params(params: dict) -> dict:
    """Every spec array present with the spec's shape -> float32 contiguous copies; KeyError / ValueError otherwise."""
    out = {}
    for name, shape in spec.param_shapes().items():
        if name not in params:
            raise KeyError(f"checkpoint has no array {name!r}")
        a = np.ascontiguousarray(params[name], dtype=np.float32)
        if a.shape != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape}, the network expects {tuple(shape)}")
        out[name] = a
    return out


def save_params(path, params: dict, **extra) -> Path:
    """Write ``params`` (+ optional optimiser state / counters as ``__name__`` arrays) atomically to ``path`` (.npz)."""
    path = Path(path)
    arrays = check_params(params)
    for k, v in extra.items():
        key = f"__{k}__"
        if key not in _EXTRA:
            raise KeyError(f"unknown checkpoint field {k!r}")
        arrays[key] = np.asarray(v)
    tmp = path.with_name(path.name + ".writing.npz")
    np.savez(tmp, **arrays)
    os.replace(tmp, path)
    return path


def load_params(path, with_extra: bool = False):
    """``path`` (.npz written by ``save_params`` / any npz of the spec arrays) -> parameter dict [, extra dict]."""
    with np.load(path) as z:
        arrays = {k: z[k] for k in z.files}
    params = check_params(arrays)
    if not with_extra:
        return params
    return params, {k[2:-2]: arrays[k] for k in _EXTRA if k in arrays}


def from_state_dict(state_dict: dict, name_map: dict) -> dict:
    """Torch-layout tensors -> this build's parameter dict.  ``name_map``: spec name -> state-dict key (or a callable
    ``state_dict -> array`` for fused / split tensors).  4-D tensors are taken as conv weights ``[cout, cin, kh, kw]``,
    2-D ones as linear weights ``[out, in]``; 1-D tensors pass through."""
    out = {}
    for name, shape in spec.param_shapes().items():
        src = name_map[name]
        a = src(state_dict) if callable(src) else state_dict[src]
        a = np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=np.float32)
        if a.ndim == 4 and len(shape) == 4:
            a = a.transpose(2, 3, 1, 0)
        elif a.ndim == 2 and len(shape) == 2:
            a = a.T
        out[name] = np.ascontiguousarray(a)
    return check_params(out)


def to_state_dict(params: dict) -> dict:
    """The inverse layout change (identity names): what a torch module mirroring ``spec`` would hold."""
    out = {}
    for name, a in check_params(params).items():
        out[name] = np.ascontiguousarray(a.transpose(3, 2, 0, 1) if a.ndim == 4 else a.T if a.ndim == 2 else a)
    return out


class TopK:
    """``save_top_model=k`` (ssl-train-av2.sh:32): keep the k checkpoints with the LOWEST validation figure in ``directory``."""

    def __init__(self, directory, k: int = 3, prefix: str = "seflowpp", resume: bool = False):
        """``resume=True`` (a run continued with ``fit(resume=...)``) adopts the ranking of the checkpoints already in
        ``directory`` -- else more than k accumulate over the restarts.  A FRESH run never adopts, ranks against or deletes
        files it did not write: checkpoints of another run in a reused directory are left alone (and a new file whose name
        collides with one of them replaces only that file)."""
        self.directory, self.k, self.prefix = Path(directory), int(k), prefix
        self.directory.mkdir(parents=True, exist_ok=True)
        self.kept = []                                       # (val, path)
        if not resume:
            return
        for path in sorted(self.directory.glob(f"{self.prefix}-epoch*-val*.npz")):
            try:
                with np.load(path) as z:
                    val = float(z["__val__"])
            except Exception:
                continue
            if np.isfinite(val):
                self.kept.append((val, path))
        self.kept.sort(key=lambda t: t[0])
        for _, old in self.kept[self.k:]:
            old.unlink(missing_ok=True)
        self.kept = self.kept[:self.k]

    def offer(self, val: float, epoch: int, params: dict, **extra):
        """Returns the path written, or None when ``val`` does not make the top k (a non-finite figure never does)."""
        val = float(val)
        if not np.isfinite(val):
            return None
        if len(self.kept) >= self.k and val >= max(v for v, _ in self.kept):
            return None
        path = save_params(self.directory / f"{self.prefix}-epoch{epoch:02d}-val{val:.4f}.npz", params, epoch=epoch, val=val, **extra)
        self.kept = [(v, p) for v, p in self.kept if p != path]          # re-offering an epoch replaces its file
        self.kept.append((val, path))
        self.kept.sort(key=lambda t: t[0])
        for _, old in self.kept[self.k:]:
            old.unlink(missing_ok=True)
        self.kept = self.kept[:self.k]
        return path

    def best(self):
        return self.kept[0][1] if self.kept else None
