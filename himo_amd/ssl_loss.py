"""Host side of stage a11: exact sweep-to-sweep nearest neighbours and the self-supervised loss terms
(`chamfer_dis`, `static_flow_loss`, `dynamic_chamfer_dis`, `cluster_based_pc0pc1`; names and unit weights from
assets/slurm/ssl-train-av2.sh:33).  PARITY UNPINNED -- the reference's `seflowppLoss` is in the absent
OpenSceneFlow submodule; definitions are in csrc/sslloss.hip, the oracle in oracle/sslloss_oracle.py.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

_lib.register({
    "himo_nn_grid_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "himo_nn_grid": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_ssl_loss_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "himo_ssl_loss": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_ssl_loss_ex": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_ssl_dyn_sizes": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "himo_ssl_loss_presized": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p]),
})

# search grid: 1 m BEV cells over the network range +- a margin (points beyond it are binned into border cells)
GRID_X0, GRID_Y0, GRID_CELL, GRID_W, GRID_H = -52.0, -52.0, 1.0, 104, 104
TERMS = ("chamfer_dis", "static_flow_loss", "dynamic_chamfer_dis", "cluster_based_pc0pc1")


def _f32(t, dev):
    return t.to(device=dev, dtype=torch.float32).contiguous()


def nn_grid(query: torch.Tensor, ref: torch.Tensor, return_index: bool = True):
    """Exact k=1 NN of (Nq,3) float32 queries in (Nr,3) references (device tensors): (squared distances, int32 rows)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    q, r = _f32(query, dev), _f32(ref, dev)
    nq, nr = q.shape[0], r.shape[0]
    d2 = torch.empty(nq, dtype=torch.float32, device=dev)
    idx = torch.empty(nq, dtype=torch.int32, device=dev) if return_index else None
    ws = torch.empty(int(lib.himo_nn_grid_workspace_bytes(max(nq, nr), GRID_W, GRID_H)), dtype=torch.uint8, device=dev)
    _lib.check(lib.himo_nn_grid(nq, _lib.ptr(q), nr, _lib.ptr(r), GRID_X0, GRID_Y0, GRID_CELL, GRID_W, GRID_H, _lib.ptr(d2),
                                _lib.ptr(idx), _lib.ptr(ws), ws.numel(), _lib.stream_handle()), "himo_nn_grid")
    return (d2, idx) if return_index else d2


class SeFlowLoss:
    """``loss(pc0, pc1, flow, label0, label1)`` -> ({term: float64 tensor}, total, d total / d flow).
    ``pc0`` must already be in pc1's frame (ego motion removed), labels: 0 static, > 0 dynamic cluster id."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        self.device = device if device is not None else _lib.require_gpu()
        self._ws = None

    def raw_neighbours(self, pc0, pc1):
        """(squared distances, int32 rows) of every pc0 point's nearest pc1 point -- the correspondences of the cluster term, which do
        not depend on the flow: a caller may compute them early (on another stream) and hand them to ``__call__`` as ``raw``.  The
        buffers are this object's own and are overwritten by the next call."""
        dev = self.device
        p0, p1 = _f32(pc0, dev)[:, :3].contiguous(), _f32(pc1, dev)[:, :3].contiguous()
        n0, n1 = p0.shape[0], p1.shape[0]
        if getattr(self, "_raw_d2", None) is None or self._raw_d2.numel() < n0:
            self._raw_d2 = torch.empty(max(n0, 1), dtype=torch.float32, device=dev)
            self._raw_idx = torch.empty(max(n0, 1), dtype=torch.int32, device=dev)
        need = int(self.lib.himo_nn_grid_workspace_bytes(max(n0, n1), GRID_W, GRID_H))
        if getattr(self, "_raw_ws", None) is None or self._raw_ws.numel() < need:
            self._raw_ws = torch.empty(need + 64, dtype=torch.uint8, device=dev)
        _lib.check(self.lib.himo_nn_grid(n0, _lib.ptr(p0), n1, _lib.ptr(p1), GRID_X0, GRID_Y0, GRID_CELL, GRID_W, GRID_H, _lib.ptr(self._raw_d2),
                                         _lib.ptr(self._raw_idx), _lib.ptr(self._raw_ws), self._raw_ws.numel(), _lib.stream_handle()), "himo_nn_grid")
        return self._raw_d2[:n0], self._raw_idx[:n0], (p0, p1)

    def dyn_sizes(self, label0, label1):
        """The sizes of the two dynamic subsets (label > 0), counted NOW on the current stream and copied to pinned host memory: hand
        the result to ``__call__`` as ``sizes`` and the loss call never blocks the host (it otherwise copies the two numbers back in
        the middle of the call, after everything enqueued before it -- a training step's whole forward pass -- has drained).  They
        depend on the labels only, so a training step counts them beside its forward pass (himo_amd/seflow/train.py).  The buffers
        are this object's own: one outstanding count at a time."""
        dev = self.device
        l0 = label0.to(device=dev, dtype=torch.int32).contiguous()
        l1 = label1.to(device=dev, dtype=torch.int32).contiguous()
        if getattr(self, "_sizes_dev", None) is None:
            self._sizes_dev = torch.zeros(2, dtype=torch.int32, device=dev)
            self._sizes_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        _lib.check(self.lib.himo_ssl_dyn_sizes(l0.shape[0], _lib.ptr(l0), l1.shape[0], _lib.ptr(l1), _lib.ptr(self._sizes_dev),
                                               _lib.stream_handle()), "himo_ssl_dyn_sizes")
        self._sizes_host.copy_(self._sizes_dev, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        return done, self._sizes_host, (l0, l1)

    def __call__(self, pc0, pc1, flow, label0, label1, n_labels: int | None = None, raw=None, sizes=None):
        dev = self.device
        p0, p1, f = _f32(pc0, dev)[:, :3].contiguous(), _f32(pc1, dev)[:, :3].contiguous(), _f32(flow, dev)
        l0 = label0.to(device=dev, dtype=torch.int32).contiguous()
        l1 = label1.to(device=dev, dtype=torch.int32).contiguous()
        n0, n1 = p0.shape[0], p1.shape[0]
        if f.shape != (n0, 3) or l0.shape != (n0,) or l1.shape != (n1,):
            raise ValueError("shape mismatch between points, flow and labels")
        if n_labels is None:
            n_labels = int(l0.max().item()) + 1 if n0 else 1
        need = int(self.lib.himo_ssl_loss_workspace_bytes(n0, n1, n_labels, GRID_W, GRID_H))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need + 64, dtype=torch.uint8, device=dev)
        loss = torch.zeros(5, dtype=torch.float64, device=dev)
        grad = torch.zeros((n0, 3), dtype=torch.float32, device=dev)
        if raw is not None and n0 > 0 and n1 > 0 and (raw[0].shape != (n0,) or raw[1].shape != (n0,)):
            raise ValueError("raw correspondences do not match pc0")
        if sizes is not None:
            sizes[0].synchronize()                               # the count's own event (long past by now), not the stream
            nd0, nd1 = int(sizes[1][0]), int(sizes[1][1])
            r0, r1 = (raw[0], raw[1]) if (raw is not None and n0 > 0 and n1 > 0) else (None, None)
            _lib.check(self.lib.himo_ssl_loss_presized(n0, n1, _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(f), _lib.ptr(l0), _lib.ptr(l1), n_labels,
                                                       GRID_X0, GRID_Y0, GRID_CELL, GRID_W, GRID_H, _lib.ptr(r0), _lib.ptr(r1), nd0, nd1,
                                                       _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(self._ws), self._ws.numel(),
                                                       _lib.stream_handle()), "himo_ssl_loss_presized")
            return {name: loss[k] for k, name in enumerate(TERMS)}, loss[4], grad
        if raw is not None and n0 > 0 and n1 > 0:
            _lib.check(self.lib.himo_ssl_loss_ex(n0, n1, _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(f), _lib.ptr(l0), _lib.ptr(l1), n_labels,
                                                 GRID_X0, GRID_Y0, GRID_CELL, GRID_W, GRID_H, _lib.ptr(raw[0]), _lib.ptr(raw[1]), _lib.ptr(loss),
                                                 _lib.ptr(grad), _lib.ptr(self._ws), self._ws.numel(), _lib.stream_handle()), "himo_ssl_loss_ex")
            return {name: loss[k] for k, name in enumerate(TERMS)}, loss[4], grad
        _lib.check(self.lib.himo_ssl_loss(n0, n1, _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(f), _lib.ptr(l0), _lib.ptr(l1), n_labels,
                                          GRID_X0, GRID_Y0, GRID_CELL, GRID_W, GRID_H, _lib.ptr(loss), _lib.ptr(grad),
                                          _lib.ptr(self._ws), self._ws.numel(), _lib.stream_handle()), "himo_ssl_loss")
        return {name: loss[k] for k, name in enumerate(TERMS)}, loss[4], grad


class _LossFn(torch.autograd.Function):
    """total loss as a differentiable function of the flow (correspondences are constants), for torch training loops."""

    @staticmethod
    def forward(ctx, flow, pc0, pc1, label0, label1, engine):
        _, total, grad = engine(pc0, pc1, flow.detach(), label0, label1)
        ctx.save_for_backward(grad)
        return total.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


def seflow_loss(flow, pc0, pc1, label0, label1, engine: SeFlowLoss | None = None):
    return _LossFn.apply(flow, pc0, pc1, label0, label1, engine or SeFlowLoss())
