"""Training side of the scene-flow network (stage a11 / BASELINE config 5): forward with saved activations, backward
pass and optimiser step, one process per GPU with a single flat gradient all-reduce over RCCL.

PARITY UNPINNED: the reference trains through ``OpenSceneFlow/train.py`` (assets/slurm/ssl-train-av2.sh:31-34), which
is absent.  Conventions of this build: BatchNorm statistics frozen (scale / shift are constants), Adam, unit-weight
SeFlow-style loss (himo_amd/ssl_loss.py).  The oracle is PyTorch CPU autograd through oracle/seflow_oracle.py.

This module currently implements the per-point HEAD (gather -> 4 GRU iterations -> MLP) forward/backward; the
convolutional backbone's backward pass is the next step.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _lib
from . import spec
from .model import ConvDesc, EPI_BIAS

c_p, c_i, c_l, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_lib.register({
    "himo_gru_gates_fwd": (c_i, [c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "himo_gru_bwd1": (c_i, [c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "himo_gru_bwd2": (c_i, [c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "himo_gru_bwd3": (c_i, [c_l, c_p, c_p, c_p, c_p, c_p]),
    "himo_affine_gelu_fwd": (c_i, [c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "himo_affine_gelu_bwd": (c_i, [c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p]),
    "himo_mask_rows": (c_i, [c_l, c_i, c_p, c_p, c_i, c_p]),
    "himo_wgrad_workspace_bytes_ex": (ctypes.c_size_t, [c_l, c_i, c_i]),
    "himo_linear_wgrad_ex": (c_i, [c_l, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_p, ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
    "himo_transpose": (c_i, [c_p, c_i, c_i, c_p, c_p]),
})


class HeadTrainer:
    """GRU head with saved states.  Parameters (device float32): ``zr.weight`` [192,256], ``zr.bias`` [256], ``q.weight``
    [192,128], ``q.bias`` [128], ``dec1.weight`` [192,32], ``dec1.bias`` [32], ``dec2.weight`` [32,4] (column 3 zero),
    ``dec2.bias`` [4]."""

    def __init__(self, params: dict, device=None):
        self.lib = _lib.load()
        self.device = device if device is not None else _lib.require_gpu()
        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        w2 = np.zeros((32, 4), np.float32); w2[:, :3] = params["head.dec2.weight"]
        b2 = np.zeros(4, np.float32); b2[:3] = params["head.dec2.bias"]
        self.p = {
            "zr.weight": t(np.concatenate([params["head.gru.z.weight"], params["head.gru.r.weight"]], axis=1)),
            "zr.bias": t(np.concatenate([params["head.gru.z.bias"], params["head.gru.r.bias"]])),
            "q.weight": t(params["head.gru.q.weight"]), "q.bias": t(params["head.gru.q.bias"]),
            "dec1.weight": t(params["head.dec1.weight"]), "dec1.bias": t(params["head.dec1.bias"]),
            "dec2.weight": t(w2), "dec2.bias": t(b2),
        }
        self.g = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.n = 0

    def _reserve(self, n):
        if n == self.n:
            return
        dev, T = self.device, spec.GRU_ITERS
        buf = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.n = n
        self.HX = [buf(n, 192) for _ in range(T + 1)]
        self.RHX = [buf(n, 192) for _ in range(T)]
        self.Z, self.R, self.Q = ([buf(n, 128) for _ in range(T)] for _ in range(3))
        self.AZR, self.AQ = buf(n, 256), buf(n, 128)
        self.PRE1, self.Y1, self.RES = buf(n, 32), buf(n, 32), buf(n, 4)
        self.A1 = buf(n, 32)
        # backward scratch
        self.DY1, self.DHX, self.DRHX = buf(n, 32), buf(n, 192), buf(n, 192)
        self.DH, self.DHP, self.DZ, self.DAQ, self.DAZR = buf(n, 128), buf(n, 128), buf(n, 128), buf(n, 128), buf(n, 256)
        self.DX = buf(n, 64)
        self.WT = {k: torch.empty((v.shape[1], v.shape[0]), dtype=torch.float32, device=dev) for k, v in self.p.items() if v.dim() == 2}
        need = int(self.lib.himo_wgrad_workspace_bytes_ex(n, 192, 256))
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)

    def _gemm(self, x, w, bias, y, cin, cout, epi=EPI_BIAS):
        d = ConvDesc()
        d.x = x.data_ptr(); d.x_pitch = x.shape[1]
        d.w = w.data_ptr(); d.bias = None if bias is None else bias.data_ptr()
        d.y = y.data_ptr(); d.y_pitch = y.shape[1]
        d.n, d.h, d.w_in, d.cin, d.cout, d.ksize, d.stride, d.epilogue = 1, 1, x.shape[0], cin, cout, 1, 1, epi
        _lib.check(self.lib.himo_conv2d(ctypes.byref(d), _lib.stream_handle()), "himo_conv2d(head)")

    def forward(self, hx0: torch.Tensor) -> torch.Tensor:
        """hx0 [n,192] = [h0 | x] -> res [n,4] (columns 0..2 = network flow, column 3 = 0)."""
        n = hx0.shape[0]
        self._reserve(n)
        lib, s, p = self.lib, _lib.stream_handle, self.p
        self.HX[0].copy_(hx0)
        for t in range(spec.GRU_ITERS):
            self._gemm(self.HX[t], p["zr.weight"], p["zr.bias"], self.AZR, 192, 256)
            _lib.check(lib.himo_gru_gates_fwd(n, 1, self.AZR.data_ptr(), None, self.HX[t].data_ptr(), self.Z[t].data_ptr(),
                                              self.R[t].data_ptr(), None, self.RHX[t].data_ptr(), s()), "gru_gates_fwd")
            self._gemm(self.RHX[t], p["q.weight"], p["q.bias"], self.AQ, 192, 128)
            _lib.check(lib.himo_gru_gates_fwd(n, 2, self.AQ.data_ptr(), self.Z[t].data_ptr(), self.HX[t].data_ptr(), None, None,
                                              self.Q[t].data_ptr(), self.HX[t + 1].data_ptr(), s()), "gru_gates_fwd")
        self._gemm(self.HX[-1], p["dec1.weight"], p["dec1.bias"], self.A1, 192, 32)
        _lib.check(lib.himo_affine_gelu_fwd(n, 32, self.A1.data_ptr(), 32, None, None, self.PRE1.data_ptr(), 32, self.Y1.data_ptr(), 32, s()), "gelu_fwd")
        self._gemm(self.Y1, p["dec2.weight"], p["dec2.bias"], self.RES, 32, 4)
        return self.RES

    def _wgrad(self, x, cin, dz, cout, name, accumulate=False):
        _lib.check(self.lib.himo_linear_wgrad_ex(x.shape[0], x.data_ptr(), x.shape[1], cin, dz.data_ptr(), dz.shape[1], cout,
                                                 self.g[f"{name}.weight"].data_ptr(), self.g[f"{name}.bias"].data_ptr(),
                                                 1 if accumulate else 0, self.ws.data_ptr(), self.ws.numel(), _lib.stream_handle()), "wgrad")

    def _transposed(self, name):
        w = self.p[f"{name}.weight"]
        _lib.check(self.lib.himo_transpose(w.data_ptr(), w.shape[0], w.shape[1], self.WT[f"{name}.weight"].data_ptr(), _lib.stream_handle()), "transpose")
        return self.WT[f"{name}.weight"]

    def backward(self, dres: torch.Tensor) -> torch.Tensor:
        """dres [n,4] (d loss / d res, column 3 ignored) -> d loss / d hx0 [n,192]; parameter gradients land in ``self.g``."""
        n = self.n
        lib, s = self.lib, _lib.stream_handle
        # dec2 / dec1
        self._wgrad(self.Y1, 32, dres, 4, "dec2")
        self._gemm(dres, self._transposed("dec2"), None, self.DY1, 4, 32)
        _lib.check(lib.himo_affine_gelu_bwd(n, 32, self.DY1.data_ptr(), 32, self.PRE1.data_ptr(), 32, None, self.DY1.data_ptr(), 32, s()), "gelu_bwd")
        self._wgrad(self.HX[-1], 192, self.DY1, 32, "dec1")
        self._gemm(self.DY1, self._transposed("dec1"), None, self.DHX, 32, 192)
        # split d[h | x] of the last state
        self.DH.copy_(self.DHX[:, :128])
        self.DX.copy_(self.DHX[:, 128:])
        wq_t, wzr_t = self._transposed("q"), self._transposed("zr")
        for t in range(spec.GRU_ITERS - 1, -1, -1):
            acc = t != spec.GRU_ITERS - 1
            _lib.check(lib.himo_gru_bwd1(n, self.DH.data_ptr(), self.Z[t].data_ptr(), self.Q[t].data_ptr(), self.HX[t].data_ptr(),
                                         self.DAQ.data_ptr(), self.DZ.data_ptr(), self.DHP.data_ptr(), s()), "gru_bwd1")
            self._wgrad(self.RHX[t], 192, self.DAQ, 128, "q", accumulate=acc)
            self._gemm(self.DAQ, wq_t, None, self.DRHX, 128, 192)
            _lib.check(lib.himo_gru_bwd2(n, self.DRHX.data_ptr(), self.HX[t].data_ptr(), self.Z[t].data_ptr(), self.R[t].data_ptr(),
                                         self.DZ.data_ptr(), self.DHP.data_ptr(), self.DAZR.data_ptr(), self.DX.data_ptr(), s()), "gru_bwd2")
            self._wgrad(self.HX[t], 192, self.DAZR, 256, "zr", accumulate=acc)
            self._gemm(self.DAZR, wzr_t, None, self.DHX, 256, 192)
            _lib.check(lib.himo_gru_bwd3(n, self.DHX.data_ptr(), self.DHP.data_ptr(), self.DH.data_ptr(), self.DX.data_ptr(), s()), "gru_bwd3")
        out = torch.empty((n, 192), dtype=torch.float32, device=self.device)
        out[:, :128].copy_(self.DH)
        out[:, 128:].copy_(self.DX)
        return out
