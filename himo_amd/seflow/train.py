"""Training side of the scene-flow network (stage a11 / BASELINE config 5): forward with saved activations, backward
pass and optimiser step, one process per GPU, the gradient exchange over RCCL bucket by bucket under the backward pass.

PARITY UNPINNED: the reference trains through ``OpenSceneFlow/train.py`` (assets/slurm/ssl-train-av2.sh:31-34), which
is absent.  Conventions of this build: BatchNorm in training mode by default (batch statistics over a pass's samples,
trainable gamma / beta; ``batchnorm="frozen"`` folds them to constants), Adam, unit-weight SeFlow-style loss
(himo_amd/ssl_loss.py).  The oracle is PyTorch CPU autograd through oracle/seflow_oracle.py.

``HeadTrainer``: the per-point head (4 GRU iterations -> MLP) with saved states and its BPTT backward pass.
``SeFlowTrainer``: the whole network -- pillar features, encoder, decoder, head -- over a per-process BATCH of samples in one
forward / backward pass (``batch=``), backward (every gradient a fixed-order reduction: no float atomics), one flat parameter /
gradient / Adam-moment buffer.
``BucketedAllReduce``: the step's data-parallel exchange, started per bucket as the backward pass completes it.
"""
from __future__ import annotations

import ctypes
import itertools
import os

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
    "himo_weight_flip": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "himo_zero_stuff2x": (c_i, [c_i, c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_l, c_i, c_p]),
    "himo_upsample2x_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "himo_conv_wgrad_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i, c_i]),
    "himo_conv_wgrad_batch_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "himo_conv3x3_wgrad_batch": (c_i, [c_i, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_l, c_i, c_i, c_i, c_p, ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
    "himo_conv3x3_wgrad_batch_bias": (c_i, [c_i, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_l, c_i, c_i, c_i, c_p, c_p, ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
    "himo_conv3x3_wgrad": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p, ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
})


# ---- stand-alone backward operators (tests): the kernels the trainer strings together ---------------------------------
def conv3x3_backward_nhwc(x: torch.Tensor, weight: torch.Tensor, dy: torch.Tensor, stride: int = 1, wgrad_flags: int = 0):
    """x [N,H,W,Cin], weight [3,3,Cin,Cout], dy [N,Ho,Wo,Cout] -> (dx [N,H,W,Cin], dw [3,3,Cin,Cout], db [Cout]) of
    y = conv3x3(x, weight, pad 1, stride) + b.  ``wgrad_flags`` 2: split-bf16 operands in the weight-gradient kernel."""
    from .model import conv2d_nhwc
    lib = _lib.load()
    n, h, w, cin = x.shape
    cout = weight.shape[3]
    ho, wo = dy.shape[1], dy.shape[2]
    s = _lib.stream_handle
    wf = torch.empty((3, 3, cout, cin), dtype=torch.float32, device=x.device)
    _lib.check(lib.himo_weight_flip(weight.data_ptr(), 3, cin, cout, wf.data_ptr(), s()), "himo_weight_flip")
    zero_b = torch.zeros(cin, dtype=torch.float32, device=x.device)
    if stride == 2:
        z = torch.empty((n, 2 * ho, 2 * wo, cout), dtype=torch.float32, device=x.device)
        _lib.check(lib.himo_zero_stuff2x(n, ho, wo, cout, dy.data_ptr(), ho * wo * cout, cout, z.data_ptr(), 4 * ho * wo * cout, cout, s()),
                   "himo_zero_stuff2x")
        dx = conv2d_nhwc(z, wf, zero_b)
    else:
        dx = conv2d_nhwc(dy, wf, zero_b)
    dw = torch.empty_like(weight)
    need = int(lib.himo_conv_wgrad_batch_workspace_bytes(n, h, w, cin, cout, stride))
    if need:                                                                                    # LDS-tiled batch kernel
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _lib.check(lib.himo_conv3x3_wgrad_batch(n, x.data_ptr(), h * w * cin, cin, h, w, cin, dy.data_ptr(), ho * wo * cout, cout, cout,
                                                stride, dw.data_ptr(), wgrad_flags, ws.data_ptr(), ws.numel(), s()), "himo_conv3x3_wgrad_batch")
        return dx, dw, dy.sum((0, 1, 2))
    ws = torch.empty(int(lib.himo_conv_wgrad_workspace_bytes(ho, wo, cin, cout)), dtype=torch.uint8, device=x.device)
    for i in range(n):
        _lib.check(lib.himo_conv3x3_wgrad(x[i].data_ptr(), cin, h, w, cin, dy[i].data_ptr(), cout, cout, stride, dw.data_ptr(),
                                          1 if i else 0, ws.data_ptr(), ws.numel(), s()), "himo_conv3x3_wgrad")
    return dx, dw, dy.sum((0, 1, 2))


def upsample2x_backward_nhwc(dy: torch.Tensor) -> torch.Tensor:
    """dy [2H,2W,C] -> dx [H,W,C]: adjoint of the bilinear x2 (align_corners) upsampling."""
    lib = _lib.load()
    h, w, c = dy.shape[0] // 2, dy.shape[1] // 2, dy.shape[2]
    dx = torch.empty((h, w, c), dtype=torch.float32, device=dy.device)
    _lib.check(lib.himo_upsample2x_bwd(dy.data_ptr(), c, h, w, c, dx.data_ptr(), c, _lib.stream_handle()), "himo_upsample2x_bwd")
    return dx


class HeadSaved(ctypes.Structure):
    """himo_head_saved (include/himo_amd.h)"""
    _fields_ = [("rows", ctypes.c_int64), ("d_hx", ctypes.c_void_p), ("d_rhx", ctypes.c_void_p), ("d_z", ctypes.c_void_p),
                ("d_r", ctypes.c_void_p), ("d_q", ctypes.c_void_p), ("d_pre1", ctypes.c_void_p), ("d_y1", ctypes.c_void_p),
                ("d_res", ctypes.c_void_p)]


class HeadTrainer:
    """GRU head with saved states.  Parameters (device float32): ``zr.weight`` [192,256], ``zr.bias`` [256], ``q.weight``
    [192,128], ``q.bias`` [128], ``dec1.weight`` [192,32], ``dec1.bias`` [32], ``dec2.weight`` [32,4] (column 3 zero),
    ``dec2.bias`` [4]."""

    @staticmethod
    def host_params(params: dict) -> dict:
        """spec parameter dict -> this class's layout (z|r fused, dec2 padded to 4 columns), numpy float32"""
        w2 = np.zeros((32, 4), np.float32); w2[:, :3] = params["head.dec2.weight"]
        b2 = np.zeros(4, np.float32); b2[:3] = params["head.dec2.bias"]
        return {
            "zr.weight": np.concatenate([params["head.gru.z.weight"], params["head.gru.r.weight"]], axis=1),
            "zr.bias": np.concatenate([params["head.gru.z.bias"], params["head.gru.r.bias"]]),
            "q.weight": params["head.gru.q.weight"], "q.bias": params["head.gru.q.bias"],
            "dec1.weight": params["head.dec1.weight"], "dec1.bias": params["head.dec1.bias"],
            "dec2.weight": w2, "dec2.bias": b2,
        }

    def __init__(self, params: dict | None = None, device=None, p: dict | None = None, g: dict | None = None,
                 precision: str = "f32"):
        """Either ``params`` (spec dict, copied to the device) or ``p`` / ``g``: device views owned by the caller.
        ``precision``: arithmetic of the 192-wide row GEMMs -- "f32" (float32 MFMA), "bf16x3" (split-bf16 forward and
        backward) or "mixed" (fp16-split forward: the GRU state and gates are O(1); split-bf16 backward).  The weights
        are re-packed on every call (a few small launches), so the owner may update them freely."""
        if precision not in ("f32", "bf16x3", "mixed"):
            raise ValueError(precision)
        self.fmt_fwd = {"f32": None, "bf16x3": 0, "mixed": 1}[precision]
        self.wgrad_flags = 2 if precision == "mixed" else 0           # split-bf16 operands in the weight-gradient GEMMs (csrc/fastnsf.hip)
        self.fmt_bwd = {"f32": None, "bf16x3": 0, "mixed": 2}[precision]      # mixed: two-term bf16 split (HIMO_PACK_BF16X2) for dX = dZ W^T
        self.lib = _lib.load()
        self.device = device if device is not None else _lib.require_gpu()
        dev = self.device
        if p is None:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
            self.p = {k: t(v) for k, v in self.host_params(params).items()}
            self.g = {k: torch.zeros_like(v) for k, v in self.p.items()}
        else:
            self.p, self.g = p, g
        self.n = 0
        self.rows = 0
        self._descs = {}
        self.WT = {k: torch.empty((v.shape[1], v.shape[0]), dtype=torch.float32, device=dev) for k, v in self.p.items() if v.dim() == 2}
        pk = lambda cin, cout: torch.empty(int(self.lib.himo_conv_packed_weight_bytes(1, cin, cout)), dtype=torch.uint8, device=dev)
        # split copies of the three wide matrices and of their transposes (dec2 is 32 x 4: stays float32)
        self.PK = {k: pk(*self.p[f"{k}.weight"].shape) for k in ("zr", "q", "dec1")} if self.fmt_fwd is not None else {}
        self.PKT = {k: pk(*self.p[f"{k}.weight"].shape[::-1]) for k in ("zr", "q", "dec1")} if self.fmt_bwd is not None else {}
        # True: the owner refreshes PK / PKT itself after every parameter update (SeFlowTrainer: one launch for the whole network)
        self.external_pack = False
        self.accumulate = False        # True: every weight / bias gradient of a backward pass is ADDED to what ``g`` holds (sample b > 0 of a batch)
        self.wgrad_stream = None       # a torch stream for the gate weight gradients of the fused backward (None: in place)
        self.fused_backward = True     # split precisions: the GRU iterations' backward sweep as one kernel (False: three element-wise
                                       # kernels around two row products per iteration -- kept as the statement the fused sweep is tested against)

    def weight_jobs(self):
        """(weight, ksize, cin, cout, format, flip, destination) per packed copy, for himo_weight_prepare_batch"""
        jobs = []
        for k in self.PK:
            cin, cout = self.p[f"{k}.weight"].shape
            jobs.append((self.p[f"{k}.weight"], 1, cin, cout, self.fmt_fwd, 0, self.PK[k]))
        for k in self.PKT:
            cin, cout = self.p[f"{k}.weight"].shape
            jobs.append((self.p[f"{k}.weight"], 1, cin, cout, self.fmt_bwd, 1, self.PKT[k]))
        return jobs

    def _reserve(self, n):
        """Buffers for n points.  Capacity only grows (real sweeps differ in size from sample to sample: reallocating ~2 GB of states per
        step would cost more than the step's head): the stacked tensors keep ``self.rows`` = the capacity as their iteration stride and
        the lists are views of the first n rows."""
        if n == self.n:
            return
        dev, T = self.device, spec.GRU_ITERS
        need_rows = (n + 63) // 64 * 64
        if need_rows > self.rows:
            rows = self.rows = (int(need_rows * 1.125) + 63) // 64 * 64 if self.rows else need_rows      # headroom once it has had to grow
            buf = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
            zbuf = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)      # (the unfused forward never writes the padding rows)
            # saved states: iterations stacked, rows padded to whole 64-row blocks (the fused forward writes whole blocks: himo_gru_head_train)
            self._HX, self._RHX = zbuf(T + 1, rows, 192), zbuf(T, rows, 192)
            self._Z, self._R, self._Q = zbuf(T, rows, 128), zbuf(T, rows, 128), zbuf(T, rows, 128)
            self._PRE1, self._Y1, self._RES = buf(rows, 32), buf(rows, 32), buf(rows, 4)
            self._AZR, self._AQ, self._A1 = buf(rows, 256), buf(rows, 128), buf(rows, 32)
            # backward scratch (the fused sweep's buffers are padded like the saved states; zeros in the padding of its input and outputs)
            self._DHX = zbuf(rows, 192)
            self._DAQ, self._DAZR, self._DHX0 = zbuf(T, rows, 128), zbuf(T, rows, 256), buf(rows, 192)
            self._DY1, self._DRHX = buf(rows, 32), buf(rows, 192)
            self._DH, self._DHP, self._DZ, self._DAQ1, self._DAZR1 = buf(rows, 128), buf(rows, 128), buf(rows, 128), buf(rows, 128), buf(rows, 256)
            self._DX = buf(rows, 64)
            self.ws = torch.empty(int(self.lib.himo_wgrad_workspace_bytes_ex(T * rows, 192, 256)), dtype=torch.uint8, device=dev)
            self._dirty = 0                              # rows of _DAQ / _DAZR / _DHX that may hold a previous sample's values
            self._descs.clear()
        self.n = n
        self.HX = [self._HX[t, :n] for t in range(T + 1)]
        self.RHX = [self._RHX[t, :n] for t in range(T)]
        self.Z, self.R, self.Q = ([x[t, :n] for t in range(T)] for x in (self._Z, self._R, self._Q))
        self.AZR, self.AQ, self.A1 = self._AZR[:n], self._AQ[:n], self._A1[:n]
        self.PRE1, self.Y1, self.RES = self._PRE1[:n], self._Y1[:n], self._RES[:n]
        self.DY1, self.DHX, self.DRHX = self._DY1[:n], self._DHX[:n], self._DRHX[:n]
        self.DH, self.DHP, self.DZ, self.DAQ, self.DAZR = self._DH[:n], self._DHP[:n], self._DZ[:n], self._DAQ1[:n], self._DAZR1[:n]
        self.DX = self._DX[:n]

    def _pack(self, w, buf, fmt):
        _lib.check(self.lib.himo_conv_pack_weights_ex(w.data_ptr(), 1, w.shape[0], w.shape[1], fmt, buf.data_ptr(), _lib.stream_handle()), "pack")
        return buf

    def _gemm(self, x, w, bias, y, cin, cout, epi=EPI_BIAS, packed=None, fmt=0):
        # descriptors are cached per call site (same buffers every step): filling a ConvDesc costs more host time
        # than the launch itself, and a training step makes ~450 launches
        key = (x.data_ptr(), x.shape[0], x.shape[1], w.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(), y.shape[1], cin, cout, epi,
               None if packed is None else packed.data_ptr(), fmt)
        d = self._descs.get(key)
        if d is None:
            d = ConvDesc()
            d.x = x.data_ptr(); d.x_pitch = x.shape[1]
            d.w = w.data_ptr(); d.bias = None if bias is None else bias.data_ptr()
            d.y = y.data_ptr(); d.y_pitch = y.shape[1]
            d.n, d.h, d.w_in, d.cin, d.cout, d.ksize, d.stride, d.epilogue = 1, 1, x.shape[0], cin, cout, 1, 1, epi
            if packed is not None:
                d.w_packed, d.packed_format = packed.data_ptr(), fmt
            if len(self._descs) > 4096:
                self._descs.clear()
            self._descs[key] = d
        _lib.check(self.lib.himo_conv2d(ctypes.byref(d), _lib.stream_handle()), "himo_conv2d(head)")

    def forward(self, hx0: torch.Tensor) -> torch.Tensor:
        """hx0 [n,192] = [h0 | x] -> res [n,4] (columns 0..2 = network flow, column 3 = 0)."""
        n = hx0.shape[0]
        self._reserve(n)
        lib, s, p = self.lib, _lib.stream_handle, self.p
        ff = self.fmt_fwd
        pk = self.PK if self.external_pack else {k: self._pack(p[f"{k}.weight"], self.PK[k], ff) for k in self.PK}
        self.HX[0].copy_(hx0)
        for t in range(spec.GRU_ITERS):
            self._gemm(self.HX[t], p["zr.weight"], p["zr.bias"], self.AZR, 192, 256, packed=pk.get("zr"), fmt=ff or 0)
            _lib.check(lib.himo_gru_gates_fwd(n, 1, self.AZR.data_ptr(), None, self.HX[t].data_ptr(), self.Z[t].data_ptr(),
                                              self.R[t].data_ptr(), None, self.RHX[t].data_ptr(), s()), "gru_gates_fwd")
            self._gemm(self.RHX[t], p["q.weight"], p["q.bias"], self.AQ, 192, 128, packed=pk.get("q"), fmt=ff or 0)
            _lib.check(lib.himo_gru_gates_fwd(n, 2, self.AQ.data_ptr(), self.Z[t].data_ptr(), self.HX[t].data_ptr(), None, None,
                                              self.Q[t].data_ptr(), self.HX[t + 1].data_ptr(), s()), "gru_gates_fwd")
        self._gemm(self.HX[-1], p["dec1.weight"], p["dec1.bias"], self.A1, 192, 32, packed=pk.get("dec1"), fmt=ff or 0)
        _lib.check(lib.himo_affine_gelu_fwd(n, 32, self.A1.data_ptr(), 32, None, None, self.PRE1.data_ptr(), 32, self.Y1.data_ptr(), 32, s()), "gelu_fwd")
        self._gemm(self.Y1, p["dec2.weight"], p["dec2.bias"], self.RES, 32, 4)
        return self.RES

    def forward_fused(self, n, pid, offsets, img0, img1, img_pitch, dec, dec_pitch, w_off, b_off, nonfinite=None) -> torch.Tensor:
        """The same forward as ``himo_head_gather`` + ``forward`` + the row mask, as ONE launch (csrc/gruhead.hip with its saves
        enabled): device addresses of the point's cell ids / offsets, the two 32-channel image groups and the decoder map in;
        res [n,4] out (zeros for dropped points), every state the backward pass reads saved.  Split precisions only."""
        if self.fmt_fwd is None:
            raise ValueError("the fused training forward needs precision 'bf16x3' or 'mixed'")
        self._reserve(n)
        if not self.external_pack:
            for k in self.PK:
                self._pack(self.p[f"{k}.weight"], self.PK[k], self.fmt_fwd)
        sv = self._saved()
        p = self.p
        _lib.check(self.lib.himo_gru_head_train(n, pid, offsets, img0, img1, img_pitch, dec, dec_pitch, w_off, b_off,
                                                self.PK["zr"].data_ptr(), p["zr.bias"].data_ptr(), self.PK["q"].data_ptr(), p["q.bias"].data_ptr(),
                                                self.PK["dec1"].data_ptr(), p["dec1.bias"].data_ptr(), p["dec2.weight"].data_ptr(), 4,
                                                p["dec2.bias"].data_ptr(), spec.GRU_ITERS, self.fmt_fwd, ctypes.byref(sv),
                                                None if nonfinite is None else nonfinite.data_ptr(), _lib.stream_handle()), "himo_gru_head_train")
        return self.RES

    def _backward_fused(self) -> torch.Tensor:
        """The GRU iterations' backward sweep as one launch (csrc/gruheadbwd.hip) + ONE weight-gradient product per matrix over the
        stacked iterations (rows of all T iterations; the sweep leaves zeros in the padding rows of the gate gradients)."""
        T, n = spec.GRU_ITERS, self.n
        n_pad = (n + 63) // 64 * 64
        sv = self._saved()
        if not self.external_pack:
            for k in ("q", "zr"):
                self._transposed_packed(k)
        if n_pad < self._dirty:                          # a larger sample left gate gradients behind the rows this sweep writes
            self._DAQ[:, n_pad:self._dirty].zero_()
            self._DAZR[:, n_pad:self._dirty].zero_()
        self._dirty = n_pad
        _lib.check(self.lib.himo_gru_head_backward(n, T, self._DHX.data_ptr(), ctypes.byref(sv), self.PKT["q"].data_ptr(),
                                                   self.PKT["zr"].data_ptr(), self.fmt_bwd, self._DAQ.data_ptr(), self._DAZR.data_ptr(),
                                                   self._DHX0.data_ptr(), _lib.stream_handle()), "himo_gru_head_backward")
        if self.rows <= n_pad + n_pad // 4:              # one product per matrix over the stacked iterations (zeros in the padding rows)
            runs = [(T * self.rows, 0, 0)]
        else:                                            # a much smaller sample than the capacity: iteration by iteration, n rows each
            runs = [(n, t, 1 if t else 0) for t in range(T)]
        def gate_weight_gradients():
            for name, x, dz, cout in (("q", self._RHX, self._DAQ, 128), ("zr", self._HX, self._DAZR, 256)):
                for rows, t, acc in runs:
                    _lib.check(self.lib.himo_linear_wgrad_ex(rows, x[t].data_ptr(), 192, 192, dz[t].data_ptr(), cout, cout,
                                                             self.g[f"{name}.weight"].data_ptr(), self.g[f"{name}.bias"].data_ptr(),
                                                             self.wgrad_flags | acc | (1 if self.accumulate else 0), self.ws.data_ptr(), self.ws.numel(),
                                                             _lib.stream_handle()), "wgrad")
        # 1.5 GB of saved states and gate gradients, read only: beside whatever the caller's stream does next (SeFlowTrainer: the
        # decoder's backward pass); the caller waits for this stream before it reads the gradients or runs the next backward
        self._aside(gate_weight_gradients)
        return self._DHX0[:n]

    def _aside(self, fn):
        """run a weight gradient (it only reads its operands; they all share ``self.ws``, so they stay in order on ONE stream) on
        ``wgrad_stream`` from the point the current stream has reached, or in place when there is none"""
        if self.wgrad_stream is None or self.fmt_bwd is None or not self.fused_backward:
            return fn()                  # (the iteration-by-iteration backward reuses its gate-gradient buffers: everything in place)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.wgrad_stream):
            self.wgrad_stream.wait_event(ready)
            fn()

    def _saved(self):
        sv = HeadSaved()
        sv.rows = self.rows
        sv.d_hx, sv.d_rhx, sv.d_z, sv.d_r, sv.d_q = (t.data_ptr() for t in (self._HX, self._RHX, self._Z, self._R, self._Q))
        sv.d_pre1, sv.d_y1, sv.d_res = self._PRE1.data_ptr(), self._Y1.data_ptr(), self._RES.data_ptr()
        return sv

    def _wgrad(self, x, cin, dz, cout, name, accumulate=False):
        _lib.check(self.lib.himo_linear_wgrad_ex(x.shape[0], x.data_ptr(), x.shape[1], cin, dz.data_ptr(), dz.shape[1], cout,
                                                 self.g[f"{name}.weight"].data_ptr(), self.g[f"{name}.bias"].data_ptr(),
                                                 (1 if (accumulate or self.accumulate) else 0) | self.wgrad_flags, self.ws.data_ptr(), self.ws.numel(),
                                                 _lib.stream_handle()), "wgrad")

    def _transposed(self, name):
        w = self.p[f"{name}.weight"]
        _lib.check(self.lib.himo_transpose(w.data_ptr(), w.shape[0], w.shape[1], self.WT[f"{name}.weight"].data_ptr(), _lib.stream_handle()), "transpose")
        return self.WT[f"{name}.weight"]

    def _transposed_packed(self, name):
        """(transposed float32 weights, their split-bf16 copy or None) for a data gradient dX = dZ W^T"""
        if self.external_pack and name in self.PKT:          # the product reads only the packed copy; the float32 address must just be valid
            return self.p[f"{name}.weight"], self.PKT[name]
        wt = self._transposed(name)
        return wt, (self._pack(wt, self.PKT[name], self.fmt_bwd) if name in self.PKT else None)

    def backward(self, dres: torch.Tensor) -> torch.Tensor:
        """dres [n,4] (d loss / d res, column 3 ignored) -> d loss / d hx0 [n,192]; parameter gradients land in ``self.g``."""
        n = self.n
        lib, s = self.lib, _lib.stream_handle
        # dec2 / dec1 (their weight gradients only read: beside the chain when a weight-gradient stream is set; ``dres`` must then stay
        # alive and unmodified until that stream has been waited for -- SeFlowTrainer.backward owns it and waits at its end)
        self._aside(lambda: self._wgrad(self.Y1, 32, dres, 4, "dec2"))
        self._gemm(dres, self._transposed("dec2"), None, self.DY1, 4, 32)
        _lib.check(lib.himo_affine_gelu_bwd(n, 32, self.DY1.data_ptr(), 32, self.PRE1.data_ptr(), 32, None, self.DY1.data_ptr(), 32, s()), "gelu_bwd")
        self._aside(lambda: self._wgrad(self.HX[-1], 192, self.DY1, 32, "dec1"))
        w1_t, w1_p = self._transposed_packed("dec1")
        self._gemm(self.DY1, w1_t, None, self.DHX, 32, 192, packed=w1_p, fmt=self.fmt_bwd or 0)
        if self.fmt_bwd is not None and self.fused_backward:
            return self._backward_fused()
        # split d[h | x] of the last state
        self.DH.copy_(self.DHX[:, :128])
        self.DX.copy_(self.DHX[:, 128:])
        (wq_t, wq_p), (wzr_t, wzr_p) = self._transposed_packed("q"), self._transposed_packed("zr")
        for t in range(spec.GRU_ITERS - 1, -1, -1):
            acc = t != spec.GRU_ITERS - 1
            _lib.check(lib.himo_gru_bwd1(n, self.DH.data_ptr(), self.Z[t].data_ptr(), self.Q[t].data_ptr(), self.HX[t].data_ptr(),
                                         self.DAQ.data_ptr(), self.DZ.data_ptr(), self.DHP.data_ptr(), s()), "gru_bwd1")
            self._wgrad(self.RHX[t], 192, self.DAQ, 128, "q", accumulate=acc)
            self._gemm(self.DAQ, wq_t, None, self.DRHX, 128, 192, packed=wq_p, fmt=self.fmt_bwd or 0)
            _lib.check(lib.himo_gru_bwd2(n, self.DRHX.data_ptr(), self.HX[t].data_ptr(), self.Z[t].data_ptr(), self.R[t].data_ptr(),
                                         self.DZ.data_ptr(), self.DHP.data_ptr(), self.DAZR.data_ptr(), self.DX.data_ptr(), s()), "gru_bwd2")
            self._wgrad(self.HX[t], 192, self.DAZR, 256, "zr", accumulate=acc)
            self._gemm(self.DAZR, wzr_t, None, self.DHX, 256, 192, packed=wzr_p, fmt=self.fmt_bwd or 0)
            _lib.check(lib.himo_gru_bwd3(n, self.DHX.data_ptr(), self.DHP.data_ptr(), self.DH.data_ptr(), self.DX.data_ptr(), s()), "gru_bwd3")
        out = torch.empty((n, 192), dtype=torch.float32, device=self.device)
        out[:, :128].copy_(self.DH)
        out[:, 128:].copy_(self.DX)
        return out


_lib.register({
    "himo_gru_head_train": (c_i, [c_l, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p,
                                  c_i, c_i, c_p, c_p, c_p]),
    "himo_gru_head_backward": (c_i, [c_l, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    "himo_weight_job_blocks": (c_i, [c_i, c_i, c_i, c_i]),
    "himo_weight_prepare_batch": (c_i, [c_p, c_i, c_i, c_p]),
    "himo_add2d": (c_i, [c_l, c_i, c_p, c_i, c_p, c_i, c_p]),
    "himo_colsum": (c_i, [c_l, c_p, c_i, c_i, c_p, ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
    "himo_pfn_backward_workspace_bytes": (ctypes.c_size_t, []),
    "himo_pfn_backward": (c_i, [c_l, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
    "himo_head_scatter": (c_i, [c_l, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "himo_adam_step": (c_i, [c_l, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_i, c_p]),
    # BatchNorm in training mode (csrc/batchnorm.hip, csrc/pillar.hip)
    "himo_bn_workspace_bytes": (ctypes.c_size_t, [c_l, c_i]),
    "himo_bn_train_fwd": (c_i, [c_i, c_l, c_i, c_p, c_l, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_l, c_i,
                                c_p, ctypes.c_size_t, c_p]),
    "himo_bn_train_bwd": (c_i, [c_i, c_l, c_i, c_p, c_l, c_i, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_p, ctypes.c_uint,
                                c_p, ctypes.c_size_t, c_p]),
    "himo_bn_train_bwd_x": (c_i, [c_i, c_l, c_i, c_p, c_l, c_i, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_p, ctypes.c_uint,
                                  c_p, ctypes.c_size_t, c_p]),
    "himo_bn_fold": (c_i, [c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p]),
    "himo_pfn_bn_workspace_bytes": (ctypes.c_size_t, []),
    "himo_pfn_bn_stats": (c_i, [c_l, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                ctypes.c_size_t, c_p]),
    "himo_pfn_bn_stats_multi": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                      ctypes.c_size_t, c_p]),
    "himo_pfn_backward_bn_multi": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                         ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
    "himo_pfn_bn_stats_groups": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                       ctypes.c_size_t, c_p]),
    "himo_pfn_backward_bn_groups": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                          ctypes.c_uint, c_p, ctypes.c_size_t, c_p]),
    "himo_pillar_features_multi": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, ctypes.c_size_t, c_i, c_p]),
    "himo_pfn_backward_bn": (c_i, [c_l, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, ctypes.c_uint,
                                   c_p, ctypes.c_size_t, c_p]),
})

BN_MOMENTUM = 0.1                 # torch.nn.BatchNorm's default; the reference's launcher sets none (ssl-train-av2.sh:31-34)


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place mean over the ranks of the default process group (no-op without one): the data-parallel exchange of a
    training step is this single collective on the flat gradient buffer."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():          # also a one-rank group: the collective itself is exercised
        if flat.is_cuda and dist.get_backend() == "gloo":
            # gloo (HIMO_DIST_BACKEND=gloo, bench.py --share-gpu: ranks sharing a device, where RCCL cannot be used) reduces host
            # memory: stage through the host; RCCL reduces the device buffer in place
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if dist.get_world_size() > 1:
            flat.div_(dist.get_world_size())
    return flat


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place SUM over the ranks of the default process group (no-op without one)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if flat.is_cuda and dist.get_backend() == "gloo":         # see allreduce_mean_
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def global_count_known(local_count: int, global_count: int | None) -> bool:
    """Does the HOST know the global batch's sample count?  Without a process group it is the local count; with one, only when the
    caller passed it (``fit`` knows every step's global size).  Raises for a step without any sample.  When the answer is no the count
    is read back from the device after the exchange -- a host wait for the whole step, every step: the launch thread then starts
    each step's enqueue only after the previous step has FINISHED (56.4 ms of a 56.6 ms step in ``.item()`` at 8 samples per pass,
    scripts/exp_train_host.py), and everything the host does between steps -- fetching the next samples from the feeder -- is device
    idle time."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        global_count = local_count
    if global_count is None:
        return False
    if global_count <= 0:
        raise ValueError("train_batch needs at least one sample on some rank")
    return True


def combine_batch_(flat_acc: torch.Tensor, flat_g: torch.Tensor, count_known: bool = False) -> torch.Tensor:
    """The data-parallel exchange of ``train_batch``: ``flat_acc`` = [this rank's gradient SUM | its sample count | its loss
    sum] -> ONE all-reduce (sum over ranks) -> ``flat_g`` = global gradient sum / global count.  Returns the global mean loss.
    Every sample of the global batch gets weight 1 / count whatever the split over the ranks (2,1,1,1 or an empty rank).
    ``count_known``: the host has checked that the global count is positive (``global_count_known``): no read-back."""
    n = flat_g.numel()
    allreduce_sum_(flat_acc)
    count = flat_acc[n]
    if not count_known and float(count.item()) <= 0.0:
        raise ValueError("train_batch needs at least one sample on some rank")
    torch.div(flat_acc[:n], count, out=flat_g)
    return (flat_acc[n + 1] / count).clone()


# The backward pass finishes the gradients in three groups, long before its last kernel: head + decoder first, then the two coarse
# encoder stages, then the fine stage and the pillar net.  A bucket is the parameters of one group = a few ranges of the flat buffer
# (the BatchNorm gamma / beta of the encoder layers live at the end of the flat layout, after everything else).
BUCKETS = (("head + decoder", ("dec", "head.")), ("encoder stages 3 + 2", ("enc3.", "enc2.")), ("encoder stage 1 + pillar net", ("enc1.", "pfn.")))


def bucket_ranges(names, offsets) -> list:
    """[[(lo, hi), ...] per bucket of ``BUCKETS``] over the flat parameter layout (``offsets[name] = (lo, hi)`` in floats, padded):
    the names of a bucket merged into maximal runs of consecutive names.  Every name falls in exactly one bucket."""
    which = []
    for k in names:
        hit = [i for i, (_, prefixes) in enumerate(BUCKETS) if k.startswith(prefixes)]
        if len(hit) != 1:
            raise ValueError(f"parameter {k!r} belongs to {len(hit)} gradient buckets")
        which.append(hit[0])
    out = [[] for _ in BUCKETS]
    for i, k in enumerate(names):
        lo, hi = offsets[k]
        runs = out[which[i]]
        if runs and i > 0 and which[i - 1] == which[i] and runs[-1][1] == lo:
            runs[-1] = (runs[-1][0], hi)
        else:
            runs.append((lo, hi))
    return out


class BucketedAllReduce:
    """The data-parallel exchange of a step as one SUM all-reduce per range of a bucket, each started as soon as the backward pass
    has finished that bucket -- on RCCL's own stream, beside the rest of the backward pass -- instead of ONE flat all-reduce after
    its last kernel (28 MB over xGMI: ~0.3 ms of an 8 ms step at 8 ranks, DESIGN section 6, and the only serial piece of the N > 1
    path).  ``launch(k, after)``: bucket k of ``flat`` is complete once the events ``after`` have passed; ``words``: a small tensor
    (sample count, loss sum) that rides along, reduced first.  ``wait()``: every range has been reduced (stream-ordered for RCCL,
    on the host for gloo).  The sums are those of the flat all-reduce, element by element."""

    def __init__(self, flat: torch.Tensor, buckets, words: torch.Tensor | None = None):
        import torch.distributed as dist
        self.flat, self.buckets, self.words = flat, buckets, words
        self.active = dist.is_available() and dist.is_initialized()
        self.backend = dist.get_backend() if self.active else None
        self.works, self.launched = [], set()
        # ONE communication stream per device and process (an exchange object is built every step: a stream per object would walk torch's
        # stream pool across the hardware queues, side_streams below)
        self.stream = comm_stream(flat.device) if (flat.is_cuda and self.active) else None

    def _reduce(self, t: torch.Tensor):
        import torch.distributed as dist
        if t.is_cuda and self.backend == "gloo":                 # ranks sharing a device (bench.py --share-gpu): through the host, blocking
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
        else:
            self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))

    def launch(self, k: int, after=()):
        if k in self.launched:
            raise RuntimeError(f"gradient bucket {k} launched twice")
        self.launched.add(k)
        if not self.active:
            return
        if self.stream is None:                                  # host tensors (tests): nothing to order against
            if self.words is not None and len(self.launched) == 1:
                self._reduce(self.words)
            for lo, hi in self.buckets[k]:
                self._reduce(self.flat[lo:hi])
            return
        with torch.cuda.stream(self.stream):
            for ev in after:
                self.stream.wait_event(ev)
            if self.backend == "gloo":
                self.stream.synchronize()
            if self.words is not None and len(self.launched) == 1:
                self._reduce(self.words)
            for lo, hi in self.buckets[k]:
                self._reduce(self.flat[lo:hi])

    def wait(self):
        if len(self.launched) != len(self.buckets):
            raise RuntimeError(f"only the gradient buckets {sorted(self.launched)} of {len(self.buckets)} were exchanged")
        for w in self.works:
            w.wait()                                             # RCCL: the current stream waits; gloo: the host does
        self.works = []


_SIDE_STREAMS = {}
_COMM_STREAMS = {}


def comm_stream(device):
    """The stream the gradient exchange runs on: one per device and process (see ``side_streams``)."""
    key = (device.type, device.index)
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COMM_STREAMS[key]


def side_streams(device) -> tuple:
    """The three side streams of the training step, ONE set per device and process: every trainer of a process shares them.  The HIP
    runtime multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 8 here) and streams that share a queue
    serialise; a process that builds trainer after trainer (bench.py's legs, a sweep over configurations) would otherwise walk
    through torch's stream pool until a side stream lands on the queue of the stream it is meant to run beside (measured: the 8-sample
    step 58 -> 71 ms in the default bench process, 58 ms in a fresh one)."""
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = tuple(torch.cuda.Stream(device=device) for _ in range(3))
    return _SIDE_STREAMS[key]


class SeFlowTrainer:
    """Whole-network training step.  PARITY UNPINNED (``OpenSceneFlow/train.py`` is absent); conventions of this build:
    ``batchnorm="batch"`` (default) is what a from-scratch job runs (assets/slurm/ssl-train-av2.sh:31-34 passes no
    checkpoint): every BatchNorm normalises with the statistics of the current forward (pillar net: the in-range points of
    each sweep; encoder layers: the sample's F frames as the batch), gamma / beta are trained and the running statistics
    follow with momentum 0.1 -- torch.nn.BatchNorm semantics; ``forward(training=False)`` (validation) uses the running
    statistics.  ``batchnorm="frozen"`` is the fine-tuning convention: running statistics AND affine folded into constant
    scale / shift, trainable = every weight and bias.

        tr = SeFlowTrainer(params)                       # spec parameter dict
        res = tr.forward(pch1, pc0, pc1, pose_h1, pose0, pose1)     # [n0, 4] network flow of pc0 rows (col 3 = 0)
        tr.backward(dres)                                # d loss / d res -> tr.flat_g (and the views tr.g[name])
        tr.allreduce(); tr.adam_step(lr)
    """

    def __init__(self, params: dict | None = None, device=None, max_points: int = 140_000, seed: int = 0,
                 precision: str = "bf16x3", batchnorm: str = "batch", batch: int = 1):
        """``batch``: samples one forward / backward pass takes (the launcher's ``batch_size=8`` on one process,
        assets/slurm/ssl-train-av2.sh:32-34): every encoder layer runs over ``batch`` x F images in ONE launch, every decoder
        layer over ``batch`` images, and training-mode BatchNorm takes its statistics over the whole batch, as torch does
        (``forward_batch`` / ``backward_batch`` / ``train_batch``); saved activations and gradient buffers are ``batch`` times
        the single-sample ones (~7 GB per sample at 120k points: 56 GB at batch = 8).  Layout: the maps whose frames are channel groups (pillar
        images, the stage outputs the decoder concatenates) hold ALL the batch's images as channel groups of one pixel-major buffer
        -- image b * F + f at channel offset C * (b * F + f), pitch C * F * batch -- so that every kernel that walks "n images with a
        stride" (convolutions, BatchNorm, weight gradients) sees the batch as n = batch * F images, and the decoder reads sample
        b's concatenation as channels [C * F * b, C * F * (b + 1)); every other map is [image][pixel][channel].

        ``precision``: "bf16x3" runs every stride-1 convolution of the forward and data-gradient passes as split-bf16
        on the matrix cores (float32-class accuracy, csrc/convbf.hip; weights are re-packed after every optimiser step);
        "mixed" runs the FORWARD convolutions as the two-term fp16 split (activations are O(1): inside fp16's range) and
        keeps the data gradient split-bf16 (gradients are far below fp16's subnormal floor); "f32" keeps float32 MFMA.
        Weight gradients are float32 MFMA either way."""
        from .model import SeFlowNet
        if precision not in ("bf16x3", "mixed", "f32"):
            raise ValueError(precision)
        if batchnorm not in ("batch", "frozen"):
            raise ValueError(batchnorm)
        if not 1 <= batch <= 16:
            raise ValueError("batch must be in 1..16")
        self.B = B = batch
        self.nb = 1
        self.precision = precision
        self.bn_batch = batchnorm == "batch"
        self.fwd_format = 1 if precision == "mixed" else 0          # HIMO_PACK_F16X2 / HIMO_PACK_BF16X3
        # mixed: the 3x3 weight gradients multiply split-bf16 operands (16 significant bits, float32 sums) on the 16-bit matrix
        # instructions (csrc/train.hip conv_wgrad_split_kernel / conv_wgrad_split2_kernel); the other modes keep float32 ones
        self.wgrad_flags = 2 if precision == "mixed" else 0
        # mixed: the data-gradient convolutions (3x3 and 1x1) run the two-term bf16 split (HIMO_PACK_BF16X2: 16 significant bits, float32
        # range, three matrix products per block) instead of the three-term one (six)
        self.bwd3_format = 2 if precision == "mixed" else 0
        self.lib = _lib.load()
        self.device = dev = device if device is not None else _lib.require_gpu()
        params = spec.init_params(seed, fresh_bn=self.bn_batch) if params is None else params      # from scratch: BatchNorm reset
        net = self.net = SeFlowNet(params, device=dev, max_points=1, precision="f32", autotune=False, max_batch=B)
        net.fold_decoder = False                          # the backward pass reads every decoder layer's own output
        net.keep_cell_lists = True
        net.use_plan = False
        net.max_points = 0
        net._reserve_points(max_points)
        H, W, F = net.H, net.W, net.F
        # ---- one flat buffer for parameters / gradients / Adam moments; net.p entries become views of it
        host = {"pfn.weight": params["pfn.weight"]}
        for name, *_ in spec.ENCODER:
            host[f"{name}.weight"], host[f"{name}.bias"] = params[f"{name}.weight"], params[f"{name}.bias"]
        for name, *_ in spec.DECODER:
            for u in ("u1", "u3", "u4", "u5"):
                host[f"{name}.{u}.weight"], host[f"{name}.{u}.bias"] = params[f"{name}.{u}.weight"], params[f"{name}.{u}.bias"]
        host["dec4.weight"], host["dec4.bias"] = params["dec4.weight"], params["dec4.bias"]
        host["head.offset.weight"], host["head.offset.bias"] = params["head.offset.weight"], params["head.offset.bias"]
        for k, v in HeadTrainer.host_params(params).items():
            host[f"head.{k}"] = v
        self.bn_layers = [("pfn.bn", spec.BN_EPS_PFN, "pfn")] + [(f"{n}.bn", spec.BN_EPS, n) for n, *_ in spec.ENCODER]
        if self.bn_batch:                                        # gamma / beta join the trainable set (at the end: the layout of
            for prefix, _, _ in self.bn_layers:                  # everything else is the frozen mode's)
                host[f"{prefix}.gamma"], host[f"{prefix}.beta"] = params[f"{prefix}.gamma"], params[f"{prefix}.beta"]
        self.names = list(host)
        sizes = [int(np.prod(host[k].shape)) for k in self.names]
        pad = lambda n: (n + 3) // 4 * 4                                   # 16-byte aligned views
        total = sum(pad(n) for n in sizes)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.p, self.g = {}, {}
        o, offsets = 0, {}
        for k, n in zip(self.names, sizes):
            shp = tuple(host[k].shape)
            self.p[k] = self.flat_p[o:o + n].view(shp)
            self.g[k] = self.flat_g[o:o + n].view(shp)
            self.p[k].copy_(torch.from_numpy(np.ascontiguousarray(host[k], dtype=np.float32)))
            offsets[k] = (o, o + pad(n))
            o += pad(n)
        self.buckets = bucket_ranges(self.names, offsets)        # the flat ranges the backward pass completes together (BucketedAllReduce)
        self.overlap_allreduce = os.environ.get("HIMO_TRAIN_OVERLAP_ALLREDUCE", "1") != "0"
        for k in self.names:
            if not k.startswith("head.") or k.startswith("head.offset"):
                net.p[k] = self.p[k]                              # incl. the BatchNorm gamma / beta views in batch mode
        hp = {k[5:]: v for k, v in self.p.items() if k.startswith("head.") and not k.startswith("head.offset")}
        hg = {k[5:]: v for k, v in self.g.items() if k.startswith("head.") and not k.startswith("head.offset")}
        # one head state per sample of the batch (saved GRU states for the backward pass: ~1.5 GB per 120k points), same parameters,
        # gradients and packed weights; sample b > 0 ADDS its weight gradients to what sample 0 wrote
        self.heads = [HeadTrainer(device=dev, p=hp, g=hg, precision=precision) for _ in range(B)]
        self.head = self.heads[0]
        for h_ in self.heads[1:]:
            h_.PK, h_.PKT, h_.WT, h_.accumulate = self.head.PK, self.head.PKT, self.head.WT, True
        self.step_count = 0
        self._descs = {}
        # ---- saved encoder activations: PRE (after BN, before GELU) and Y per layer, frames as the batch
        buf = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        self.layers = []          # (name, cin, cout, stride, h_in, w_in, ho, wo, last_of_stage)
        h, w = H, W
        for i, (name, cin, cout, stride) in enumerate(spec.ENCODER):
            ho, wo = (h // 2, w // 2) if stride == 2 else (h, w)
            last = i + 1 == len(spec.ENCODER) or spec.ENCODER[i + 1][3] == 2
            self.layers.append((name, cin, cout, stride, h, w, ho, wo, last))
            h, w = ho, wo
        BF = B * F
        self.PRE = [buf(BF, L[6] * L[7], L[2]) for L in self.layers]
        self.Y = [None if L[8] else buf(BF, L[6] * L[7], L[2]) for L in self.layers]
        # the maps whose frames are channel groups: pixel-major, ALL the batch's images as channel groups (class docstring); the
        # network object's [sample][pixel][C * F] buffers have exactly this many elements and are used as raw storage
        self.B0p, self.F1p, self.F2p, self.F3p = (t.data_ptr() for t in (net._B0, net.F1, net.F2, net.F3))
        # ---- gradient buffers
        self.dB0, self.dDEC = buf(H * W, 32 * BF), buf(B * H * W, 64)
        self.dF1, self.dF2, self.dF3 = buf(H * W // 4, 64 * BF), buf(H * W // 16, 128 * BF), buf(H * W // 64, 256 * BF)
        big = BF * (H // 2) * (W // 2) * 64                      # the largest encoder activation (floats)
        self.dA, self.dB, self.DP, self.DP2 = buf(big), buf(big), buf(big), buf(big)
        self._Z = None                                           # zero-stuffed dY of the stride-2 layers (largest: enc1.0): only the fallback path
        self.TMP = buf(max(B * H * W * 128, BF * H * W * 32))   # a decoder-sized scratch / the enc1.0 data gradient
        self.dWORK = [buf(B * H * W * 64) for _ in range(2)]     # d work[0] and d (block input) ping-pong
        self.dCAT = buf(B * H * W * 128)
        self.dTMPc = buf(B * (H // 2) * (W // 2) * 64)           # gradient of the 1x1-projected coarse map
        self.dCO = [buf(B * (H // 2) * (W // 2) * 128) for _ in range(2)]   # d coarse of dec3 / dec2
        self.WF = buf(3 * 3 * 512 * 256) if precision == "f32" else None      # flipped weights of the layer being differentiated
        ws = max(int(self.lib.himo_conv_wgrad_workspace_bytes(H // 4, W // 4, 512, 256)),
                 max(int(self.lib.himo_conv_wgrad_batch_workspace_bytes(n, h_, w_, ci, co, st)) for n, h_, w_, ci, co, st in
                     [(BF, H // 2, W // 2, 64, 64, 1), (BF, H // 4, W // 4, 128, 128, 1), (BF, H // 8, W // 8, 256, 256, 1),
                      (B, H // 4, W // 4, 512, 256, 1), (B, H // 4, W // 4, 256, 256, 1), (B, H // 2, W // 2, 256, 128, 1),
                      (B, H // 2, W // 2, 128, 128, 1), (B, H, W, 128, 64, 1), (B, H, W, 64, 64, 1),
                      (BF, H, W, 32, 64, 2), (BF, H // 2, W // 2, 64, 128, 2), (BF, H // 4, W // 4, 128, 256, 2)]),
                 int(self.lib.himo_conv_wgrad_workspace_bytes(H, W, 128, 64)),
                 int(self.lib.himo_conv_wgrad_workspace_bytes(H // 2, W // 2, 256, 128)),
                 int(self.lib.himo_wgrad_workspace_bytes_ex(B * H * W, 96, 64)),
                 int(self.lib.himo_wgrad_workspace_bytes_ex(B * H * W // 4, 192, 128)),
                 int(self.lib.himo_wgrad_workspace_bytes_ex(B * H * W // 16, 384, 256)),
                 int(self.lib.himo_wgrad_workspace_bytes_ex(max_points, 192, 256)),
                 int(self.lib.himo_pfn_backward_workspace_bytes()))
        ws = max(ws, BF * int(self.lib.himo_pfn_bn_workspace_bytes()),
                 max(int(self.lib.himo_bn_workspace_bytes(BF * L[6] * L[7], L[2])) for L in self.layers))
        self.ws = torch.empty(ws + 64, dtype=torch.uint8, device=dev)
        # the encoder's weight gradients run on a SIDE stream under the data-gradient chain (backward): own workspace, and the
        # pre-activation gradient they read alternates between two buffers so the chain never waits for them
        self.overlap_wgrad = self._side_streams_built = os.environ.get("HIMO_TRAIN_SIDE_STREAM", "1") != "0"
        # 3x3 convolutions of the forward and data-gradient passes: the fastest of the library's tile variants per layer shape,
        # timed once at the shape's first launch (the variants return identical bits: tests/test_train_gpu.py)
        self.tune_tiles = os.environ.get("HIMO_TRAIN_TUNE_TILES", "1") != "0"
        self._tile_hints = {}
        if self.tune_tiles and precision != "f32":   # the decoder's forward runs through net._conv: the same restricted candidates
            net.autotune = True
            net._tune = lambda d: self._tune_tile(d) if (d.ksize == 3 and d.w_packed) else 0
        self.side, self.side2, self.side3 = side_streams(dev)
        self.ws_side = torch.empty(ws + 64, dtype=torch.uint8, device=dev)
        # ... and the DECODER's weight gradients on a second side stream (the encoder's ping-pong waits on the first: a backlog of
        # decoder work there would stall the chain).  Their gradient operands get a buffer per block instead of the shared scratch
        self.overlap_decoder = self.overlap_wgrad and os.environ.get("HIMO_TRAIN_SIDE_STREAM_DECODER", "1") != "0"
        self._skip_done = []
        self.ws_side2 = torch.empty(ws + 64, dtype=torch.uint8, device=dev) if self.overlap_decoder else None
        for h_ in self.heads:
            h_.wgrad_stream = self.side2 if self.overlap_decoder else None      # the GRU gates' weight gradients go there too
        if self.overlap_decoder:
            self.TMP3 = buf(H * W * 32 * BF)                     # dec3's skip gradient before it is added to the head's (third side stream)
            self.dIN = {"dec3": buf(B * H * W * 64), "dec2": buf(B * H * W // 4 * 128), "dec1": buf(B * H * W // 16 * 256)}
            self.dCATb = {"dec3": self.dCAT, "dec2": buf(B * H * W // 4 * 256), "dec1": buf(B * H * W // 16 * 512)}
            self.dTMPb = {"dec3": self.dTMPc, "dec2": buf(B * H * W // 16 * 128), "dec1": buf(B * H * W // 64 * 256)}
        self.zero_bias = torch.zeros(1024, dtype=torch.float32, device=dev)
        self._bn_bias_zeroed = set()                              # see _zero_bn_bias
        self.wgrad_leave_room = True         # 3x3 weight gradients on a side stream: one block per CU (himo_conv3x3_wgrad_batch flags bit 2)
        self.bn_from_x = True                # batch mode: the forward pass writes no xhat, the backward pass re-forms it (himo_bn_train_bwd_x)
        self._bn_fwd_from_x = True
        self.stuffed_dgrad = True            # stride-2 data gradients read dY as its zero-stuffed image (HIMO_ACT_STUFFED_2X); False: a stuffed copy first
        # BatchNorm in training mode: per-layer batch statistics kept for the backward pass; the pillar net has one set per sweep
        self.bn_mean = torch.zeros((len(self.layers), 256), dtype=torch.float32, device=dev)
        self.bn_invstd = torch.zeros((len(self.layers), 256), dtype=torch.float32, device=dev)
        self.pfn_scale, self.pfn_shift, self.pfn_mean, self.pfn_invstd = (torch.zeros((F, 32), dtype=torch.float32, device=dev) for _ in range(4))
        self._dres = [None] * B                                  # the loss gradients a backward pass reads (kept alive until it has finished)
        self._bn_folded = True                                   # net.p[*.scale / *.shift] match the running statistics
        # split-bf16 copies of the convolution weights (forward) and a scratch for the flipped ones (data gradient)
        self.packed = {}
        if precision != "f32":
            for k in self.names:
                v = self.p[k]
                if k.endswith(".weight") and v.dim() == 4:
                    ks, _, cin, cout = v.shape
                    self.packed[k] = torch.empty(int(self.lib.himo_conv_packed_weight_bytes(ks, cin, cout)), dtype=torch.uint8, device=dev)
            # ... and of every layer's DATA-GRADIENT weights (taps mirrored, channel roles swapped), same byte count
            self.packed_flip = {k: torch.empty_like(v) for k, v in self.packed.items()}
            self._flip_ptrs = {v.data_ptr() for v in self.packed_flip.values()}
            net.packed = self.packed                      # the decoder forward runs through net._conv
            net.packed_format = self.fwd_format
            jobs = []
            for k, buf in self.packed.items():
                ks, _, cin, cout = self.p[k].shape
                jobs.append((self.p[k], ks, cin, cout, self.fwd_format, 0, buf))
                jobs.append((self.p[k], ks, cin, cout, self.bwd3_format, 1, self.packed_flip[k]))
            jobs += self.head.weight_jobs()
            for h_ in self.heads:
                h_.external_pack = True
            table = np.zeros(len(jobs), dtype=np.dtype([("w", "<u8"), ("packed", "<u8"), ("ksize", "<i4"), ("cin", "<i4"), ("cout", "<i4"),
                                                        ("format", "<i4"), ("flip", "<i4"), ("first_block", "<i4")]))   # himo_weight_job
            blocks = 0
            for i, (w, ks, cin, cout, fmt, flip, dst) in enumerate(jobs):
                nb = int(self.lib.himo_weight_job_blocks(ks, cin, cout, flip))
                if nb < 1:
                    raise ValueError(f"weight job {i}: {ks}x{ks} {cin}->{cout}")
                table[i] = (w.data_ptr(), dst.data_ptr(), ks, cin, cout, fmt, flip, blocks)
                blocks += nb
            self._jobs = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
            self._n_jobs, self._job_blocks = len(jobs), blocks
        self._repack()

    def _repack(self):
        """refresh EVERY packed weight copy -- forward, data-gradient (flipped) and the head's -- in one launch: after construction,
        after every optimiser step and after loading parameters.  Code that edits ``self.p`` by hand must call it too."""
        if self.precision == "f32":
            return
        _lib.check(self.lib.himo_weight_prepare_batch(self._jobs.data_ptr(), self._n_jobs, self._job_blocks, _lib.stream_handle()),
                   "weight_prepare_batch")

    # ---- launch helpers (raw device addresses: most operands are channel groups of wider buffers) --------------
    def _conv(self, x, x_bs, x_pitch, w, bias, y, y_bs, y_pitch, n, h, wd, cin, cout, ks, stride=1, packed=None, fmt=0, accumulate=False,
              stuffed=False):
        """``stuffed``: x is the compact [h / 2][wd / 2] map read as its zero-stuffed x2 image (HIMO_ACT_STUFFED_2X)"""
        if packed is not None and self.precision != "f32" and packed in self._flip_ptrs:
            fmt = self.bwd3_format                    # a data-gradient copy (_flip)
        key = (x, x_bs, x_pitch, w, bias, y, y_bs, y_pitch, n, h, wd, cin, cout, ks, stride, packed, fmt, accumulate, stuffed)
        d = self._descs.get(key)                 # cached per call site: see HeadTrainer._gemm
        if d is None:
            d = ConvDesc()
            d.x, d.x_batch_stride, d.x_pitch = x, x_bs, x_pitch
            d.w = w; d.bias = bias
            d.w_packed = packed; d.packed_format = fmt
            d.y, d.y_batch_stride, d.y_pitch = y, y_bs, y_pitch
            d.n, d.h, d.w_in, d.cin, d.cout, d.ksize, d.stride, d.epilogue = n, h, wd, cin, cout, ks, stride, EPI_BIAS
            d.act_layout = (8 if accumulate else 0) | (16 if stuffed else 0)     # HIMO_ACT_ACCUMULATE | HIMO_ACT_STUFFED_2X (two-term bf16 3x3 kernel)
            if self.tune_tiles and ks == 3 and packed is not None and self.precision != "f32":
                tkey = (n, h, wd, cin, cout, stride, fmt, bool(accumulate), bool(stuffed))
                if tkey not in self._tile_hints:
                    self._tile_hints[tkey] = self._tune_tile(d)
                d.tile_hint = self._tile_hints[tkey]
            if len(self._descs) > 4096:
                self._descs.clear()
            self._descs[key] = d
        _lib.check(self.lib.himo_conv2d(ctypes.byref(d), _lib.stream_handle()), "himo_conv2d(train)")

    # library heuristic, then the weights-from-L2 structure pinned to 4 / 2 / 1 rows per wave, to two pixel groups x two channel tiles
    # (5 / 6: 1 / 2 rows per wave) and to four pixel groups x one 32-channel tile (9 / 10): the waves of a block share their weight fragments
    TILE_HINTS = (0, 0x1004, 0x1002, 0x1001, 0x1005, 0x1006, 0x1009, 0x100A)

    def _tune_tile(self, d) -> int:
        """The fastest tile variant of one 3x3 layer shape (csrc/convsp.hip; all return the same bits).  Runs on the layer's own
        operands at its first launch: the convolution overwrites its output, so repeating it is harmless -- except with
        HIMO_ACT_ACCUMULATE (y += result), which is timed on a twin writing a dense scratch image instead."""
        t = ConvDesc.from_buffer_copy(d)
        if t.act_layout & 8:
            ho, wo = ((t.h + 1) // 2, (t.w_in + 1) // 2) if t.stride == 2 else (t.h, t.w_in)
            if t.n * ho * wo * t.cout > self.TMP.numel():
                return 0
            t.act_layout &= ~8
            t.y, t.y_batch_stride, t.y_pitch = self.TMP.data_ptr(), ho * wo * t.cout, t.cout
        stream = _lib.stream_handle()
        best, best_ms = 0, float("inf")
        for _round in range(2):                       # two interleaved rounds (the clock drifts with load), best time of each variant
            for hint in self.TILE_HINTS:
                t.tile_hint = hint
                if self.lib.himo_conv2d(ctypes.byref(t), stream) != 0:
                    continue                          # a variant this shape does not admit
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    self.lib.himo_conv2d(ctypes.byref(t), stream)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                if ms < best_ms:
                    best, best_ms = hint, ms
        return best

    def _wgrad3_batch(self, n, x, x_bs, x_pitch, h, w, cin, dy, dy_bs, dy_pitch, cout, gname, stride=1, ws=None):
        """weight gradient over n images in one launch (LDS-tiled kernel)"""
        flags = self.wgrad_flags | self._beside_flag(ws)
        ws = self.ws if ws is None else ws
        _lib.check(self.lib.himo_conv3x3_wgrad_batch(n, x, x_bs, x_pitch, h, w, cin, dy, dy_bs, dy_pitch, cout, stride,
                                                     self.g[gname].data_ptr(), flags, ws.data_ptr(), ws.numel(),
                                                     _lib.stream_handle()), "conv3x3_wgrad_batch")

    def _beside_flag(self, ws) -> int:
        """flags bit 2 of the 3x3 weight gradients: a launch on a side stream (it was handed a side workspace) takes ONE block per CU
        and leaves the other half of every CU's registers and LDS to the data-gradient chain on the main stream"""
        return 4 if (self.wgrad_leave_room and ws is not None and ws is not self.ws) else 0

    def _wgrad3(self, x, x_pitch, h, w, cin, dy, dy_pitch, cout, stride, gname, acc, ws=None):
        if stride == 1 and not acc:
            return self._wgrad3_batch(1, x, 0, x_pitch, h, w, cin, dy, 0, dy_pitch, cout, gname, ws=ws)
        ws = self.ws if ws is None else ws
        _lib.check(self.lib.himo_conv3x3_wgrad(x, x_pitch, h, w, cin, dy, dy_pitch, cout, stride, self.g[gname].data_ptr(),
                                               1 if acc else 0, ws.data_ptr(), ws.numel(), _lib.stream_handle()), "conv3x3_wgrad")

    def _wgrad3_bias(self, n, x, x_bs, x_pitch, h, w, cin, dy, dy_bs, dy_pitch, cout, wname, bname, ws=None):
        """stride-1 3x3 weight gradient AND the bias gradient (column sums of dY) over n images ([image][pixel][channel] maps): one
        pass over dY in the split-bf16 kernel (mixed precision); the float32 kernels keep the separate column-sum launch"""
        w_ = self.ws if ws is None else ws
        if self.wgrad_flags & 2:
            st = self.lib.himo_conv3x3_wgrad_batch_bias(n, x, x_bs, x_pitch, h, w, cin, dy, dy_bs, dy_pitch, cout, 1, self.g[wname].data_ptr(),
                                                        self.g[bname].data_ptr(), self.wgrad_flags | self._beside_flag(ws), w_.data_ptr(), w_.numel(),
                                                        _lib.stream_handle())
            if st == 0:
                return
        if n == 1:
            self._wgrad3(x, x_pitch, h, w, cin, dy, dy_pitch, cout, 1, wname, False, ws=ws)
        else:
            self._wgrad3_batch(n, x, x_bs, x_pitch, h, w, cin, dy, dy_bs, dy_pitch, cout, wname, ws=ws)
        if n > 1 and (dy_bs != h * w * dy_pitch):
            raise ValueError("bias gradient over a batch needs consecutive images")
        self._colsum(n * h * w, dy, dy_pitch, cout, bname, ws=ws)

    def set_side_streams(self, on: bool):
        """switch the weight-gradient overlap (both side streams) off / back on at run time: the same kernels in the same order per
        stream either way, so the same bits (with ``wgrad_leave_room`` off: on, the side-stream launches split their pixel chunks over
        half as many blocks, another summation order) -- bench.py times the step's dominant kernel with it off (a launch's HIP-event time
        otherwise includes whatever the other streams co-run)"""
        self.overlap_wgrad = bool(on) and self._side_streams_built
        self.overlap_decoder = self.overlap_wgrad and self.ws_side2 is not None
        for h_ in self.heads:
            h_.wgrad_stream = self.side2 if self.overlap_decoder else None

    def _beside(self, fn):
        """a decoder weight gradient: it only READS its operands, so it runs on the second side stream from the point the main stream
        has reached (fn(workspace)); without the overlap it runs in place"""
        if not self.overlap_decoder:
            return fn(self.ws)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side2):
            self.side2.wait_event(ready)
            fn(self.ws_side2)

    def _zero_bn_bias(self, name):
        """batch mode: the bias gradient of a convolution in front of a training-mode BatchNorm is exactly zero.  Its entries of flat_g
        are zeroed once and stay zero until something writes them -- ``_colsum`` does (a backward after a frozen-statistics forward)
        and forgets the layer here -- instead of one fill launch per layer and step (16 of a step's ~340 launches, each with its
        launch gap on the data-gradient chain)."""
        if name not in self._bn_bias_zeroed:
            self.g[f"{name}.bias"].zero_()
            self._bn_bias_zeroed.add(name)

    def _colsum(self, rows, z, pitch, cout, gname, acc=False, ws=None):
        self._bn_bias_zeroed.discard(gname.rsplit(".", 1)[0])
        ws = self.ws if ws is None else ws
        _lib.check(self.lib.himo_colsum(rows, z, pitch, cout, self.g[gname].data_ptr(), 1 if acc else 0, ws.data_ptr(),
                                        ws.numel(), _lib.stream_handle()), "colsum")

    def _wgrad1(self, rows, x, x_pitch, cin, dz, z_pitch, cout, name, ws=None, acc=False):
        ws = self.ws if ws is None else ws
        _lib.check(self.lib.himo_linear_wgrad_ex(rows, x, x_pitch, cin, dz, z_pitch, cout, self.g[f"{name}.weight"].data_ptr(),
                                                 self.g[f"{name}.bias"].data_ptr(), self.wgrad_flags | (1 if acc else 0), ws.data_ptr(), ws.numel(),
                                                 _lib.stream_handle()), "linear_wgrad")

    def _flip(self, name, ks, cin, cout):
        """flipped / transposed weights of one layer for its data gradient: (float32 address, split-bf16 address or None)"""
        if self.precision != "f32":      # prepared by _repack; the product reads only the packed copy, the float32 address must just be valid
            return self.p[f"{name}.weight"].data_ptr(), self.packed_flip[f"{name}.weight"].data_ptr()
        _lib.check(self.lib.himo_weight_flip(self.p[f"{name}.weight"].data_ptr(), ks, cin, cout, self.WF.data_ptr(), _lib.stream_handle()), "flip")
        return self.WF.data_ptr(), None

    def _add2d(self, rows, cols, b, b_pitch, y, y_pitch):
        _lib.check(self.lib.himo_add2d(rows, cols, b, b_pitch, y, y_pitch, _lib.stream_handle()), "add2d")

    # ---- forward -------------------------------------------------------------------------------------------------
    def fold_batchnorm(self):
        """eval-mode constants (net.p[*.scale / *.shift]) from the CURRENT gamma / beta / running statistics"""
        net = self.net
        for prefix, eps, out in self.bn_layers:
            ch = net.p[f"{prefix}.gamma"].numel()
            _lib.check(self.lib.himo_bn_fold(ch, net.p[f"{prefix}.gamma"].data_ptr(), net.p[f"{prefix}.beta"].data_ptr(),
                                             net.p[f"{prefix}.mean"].data_ptr(), net.p[f"{prefix}.var"].data_ptr(), eps,
                                             net.p[f"{out}.scale"].data_ptr(), net.p[f"{out}.shift"].data_ptr(), _lib.stream_handle()), "bn_fold")
        self._bn_folded = True

    def forward(self, pch1, pc0, pc1, pose_h1, pose0, pose1, training: bool = True, after_pillarize=None) -> torch.Tensor:
        """One sample (``forward_batch`` with a batch of one): res [n0, 4]."""
        return self.forward_batch([(pch1, pc0, pc1, pose_h1, pose0, pose1)], training=training, after_pillarize=after_pillarize)[0]

    def _pillar_stage(self, jobs, constants=None):
        """The pillar stage for the sweeps of ``jobs`` = [(sample b, F sweeps, F transforms)]: image b * F + slot of the pixel-major
        pillar-image buffer, up to 12 sweeps per launch group.  ``constants`` = (scale, shift) [n sweeps][32]: the feature kernel
        ALONE with per-sweep BatchNorm constants (the cell lists are those of the pass that ran before)."""
        net, lib, F = self.net, self.lib, self.net.F
        from .model import HimoSweep, _f32x
        pitch = 32 * F * self.B
        flat = [(b, slot, pts, T) for b, sweeps, transforms in jobs for slot, (pts, T) in enumerate(zip(sweeps, transforms))]
        flags = (1 if net.split_acts else 0) | (2 if net.incremental_images else 0)
        for lo in range(0, len(flat), net.MAX_SWEEPS):
            grp = flat[lo:lo + net.MAX_SWEEPS]
            arr = (HimoSweep * len(grp))()
            for j, (b, slot, pts, T) in enumerate(grp):
                st, w = net._pt[b], arr[j]
                w.n, w.d_pts, w.pc_stride = pts.shape[0], pts.data_ptr(), pts.shape[1]
                w.transform = _f32x(np.asarray(T, dtype=np.float32).reshape(-1))
                w.d_xyz_t, w.d_pid, w.d_offsets = st["xyz_t"][slot].data_ptr(), st["pid"][slot].data_ptr(), st["offsets"][slot].data_ptr()
                w.d_image = self.B0p + 4 * 32 * (b * F + slot)
                w.d_workspace = st["ws_slots"][slot].data_ptr()
            ws_bytes = net._pt[0]["ws_slots"][0].numel()
            if constants is None:
                _lib.check(lib.himo_pillarize_multi_ex(len(arr), ctypes.addressof(arr), net._range, net._voxel, net._centre, net.W, net.H,
                                                       self.p["pfn.weight"].data_ptr(), net.p["pfn.scale"].data_ptr(), net.p["pfn.shift"].data_ptr(),
                                                       pitch, ws_bytes, flags, _lib.stream_handle()), "himo_pillarize_multi_ex")
            else:
                _lib.check(lib.himo_pillar_features_multi(len(arr), ctypes.addressof(arr), net._range, net._voxel, net._centre, net.W, net.H,
                                                          self.p["pfn.weight"].data_ptr(), constants[0].data_ptr() + 4 * 32 * lo,
                                                          constants[1].data_ptr() + 4 * 32 * lo, pitch, ws_bytes, flags, _lib.stream_handle()),
                           "himo_pillar_features_multi")

    def _per_sweep(self, t: torch.Tensor) -> torch.Tensor:
        """[F][32] constants of the frame slots -> one row per sweep of the batch (sweep b * F + f takes row f)"""
        return t if self.nb == 1 else t.repeat(self.nb, 1)

    def forward_batch(self, samples, training: bool = True, after_pillarize=None) -> list:
        """``samples``: 1 .. ``batch`` tuples (pch1, pc0, pc1, pose_h1, pose0, pose1) -> [res [n0_b, 4]] per sample (columns 0..2 =
        network flow of the pc0 rows, column 3 = 0).  ``training`` (only meaningful with batchnorm="batch"): True normalises with
        the statistics of THIS batch (all its sweeps per frame slot in the pillar net, all its images in the encoder), saves them
        for the backward pass and moves the running statistics; False (validation) uses the running statistics."""
        net, lib, s = self.net, self.lib, _lib.stream_handle
        dev = self.device
        nb = len(samples)
        if not 1 <= nb <= self.B:
            raise ValueError(f"forward_batch takes 1..{self.B} samples (SeFlowTrainer(batch=...))")
        self.nb = nb
        batch = self.bn_batch and training
        self._fwd_batch = batch
        self._bn_fwd_from_x = self.bn_from_x                    # what THIS forward pass leaves in PRE (read by backward)
        if self.bn_batch and not training and not self._bn_folded:
            self.fold_batchnorm()
        to_dev = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(dev, torch.float32).contiguous()
        jobs, self.n_pts_b = [], []
        for b_, (pch1, pc0, pc1, pose_h1, pose0, pose1) in enumerate(samples):
            pch1, pc0, pc1 = to_dev(pch1), to_dev(pc0), to_dev(pc1)
            inv1 = np.linalg.inv(np.asarray(pose1, np.float64))
            jobs.append((b_, (pch1, pc0, pc1), (inv1 @ np.asarray(pose_h1, np.float64), inv1 @ np.asarray(pose0, np.float64), np.eye(4))))
            self.n_pts_b.append([pch1.shape[0], pc0.shape[0], pc1.shape[0]])
            for t in (pch1, pc0, pc1):
                net._reserve_points(t.shape[0])
        self.n_pts = self.n_pts_b[0]
        self._fwd_jobs = jobs                                    # (keeps the converted sweeps alive until the next pass)
        self._pillar_stage(jobs)
        if after_pillarize is not None:                          # the sweeps are in the common frame (net._pt[b]["xyz_t"]): input-only work may start
            after_pillarize()
        F, B = net.F, self.B
        NI = nb * F                                              # images the encoder's launches run over
        if batch:
            # the pass above left every sweep's cell lists (its images were written with stale constants): batch statistics of each
            # frame slot over the batch's sweeps -- one call of the embedder per slot, in slot order for the running estimate --
            # then the feature kernel again with the slot's constants
            self._bn_folded = False
            g, b = net.p["pfn.bn.gamma"].data_ptr(), net.p["pfn.bn.beta"].data_ptr()
            rm, rv = net.p["pfn.bn.mean"].data_ptr(), net.p["pfn.bn.var"].data_ptr()
            ns, xs, wss = self._pfn_sweep_arrays()
            _lib.check(lib.himo_pfn_bn_stats_groups(NI, F, ns, xs, wss, net._voxel, net._centre, net.W, net.H, self.p["pfn.weight"].data_ptr(),
                                                    g, b, spec.BN_EPS_PFN, BN_MOMENTUM, rm, rv, self.pfn_scale.data_ptr(),
                                                    self.pfn_shift.data_ptr(), self.pfn_mean.data_ptr(), self.pfn_invstd.data_ptr(),
                                                    self.ws.data_ptr(), self.ws.numel(), s()), "pfn_bn_stats_groups")
            self._pillar_stage(jobs, constants=(self._per_sweep(self.pfn_scale), self._per_sweep(self.pfn_shift)))
        # encoder with saved activations
        src, src_bs, src_pitch = self.B0p, 32, 32 * F * B
        cat = {64: self.F1p, 128: self.F2p, 256: self.F3p}
        self.inputs = []
        for li, (name, cin, cout, stride, h, w, ho, wo, last) in enumerate(self.layers):
            self.inputs.append((src, src_bs, src_pitch))
            pre = self.PRE[li]
            pk = self.packed.get(f"{name}.weight")
            self._conv(src, src_bs, src_pitch, self.p[f"{name}.weight"].data_ptr(), self.p[f"{name}.bias"].data_ptr(),
                       pre.data_ptr(), ho * wo * cout, cout, NI, h, w, cin, cout, 3, stride, packed=None if pk is None else pk.data_ptr(),
                       fmt=self.fwd_format)
            if last:
                y_ptr, y_bs, y_pitch = cat[cout], cout, cout * F * B          # images as channel groups of the concat buffer
            else:
                y_ptr, y_bs, y_pitch = self.Y[li].data_ptr(), ho * wo * cout, cout
            if batch:                                            # PRE keeps the convolution's output x: the backward pass re-forms
                pfx = f"{name}.bn"                               # xhat from it and the saved mean / invstd (himo_bn_train_bwd_x)
                _lib.check(lib.himo_bn_train_fwd(NI, ho * wo, cout, pre.data_ptr(), ho * wo * cout, cout, net.p[f"{pfx}.gamma"].data_ptr(),
                                                 net.p[f"{pfx}.beta"].data_ptr(), spec.BN_EPS, BN_MOMENTUM, net.p[f"{pfx}.mean"].data_ptr(),
                                                 net.p[f"{pfx}.var"].data_ptr(), self.bn_mean[li].data_ptr(), self.bn_invstd[li].data_ptr(),
                                                 None if self.bn_from_x else pre.data_ptr(), ho * wo * cout, cout, y_ptr, y_bs, y_pitch,
                                                 self.ws.data_ptr(), self.ws.numel(), s()), "bn_train_fwd")
            else:                                                # PRE keeps the affine pre-activation
                sc, sh = net.p[f"{name}.scale"].data_ptr(), net.p[f"{name}.shift"].data_ptr()
                if last:
                    for i in range(NI):
                        _lib.check(lib.himo_affine_gelu_fwd(ho * wo, cout, pre[i].data_ptr(), cout, sc, sh, pre[i].data_ptr(), cout,
                                                            y_ptr + 4 * cout * i, y_pitch, s()), "affine_gelu_fwd")
                else:
                    _lib.check(lib.himo_affine_gelu_fwd(NI * ho * wo, cout, pre.data_ptr(), cout, sc, sh, pre.data_ptr(), cout,
                                                        y_ptr, cout, s()), "affine_gelu_fwd")
            src, src_bs, src_pitch = y_ptr, y_bs, y_pitch
        self._decoder_fwd()
        # head: gather -> GRU with saved states, sample by sample (120k rows fill the device: nothing to gain from one launch)
        H, W = net.H, net.W
        out = []
        self.n0_b = []
        for b_, (_, (pch1, pc0, pc1), _) in enumerate(jobs):
            st, head = net._pt[b_], self.heads[b_]
            n0 = pc0.shape[0]
            self.n0_b.append(n0)
            img0, img1 = self.B0p + 4 * 32 * (b_ * F + 1), self.B0p + 4 * 32 * (b_ * F + 2)
            dec = net.DEC.data_ptr() + 4 * b_ * H * W * 64
            if self.precision != "f32" and n0 > 0:    # gather + GRU + decoder + row mask as one launch, states saved for backward
                out.append(head.forward_fused(n0, st["pid"][1].data_ptr(), st["offsets"][1].data_ptr(), img0, img1, 32 * F * B, dec, 64,
                                              self.p["head.offset.weight"].data_ptr(), self.p["head.offset.bias"].data_ptr()))
                continue
            hx0 = torch.empty((n0, 192), dtype=torch.float32, device=dev)
            rhx = torch.empty((n0, 192), dtype=torch.float32, device=dev)
            _lib.check(lib.himo_head_gather(n0, st["pid"][1].data_ptr(), st["offsets"][1].data_ptr(), img0, img1, 32 * F * B, dec, 64,
                                            self.p["head.offset.weight"].data_ptr(), self.p["head.offset.bias"].data_ptr(),
                                            hx0.data_ptr(), rhx.data_ptr(), 192, s()), "himo_head_gather")
            res = head.forward(hx0)
            _lib.check(lib.himo_mask_rows(n0, 4, st["pid"][1].data_ptr(), res.data_ptr(), 4, s()), "mask_rows")
            out.append(res)
        self.n0 = self.n0_b[0]
        return out

    def _fwd_conv(self, x, x_bs, x_pitch, name, y, y_bs, y_pitch, h, w, cin, cout, ks):
        """a decoder layer of the forward pass over the batch's ``nb`` images"""
        pk = self.packed.get(f"{name}.weight")
        self._conv(x, x_bs, x_pitch, self.p[f"{name}.weight"].data_ptr(), self.p[f"{name}.bias"].data_ptr(), y, y_bs, y_pitch, self.nb, h, w,
                   cin, cout, ks, 1, packed=None if pk is None else pk.data_ptr(), fmt=self.fwd_format)

    def _decoder_fwd(self):
        """pillar images + the three stage outputs (pixel-major, the batch's images as channel groups) -> DEC [sample][pixel][64];
        every intermediate keeps its own [sample][pixel][channel] buffer of the network object (the backward pass reads them)."""
        net, nb = self.net, self.nb
        H, W, F, B = net.H, net.W, net.F, self.B
        conv = self._fwd_conv

        def block(name, coarse, coarse_bs, coarse_pitch, c_in, ch, cw, tmp, cat, skip, skip_c, lat, out, work):
            Pc, P = ch * cw, 4 * ch * cw
            conv(coarse, coarse_bs, coarse_pitch, f"{name}.u1", tmp.data_ptr(), Pc * lat, lat, 1, Pc, c_in, lat, 1)
            _lib.check(self.lib.himo_upsample2x_batch_ex(nb, tmp.data_ptr(), Pc * lat if nb > 1 else 0, lat, ch, cw, lat, cat.data_ptr(),
                                                         P * 2 * lat if nb > 1 else 0, 2 * lat, 0, _lib.stream_handle()), "himo_upsample2x_batch_ex")
            conv(skip, skip_c, skip_c * B, f"{name}.u3", cat.data_ptr() + 4 * lat, P * 2 * lat, 2 * lat, 1, P, skip_c, lat, 1)
            conv(cat.data_ptr(), P * 2 * lat, 2 * lat, f"{name}.u4", work[0].data_ptr(), P * out, out, 2 * ch, 2 * cw, 2 * lat, out, 3)
            conv(work[0].data_ptr(), P * out, out, f"{name}.u5", work[1].data_ptr(), P * out, out, 2 * ch, 2 * cw, out, out, 3)
            return work[1]
        s_ = block("dec1", self.F3p, 256 * F, 256 * F * B, 256 * F, H // 8, W // 8, net.T1, net.CAT1, self.F2p, 128 * F, 256, 256, net.S)
        t_ = block("dec2", s_.data_ptr(), (H // 4) * (W // 4) * 256, 256, 256, H // 4, W // 4, net.T2, net.CAT2, self.F1p, 64 * F, 128, 128, net.T)
        u_ = block("dec3", t_.data_ptr(), (H // 2) * (W // 2) * 128, 128, 128, H // 2, W // 2, net.T3, net.CAT3, self.B0p, 32 * F, 64, 64, net.U)
        conv(u_.data_ptr(), H * W * 64, 64, "dec4", net.DEC.data_ptr(), H * W * 64, 64, H, W, 64, 64, 3)

    # ---- backward ----------------------------------------------------------------------------------------------
    def _block_bwd(self, name, coarse, coarse_bs, coarse_pitch, c_in, ch, cw, skip, skip_c, lat, out, cat, work0, d_out, d_in,
                   d_coarse, dc_bs, dc_pitch, d_skip, skip_acc):
        """UpsampleSkip block over the batch's ``nb`` images: d_out (gradient of its output, [image][P][out]) -> parameter gradients,
        d_coarse ([ch*cw][c_in] per image: batch stride ``dc_bs``, pitch ``dc_pitch``), d_skip (pixel-major, the batch's images as
        channel groups of ``skip_c`` channels; added when ``skip_acc``).  ``coarse`` / ``skip``: device addresses of the forward
        pass's inputs (``skip`` in the pixel-major layout).  ``d_in`` is scratch for the gradient of work[0]."""
        lib, s = self.lib, _lib.stream_handle
        nb, B = self.nb, self.B
        h2, w2 = 2 * ch, 2 * cw
        P, Pc = h2 * w2, ch * cw
        zb = self.zero_bias.data_ptr()
        if self.overlap_decoder:                                 # this block's own gradient buffers: its weight gradients read them later
            d_in = self.dIN[name].data_ptr()
        dcat = (self.dCATb[name] if self.overlap_decoder else self.dCAT).data_ptr()
        dtmp = (self.dTMPb[name] if self.overlap_decoder else self.dTMPc).data_ptr()
        # u5
        self._beside(lambda ws: self._wgrad3_bias(nb, work0.data_ptr(), P * out, out, h2, w2, out, d_out, P * out, out, out,
                                                  f"{name}.u5.weight", f"{name}.u5.bias", ws=ws))
        wf, wp = self._flip(f"{name}.u5", 3, out, out)
        self._conv(d_out, P * out, out, wf, zb, d_in, P * out, out, nb, h2, w2, out, out, 3, packed=wp)
        # u4
        self._beside(lambda ws: self._wgrad3_bias(nb, cat.data_ptr(), P * 2 * lat, 2 * lat, h2, w2, 2 * lat, d_in, P * out, out, out,
                                                  f"{name}.u4.weight", f"{name}.u4.bias", ws=ws))
        wf, wp = self._flip(f"{name}.u4", 3, 2 * lat, out)
        self._conv(d_in, P * out, out, wf, zb, dcat, P * 2 * lat, 2 * lat, nb, h2, w2, out, 2 * lat, 3, packed=wp)
        # u3 (1x1 on the skip): gradient rows are the right half of dCAT; the skip rows of sample b are its channel groups of the
        # pixel-major map -- one product per sample, the later ones added to the first
        def u3_weight_gradient(ws):
            for b_ in range(nb):
                self._wgrad1(P, skip + 4 * skip_c * b_, skip_c * B, skip_c, dcat + 4 * (b_ * P * 2 * lat + lat), 2 * lat, lat, f"{name}.u3", ws=ws,
                             acc=b_ > 0)
        self._beside(u3_weight_gradient)
        wt, wp = self._flip(f"{name}.u3", 1, skip_c, lat)                   # [lat][skip_c]

        def skip_gradient(tmp):
            if skip_acc and wp is not None:                      # added to what d_skip holds in the product's own epilogue (HIMO_ACT_ACCUMULATE)
                self._conv(dcat + 4 * lat, P * 2 * lat, 2 * lat, wt, zb, d_skip, skip_c, skip_c * B, nb, 1, P, lat, skip_c, 1, packed=wp, accumulate=True)
            elif skip_acc:                                       # (float32 matrix instructions: a scratch map, pixel-major like d_skip, and an add pass)
                self._conv(dcat + 4 * lat, P * 2 * lat, 2 * lat, wt, zb, tmp, skip_c, skip_c * B, nb, 1, P, lat, skip_c, 1, packed=wp)
                self._add2d(P, skip_c * nb, tmp, skip_c * B, d_skip, skip_c * B)
            else:
                self._conv(dcat + 4 * lat, P * 2 * lat, 2 * lat, wt, zb, d_skip, skip_c, skip_c * B, nb, 1, P, lat, skip_c, 1, packed=wp)
        if self.overlap_decoder and self.precision != "f32":     # (float32: _flip's ONE scratch for flipped weights is rewritten by the next layer)
            # the skip connection's gradient is not on the chain (the encoder's backward pass reads it much later): third side
            # stream, own scratch; backward() waits for the three events before the encoder loop touches a skip gradient
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side3):
                self.side3.wait_event(ready)
                skip_gradient(self.TMP3.data_ptr())
                ev = torch.cuda.Event()
                ev.record(self.side3)
            self._skip_done.append(ev)
        else:
            skip_gradient(self.TMP.data_ptr())
        # upsample, u1 (1x1 on the coarse map)
        for b_ in range(nb):
            _lib.check(lib.himo_upsample2x_bwd(dcat + 4 * b_ * P * 2 * lat, 2 * lat, ch, cw, lat, dtmp + 4 * b_ * Pc * lat, lat, s()), "upsample2x_bwd")
        if coarse_bs == Pc * c_in and coarse_pitch == c_in:      # the coarse maps of the batch are consecutive rows: one product
            self._beside(lambda ws: self._wgrad1(nb * Pc, coarse, c_in, c_in, dtmp, lat, lat, f"{name}.u1", ws=ws))
        else:                                                    # (dec1: the coarse map is the pixel-major stage output)
            def u1_weight_gradient(ws):
                for b_ in range(nb):
                    self._wgrad1(Pc, coarse + 4 * coarse_bs * b_, coarse_pitch, c_in, dtmp + 4 * b_ * Pc * lat, lat, lat, f"{name}.u1", ws=ws, acc=b_ > 0)
            self._beside(u1_weight_gradient)
        wt, wp = self._flip(f"{name}.u1", 1, c_in, lat)
        self._conv(dtmp, Pc * lat, lat, wt, zb, d_coarse, dc_bs, dc_pitch, nb, 1, Pc, lat, c_in, 1, packed=wp)

    def backward(self, dres: torch.Tensor):
        """One sample (``backward_batch`` after a ``forward`` of one sample)."""
        return self.backward_batch([dres])

    def _bucket_done(self, exchange, k: int, streams):
        """bucket k of ``flat_g`` is complete once everything enqueued so far on ``streams`` has run: start its exchange there"""
        if exchange is None:
            return
        events = []
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            events.append(ev)
        exchange.launch(k, events)

    def backward_batch(self, dres_list, exchange: BucketedAllReduce | None = None):
        """``dres_list[b]`` [n0_b, 4] = d loss / d res of sample b of the last ``forward_batch``.  The gradients of every trainable
        tensor -- of the SUM of the samples' losses -- land in ``self.flat_g``.  ``exchange``: the step's data-parallel exchange, started
        bucket by bucket as the pass completes them (head + decoder after the last decoder block, the coarse encoder stages after
        enc2.0, the rest at the end)."""
        net, lib, s = self.net, self.lib, _lib.stream_handle
        H, W, F, B, nb = net.H, net.W, net.F, self.B, self.nb
        if len(dres_list) != nb:
            raise ValueError(f"backward_batch: {len(dres_list)} gradients for the {nb} samples of the last forward pass")
        NI = nb * F
        for b_ in range(nb):
            st, head, n0 = net._pt[b_], self.heads[b_], self.n0_b[b_]
            dres = self._dres[b_] = dres_list[b_].contiguous().clone()
            _lib.check(lib.himo_mask_rows(n0, 4, st["pid"][1].data_ptr(), dres.data_ptr(), 4, s()), "mask_rows")
            head.accumulate = b_ > 0
            dhx0 = head.backward(dres)
            # offset embedding x = offsets @ W + b
            self._beside(lambda ws, n0=n0, st=st, dhx0=dhx0, acc=1 if b_ > 0 else 0: _lib.check(lib.himo_linear_wgrad_ex(
                n0, st["offsets"][1].data_ptr(), 3, 3, dhx0.data_ptr() + 4 * 128, 192, 64, self.g["head.offset.weight"].data_ptr(),
                self.g["head.offset.bias"].data_ptr(), acc, ws.data_ptr(), ws.numel(), s()), "offset wgrad"))
            # gather adjoint: per-point rows -> image gradients (pc0 is slot 1; it gathers from groups 1 and 2 of its sample and from DEC)
            _lib.check(lib.himo_head_scatter(n0, W, H, st["ws_slots"][1].data_ptr(), dhx0.data_ptr(), 192, self.dB0.data_ptr() + 4 * 32 * F * b_,
                                             32 * F * B, 1, 2, F, self.dDEC.data_ptr() + 4 * b_ * H * W * 64, 64, s()), "head_scatter")
        zb = self.zero_bias.data_ptr()
        P, P2, P4, P8 = H * W, (H // 2) * (W // 2), (H // 4) * (W // 4), (H // 8) * (W // 8)
        # dec4
        u = net.U[1]
        self._beside(lambda ws: self._wgrad3_bias(nb, u.data_ptr(), P * 64, 64, H, W, 64, self.dDEC.data_ptr(), P * 64, 64, 64, "dec4.weight", "dec4.bias", ws=ws))
        d_u = self.dWORK[0].data_ptr()
        wf, wp = self._flip("dec4", 3, 64, 64)
        self._conv(self.dDEC.data_ptr(), P * 64, 64, wf, zb, d_u, P * 64, 64, nb, H, W, 64, 64, 3, packed=wp)
        # decoder blocks, last to first
        d_t, d_s = self.dCO[0].data_ptr(), self.dCO[1].data_ptr()
        self._block_bwd("dec3", net.T[1].data_ptr(), P2 * 128, 128, 128, H // 2, W // 2, self.B0p, 32 * F, 64, 64, net.CAT3, net.U[0], d_u,
                        self.dWORK[1].data_ptr(), d_t, P2 * 128, 128, self.dB0.data_ptr(), True)
        self._block_bwd("dec2", net.S[1].data_ptr(), P4 * 256, 256, 256, H // 4, W // 4, self.F1p, 64 * F, 128, 128, net.CAT2, net.T[0], d_t,
                        self.dWORK[1].data_ptr(), d_s, P4 * 256, 256, self.dF1.data_ptr(), False)
        self._block_bwd("dec1", self.F3p, 256 * F, 256 * F * B, 256 * F, H // 8, W // 8, self.F2p, 128 * F, 256, 256, net.CAT1, net.S[0], d_s,
                        self.dWORK[1].data_ptr(), self.dF3.data_ptr(), 256 * F, 256 * F * B, self.dF2.data_ptr(), False)
        # encoder, last layer to first
        dcat = {256: self.dF3, 128: self.dF2, 64: self.dF1, 32: self.dB0}
        dy = None
        main = torch.cuda.current_stream(self.device)
        # head + decoder: their weight gradients are on the second side stream (or on this one), everything enqueued
        self._bucket_done(exchange, 0, (main, self.side2))
        while self._skip_done:                                   # the decoder blocks' skip gradients (third side stream) are complete
            main.wait_event(self._skip_done.pop())
        side_done = {}                                           # layer -> event: its side-stream launches have finished reading dp
        for li in range(len(self.layers) - 1, -1, -1):
            name, cin, cout, stride, h, w, ho, wo, last = self.layers[li]
            pre = self.PRE[li]
            sc = net.p[f"{name}.scale"].data_ptr()
            dp = (self.DP2 if (self.overlap_wgrad and li & 1) else self.DP).data_ptr()
            if li + 2 in side_done:                              # this buffer's previous reader (two layers up) must be through
                main.wait_event(side_done.pop(li + 2))
            if self._fwd_batch:                                  # through GELU and the batch statistics; also d gamma / d beta
                pfx = f"{name}.bn"
                dy_ptr, dy_bs, dy_pitch = (dcat[cout].data_ptr(), cout, cout * F * B) if last else (dy, ho * wo * cout, cout)
                if not self._bn_fwd_from_x:                      # PRE holds xhat (a forward pass with bn_from_x off)
                    _lib.check(lib.himo_bn_train_bwd(NI, ho * wo, cout, dy_ptr, dy_bs, dy_pitch, pre.data_ptr(), ho * wo * cout, cout,
                                                     net.p[f"{pfx}.gamma"].data_ptr(), net.p[f"{pfx}.beta"].data_ptr(), self.bn_invstd[li].data_ptr(),
                                                     dp, ho * wo * cout, cout, self.g[f"{pfx}.gamma"].data_ptr(), self.g[f"{pfx}.beta"].data_ptr(),
                                                     0, self.ws.data_ptr(), self.ws.numel(), s()), "bn_train_bwd")
                else:
                    _lib.check(lib.himo_bn_train_bwd_x(NI, ho * wo, cout, dy_ptr, dy_bs, dy_pitch, pre.data_ptr(), ho * wo * cout, cout,
                                                       net.p[f"{pfx}.gamma"].data_ptr(), net.p[f"{pfx}.beta"].data_ptr(), self.bn_mean[li].data_ptr(),
                                                       self.bn_invstd[li].data_ptr(), dp, ho * wo * cout, cout, self.g[f"{pfx}.gamma"].data_ptr(),
                                                       self.g[f"{pfx}.beta"].data_ptr(), 0, self.ws.data_ptr(), self.ws.numel(), s()), "bn_train_bwd_x")
            elif last:                                           # gradient arrives in the concat layout
                src = dcat[cout]
                for i in range(NI):
                    _lib.check(lib.himo_affine_gelu_bwd(ho * wo, cout, src.data_ptr() + 4 * cout * i, cout * F * B, pre[i].data_ptr(), cout, sc,
                                                        dp + 4 * i * ho * wo * cout, cout, s()), "affine_gelu_bwd")
            else:
                _lib.check(lib.himo_affine_gelu_bwd(NI * ho * wo, cout, dy, cout, pre.data_ptr(), cout, sc, dp, cout, s()), "affine_gelu_bwd")
            # a bias in front of a training-mode BatchNorm has exactly zero gradient (the batch mean absorbs it); the column sums
            # of dp would be rounding noise that Adam's normalisation turns into full-size random steps.  Nothing ever writes
            # these entries of flat_g in batch mode -- but a backward() after a frozen-statistics forward does (the branch
            # below), and train_batch ADDS flat_g into its accumulator: so batch mode zeroes them, it does not assume them zero.
            x, x_bs, x_pitch = self.inputs[li]
            if self.overlap_wgrad:
                # the weight (and bias) gradient only READS dp and the saved layer input: it runs beside the data-gradient chain,
                # which is the critical path (same kernels; with `wgrad_leave_room` one block per CU, so that the chain's blocks fit beside)
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ready)
                    if not self._fwd_batch:
                        self._colsum(NI * ho * wo, dp, cout, cout, f"{name}.bias", ws=self.ws_side)
                    self._wgrad3_batch(NI, x, x_bs, x_pitch, h, w, cin, dp, ho * wo * cout, cout, cout, f"{name}.weight", stride, ws=self.ws_side)
                    side_done[li] = torch.cuda.Event()
                    side_done[li].record(self.side)
                if self._fwd_batch:
                    self._zero_bn_bias(name)
            else:
                if not self._fwd_batch:
                    self._colsum(NI * ho * wo, dp, cout, cout, f"{name}.bias")
                else:
                    self._zero_bn_bias(name)
                self._wgrad3_batch(NI, x, x_bs, x_pitch, h, w, cin, dp, ho * wo * cout, cout, cout, f"{name}.weight", stride)
            if name == "enc2.0":                                 # stages 3 and 2: weight gradients on the first side stream, d gamma / d beta on this one
                self._bucket_done(exchange, 1, (main, self.side))
            wf, wp = self._flip(name, 3, cin, cout)
            if stride == 2:
                dst = dcat[cin]                                  # fan-in: add to the decoder's skip gradient
                if self.bwd3_format == 2 and self.stuffed_dgrad and h % 2 == 0 and w % 2 == 0:
                    # ... in the convolution's own epilogue (images = channel groups of dst), and the convolution reads dp as its own
                    # zero-stuffed image: no stuffed copy is written or read, and the all-zero rows cost no matrix instructions
                    self._conv(dp, ho * wo * cout, cout, wf, zb, dst.data_ptr(), cin, cin * F * B, NI, h, w, cout, cin, 3, packed=wp, accumulate=True,
                               stuffed=True)
                    continue
                if self._Z is None:                              # (the fallback: a zero-stuffed copy of dY first)
                    self._Z = torch.empty(B * F * H * W * 64, dtype=torch.float32, device=self.device)
                z = self._Z.data_ptr()
                _lib.check(lib.himo_zero_stuff2x(NI, ho, wo, cout, dp, ho * wo * cout, cout, z, h * w * cout, cout, s()), "zero_stuff")
                if self.bwd3_format == 2:
                    self._conv(z, h * w * cout, cout, wf, zb, dst.data_ptr(), cin, cin * F * B, NI, h, w, cout, cin, 3, packed=wp, accumulate=True)
                else:
                    tmp = self.TMP.data_ptr()
                    self._conv(z, h * w * cout, cout, wf, zb, tmp, h * w * cin, cin, NI, h, w, cout, cin, 3, packed=wp)
                    for i in range(NI):
                        self._add2d(h * w, cin, tmp + 4 * i * h * w * cin, cin, dst.data_ptr() + 4 * cin * i, cin * F * B)
            else:
                nxt = self.dA.data_ptr() if dy != self.dA.data_ptr() else self.dB.data_ptr()
                self._conv(dp, ho * wo * cout, cout, wf, zb, nxt, h * w * cin, cin, NI, h, w, cout, cin, 3, packed=wp)
                dy = nxt
        # pillar feature net
        if self._fwd_batch:                                      # the sweeps' walks share their launches; statistics per frame slot
            ns, xs, wss = self._pfn_sweep_arrays()
            dimg = (ctypes.c_void_p * NI)(*[self.dB0.data_ptr() + 4 * 32 * i for i in range(NI)])
            _lib.check(lib.himo_pfn_backward_bn_groups(NI, F, ns, xs, wss, dimg, 32 * F * B, net._voxel, net._centre, W, H, self.p["pfn.weight"].data_ptr(),
                                                       self.pfn_scale.data_ptr(), self.pfn_shift.data_ptr(), self.pfn_mean.data_ptr(),
                                                       self.pfn_invstd.data_ptr(), self.g["pfn.weight"].data_ptr(),
                                                       self.g["pfn.bn.gamma"].data_ptr(), self.g["pfn.bn.beta"].data_ptr(), 0,
                                                       self.ws.data_ptr(), self.ws.numel(), s()), "pfn_backward_bn_groups")
        else:
            for i in range(NI):
                b_, slot = divmod(i, F)
                st = net._pt[b_]
                _lib.check(lib.himo_pfn_backward(self.n_pts_b[b_][slot], net._voxel, net._centre, W, H, self.p["pfn.weight"].data_ptr(),
                                                 net.p["pfn.scale"].data_ptr(), net.p["pfn.shift"].data_ptr(), st["xyz_t"][slot].data_ptr(),
                                                 st["ws_slots"][slot].data_ptr(), self.dB0.data_ptr() + 4 * 32 * i, 32 * F * B,
                                                 self.g["pfn.weight"].data_ptr(), 1 if i else 0, self.ws.data_ptr(), self.ws.numel(), s()),
                           "pfn_backward")
        self._bucket_done(exchange, 2, (main, self.side))
        main.wait_stream(self.side)                             # every gradient is in flat_g when this stream goes on
        main.wait_stream(self.side2)
        main.wait_stream(self.side3)
        if exchange is not None:
            exchange.wait()                                      # ... and summed over the ranks

    def _pfn_sweep_arrays(self):
        """host arrays of the batch's sweeps (sweep b * F + f) for the pillar-net calls: point counts, transformed points, pillar workspaces"""
        net, F, nb = self.net, self.net.F, self.nb
        n = nb * F
        return ((ctypes.c_int64 * n)(*[self.n_pts_b[b][k] for b in range(nb) for k in range(F)]),
                (ctypes.c_void_p * n)(*[net._pt[b]["xyz_t"][k].data_ptr() for b in range(nb) for k in range(F)]),
                (ctypes.c_void_p * n)(*[net._pt[b]["ws_slots"][k].data_ptr() for b in range(nb) for k in range(F)]))

    # ---- optimiser / data parallel -----------------------------------------------------------------------------
    def sync_running_stats(self, src: int = 0):
        """Every rank adopts rank ``src``'s BatchNorm running statistics (one small broadcast; what DDP's default
        ``broadcast_buffers`` does before each forward, done here once per epoch: batch statistics -- not these -- drive the
        training forward, so only validation and the checkpoints see them)."""
        import torch.distributed as dist
        if not (self.bn_batch and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        stats = [self.net.p[f"{prefix}.{k}"] for prefix, _, _ in self.bn_layers for k in ("mean", "var")]
        flat = torch.cat([t.reshape(-1) for t in stats])
        if flat.is_cuda and dist.get_backend() == "gloo":
            host = flat.cpu()
            dist.broadcast(host, src=src)
            flat = host.to(self.device)
        else:
            dist.broadcast(flat, src=src)
        o = 0
        for t in stats:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()
        self._bn_folded = False

    def allreduce(self):
        """Mean of the flat gradient over ranks: ONE collective per step (RCCL over xGMI)."""
        allreduce_mean_(self.flat_g)

    def _loss_engine(self, b: int):
        """sample slot b's loss engine (each keeps its own neighbour / count buffers: a batch's searches run ahead of its losses)"""
        from ..ssl_loss import SeFlowLoss
        if not hasattr(self, "losses"):
            self.losses = [SeFlowLoss(device=self.device) for _ in range(self.B)]
            self.loss = self.losses[0]
        return self.losses[b]

    def loss_and_grad(self, pch1, pc0, pc1, pose_h1, pose0, pose1, label0, label1, n_labels: int | None = None):
        """One sample (``loss_and_grad_batch`` with a batch of one).  Returns (terms, total)."""
        terms, totals = self.loss_and_grad_batch([(pch1, pc0, pc1, pose_h1, pose0, pose1, label0, label1, n_labels)])
        return terms[0], totals[0]

    def loss_and_grad_batch(self, samples, exchange=None):
        """forward over the batch + self-supervised loss per sample (himo_amd/ssl_loss.py; the flow it scores is the network's residual
        flow of pc0 in pc1's frame) + backward: the gradient of the SUM of the samples' losses lands in ``flat_g``.  ``samples``: 1 ..
        ``batch`` tuples (pch1, pc0, pc1, pose_h1, pose0, pose1, label0, label1, n_labels).  Returns ([terms per sample], [total per
        sample]) as 0-d float64 device tensors."""
        samples = list(samples)
        nb = len(samples)
        engines = [self._loss_engine(b) for b in range(nb)]
        hook = None
        raws, sizes_all = [None] * nb, [None] * nb
        if self.overlap_decoder:
            # the cluster term's correspondences pc0 -> pc1 depend on the sweeps only: searched on the second side stream UNDER the
            # forward pass instead of between forward and backward (a grid build + a 120k-point query: ~0.17 ms of the chain)
            def hook():
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.side2):
                    self.side2.wait_event(ready)
                    for b in range(nb):
                        m0, m1 = self.n_pts_b[b][1], self.n_pts_b[b][2]
                        st = self.net._pt[b]
                        raws[b] = engines[b].raw_neighbours(st["xyz_t"][1][:m0], st["xyz_t"][2][:m1])
                        # ... and the sizes of the two dynamic subsets (labels only): counted here, read from pinned memory at the loss,
                        # which then never blocks -- the host enqueues the backward pass while the forward pass is still running
                        sizes_all[b] = engines[b].dyn_sizes(samples[b][6], samples[b][7])
                    self._raw_done = torch.cuda.Event()
                    self._raw_done.record(self.side2)
        res = self.forward_batch([smp[:6] for smp in samples], after_pillarize=hook)
        if hook is not None:                                     # (the side stream's work, the converted labels of dyn_sizes included)
            torch.cuda.current_stream(self.device).wait_event(self._raw_done)
        terms_all, totals, dres_list = [], [], []
        for b, smp in enumerate(samples):
            label0, label1, n_labels = smp[6], smp[7], smp[8] if len(smp) > 8 else None
            n0, n1 = self.n_pts_b[b][1], self.n_pts_b[b][2]
            st = self.net._pt[b]
            raw = raws[b][:2] if (raws[b] is not None and n0 > 0 and n1 > 0) else None
            sizes = None
            if sizes_all[b] is not None and n_labels is not None:
                sizes, (label0, label1) = sizes_all[b][:2], sizes_all[b][2]
            terms, total, grad = engines[b](st["xyz_t"][1][:n0], st["xyz_t"][2][:n1], res[b][:, :3].contiguous(), label0, label1, n_labels,
                                            raw=raw, sizes=sizes)
            dres = torch.zeros((n0, 4), dtype=torch.float32, device=self.device)
            dres[:, :3] = grad
            terms_all.append(terms); totals.append(total); dres_list.append(dres)
        if exchange is not None and exchange.words is not None:      # [sample count | loss sum] of this rank ride along with the first bucket
            exchange.words[0] = float(nb)
            exchange.words[1] = torch.stack(totals).sum().to(exchange.words.dtype)
        self.backward_batch(dres_list, exchange=exchange)
        return terms_all, totals

    def loss_only(self, pch1, pc0, pc1, pose_h1, pose0, pose1, label0, label1, n_labels: int | None = None):
        """forward + loss without a backward pass (validation): the total as a 0-d float64 device tensor"""
        res = self.forward(pch1, pc0, pc1, pose_h1, pose0, pose1, training=False)
        n0, n1 = self.n_pts[1], self.n_pts[2]
        _, total, _ = self._loss_engine(0)(self.net.xyz_t[1][:n0], self.net.xyz_t[2][:n1], res[:, :3].contiguous(), label0, label1, n_labels)
        return total

    def train_step(self, pch1, pc0, pc1, pose_h1, pose0, pose1, label0, label1, n_labels: int | None = None, lr: float = 6e-5):
        """One optimisation step on one sample per rank: forward, loss, backward, gradient all-reduce, Adam
        (``lr`` default = the reference launcher's ``optimizer.lr=6e-5``, assets/slurm/ssl-train-av2.sh:33).
        Returns ({term: 0-d float64 device tensor}, total)."""
        terms, total = self.loss_and_grad(pch1, pc0, pc1, pose_h1, pose0, pose1, label0, label1, n_labels)
        self.allreduce()
        self.adam_step(lr)
        return terms, total

    def train_batch(self, samples, lr: float = 6e-5, bucketed: bool = False, global_count: int | None = None):
        """One optimisation step on SEVERAL samples per rank (the launcher's ``batch_size=8``): ``samples`` = iterable of (pch1, pc0,
        pc1, pose_h1, pose0, pose1, label0, label1, n_labels).  They go through the network ``batch`` at a time (the constructor's
        capacity: ONE forward / backward pass over up to that many samples, BatchNorm statistics over the pass's samples -- torch's
        semantics for a per-process batch; with ``batch=1`` sample by sample, statistics per sample).  Every sample of the GLOBAL batch
        weighs the same whatever the split over the ranks and passes: each rank sums its passes' gradients (and its sample count and
        loss), the ranks' sums are added, and the sum is divided by the global count once.  A rank may hold no sample of a partial
        last batch -- it still enters the collectives, with zeros.  Returns the mean loss over the global batch (the same number on
        every rank).

        ``bucketed``: HOW the ranks' sums are added -- and EVERY rank must pass the same value, since the two forms are different
        sequences of collectives.  False: ONE flat all-reduce after the last pass.  True: bucket by bucket UNDER the backward pass
        (``BucketedAllReduce``) -- possible when every rank's share of the step is at most one pass (<= ``batch`` samples; 0 is fine:
        that rank sends zeros through the same collectives); the caller decides from what all ranks know (``fit``: the global batch
        size and the world size).  Same sums, element by element.

        ``global_count``: the number of samples of the step over ALL ranks when the caller knows it (ignored without a process group:
        it is ``len(samples)``).  Without it the count is read back from the device after the exchange -- a host wait for the whole
        step (``global_count_known``)."""
        n = self.flat_g.numel()
        if not hasattr(self, "flat_acc") or self.flat_acc.numel() != n + 2:
            self.flat_acc = torch.zeros(n + 2, dtype=self.flat_g.dtype, device=self.flat_g.device)   # [gradient sum | count | loss sum]
        acc = self.flat_acc[:n]
        samples = list(samples)
        known = global_count_known(len(samples), global_count)
        it, passes = iter(samples), 0
        import torch.distributed as dist
        if bucketed and self.overlap_allreduce and dist.is_available() and dist.is_initialized():
            first = list(itertools.islice(it, self.B + 1))
            if len(first) > self.B:
                raise ValueError(f"train_batch(bucketed=True) takes at most batch = {self.B} samples per rank and step")
            words = self.flat_acc[n:]
            exchange = BucketedAllReduce(self.flat_g, self.buckets, words=words)
            if first:
                self.loss_and_grad_batch(first, exchange=exchange)
            else:                                                # no sample on this rank: zeros through the same collectives, in the same order
                self.flat_g.zero_()
                words.zero_()
                for k in range(len(self.buckets)):
                    exchange.launch(k)
                exchange.wait()
            count = words[0]
            if not known and float(count.item()) <= 0.0:
                raise ValueError("train_batch needs at least one sample on some rank")
            self.flat_g.div_(count)
            loss = (words[1] / count).clone()
            self.adam_step(lr)
            return loss
        while True:
            chunk = list(itertools.islice(it, self.B))
            if not chunk:
                break
            _, totals = self.loss_and_grad_batch(chunk)
            if passes == 0:
                self.flat_acc[n:].zero_()
                acc.copy_(self.flat_g)
            else:
                acc.add_(self.flat_g)
            passes += 1
            self.flat_acc[n] += float(len(chunk))
            self.flat_acc[n + 1] += torch.stack(totals).sum().to(self.flat_acc.dtype)
        if passes == 0:
            self.flat_acc.zero_()
        loss = combine_batch_(self.flat_acc, self.flat_g, count_known=known)
        self.adam_step(lr)
        return loss

    @staticmethod
    def step_lr(epoch: int, base_lr: float = 6e-5, step_size: int = 3, gamma: float = 0.5) -> float:
        """The launcher's ``StepLR(3, 0.5)`` schedule (assets/slurm/ssl-train-av2.sh:33): lr of a 0-based epoch."""
        return base_lr * gamma ** (epoch // step_size)

    def adam_step(self, lr: float = 2e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        self.step_count += 1
        _lib.check(self.lib.himo_adam_step(self.flat_p.numel(), self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                           self.flat_v.data_ptr(), lr, beta1, beta2, eps, self.step_count, _lib.stream_handle()), "adam")
        self._repack()

    # ---- checkpoints (seflow/checkpoint.py): parameters + Adam moments + step counter ---------------------------------
    def save_checkpoint(self, path, **extra):
        """Parameters in the spec layout + the flat Adam moments and the step counter (``extra``: epoch=, val=)."""
        from .checkpoint import save_params
        return save_params(path, self.export_params(), adam_m=self.flat_m.cpu().numpy(), adam_v=self.flat_v.cpu().numpy(),
                           step=self.step_count, **extra)

    def load_checkpoint(self, path) -> dict:
        """Restore parameters (and, when the file carries them, optimiser state) written by ``save_checkpoint`` -- training
        continues bit for bit where the saved run stood.  Returns the file's extra fields (epoch, val, ...)."""
        from .checkpoint import load_params
        params, extra = load_params(path, with_extra=True)
        self.import_params(params)
        if "adam_m" in extra and "adam_v" in extra:
            if extra["adam_m"].shape != tuple(self.flat_m.shape):
                raise ValueError("checkpoint optimiser state does not match this network's parameter count")
            self.flat_m.copy_(torch.from_numpy(extra["adam_m"]))
            self.flat_v.copy_(torch.from_numpy(extra["adam_v"]))
            self.step_count = int(extra.get("step", 0))
        return extra

    def import_params(self, params: dict):
        """spec-layout parameter dict -> the flat buffer, plus the (frozen) BatchNorm tensors and the scale / shift constants
        folded from them exactly as SeFlowNet folds them: the whole network becomes the checkpoint's."""
        host = {k: params[k] for k in self.names if k in params}
        host.update({f"head.{k}": v for k, v in HeadTrainer.host_params(params).items()})
        for k in self.names:
            self.p[k].copy_(torch.from_numpy(np.ascontiguousarray(host[k], dtype=np.float32)))
        t = lambda k: torch.from_numpy(np.ascontiguousarray(params[k], dtype=np.float32))
        for prefix, eps, out in [("pfn.bn", spec.BN_EPS_PFN, "pfn")] + [(f"{n}.bn", spec.BN_EPS, n) for n, *_ in spec.ENCODER]:
            scale = t(f"{prefix}.gamma") / torch.sqrt(t(f"{prefix}.var") + eps)
            shift = t(f"{prefix}.beta") - t(f"{prefix}.mean") * scale
            self.net.p[f"{out}.scale"].copy_(scale)
            self.net.p[f"{out}.shift"].copy_(shift)
            for k in ("gamma", "beta", "mean", "var"):
                self.net.p[f"{prefix}.{k}"].copy_(t(f"{prefix}.{k}"))
        self._bn_folded = True
        self._repack()

    def export_params(self) -> dict:
        """spec-layout numpy parameter dict (BatchNorm constants unchanged) for SeFlowNet / the oracle."""
        out = {k: v.cpu().numpy().copy() for k, v in self.net.p.items() if k in spec.param_shapes()}
        for k in self.names:
            if k in out:
                out[k] = self.p[k].cpu().numpy().copy()
        zr_w, zr_b = self.p["head.zr.weight"].cpu().numpy(), self.p["head.zr.bias"].cpu().numpy()
        out["head.gru.z.weight"], out["head.gru.r.weight"] = zr_w[:, :128].copy(), zr_w[:, 128:].copy()
        out["head.gru.z.bias"], out["head.gru.r.bias"] = zr_b[:128].copy(), zr_b[128:].copy()
        out["head.gru.q.weight"], out["head.gru.q.bias"] = self.p["head.q.weight"].cpu().numpy().copy(), self.p["head.q.bias"].cpu().numpy().copy()
        out["head.dec1.weight"], out["head.dec1.bias"] = self.p["head.dec1.weight"].cpu().numpy().copy(), self.p["head.dec1.bias"].cpu().numpy().copy()
        out["head.dec2.weight"], out["head.dec2.bias"] = self.p["head.dec2.weight"].cpu().numpy()[:, :3].copy(), self.p["head.dec2.bias"].cpu().numpy()[:3].copy()
        return out
