"""Stage a12: optimisation-based scene flow ("fastnsf"): a per-scene coordinate MLP fitted at run time.

PARITY UNPINNED.  The reference only names the method (``python save.py model=fastnsf``, README.md:53; result keys
``fastnsf10`` / ``nsfp`` at tools/view_instance.py:155); its implementation is in the absent OpenSceneFlow submodule.
This build's own specification, after the published Neural Scene Flow Prior family:

  * flow field  f_theta: R^3 -> R^3, an MLP 3 -> 128 (x8 hidden layers, ReLU) -> 3, weights U(-1/sqrt(fan_in), ..);
  * pc0 is first brought into pc1's frame (p' = R p + t with inv(pose1) @ pose0, float32 -- as seflow/spec.py step 0);
  * objective   ``objective="dt"`` (default; what makes FastNSF fast): ONCE per pair pc1 becomes a distance-transform volume --
                cells of 0.1 m over the network range grown by tau, D[c] = distance from cell c to the nearest cell holding a
                pc1 point, exact up to tau (csrc/dtloss.hip) -- and L = mean_i [D(m_i) <= tau] D(m_i) with m_i = p'_i + f(p'_i)
                and D(.) the trilinear interpolation of the volume: one lookup kernel per iteration, no search.
                ``objective="nn"`` (the NSFP objective, kept for the parity tests of round 1-2):
                L = mean_i [d_i <= tau^2] d_i + mean_j [e_j <= tau^2] e_j, with d_i the squared distance from
                m_i to its nearest pc1 point, e_j the squared distance from pc1_j to its nearest moved
                point; correspondences are exact (csrc/nngrid.hip) and constant within an iteration; tau = 2 m in both;
  * optimiser   Adam(lr 1e-3, betas (0.9, 0.999), eps 1e-8), ``iters`` steps, optional early stop on the loss;
  * output      (N,3) float32 flow INCLUDING ego motion, row-aligned with pc0 -- the h5 ``<res_name>`` payload.

Everything that touches point data runs in HIP.  The default configuration (mixed precision, "dt" objective) runs an optimiser
iteration as THREE launches (csrc/nsffused.hip): forward + objective, backward + every weight gradient, reduce + Adam + re-pack.
The other configurations ("nn" objective, float32 products, ``fused=False``) keep the layer-by-layer kernels of rounds 1-3: the
forward / input-gradient products on the matrix cores (csrc/conv.hip row GEMM with BIAS_RELU / RELU_MASK epilogues, or
csrc/mlpfused.hip), the weight gradients as a split-K MFMA product (csrc/fastnsf.hip), the objective, Adam.
Python sequences launches and owns no arithmetic.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import _lib
from .seflow.model import ConvDesc
from .ssl_loss import GRID_CELL, GRID_H, GRID_W, GRID_X0, GRID_Y0

EPI_BIAS, EPI_BIAS_RELU, EPI_RELU_MASK = 0, 5, 6
HIDDEN, N_HIDDEN, TRUNC = 128, 8, 2.0

_lib.register({
    "himo_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "himo_linear_wgrad": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_transpose": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "himo_wgrad_workspace_bytes_ex": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "himo_linear_wgrad_ex": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_adam_step": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]),
    "himo_chamfer_trunc_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "himo_chamfer_trunc": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_rows_add": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "himo_rigid_transform": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p]),
    "himo_mlp_repack": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "himo_mlp_forward_fused": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "himo_mlp_backward_fused": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p]),
    "himo_mlp_bias_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "himo_mlp_backward_fused_bias": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_nsf_padded_rows": (ctypes.c_int64, [ctypes.c_int64]),
    "himo_nsf_spill_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "himo_nsf_backward_blocks": (ctypes.c_int, [ctypes.c_int64]),
    "himo_nsf_forward": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "himo_nsf_backward": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "himo_nsf_update": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "himo_dt_volume_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "himo_dt_build": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_dt_loss_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "himo_dt_loss": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                    ctypes.c_void_p]),
})

DT_CELL = 0.1                       # metres per cell of the distance-transform volume


def dt_grid(trunc: float = TRUNC, cell: float = DT_CELL, box=None):
    """(origin float32[3], dims int32[3], window) of the distance-transform volume: ``box`` = (xmin, ymin, zmin, xmax, ymax, zmax),
    default the network range (seflow/spec.py POINT_CLOUD_RANGE) grown by ``trunc`` on every side; the window -- the distance in
    cells up to which the transform is exact -- covers ``trunc`` plus one cell of interpolation margin."""
    from .seflow import spec
    if box is None:
        r = spec.POINT_CLOUD_RANGE
        box = (r[0] - trunc, r[1] - trunc, r[2] - trunc, r[3] + trunc, r[4] + trunc, r[5] + trunc)
    origin = np.asarray(box[:3], np.float32)
    dims = np.asarray([max(1, int(math.ceil((box[3 + k] - box[k]) / cell - 1e-6))) for k in range(3)], np.int32)
    return origin, dims, int(math.ceil(trunc / cell)) + 1



def init_mlp(seed: int = 0) -> list:
    """[(W [in,out], b [out])] float32 numpy, torch.nn.Linear's default init; in/out of the first/last layer are 3."""
    rng = np.random.default_rng(seed)
    dims = [3] + [HIDDEN] * N_HIDDEN + [3]
    out = []
    for cin, cout in zip(dims[:-1], dims[1:]):
        bound = 1.0 / math.sqrt(cin)
        out.append((rng.uniform(-bound, bound, (cin, cout)).astype(np.float32), rng.uniform(-bound, bound, cout).astype(np.float32)))
    return out


class FastNSF:
    def __init__(self, device=None, lr: float = 1e-3, iters: int = 100, seed: int = 0, trunc: float = TRUNC,
                 early_patience: int = 0, early_min_delta: float = 1e-4, objective: str = "dt", dt_cell: float = DT_CELL, dt_box=None,
                 precision: str = "mixed", fused: bool = True, three_launch: bool = True):
        """``fused`` (mixed precision only): the whole forward pass and the whole chain of input gradients as ONE kernel each
        instead of a row GEMM per layer and direction; with ``three_launch`` (default; "dt" objective) also the objective, every
        weight gradient and the optimiser step: three launches per iteration (csrc/nsffused.hip) -- ``three_launch=False`` keeps
        round 3's kernels (csrc/mlpfused.hip + a split-K weight-gradient product per layer), which the tests compare with.
        ``precision``: "mixed" (default) runs the MLP's products on the 16-bit matrix instructions with split operands -- forward
        fp16 split (x = h + l: 22-bit products; activations are O(1) and coordinates < 64 m), input gradients and weight gradients
        two-term bf16 (16 significant bits at float32's range: gradients sit far below fp16's subnormal floor), float32 sums
        throughout -- as the training step does (seflow/train.py); "f32": float32 matrix instructions everywhere."""
        if objective not in ("dt", "nn"):
            raise ValueError(objective)
        if precision not in ("mixed", "f32"):
            raise ValueError(precision)
        self.mixed = precision == "mixed"
        self.fused = bool(fused) and self.mixed
        self.three_launch = bool(three_launch)
        self.lib = _lib.load()
        self.device = device if device is not None else _lib.require_gpu()
        self.lr, self.iters, self.seed, self.trunc = lr, iters, seed, trunc
        self.objective, self.dt_cell, self.dt_box = objective, dt_cell, dt_box
        self._dt_vol = None                                   # the volume buffer is kept between fits (0.45 GB at the default box)
        self.early_patience, self.early_min_delta = early_patience, early_min_delta
        self.loss_history = []
        self._descs = {}
        self._defer_finish, self._finish = False, None

    # ---- parameters: stored padded to multiples of 4 channels (3 -> 4) -----------------------------------------------
    def _load(self, layers):
        """Parameters, gradients and Adam moments live in ONE flat buffer each (views per layer): the optimiser step of
        an iteration is a single launch instead of one per tensor."""
        dev = self.device
        shapes = []
        for w, b in layers:
            cin, cout = w.shape
            shapes.append(((cin + 3) // 4 * 4, (cout + 3) // 4 * 4))
        total = sum(pi * po + po for pi, po in shapes)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g, self.flat_m, self.flat_v = (torch.zeros_like(self.flat_p) for _ in range(3))
        self.W, self.b, self.gW, self.gb, self.Wt = [], [], [], [], []
        self.off_w, self.off_b = [], []
        host = np.zeros(total, np.float32)
        o = 0
        for (w, b), (pin, pout) in zip(layers, shapes):
            cin, cout = w.shape
            self.off_w.append(o); self.off_b.append(o + pin * pout)
            wp = host[o:o + pin * pout].reshape(pin, pout); wp[:cin, :cout] = w
            self.W.append(self.flat_p[o:o + pin * pout].view(pin, pout)); self.gW.append(self.flat_g[o:o + pin * pout].view(pin, pout))
            o += pin * pout
            host[o:o + cout] = b
            self.b.append(self.flat_p[o:o + pout]); self.gb.append(self.flat_g[o:o + pout])
            o += pout
            self.Wt.append(torch.empty((pout, pin), dtype=torch.float32, device=dev))
        self.flat_p.copy_(torch.from_numpy(host))
        self._descs = {}
        self.pk_fwd, self.pk_bwd, self._repack_args = [], [], None
        if self.mixed:
            for k, (pin, pout) in enumerate(shapes):
                nb = lambda ci, co: int(self.lib.himo_conv_packed_weight_bytes(1, ci, co))
                self.pk_fwd.append(torch.empty(nb(pin, pout), dtype=torch.uint8, device=dev))
                self.pk_bwd.append(torch.empty(nb(pout, pin), dtype=torch.uint8, device=dev) if k > 0 else None)
            L = len(shapes)
            P, I = ctypes.c_void_p * L, ctypes.c_int * L
            self._repack_args = (L, P(*[w.data_ptr() for w in self.W]), I(*[p for p, _ in shapes]), I(*[q for _, q in shapes]),
                                 P(*[b.data_ptr() for b in self.pk_fwd]), P(*[None if b is None else b.data_ptr() for b in self.pk_bwd]))
            self._repack()

    def _repack(self):
        """refresh the split copies of every layer's W (forward) and W^T (input gradient): one launch (csrc/convbf.hip)"""
        if self._repack_args is not None:
            _lib.check(self.lib.himo_mlp_repack(*self._repack_args, _lib.stream_handle()), "himo_mlp_repack")

    def _gemm(self, x, w, bias, y, n, cin, cout, epi, aux=None, packed=None, fmt=0):
        key = (x.data_ptr(), w.data_ptr(), y.data_ptr(), epi)          # descriptors are cached per call site
        d = self._descs.get(key)
        if d is None:
            d = ConvDesc()
            d.x = x.data_ptr(); d.x_batch_stride = 0; d.x_pitch = x.shape[1]
            d.w = w.data_ptr(); d.bias = None if bias is None else bias.data_ptr()
            d.y = y.data_ptr(); d.y_batch_stride = 0; d.y_pitch = y.shape[1]
            d.n, d.h, d.w_in, d.cin, d.cout, d.ksize, d.stride, d.epilogue = 1, 1, n, cin, cout, 1, 1, epi
            if aux is not None:
                d.aux_in = aux.data_ptr(); d.aux_in_pitch = aux.shape[1]
            if packed is not None:
                d.w_packed, d.packed_format = packed.data_ptr(), fmt
            self._descs[key] = d
        _lib.check(self.lib.himo_conv2d(ctypes.byref(d), _lib.stream_handle()), "himo_conv2d(mlp)")

    def _fused_args(self):
        """pointer tables of the fused kernels (buffers are fixed for the duration of a fit)"""
        L = len(self.W)                                         # 1 + (N_HIDDEN - 1) + 1 layers
        P = ctypes.c_void_p * N_HIDDEN
        hid = lambda bufs: P(*([None] + [bufs[k].data_ptr() for k in range(1, L - 1)]))
        self._fz = dict(wf=hid(self.pk_fwd), wb=hid(self.pk_bwd), bias=hid(self.b), H=P(*[h.data_ptr() for h in self.H]),
                        dZ=P(*[z.data_ptr() for z in self.dZ]), gb=P(*[self.gb[k].data_ptr() for k in range(N_HIDDEN)]))

    def _forward(self, n):
        L = len(self.W)
        if self.fused:
            f = self._fz
            _lib.check(self.lib.himo_mlp_forward_fused(n, self.X0.data_ptr(), N_HIDDEN, self.W[0].data_ptr(), self.b[0].data_ptr(), f["wf"], f["bias"],
                                                       self.W[L - 1].data_ptr(), self.b[L - 1].data_ptr(), f["H"], self.OUT.data_ptr(),
                                                       _lib.stream_handle()), "himo_mlp_forward_fused")
            return
        for k in range(L):
            x = self.X0 if k == 0 else self.H[k - 1]
            y = self.OUT if k == L - 1 else self.H[k]
            self._gemm(x, self.W[k], self.b[k], y, n, self.W[k].shape[0], self.W[k].shape[1], EPI_BIAS if k == L - 1 else EPI_BIAS_RELU,
                       packed=self.pk_fwd[k] if self.mixed else None, fmt=1)

    def fit(self, pc0, pc1, pose0=None, pose1=None, layers=None) -> torch.Tensor:
        """-> (N0,3) float32 device tensor: flow of every pc0 row including ego motion."""
        lib, dev, s = self.lib, self.device, _lib.stream_handle
        up = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(dev, torch.float32)
        p0_raw, p1 = up(pc0)[:, :3].contiguous(), up(pc1)[:, :3].contiguous()
        n, n1 = p0_raw.shape[0], p1.shape[0]
        if self.fused and self.three_launch and self.objective == "dt" and n > 0:
            return self._fit_three_launches(p0_raw, p1, pose0, pose1, layers)
        # ego transform (host 4x4, float32 like seflow/spec.py step 0), applied by the pillar front end's rule
        T = np.eye(4) if pose0 is None else np.linalg.inv(np.asarray(pose1, np.float64)) @ np.asarray(pose0, np.float64)
        T32 = torch.from_numpy(np.ascontiguousarray(T, dtype=np.float32)).to(dev)
        n_pad = (n + 63) // 64 * 64                            # the fused kernels work on whole 64-row blocks (csrc/mlpfused.hip)
        self.X0 = torch.zeros((n_pad, 4), dtype=torch.float32, device=dev)[:n]                  # [x', y', z', 0]
        _lib.check(lib.himo_rigid_transform(n, p0_raw.data_ptr(), 3, T32.data_ptr(), self.X0.data_ptr(), 4, s()), "rigid")
        self._load(init_mlp(self.seed) if layers is None else layers)
        buf = lambda c: torch.zeros((n_pad, c), dtype=torch.float32, device=dev)[:n]
        self.H = [buf(HIDDEN) for _ in range(N_HIDDEN)]
        self.dH = [buf(HIDDEN) for _ in range(2)]
        self.OUT, self.dOUT = buf(4), buf(4)
        if self.fused:
            self.dZ = [buf(HIDDEN) for _ in range(N_HIDDEN)]    # masked gradients at every hidden layer's output (weight-gradient operands)
            self._fused_args()
        moved, gmoved = torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n, 3), dtype=torch.float32, device=dev)
        dt = self.objective == "dt"
        if dt:
            # the target sweep as a distance-transform volume, built once for this pair
            origin, dims, window = dt_grid(self.trunc, self.dt_cell, self.dt_box)
            o_c, d_c = (ctypes.c_float * 3)(*origin.tolist()), (ctypes.c_int * 3)(*dims.tolist())
            need = int(lib.himo_dt_volume_bytes(d_c))
            if self._dt_vol is None or self._dt_vol.numel() < need:
                self._dt_vol = torch.empty(need, dtype=torch.uint8, device=dev)
            _lib.check(lib.himo_dt_build(n1, p1.data_ptr(), o_c, self.dt_cell, d_c, window, self._dt_vol.data_ptr(), self._dt_vol.numel(), s()),
                       "himo_dt_build")
            ch_ws = torch.empty(int(lib.himo_dt_loss_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        else:
            d_a, i_a = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
            d_b, i_b = torch.empty(n1, dtype=torch.float32, device=dev), torch.empty(n1, dtype=torch.int32, device=dev)
            nn_ws = torch.empty(int(lib.himo_nn_grid_workspace_bytes(max(n, n1), GRID_W, GRID_H)), dtype=torch.uint8, device=dev)
            ch_ws = torch.empty(int(lib.himo_chamfer_trunc_workspace_bytes(n, n1)), dtype=torch.uint8, device=dev)
        wg_ws = torch.empty(int(lib.himo_wgrad_workspace_bytes_ex(n, HIDDEN, HIDDEN)), dtype=torch.uint8, device=dev)
        mb_ws = torch.empty(int(lib.himo_mlp_bias_workspace_bytes(n, N_HIDDEN)), dtype=torch.uint8, device=dev)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self._moved, self._gmoved = moved, gmoved             # (tests read the objective's gradient)
        self.loss_history, best, stale = [], float("inf"), 0
        L = len(self.W)

        def nn(q, nq, r, nr, d, i):
            _lib.check(lib.himo_nn_grid(nq, q.data_ptr(), nr, r.data_ptr(), GRID_X0, GRID_Y0, GRID_CELL, GRID_W, GRID_H, d.data_ptr(),
                                        i.data_ptr(), nn_ws.data_ptr(), nn_ws.numel(), s()), "himo_nn_grid")

        for it in range(1, self.iters + 1):
            self._forward(n)
            _lib.check(lib.himo_rows_add(n, 3, self.X0.data_ptr(), 4, self.OUT.data_ptr(), 4, 1.0, moved.data_ptr(), 3, 0, s()), "rows_add")
            if dt:
                _lib.check(lib.himo_dt_loss(n, moved.data_ptr(), o_c, self.dt_cell, d_c, window, self._dt_vol.data_ptr(), self.trunc,
                                            loss.data_ptr(), gmoved.data_ptr(), ch_ws.data_ptr(), ch_ws.numel(), s()), "himo_dt_loss")
            else:
                nn(moved, n, p1, n1, d_a, i_a)
                nn(p1, n1, moved, n, d_b, i_b)
                _lib.check(lib.himo_chamfer_trunc(n, n1, moved.data_ptr(), p1.data_ptr(), d_a.data_ptr(), i_a.data_ptr(), d_b.data_ptr(),
                                                  i_b.data_ptr(), self.trunc, loss.data_ptr(), gmoved.data_ptr(), ch_ws.data_ptr(),
                                                  ch_ws.numel(), s()), "himo_chamfer_trunc")
            _lib.check(lib.himo_rows_add(n, 3, gmoved.data_ptr(), 3, None, 0, 0.0, self.dOUT.data_ptr(), 4, 1, s()), "rows_add")
            # backward: dZ_k is the gradient at layer k's output (post-mask for hidden layers)
            if self.fused:
                f = self._fz
                # ... which also leaves the hidden layers' bias gradients (column sums of dZ_k, taken inside the kernel)
                _lib.check(lib.himo_mlp_backward_fused_bias(n, self.dOUT.data_ptr(), N_HIDDEN, f["wb"], self.W[L - 1].data_ptr(), f["H"], f["dZ"],
                                                            f["gb"], mb_ws.data_ptr(), mb_ws.numel(), s()), "himo_mlp_backward_fused_bias")
                for k in range(L):
                    xk = self.X0 if k == 0 else self.H[k - 1]
                    dz = self.dOUT if k == L - 1 else self.dZ[k]
                    cin, cout = self.W[k].shape
                    _lib.check(lib.himo_linear_wgrad_ex(n, xk.data_ptr(), xk.shape[1], cin, dz.data_ptr(), dz.shape[1], cout,
                                                        self.gW[k].data_ptr(), self.gb[k].data_ptr() if k == L - 1 else None, 2,
                                                        wg_ws.data_ptr(), wg_ws.numel(), s()), "wgrad")
            dz = self.dOUT
            for k in (range(L - 1, -1, -1) if not self.fused else ()):
                xk = self.X0 if k == 0 else self.H[k - 1]
                cin, cout = self.W[k].shape
                _lib.check(lib.himo_linear_wgrad_ex(n, xk.data_ptr(), xk.shape[1], cin, dz.data_ptr(), dz.shape[1], cout,
                                                    self.gW[k].data_ptr(), self.gb[k].data_ptr(), 2 if self.mixed else 0, wg_ws.data_ptr(),
                                                    wg_ws.numel(), s()), "wgrad")
                if k > 0:
                    nxt = self.dH[k % 2]
                    if self.mixed:                              # W^T's two-term bf16 copy was refreshed by _repack
                        self._gemm(dz, self.Wt[k], None, nxt, n, cout, cin, EPI_RELU_MASK, aux=self.H[k - 1], packed=self.pk_bwd[k], fmt=2)
                    else:
                        _lib.check(lib.himo_transpose(self.W[k].data_ptr(), cin, cout, self.Wt[k].data_ptr(), s()), "transpose")
                        self._gemm(dz, self.Wt[k], None, nxt, n, cout, cin, EPI_RELU_MASK, aux=self.H[k - 1])
                    dz = nxt
            _lib.check(lib.himo_adam_step(self.flat_p.numel(), self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                          self.flat_v.data_ptr(), self.lr, 0.9, 0.999, 1e-8, it, s()), "adam")
            self._repack()
            if self.early_patience > 0 or it == self.iters or it <= 3:
                lv = float(loss.item())
                self.loss_history.append((it, lv))
                if self.early_patience > 0:
                    if lv < best - self.early_min_delta:
                        best, stale = lv, 0
                    else:
                        stale += 1
                        if stale >= self.early_patience:
                            break
        self._forward(n)
        flow = torch.empty((n, 3), dtype=torch.float32, device=dev)
        # flow incl. ego motion = (p' + f(p')) - p
        _lib.check(lib.himo_rows_add(n, 3, self.X0.data_ptr(), 4, self.OUT.data_ptr(), 4, 1.0, moved.data_ptr(), 3, 0, s()), "rows_add")
        _lib.check(lib.himo_rows_add(n, 3, moved.data_ptr(), 3, p0_raw.data_ptr(), 3, -1.0, flow.data_ptr(), 3, 0, s()), "rows_add")
        return flow

    def _fit_three_launches(self, p0_raw, p1, pose0, pose1, layers) -> torch.Tensor:
        """The default path: per iteration himo_nsf_forward -> himo_nsf_backward -> himo_nsf_update (csrc/nsffused.hip)."""
        lib, dev, s = self.lib, self.device, _lib.stream_handle
        n, n1 = p0_raw.shape[0], p1.shape[0]
        T = np.eye(4) if pose0 is None else np.linalg.inv(np.asarray(pose1, np.float64)) @ np.asarray(pose0, np.float64)
        T32 = torch.from_numpy(np.ascontiguousarray(T, dtype=np.float32)).to(dev)
        n_pad = int(lib.himo_nsf_padded_rows(n))
        tiles, blocks = n_pad // 64, int(lib.himo_nsf_backward_blocks(n))
        self.X0 = torch.zeros((n_pad, 4), dtype=torch.float32, device=dev)[:n]                 # [x', y', z', 0]; padding rows zero
        _lib.check(lib.himo_rigid_transform(n, p0_raw.data_ptr(), 3, T32.data_ptr(), self.X0.data_ptr(), 4, s()), "rigid")
        self._load(init_mlp(self.seed) if layers is None else layers)
        L = len(self.W)                                         # 1 + (N_HIDDEN - 1) + 1 layers
        total = self.flat_p.numel()
        stride = (total + 63) // 64 * 64
        self.OUT = torch.zeros((n_pad, 4), dtype=torch.float32, device=dev)[:n]               # (views of the padded buffers)
        self.dOUT = torch.zeros((n_pad, 4), dtype=torch.float32, device=dev)[:n]
        spill = torch.empty(int(lib.himo_nsf_spill_bytes(n, N_HIDDEN)), dtype=torch.uint8, device=dev)
        partial = torch.empty((blocks, stride), dtype=torch.float32, device=dev)
        loss_partial = torch.zeros(tiles, dtype=torch.float64, device=dev)
        count_partial = torch.zeros(tiles, dtype=torch.int32, device=dev)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        # the target sweep as a distance-transform volume, built once for this pair
        origin, dims, window = dt_grid(self.trunc, self.dt_cell, self.dt_box)
        o_c, d_c = (ctypes.c_float * 3)(*origin.tolist()), (ctypes.c_int * 3)(*dims.tolist())
        need = int(lib.himo_dt_volume_bytes(d_c))
        if self._dt_vol is None or self._dt_vol.numel() < need:
            self._dt_vol = torch.empty(need, dtype=torch.uint8, device=dev)
        _lib.check(lib.himo_dt_build(n1, p1.data_ptr(), o_c, self.dt_cell, d_c, window, self._dt_vol.data_ptr(), self._dt_vol.numel(), s()),
                   "himo_dt_build")
        P = ctypes.c_void_p * N_HIDDEN
        hid = lambda bufs: P(*([None] + [bufs[k].data_ptr() for k in range(1, L - 1)]))
        wf, wb, bias = hid(self.pk_fwd), hid(self.pk_bwd), hid(self.b)
        I = ctypes.c_int * L
        off_w, off_b = I(*self.off_w), I(*self.off_b)

        def forward(with_objective: bool):
            _lib.check(lib.himo_nsf_forward(n, self.X0.data_ptr(), N_HIDDEN, self.W[0].data_ptr(), self.b[0].data_ptr(), wf, bias,
                                            self.W[L - 1].data_ptr(), self.b[L - 1].data_ptr(), spill.data_ptr(), self.OUT.data_ptr(),
                                            o_c, self.dt_cell, d_c, window, self._dt_vol.data_ptr(), self.trunc,
                                            self.dOUT.data_ptr() if with_objective else None, loss_partial.data_ptr(), count_partial.data_ptr(),
                                            s()), "himo_nsf_forward")

        self.loss_history, best, stale = [], float("inf"), 0
        loss_hist = torch.zeros(max(self.iters, 1), dtype=torch.float64, device=dev)       # one slot per iteration: read back ONCE, after the fit
        done = 0
        for it in range(1, self.iters + 1):
            forward(True)
            _lib.check(lib.himo_nsf_backward(n, self.X0.data_ptr(), self.dOUT.data_ptr(), N_HIDDEN, wb, self.W[L - 1].data_ptr(), spill.data_ptr(),
                                             off_w, off_b, stride, partial.data_ptr(), s()), "himo_nsf_backward")
            _lib.check(lib.himo_nsf_update(total, blocks, stride, partial.data_ptr(), tiles, loss_partial.data_ptr(), count_partial.data_ptr(),
                                           self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                           self.lr, 0.9, 0.999, 1e-8, it, N_HIDDEN, off_w, wf, wb, loss_hist.data_ptr() + 8 * (it - 1), count.data_ptr(), s()),
                       "himo_nsf_update")
            done = it
            if self.early_patience > 0:                         # early stopping watches the loss: one host sync per iteration
                lv = float(loss_hist[it - 1].item())
                self.loss_history.append((it, lv))
                if lv < best - self.early_min_delta:
                    best, stale = lv, 0
                else:
                    stale += 1
                    if stale >= self.early_patience:
                        break
        # (tests read the objective's gradient of the LAST iteration: d loss / d moved = dOUT / points in the volume -- on demand)
        self._gm_lazy = (self.dOUT, count)
        forward(False)
        moved = torch.empty((n, 3), dtype=torch.float32, device=dev)
        flow = torch.empty((n, 3), dtype=torch.float32, device=dev)
        # flow incl. ego motion = (p' + f(p')) - p
        _lib.check(lib.himo_rows_add(n, 3, self.X0.data_ptr(), 4, self.OUT.data_ptr(), 4, 1.0, moved.data_ptr(), 3, 0, s()), "rows_add")
        _lib.check(lib.himo_rows_add(n, 3, moved.data_ptr(), 3, p0_raw.data_ptr(), 3, -1.0, flow.data_ptr(), 3, 0, s()), "rows_add")
        self._moved = moved

        def finish():
            # the loss trajectory and the count, read once the whole fit is queued: no bubble inside it (fit_async: not even here)
            if self.early_patience <= 0 and done:
                hist = loss_hist[:done].cpu().numpy()
                self.loss_history = [(it, float(hist[it - 1])) for it in range(1, done + 1) if it <= 3 or it == done]
            self.points_in_volume = max(int(count.item()), 1)
        self._finish = finish
        if not self._defer_finish:
            self.wait()
        return flow

    def fit_async(self, pc0, pc1, pose0=None, pose1=None, layers=None) -> torch.Tensor:
        """``fit`` without its closing host reads: every launch of the fit is queued on the current stream and the flow tensor is
        returned at once (valid in stream order); ``wait()`` reads the loss trajectory and the point count.  Two engines on two
        streams keep two fits in flight this way (``OverlappedFastNSF``).  Early stopping reads the loss every iteration and
        therefore runs synchronously."""
        self._defer_finish = True
        try:
            return self.fit(pc0, pc1, pose0, pose1, layers)
        finally:
            self._defer_finish = False

    def wait(self):
        """completes a ``fit_async``: ``loss_history`` and ``points_in_volume`` are valid afterwards (blocks on the fit's stream)"""
        f, self._finish = self._finish, None
        if f is not None:
            f()

    @property
    def _gmoved(self):
        """d loss / d moved of the last iteration (tests)"""
        if getattr(self, "_gm_lazy", None) is not None:
            dout, count = self._gm_lazy
            return dout[:, :3] / float(max(int(count.item()), 1))
        return self._gm_value

    @_gmoved.setter
    def _gmoved(self, v):
        self._gm_lazy, self._gm_value = None, v

    def layers(self) -> list:
        """Current parameters as [(W [in,out], b [out])] numpy with the padding removed."""
        dims = [3] + [HIDDEN] * N_HIDDEN + [3]
        return [(w.cpu().numpy()[:ci, :co].copy(), b.cpu().numpy()[:co].copy()) for w, b, ci, co in zip(self.W, self.b, dims[:-1], dims[1:])]


class OverlappedFastNSF:
    """Two FastNSF engines on two HIP streams: the fit of sweep pair k + 1 is queued while the fit of pair k runs, so one fit's
    kernel boundaries and partial last waves (the forward kernel's 1875 blocks are 2.4 rounds of the chip, the iteration is a
    strict chain forward -> backward -> update) are filled by the other's kernels.  Fits of different pairs are independent
    (the model is per scene pair: README.md:53 ``model=fastnsf``); each is the same launch sequence as ``FastNSF.fit`` -- same bits.

        nsf = OverlappedFastNSF(device=dev, iters=100)
        for flow in nsf.fits(pairs):          # pairs: iterable of (pc0, pc1, pose0, pose1); flows come back in order
            ...
    """

    def __init__(self, device=None, engines: int = 2, **kw):
        self.device = device if device is not None else _lib.require_gpu()
        self.engines = [FastNSF(device=self.device, **kw) for _ in range(engines)]
        from .pipeline import batch_streams
        self.streams = batch_streams(self.device, engines)          # (one list per process: pipeline.batch_streams)
        self._turn = 0
        self._busy = [False] * engines

    def submit(self, pc0, pc1, pose0=None, pose1=None):
        """queue one fit on the next engine (after completing that engine's previous fit); -> (engine index, flow tensor).  The
        flow is valid once ``collect(index)`` has returned (or in the engine's stream order)."""
        k = self._turn % len(self.engines)
        self._turn += 1
        self.collect(k)
        st, caller = self.streams[k], torch.cuda.current_stream(self.device)
        st.wait_stream(caller)                                   # the inputs were produced on the caller's stream
        with torch.cuda.stream(st):
            flow = self.engines[k].fit_async(pc0, pc1, pose0, pose1)
        flow.record_stream(caller)          # ... and the flow is consumed there: its memory must not go back to the engine's stream early
        for t in (pc0, pc1):                # ... and the sweeps are read on the engine's stream: a caller that drops them right after this call
            if isinstance(t, torch.Tensor) and t.is_cuda and t.numel():        # (a generator of pairs) must not see their memory handed out again before
                t.record_stream(st)
        self._busy[k] = True
        return k, flow

    def collect(self, k: int):
        """wait for engine k's fit in flight (no-op when it has none): its ``loss_history`` / ``points_in_volume`` are valid after"""
        if self._busy[k]:
            with torch.cuda.stream(self.streams[k]):
                self.engines[k].wait()
            self.streams[k].synchronize()
            self._busy[k] = False

    def drain(self):
        for k in range(len(self.engines)):
            self.collect(k)

    sync_check = drain                       # (the name OverlappedPipeline gives the same thing)

    def fits(self, pairs):
        pending = []
        for pair in pairs:
            pending.append(self.submit(*pair))
            if len(pending) == len(self.engines):
                k, flow = pending.pop(0)
                self.collect(k)
                yield flow
        for k, flow in pending:
            self.collect(k)
            yield flow

