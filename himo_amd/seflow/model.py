"""Host orchestration of the scene-flow network (stage a10) on one MI355X.

Python only sequences C-ABI kernel launches (include/himo_amd.h) on torch-owned HBM buffers; there
is no torch arithmetic on the forward path.  Architecture and parameter names: seflow/spec.py
(self-specified -- the reference's network source is absent).

Memory plan (NHWC float32, one sample = 3 sweeps): all feature maps are preallocated once; channel
concatenation is done by writing producers straight into channel groups of the wider consumer
buffer (``pitch`` / ``batch_stride`` addressing), so the decoder never copies.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _lib
from . import spec

EPI_BIAS, EPI_BIAS_BN_GELU, EPI_BIAS_GELU, EPI_GRU_ZR, EPI_GRU_Q = range(5)


class ConvDesc(ctypes.Structure):
    """include/himo_amd.h: himo_conv_desc."""
    _fields_ = [("x", ctypes.c_void_p), ("x_batch_stride", ctypes.c_int64), ("x_pitch", ctypes.c_int),
                ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
                ("y", ctypes.c_void_p), ("y_batch_stride", ctypes.c_int64), ("y_pitch", ctypes.c_int),
                ("n", ctypes.c_int), ("h", ctypes.c_int), ("w_in", ctypes.c_int), ("cin", ctypes.c_int),
                ("cout", ctypes.c_int), ("ksize", ctypes.c_int), ("stride", ctypes.c_int), ("epilogue", ctypes.c_int),
                ("aux_in", ctypes.c_void_p), ("aux_in_pitch", ctypes.c_int),
                ("aux_out", ctypes.c_void_p), ("aux_out_pitch", ctypes.c_int), ("w_packed", ctypes.c_void_p),
                ("tile_hint", ctypes.c_int), ("packed_format", ctypes.c_int),
                ("n_outer", ctypes.c_int), ("x_outer_stride", ctypes.c_int64), ("y_outer_stride", ctypes.c_int64),
                ("act_layout", ctypes.c_int), ("range_seen", ctypes.c_void_p)]


ACT_SPLIT_IN, ACT_SPLIT_OUT = 1, 2      # himo_conv_desc.act_layout: x / y in the split activation format (csrc/convsg.hip)
ACT_ACCUMULATE = 8                      # y += result (two-term bf16 3x3 kernel, float32 maps)
ACT_STUFFED_2X = 16                     # x is a compact [H/2][W/2] map read as its zero-stuffed x2 image (same kernel)


_lib.register({
    "himo_pillar_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "himo_pillar_occupancy_reset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "himo_pillarize": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,                    # pfn weight/scale/shift
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,   # xyz_t, pid, offsets, image
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "himo_conv2d": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_void_p]),
    "himo_conv_packed_weight_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "himo_conv_pack_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                              ctypes.c_void_p]),
    "himo_upsample2x": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "himo_upsample2x_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "himo_upsample2x_batch_ex": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "himo_conv_pack_weights_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                 ctypes.c_void_p]),
    "himo_gru_head": (ctypes.c_int, [ctypes.c_int64] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
                      + [ctypes.c_void_p] * 12 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "himo_gru_head_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 10
                            + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "himo_gru_head_batch_folded": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8
                                   + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "himo_gru_head_batch_guarded": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 10
                                    + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "himo_clear_u32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "himo_head_gather": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "himo_head_final": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p]),
})


class HimoHeadSample(ctypes.Structure):
    """mirror of `himo_head_sample` (include/himo_amd.h)"""
    _fields_ = [("n", ctypes.c_int64), ("d_pid", ctypes.c_void_p), ("d_offsets", ctypes.c_void_p), ("d_img0", ctypes.c_void_p),
                ("d_img1", ctypes.c_void_p), ("d_dec", ctypes.c_void_p), ("d_xyz_t", ctypes.c_void_p), ("d_pts", ctypes.c_void_p),
                ("pc_stride", ctypes.c_int), ("d_flow", ctypes.c_void_p)]


class HimoSweep(ctypes.Structure):
    """mirror of `himo_sweep` (include/himo_amd.h)"""
    _fields_ = [("n", ctypes.c_int64), ("d_pts", ctypes.c_void_p), ("pc_stride", ctypes.c_int), ("transform", ctypes.c_float * 16),
                ("d_xyz_t", ctypes.c_void_p), ("d_pid", ctypes.c_void_p), ("d_offsets", ctypes.c_void_p), ("d_image", ctypes.c_void_p),
                ("d_workspace", ctypes.c_void_p)]


class HimoOp(ctypes.Structure):
    """mirror of `himo_op` (include/himo_amd.h)"""
    _fields_ = [("kind", ctypes.c_int), ("conv", ConvDesc),
                ("up_x", ctypes.c_void_p), ("up_x_pitch", ctypes.c_int), ("up_h", ctypes.c_int), ("up_w", ctypes.c_int),
                ("up_c", ctypes.c_int), ("up_y", ctypes.c_void_p), ("up_y_pitch", ctypes.c_int),
                ("up_n", ctypes.c_int), ("up_x_batch_stride", ctypes.c_int64), ("up_y_batch_stride", ctypes.c_int64),
                ("up_out_split", ctypes.c_int)]


_lib.register({
    "himo_pillarize_multi": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t,
                                            ctypes.c_void_p]),
    "himo_pillarize_multi_ex": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t,
                                               ctypes.c_int, ctypes.c_void_p]),
    "himo_run_ops": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]),
    "himo_ops_release": (None, [ctypes.c_void_p]),
})


def _f32x(values):
    return (ctypes.c_float * len(values))(*[float(v) for v in values])


class SeFlowNet:
    """``forward(pch1, pc0, pc1, pose_h1, pose0, pose1)`` -> (N0,3) float32 device tensor: the flow of every
    pc0 row INCLUDING ego motion (the h5 ``<res_name>`` dataset that save_zip.py:117 reads)."""

    def __init__(self, params: dict | None = None, device=None, max_points: int = 140_000, seed: int = 0,
                 precision: str = "bf16x3", autotune: bool = True, max_batch: int = 1):
        """``precision``: "bf16x3" = split-bf16 matrix instructions for every stride-1 convolution / GEMM (float32-class
        accuracy, float32 range; csrc/convbf.hip); "f16x2" = two-term fp16 split (x = h + l, one accumulator) for the
        convolutions (22-bit products, HALF the matrix instructions of bf16x3; activations and weights must stay
        below fp16's 65504 -- true for this normalised network; the head keeps bf16x3); "f32" = float32 MFMA everywhere."""
        if precision not in ("bf16x3", "f16x2", "f32"):
            raise ValueError(precision)
        if max_batch < 1:
            raise ValueError("max_batch")
        # ``max_batch`` samples go through every backbone layer in ONE launch (forward_batch): the low-resolution layers
        # of a single sample are a fraction of one round of blocks on 256 CUs; activation buffers are ~1 GB per sample
        self.max_batch = max_batch
        self.precision = precision
        self.autotune = autotune
        self.keep_cell_lists = False
        self.fused_head = precision != "f32"          # one kernel for gather + GRU + output (csrc/gruhead.hip)
        self._incremental_images = True               # pillar images: write only the cells that changed since the last forward
        self.use_plan = True                          # replay the backbone's operator list from one call (csrc/plan.hip)
        self.use_graph = True                         # ... as a captured hipGraph
        self._plans = {}                              # samples per launch -> recorded operator list
        self._recording = None
        self._nb = 1                                  # samples the backbone currently runs over
        self.packed_format = 1 if precision == "f16x2" else 0
        # fp16 split: every map of the backbone (pillar images, encoder / decoder maps; not the decoder output the head
        # gathers) is stored already split -- fp16 pairs in place of floats, csrc/convsg.hip -- and staged by LDS-DMA;
        # results are bit-identical either way.  False keeps every activation buffer float32 (the training pass reads them)
        self._split_acts = precision == "f16x2"
        self.tiles = {}
        self.lib = _lib.load()
        self.device = device if device is not None else _lib.require_gpu()
        params = spec.init_params(seed) if params is None else params
        shapes = spec.param_shapes()
        for k, shp in shapes.items():
            if k not in params or tuple(params[k].shape) != tuple(shp):
                raise KeyError(f"parameter {k} missing or of wrong shape (want {shp})")
        cpu = {k: torch.from_numpy(np.ascontiguousarray(params[k], dtype=np.float32)) for k in shapes}
        # BatchNorm(eval) folded to scale / shift in float32, formed exactly as the oracle forms them
        def fold(prefix, eps):
            scale = cpu[f"{prefix}.gamma"] / torch.sqrt(cpu[f"{prefix}.var"] + eps)
            return scale, cpu[f"{prefix}.beta"] - cpu[f"{prefix}.mean"] * scale
        derived = {}
        derived["pfn.scale"], derived["pfn.shift"] = fold("pfn.bn", spec.BN_EPS_PFN)
        for name, *_ in spec.ENCODER:
            derived[f"{name}.scale"], derived[f"{name}.shift"] = fold(f"{name}.bn", spec.BN_EPS)
        # folded decoder joints: a block's last 3x3 conv (u5) feeds ONLY the next block's 1x1 conv (u1) and nothing stands between
        # them (spec step 4: no activations), so the pair is one 3x3 conv with W = W_u5 W_u1, b = b_u5 W_u1 + b_u1 (formed in
        # float64): half the output channels of u5, and the 1x1 launch with its float32 round trip disappears.  Exact at the
        # borders too (the 1x1 is pointwise).  The training pass keeps the layers apart (it needs u5's output).
        self.fold_decoder = True
        for a, b in (("dec1", "dec2"), ("dec2", "dec3")):
            w5, b5 = cpu[f"{a}.u5.weight"].double().numpy(), cpu[f"{a}.u5.bias"].double().numpy()
            w1, b1 = cpu[f"{b}.u1.weight"].double().numpy()[0, 0], cpu[f"{b}.u1.bias"].double().numpy()
            derived[f"{a}.u5>{b}.u1.weight"] = torch.from_numpy(np.ascontiguousarray(w5 @ w1, dtype=np.float32))
            derived[f"{a}.u5>{b}.u1.bias"] = torch.from_numpy(np.ascontiguousarray(b5 @ w1 + b1, dtype=np.float32))
        derived["head.gru.zr.weight"] = torch.cat([cpu["head.gru.z.weight"], cpu["head.gru.r.weight"]], dim=1).contiguous()
        derived["head.gru.zr.bias"] = torch.cat([cpu["head.gru.z.bias"], cpu["head.gru.r.bias"]]).contiguous()
        self.p = {k: v.to(self.device) for k, v in {**cpu, **derived}.items()}
        self.packed = {}
        if precision != "f32":
            for k, v in self.p.items():
                if k.endswith(".weight") and k != "pfn.weight" and not k.startswith("head.offset") and k != "head.dec2.weight":
                    w = v if v.dim() == 4 else v.reshape(1, 1, *v.shape)          # linears are 1x1 convolutions
                    ks, _, cin, cout = w.shape
                    fmt = self.packed_format
                    buf = torch.empty(int(self.lib.himo_conv_packed_weight_bytes(ks, cin, cout)), dtype=torch.uint8, device=self.device)
                    _lib.check(self.lib.himo_conv_pack_weights_ex(w.contiguous().data_ptr(), ks, cin, cout, fmt, buf.data_ptr(),
                                                                  _lib.stream_handle()), "himo_conv_pack_weights_ex")
                    self.packed[k] = buf

        # folded head (csrc/gruhead.hip, himo_gru_head_batch_folded): x = Linear(3,64)(offset) is affine in the offset and constant
        # over the GRU iterations, so the 64 x-rows of every head matrix collapse (in float64, here) to 4 rows -- W_off W_x and
        # b_off W_x -- that meet (o0, o1, o2, 1) in the A operand: [144][cout] matrices, 9 slabs per GEMM instead of 12
        self.fold_head = precision != "f32"
        if self.fold_head:
            w_off, b_off = cpu["head.offset.weight"].double().numpy(), cpu["head.offset.bias"].double().numpy()
            for name in ("head.gru.zr", "head.gru.q", "head.dec1"):
                wfull = (derived if name.endswith("zr") else cpu)[f"{name}.weight"].double().numpy()
                wh, wx = wfull[:spec.HIDDEN], wfull[spec.HIDDEN:]
                folded = np.concatenate([wh, w_off @ wx, (b_off @ wx)[None], np.zeros((12, wfull.shape[1]))], axis=0)      # [144][cout]
                wdev = torch.from_numpy(np.ascontiguousarray(folded, dtype=np.float32)).to(self.device)
                buf = torch.empty(int(self.lib.himo_conv_packed_weight_bytes(1, 144, wdev.shape[1])), dtype=torch.uint8, device=self.device)
                _lib.check(self.lib.himo_conv_pack_weights_ex(wdev.data_ptr(), 1, 144, wdev.shape[1], self.packed_format, buf.data_ptr(),
                                                              _lib.stream_handle()), "himo_conv_pack_weights_ex")
                torch.cuda.synchronize(self.device)                                          # wdev may be freed once packed
                self.packed[f"{name}.weight.h"] = buf

        H, W = spec.GRID
        F = spec.NUM_FRAMES
        self.H, self.W, self.F = H, W, F
        dev = self.device
        B = self.max_batch
        buf = lambda *shape: torch.empty((B, *shape), dtype=torch.float32, device=dev)      # [sample][...]
        self._B0 = buf(H * W, 32 * F)                                  # 3 pillar images as channel groups
        self.E1 = [buf(F, (H // 2) * (W // 2), 64) for _ in range(2)]
        self.F1 = buf((H // 2) * (W // 2), 64 * F)
        self.E2 = [buf(F, (H // 4) * (W // 4), 128) for _ in range(2)]
        self.F2 = buf((H // 4) * (W // 4), 128 * F)
        self.E3 = [buf(F, (H // 8) * (W // 8), 256) for _ in range(2)]
        self.F3 = buf((H // 8) * (W // 8), 256 * F)
        self.T1 = buf((H // 8) * (W // 8), 256)
        self.CAT1 = buf((H // 4) * (W // 4), 512)
        self.S = [buf((H // 4) * (W // 4), 256) for _ in range(2)]
        self.T2 = buf((H // 4) * (W // 4), 128)
        self.CAT2 = buf((H // 2) * (W // 2), 256)
        self.T = [buf((H // 2) * (W // 2), 128) for _ in range(2)]
        self.T3 = buf((H // 2) * (W // 2), 64)
        self.CAT3 = buf(H * W, 128)
        self.U = [buf(H * W, 64) for _ in range(2)]
        self.DEC = buf(H * W, 64)
        self.max_points = 0
        self._reserve_points(max_points)
        # guard words of the fp16 split (one buffer, one clear, one read-back).  [0] finite-flow guard: the fused head ORs 1 into it
        # when it writes a NaN / inf flow value (csrc/gruhead.hip).  [1 + k] low-side guard of split-output layer k
        # (himo_conv_desc.d_range_seen): set by the layer when it sees an output of magnitude >= 2^-6; a word still 0 after a
        # forward pass = that layer's activations sit on the split's absolute floor (``range_ok``)
        # [-1]: "a backbone pass ran since the words were cleared" (set by ``backbone``): with no pass in between -- an empty batch --
        # the low-side words are still 0 and must not read as "underflow" 
        self.guard = torch.zeros(2 + self.MAX_RANGE_SLOTS, dtype=torch.int32, device=dev)
        self.nonfinite = self.guard[:1]
        self._ran_word = self.guard[-1:]
        self._range_slots = {}
        self._range = _f32x(spec.POINT_CLOUD_RANGE[:3])
        self._voxel = _f32x(spec.VOXEL_SIZE)
        r, v = spec.POINT_CLOUD_RANGE, spec.VOXEL_SIZE
        self._centre = _f32x([v[0] / 2 + r[0], v[1] / 2 + r[1], v[2] / 2 + r[2]])

    def _reserve_points(self, n: int):
        if n <= self.max_points:
            return
        dev = self.device
        self.max_points = n
        need = int(self.lib.himo_pillar_workspace_bytes(n, self.W, self.H))
        # per sample of a batch: one pillar workspace per frame slot (the three sweeps of a sample share their launches,
        # himo_pillarize_multi; training keeps every sweep's cell lists for the backward pass) and the per-point outputs
        self._pt = [{"ws_slots": [torch.empty(need + 64, dtype=torch.uint8, device=dev) for _ in range(self.F)],
                     "xyz_t": torch.empty((self.F, n, 3), dtype=torch.float32, device=dev),
                     "pid": torch.empty((self.F, n), dtype=torch.int32, device=dev),
                     "offsets": torch.empty((self.F, n, 3), dtype=torch.float32, device=dev)} for _ in range(self.max_batch)]
        # incremental pillar images (csrc/pillar.hip, HIMO_IMAGE_INCREMENTAL): a sweep's image channels persist in B0 between
        # forwards and only changed cells are written; fresh workspaces start with every cell marked dirty
        for st in self._pt:
            for ws in st["ws_slots"]:
                _lib.check(self.lib.himo_pillar_occupancy_reset(ws.data_ptr(), ws.numel(), self.W, self.H, _lib.stream_handle()),
                           "himo_pillar_occupancy_reset")
        self._use_sample(0)
        self.hx = torch.empty((n, 192), dtype=torch.float32, device=dev)
        self.rhx = torch.empty((n, 192), dtype=torch.float32, device=dev)
        self.zbuf = torch.empty((n, 128), dtype=torch.float32, device=dev)
        self.y1 = torch.empty((n, 32), dtype=torch.float32, device=dev)

    # The incremental pillar images rest on one invariant: the occupancy bits at the tail of every sweep workspace describe
    # what B0's slot of that sweep holds RIGHT NOW, in the current activation format.  Everything that can break it goes
    # through a setter that marks all cells dirty again (the next forward then rewrites the whole image).
    @property
    def incremental_images(self) -> bool:
        return self._incremental_images

    @incremental_images.setter
    def incremental_images(self, on: bool):
        on = bool(on)
        if on != self._incremental_images:
            self._incremental_images = on
            self.reset_images()                        # whatever was written while it was off is not in the bitmap

    @property
    def split_acts(self) -> bool:
        return self._split_acts

    @split_acts.setter
    def split_acts(self, on: bool):
        on = bool(on)
        if on != self._split_acts:
            self._split_acts = on
            self.drop_plan()                           # recorded operator lists carry the old layout flags
            self.reset_images()                        # the persisted image bytes are in the other format

    @property
    def B0(self):
        return self._B0

    @B0.setter
    def B0(self, buf):
        self._B0 = buf                                 # a rebound / reallocated image buffer holds arbitrary bytes
        self.drop_plan()
        self.reset_images()

    MAX_RANGE_SLOTS = 63

    def clear_nonfinite(self):
        """zero the guard words -- finite flow and the layers' low-side words -- stream-ordered; no host sync"""
        _lib.check(self.lib.himo_clear_u32(self.guard.data_ptr(), self.guard.numel(), _lib.stream_handle()), "himo_clear_u32")

    def _range_word(self, wname: str) -> int:
        """device address of layer ``wname``'s low-side guard word (fp16 split, split-output layers)"""
        k = self._range_slots.setdefault(wname, len(self._range_slots))
        if k >= self.MAX_RANGE_SLOTS:
            raise RuntimeError("more split-output layers than guard words")
        return self.guard.data_ptr() + 4 * (1 + k)

    def guard_verdict(self, words) -> str | None:
        """``words``: host copy of ``self.guard`` taken after a forward pass that followed ``clear_nonfinite``.  None = fine;
        "overflow" = a non-finite flow value (an activation left fp16's range); "underflow" = some split-output layer never
        produced a value of magnitude >= 2^-6 (its activations sit on the split's absolute floor)."""
        w = [int(v) for v in words]
        if w[0] != 0:
            return "overflow"
        if w[-1] != 0 and any(w[1 + k] == 0 for k in self._range_slots.values()):
            return "underflow"
        return None

    def reset_images(self):
        """Mark every pillar-image cell dirty (the next forward rewrites the whole image).  Needed only after something other
        than this network's own pillar stage wrote into B0, or after switching ``incremental_images`` back on."""
        for st in getattr(self, "_pt", ()):
            for ws in st["ws_slots"]:
                _lib.check(self.lib.himo_pillar_occupancy_reset(ws.data_ptr(), ws.numel(), self.W, self.H, _lib.stream_handle()),
                           "himo_pillar_occupancy_reset")

    def _use_sample(self, i: int):
        """bind the per-point buffers (and cell lists) of sample ``i`` of the batch"""
        st = self._pt[i]
        self._sample = i
        self.ws_slots, self.xyz_t, self.pid, self.offsets = st["ws_slots"], st["xyz_t"], st["pid"], st["offsets"]

    # ---- launch helpers ---------------------------------------------------------------------------
    def _conv(self, x, x_bs, x_pitch, wname, y, y_bs, y_pitch, n, h, w, cin, cout, ks, stride, epi, x_off=0, y_off=0,
              scale=None, shift=None, aux_in=None, aux_in_pitch=0, aux_out=None, aux_out_pitch=0, bias=None, batched=True,
              act=0):
        """``x`` / ``y``: activation buffers [sample][...] -- with ``batched`` the layer runs over the first
        ``self._nb`` samples in one launch (outer stride = one sample of the buffer).  ``act``: ACT_SPLIT_IN / _OUT when
        x / y are in the split activation format (only honoured with ``self.split_acts``)."""
        d = ConvDesc()
        d.x = x.data_ptr() + 4 * x_off; d.x_batch_stride = x_bs; d.x_pitch = x_pitch
        if batched and self._nb > 1:
            d.n_outer, d.x_outer_stride, d.y_outer_stride = self._nb, x.stride(0), y.stride(0)
        d.w = self.p[f"{wname}.weight"].data_ptr()
        d.bias = (self.p[f"{wname}.bias"] if bias is None else bias).data_ptr()
        d.scale = None if scale is None else scale.data_ptr()
        d.shift = None if shift is None else shift.data_ptr()
        d.y = y.data_ptr() + 4 * y_off; d.y_batch_stride = y_bs; d.y_pitch = y_pitch
        d.n, d.h, d.w_in, d.cin, d.cout, d.ksize, d.stride, d.epilogue = n, h, w, cin, cout, ks, stride, epi
        d.aux_in = None if aux_in is None else aux_in.data_ptr(); d.aux_in_pitch = aux_in_pitch
        d.aux_out = None if aux_out is None else aux_out.data_ptr(); d.aux_out_pitch = aux_out_pitch
        pk = self.packed.get(f"{wname}.weight")
        d.w_packed = None if pk is None else pk.data_ptr()
        d.packed_format = self.packed_format
        d.act_layout = act if self.split_acts else 0
        if (d.act_layout & ACT_SPLIT_OUT) and self.packed_format == 1 and epi in (EPI_BIAS, EPI_BIAS_BN_GELU):
            d.range_seen = self._range_word(wname)
        key = (n * max(d.n_outer, 1), h, w, cin, cout, ks, stride, epi, pk is not None, d.act_layout)
        if self.autotune and key not in self.tiles:
            self.tiles[key] = self._tune(d)
        d.tile_hint = self.tiles.get(key, 0)
        if self._recording is not None and epi not in (EPI_GRU_ZR, EPI_GRU_Q):
            op = HimoOp(); op.kind = 0; op.conv = d
            self._recording.append(op)
        _lib.check(self.lib.himo_conv2d(ctypes.byref(d), _lib.stream_handle()), f"himo_conv2d({wname})")

    def _tune(self, d: "ConvDesc") -> int:
        """Time the tile variants of one layer shape once (the kernels are idempotent for the non-GRU epilogues; the
        GRU ones update state in place, so they keep the library heuristic) and remember the fastest."""
        if d.epilogue in (EPI_GRU_ZR, EPI_GRU_Q):
            return 0
        stream = _lib.stream_handle()
        cands = [(bn << 4) | mi for bn in (128, 64) if not (bn == 128 and d.cout % 128) for mi in (2, 1)]
        if d.act_layout:                                  # split activation format: only the weights-from-L2 structures
            if d.stride == 2 and (d.act_layout & ACT_SPLIT_IN):
                return 0                                  # one variant
            cands = [0x1000 | 4, 0x1000 | 2, 0x1000 | 1] if d.ksize == 1 else []
        if d.w_packed and d.ksize == 3:
            cands += [0x1000 | 4, 0x1000 | 2, 0x1000 | 1] if d.stride == 1 else [0x1000 | 2, 0x1000 | 1]   # weights-from-L2 structure (csrc/convsp.hip)
            if d.stride == 1 and (d.act_layout & ACT_SPLIT_IN) and self.packed_format == 1:
                # two 32-channel column tiles per wave (convsg.hip): 8-row tiles for the 64-channel layers, 4 rows x 64 channels
                # per wave for the wide ones
                cands.append(0x1000 | (8 if d.cout <= 64 else 12))
        if len(cands) < 2:
            return cands[0] if cands else 0
        times = {hint: float("inf") for hint in cands}
        for _round in range(2):                           # two interleaved rounds, best of each candidate: the chip's clock
            for hint in cands:                            # drifts with load, and a one-shot ranking of near-equal tiles flips
                d.tile_hint = hint
                for _ in range(2):
                    self.lib.himo_conv2d(ctypes.byref(d), stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    self.lib.himo_conv2d(ctypes.byref(d), stream)
                e1.record()
                e1.synchronize()
                times[hint] = min(times[hint], e0.elapsed_time(e1))
        best = min(cands, key=lambda hint: times[hint])
        return best

    def _up(self, x, x_pitch, h, w, c, y, y_pitch, out_split=False):
        nb = self._nb
        osp = 1 if (out_split and self.split_acts) else 0
        if self._recording is not None:
            op = HimoOp(); op.kind = 1
            op.up_x, op.up_x_pitch, op.up_h, op.up_w, op.up_c, op.up_y, op.up_y_pitch = x.data_ptr(), x_pitch, h, w, c, y.data_ptr(), y_pitch
            op.up_n, op.up_x_batch_stride, op.up_y_batch_stride = nb, x.stride(0), y.stride(0)
            op.up_out_split = osp
            self._recording.append(op)
        _lib.check(self.lib.himo_upsample2x_batch_ex(nb, x.data_ptr(), x.stride(0) if nb > 1 else 0, x_pitch, h, w, c, y.data_ptr(),
                                                     y.stride(0) if nb > 1 else 0, y_pitch, osp, _lib.stream_handle()), "himo_upsample2x_batch_ex")

    # ---- stages ---------------------------------------------------------------------------------------
    def backbone(self, n_samples: int = 1):
        """B0 (3 pillar images per sample) -> DEC (64 x H x W per sample) for the first ``n_samples`` samples of the
        activation buffers, every layer ONE launch.  The first call per batch size runs (and tile-tunes) layer by layer
        while recording the operator list; later calls replay it from one C call / one hipGraph launch."""
        if not 1 <= n_samples <= self.max_batch:
            raise ValueError(f"n_samples must be in 1..{self.max_batch}")
        self._nb = n_samples
        if self.packed_format == 1:
            self._ran_word.fill_(1)                       # (stream-ordered: the guard words' read-back sees it with this pass's words)
        plan = self._plans.get(n_samples)
        if self.use_plan and plan is not None:
            ops, n = plan
            _lib.check(self.lib.himo_run_ops(ctypes.addressof(ops), n, 1 if self.use_graph else 0, _lib.stream_handle()), "himo_run_ops")
            return self.DEC
        self._recording = [] if self.use_plan else None
        self.encoder()
        self.decoder()
        if self._recording is not None:
            rec, self._recording = self._recording, None
            self._plans[n_samples] = ((HimoOp * len(rec))(*rec), len(rec))
        return self.DEC

    def __del__(self):
        try:
            self.drop_plan()
        except Exception:
            pass

    def drop_plan(self):
        """forget the recorded operator lists (call after changing weights buffers, precision or tile choices)"""
        for plan in self._plans.values():
            self.lib.himo_ops_release(ctypes.addressof(plan[0]))
        self._plans = {}

    def encoder(self):
        """B0 -> the three concat buffers F1 / F2 / F3 (frames stacked on channels)."""
        H, W, F = self.H, self.W, self.F
        p = self.p
        # encoder: frames are the batch; the last conv of a stage writes into the concat buffer
        stages = [("enc1", 4, 32, 64, H, W, self.E1, self.F1), ("enc2", 6, 64, 128, H // 2, W // 2, self.E2, self.F2),
                  ("enc3", 6, 128, 256, H // 4, W // 4, self.E3, self.F3)]
        src, src_bs, src_pitch = self.B0, 32, 32 * F
        for stage, n_conv, cin, cout, h, w, pingpong, catbuf in stages:
            ho, wo = h // 2, w // 2
            for i in range(n_conv):
                name = f"{stage}.{i}"
                last = i == n_conv - 1
                dst = catbuf if last else pingpong[i % 2]
                dst_bs, dst_pitch = (cout, cout * F) if last else (ho * wo * cout, cout)
                # with split_acts every map of the backbone travels in the split activation format (csrc/convsg.hip)
                act = ACT_SPLIT_IN | ACT_SPLIT_OUT
                if i == 0:
                    self._conv(src, src_bs, src_pitch, name, dst, dst_bs, dst_pitch, F, h, w, cin, cout, 3, 2,
                               EPI_BIAS_BN_GELU, scale=p[f"{name}.scale"], shift=p[f"{name}.shift"], act=act)
                else:
                    self._conv(src, src_bs, src_pitch, name, dst, dst_bs, dst_pitch, F, ho, wo, cout, cout, 3, 1,
                               EPI_BIAS_BN_GELU, scale=p[f"{name}.scale"], shift=p[f"{name}.shift"], act=act)
                src, src_bs, src_pitch = dst, dst_bs, dst_pitch

    def decoder(self):
        """B0, F1, F2, F3 -> DEC; every intermediate keeps its own buffer (the training backward pass reads them)."""
        H, W, F = self.H, self.W, self.F
        IN, IO = ACT_SPLIT_IN, ACT_SPLIT_IN | ACT_SPLIT_OUT
        fold = self.fold_decoder
        def block(name, coarse, c_in, ch, cw, tmp, cat, skip, skip_c, lat, out, work, nxt=None):
            # the 1x1 output that feeds the bilinear upsampling stays float32 (the interpolation reads float32); its result
            # and everything else is written split
            if coarse is not None:
                self._conv(coarse, 0, c_in, f"{name}.u1", tmp, 0, lat, 1, 1, ch * cw, c_in, lat, 1, 1, EPI_BIAS, act=IN)
            self._up(tmp, lat, ch, cw, lat, cat, 2 * lat, out_split=True)
            self._conv(skip, 0, skip_c, f"{name}.u3", cat, 0, 2 * lat, 1, 1, 4 * ch * cw, skip_c, lat, 1, 1, EPI_BIAS, y_off=lat, act=IO)
            self._conv(cat, 0, 2 * lat, f"{name}.u4", work[0], 0, out, 1, 2 * ch, 2 * cw, 2 * lat, out, 3, 1, EPI_BIAS, act=IO)
            if fold and nxt is not None:
                # u5 and the next block's u1 as one 3x3 conv straight into that block's (float32) upsampling source
                nname, ntmp, nlat = nxt
                self._conv(work[0], 0, out, f"{name}.u5>{nname}.u1", ntmp, 0, nlat, 1, 2 * ch, 2 * cw, out, nlat, 3, 1, EPI_BIAS, act=IN)
                return None
            self._conv(work[0], 0, out, f"{name}.u5", work[1], 0, out, 1, 2 * ch, 2 * cw, out, out, 3, 1, EPI_BIAS, act=IO)
            return work[1]
        s = block("dec1", self.F3, 256 * F, H // 8, W // 8, self.T1, self.CAT1, self.F2, 128 * F, 256, 256, self.S, ("dec2", self.T2, 128))
        t = block("dec2", s, 256, H // 4, W // 4, self.T2, self.CAT2, self.F1, 64 * F, 128, 128, self.T, ("dec3", self.T3, 64))
        u = block("dec3", t, 128, H // 2, W // 2, self.T3, self.CAT3, self.B0, 32 * F, 64, 64, self.U)
        self._conv(u, 0, 64, "dec4", self.DEC, 0, 64, 1, H, W, 64, 64, 3, 1, EPI_BIAS, act=IN)      # DEC stays float32: the head gathers it
        return self.DEC

    def head(self, pc0: torch.Tensor, slot0: int = 1, slot1: int = 2, out: torch.Tensor | None = None) -> torch.Tensor:
        n = pc0.shape[0]
        p = self.p
        F = self.F
        B0, DEC = self.B0[self._sample], self.DEC[self._sample]          # this sample's images
        if self.fused_head:
            flow = out if out is not None else torch.empty((n, 3), dtype=torch.float32, device=self.device)
            if flow.shape != (n, 3) or flow.dtype != torch.float32 or not flow.is_contiguous():
                raise ValueError("out must be a contiguous (N0,3) float32 tensor")
            self.head_batch([pc0], [flow], slot0, slot1, samples=[self._sample])
            return flow
        if self.split_acts:
            raise RuntimeError("the multi-launch head reads float32 pillar images: set split_acts = False (or use the fused head)")
        st = self.lib.himo_head_gather(n, self.pid[slot0].data_ptr(), self.offsets[slot0].data_ptr(),
                                       B0.data_ptr() + 4 * 32 * slot0, B0.data_ptr() + 4 * 32 * slot1, 32 * F,
                                       DEC.data_ptr(), 64, p["head.offset.weight"].data_ptr(),
                                       p["head.offset.bias"].data_ptr(), self.hx.data_ptr(), self.rhx.data_ptr(), 192,
                                       _lib.stream_handle())
        _lib.check(st, "himo_head_gather")
        for _ in range(spec.GRU_ITERS):
            self._conv(self.hx, 0, 192, "head.gru.zr", self.zbuf, 0, 128, 1, 1, n, 192, 256, 1, 1, EPI_GRU_ZR,
                       aux_in=self.hx, aux_in_pitch=192, aux_out=self.rhx, aux_out_pitch=192, batched=False)
            self._conv(self.rhx, 0, 192, "head.gru.q", self.zbuf, 0, 128, 1, 1, n, 192, 128, 1, 1, EPI_GRU_Q,
                       aux_in=self.zbuf, aux_in_pitch=128, aux_out=self.hx, aux_out_pitch=192, batched=False)
        self._conv(self.hx, 0, 192, "head.dec1", self.y1, 0, 32, 1, 1, n, 192, 32, 1, 1, EPI_BIAS_GELU, batched=False)
        flow = out if out is not None else torch.empty((n, 3), dtype=torch.float32, device=self.device)
        if flow.shape != (n, 3) or flow.dtype != torch.float32 or not flow.is_contiguous():
            raise ValueError("out must be a contiguous (N0,3) float32 tensor")
        st = self.lib.himo_head_final(n, self.y1.data_ptr(), 32, p["head.dec2.weight"].data_ptr(),
                                      p["head.dec2.bias"].data_ptr(), self.pid[slot0].data_ptr(),
                                      self.xyz_t[slot0].data_ptr(), pc0.data_ptr(), pc0.shape[1], flow.data_ptr(),
                                      _lib.stream_handle())
        _lib.check(st, "himo_head_final")
        return flow

    def forward(self, pch1, pc0, pc1, pose_h1, pose0, pose1) -> torch.Tensor:
        dev = self.device
        to_dev = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(dev, torch.float32).contiguous()
        return self.forward_device(to_dev(pch1), to_dev(pc0), to_dev(pc1), pose_h1, pose0, pose1)

    def forward_device(self, pch1, pc0, pc1, pose_h1, pose0, pose1, out: torch.Tensor | None = None) -> torch.Tensor:
        """Sweeps already in HBM as contiguous float32 (N,>=3) tensors; ``out`` may be a slice of a batch buffer."""
        inv1 = np.linalg.inv(np.asarray(pose1, np.float64))
        T0 = inv1 @ np.asarray(pose0, np.float64)
        Th = inv1 @ np.asarray(pose_h1, np.float64)
        self._use_sample(0)
        self.pillarize_all((pch1, pc0, pc1), (Th, T0, np.eye(4)))
        self.backbone(1)
        return self.head(pc0, out=out)

    def forward_batch(self, samples, outs) -> None:
        """Up to ``max_batch`` samples -- (pch1, pc0, pc1, pose_h1, pose0, pose1) device sweeps + host poses -- through
        the network with every backbone layer ONE launch over all of them; ``outs[k]``: (N0_k,3) float32 flow buffers."""
        if not 1 <= len(samples) <= self.max_batch:
            raise ValueError(f"forward_batch takes 1..{self.max_batch} samples")
        jobs = []
        for k, (pch1, pc0, pc1, pose_h1, pose0, pose1) in enumerate(samples):
            inv1 = np.linalg.inv(np.asarray(pose1, np.float64))
            jobs.append((k, (pch1, pc0, pc1), (inv1 @ np.asarray(pose_h1, np.float64), inv1 @ np.asarray(pose0, np.float64), np.eye(4))))
        self.pillarize_many(jobs)
        self.backbone(len(samples))
        if self.fused_head:
            self.head_batch([smp[1] for smp in samples], outs)
            return
        for k, smp in enumerate(samples):
            self._use_sample(k)
            self.head(smp[1], out=outs[k])

    MAX_HEAD_SAMPLES = 16                 # kGhMaxSamples of csrc/gruhead.hip

    def head_batch(self, pc0s, outs, slot0: int = 1, slot1: int = 2, samples=None) -> None:
        """The fused head (csrc/gruhead.hip) over samples ``samples`` (default 0..len(pc0s)-1) of the activation buffers
        in ONE launch."""
        p, pk, F = self.p, self.packed, self.F
        samples = list(range(len(pc0s))) if samples is None else list(samples)
        for lo in range(0, len(pc0s), self.MAX_HEAD_SAMPLES):
            grp = range(lo, min(lo + self.MAX_HEAD_SAMPLES, len(pc0s)))
            arr = (HimoHeadSample * len(grp))()
            for j, i in enumerate(grp):
                k = samples[i]
                pc0, flow, st = pc0s[i], outs[i], self._pt[k]
                n = pc0.shape[0]
                if flow.shape != (n, 3) or flow.dtype != torch.float32 or not flow.is_contiguous():
                    raise ValueError("out must be a contiguous (N0,3) float32 tensor")
                h = arr[j]
                h.n, h.d_pid, h.d_offsets = n, st["pid"][slot0].data_ptr(), st["offsets"][slot0].data_ptr()
                h.d_img0, h.d_img1 = self.B0[k].data_ptr() + 4 * 32 * slot0, self.B0[k].data_ptr() + 4 * 32 * slot1
                h.d_dec, h.d_xyz_t = self.DEC[k].data_ptr(), st["xyz_t"][slot0].data_ptr()
                h.d_pts, h.pc_stride, h.d_flow = pc0.data_ptr(), pc0.shape[1], flow.data_ptr()
            tail = (p["head.dec2.weight"].data_ptr(), p["head.dec2.bias"].data_ptr(),
                    spec.GRU_ITERS, self.packed_format, 1 if self.split_acts else 0, self.nonfinite.data_ptr(), _lib.stream_handle())
            if self.fold_head:
                st = self.lib.himo_gru_head_batch_guarded(len(arr), ctypes.addressof(arr), 32 * F, 64, None, None,
                                                          pk["head.gru.zr.weight.h"].data_ptr(), p["head.gru.zr.bias"].data_ptr(),
                                                          pk["head.gru.q.weight.h"].data_ptr(), p["head.gru.q.bias"].data_ptr(),
                                                          pk["head.dec1.weight.h"].data_ptr(), p["head.dec1.bias"].data_ptr(), *tail)
            else:
                st = self.lib.himo_gru_head_batch_guarded(len(arr), ctypes.addressof(arr), 32 * F, 64,
                                                          p["head.offset.weight"].data_ptr(), p["head.offset.bias"].data_ptr(),
                                                          pk["head.gru.zr.weight"].data_ptr(), p["head.gru.zr.bias"].data_ptr(),
                                                          pk["head.gru.q.weight"].data_ptr(), p["head.gru.q.bias"].data_ptr(),
                                                          pk["head.dec1.weight"].data_ptr(), p["head.dec1.bias"].data_ptr(), *tail)
            _lib.check(st, "himo_gru_head_batch")

    def pillarize_all(self, sweeps, transforms):
        """The F sweeps of the current sample -> the F channel groups of its B0, sharing every launch of the stage."""
        self.pillarize_many([(self._sample, sweeps, transforms)])

    MAX_SWEEPS = 12                       # kMaxSweeps of csrc/pillar.hip

    def pillarize_many(self, jobs):
        """``jobs``: [(sample index, F sweeps, F transforms)] -- up to 12 sweeps (4 samples) share the stage's launches."""
        for _, sweeps, _ in jobs:
            for t in sweeps:
                self._reserve_points(t.shape[0])
        per_call = self.MAX_SWEEPS // self.F
        for lo in range(0, len(jobs), per_call):
            grp = jobs[lo:lo + per_call]
            arr = (HimoSweep * (len(grp) * self.F))()
            for j, (sample, sweeps, transforms) in enumerate(grp):
                st = self._pt[sample]
                for slot, (pts, T) in enumerate(zip(sweeps, transforms)):
                    w = arr[j * self.F + slot]
                    w.n, w.d_pts, w.pc_stride = pts.shape[0], pts.data_ptr(), pts.shape[1]
                    w.transform = _f32x(np.asarray(T, dtype=np.float32).reshape(-1))
                    w.d_xyz_t, w.d_pid, w.d_offsets = st["xyz_t"][slot].data_ptr(), st["pid"][slot].data_ptr(), st["offsets"][slot].data_ptr()
                    w.d_image = self.B0[sample].data_ptr() + 4 * 32 * slot
                    w.d_workspace = st["ws_slots"][slot].data_ptr()
            status = self.lib.himo_pillarize_multi_ex(len(arr), ctypes.addressof(arr), self._range, self._voxel, self._centre, self.W, self.H,
                                                      self.p["pfn.weight"].data_ptr(), self.p["pfn.scale"].data_ptr(),
                                                      self.p["pfn.shift"].data_ptr(), 32 * self.F, self._pt[0]["ws_slots"][0].numel(),
                                                      (1 if self.split_acts else 0) | (2 if self.incremental_images else 0), _lib.stream_handle())
            _lib.check(status, "himo_pillarize_multi_ex")

    def pillar_features(self, sweeps, transforms, scale: torch.Tensor, shift: torch.Tensor):
        """Training mode (csrc/pillar.hip himo_pillar_features_multi): the feature kernel ALONE for the F sweeps of the current
        sample, with per-sweep BatchNorm constants ``scale`` / ``shift`` [F][32] -- ``pillarize_all`` has just built the cell
        lists of exactly these sweeps (same tensors, same transforms)."""
        arr = (HimoSweep * self.F)()
        st = self._pt[self._sample]
        for slot, (pts, T) in enumerate(zip(sweeps, transforms)):
            w = arr[slot]
            w.n, w.d_pts, w.pc_stride = pts.shape[0], pts.data_ptr(), pts.shape[1]
            w.transform = _f32x(np.asarray(T, dtype=np.float32).reshape(-1))
            w.d_xyz_t, w.d_pid, w.d_offsets = st["xyz_t"][slot].data_ptr(), st["pid"][slot].data_ptr(), st["offsets"][slot].data_ptr()
            w.d_image = self.B0[self._sample].data_ptr() + 4 * 32 * slot
            w.d_workspace = st["ws_slots"][slot].data_ptr()
        status = self.lib.himo_pillar_features_multi(self.F, ctypes.addressof(arr), self._range, self._voxel, self._centre, self.W, self.H,
                                                     self.p["pfn.weight"].data_ptr(), scale.data_ptr(), shift.data_ptr(), 32 * self.F,
                                                     st["ws_slots"][0].numel(),
                                                     (1 if self.split_acts else 0) | (2 if self.incremental_images else 0), _lib.stream_handle())
        _lib.check(status, "himo_pillar_features_multi")

    def pillarize_into(self, slot: int, pts: torch.Tensor, transform):
        """Sweep -> channel group ``slot`` of B0 (pitch 96)."""
        n = pts.shape[0]
        self._reserve_points(n)
        arr = (HimoSweep * 1)()
        w = arr[0]
        w.n, w.d_pts, w.pc_stride = n, pts.data_ptr(), pts.shape[1]
        w.transform = _f32x(np.asarray(transform, dtype=np.float32).reshape(-1))
        w.d_xyz_t, w.d_pid, w.d_offsets = self.xyz_t[slot].data_ptr(), self.pid[slot].data_ptr(), self.offsets[slot].data_ptr()
        w.d_image = self.B0[self._sample].data_ptr() + 4 * 32 * slot
        w.d_workspace = self.ws_slots[slot].data_ptr()
        st = self.lib.himo_pillarize_multi_ex(1, ctypes.addressof(arr), self._range, self._voxel, self._centre, self.W, self.H,
                                              self.p["pfn.weight"].data_ptr(), self.p["pfn.scale"].data_ptr(),
                                              self.p["pfn.shift"].data_ptr(), 32 * self.F, self.ws_slots[slot].numel(),
                                              (1 if self.split_acts else 0) | (2 if self.incremental_images else 0), _lib.stream_handle())
        _lib.check(st, "himo_pillarize_multi_ex")


# ---- stand-alone operators (tests / experiments): the same kernels on caller-provided tensors -------------------
def conv2d_nhwc(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, stride: int = 1, epilogue: int = EPI_BIAS,
                scale: torch.Tensor | None = None, shift: torch.Tensor | None = None, precision: str = "f32",
                tile_hint: int = 0, act_layout: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
    """x [N,H,W,Cin] float32 (contiguous, device), weight [k,k,Cin,Cout] -> y [N,Ho,Wo,Cout].  ``act_layout``:
    ACT_SPLIT_IN / ACT_SPLIT_OUT -- x / y hold the split activation format (same shape, fp16 pairs; f16x2 3x3 layers only);
    ACT_ACCUMULATE with ``out`` given: out += result (bf16x2 3x3 stride-1 layers); ACT_STUFFED_2X: x [N,H/2,W/2,Cin] is read as its
    zero-stuffed [N,H,W,Cin] image (x[2i][2j] = map[i][j]; the data gradient of a stride-2 layer) -> y [N,H,W,Cout]."""
    lib = _lib.load()
    n, h, w, cin = x.shape
    k, _, _, cout = weight.shape
    stuffed = bool(act_layout & ACT_STUFFED_2X)
    if stuffed:
        h, w = 2 * h, 2 * w
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device) if out is None else out
    d = ConvDesc()
    d.x = x.data_ptr(); d.x_batch_stride = (h * w * cin) // (4 if stuffed else 1); d.x_pitch = cin
    d.w = weight.data_ptr(); d.bias = bias.data_ptr()
    d.scale = None if scale is None else scale.data_ptr(); d.shift = None if shift is None else shift.data_ptr()
    d.y = y.data_ptr(); d.y_batch_stride = ho * wo * cout; d.y_pitch = cout
    if k == 1:
        d.n, d.h, d.w_in = n, 1, h * w
    else:
        d.n, d.h, d.w_in = n, h, w
    d.cin, d.cout, d.ksize, d.stride, d.epilogue = cin, cout, k, stride, epilogue
    d.tile_hint = tile_hint
    d.act_layout = act_layout
    if precision in ("bf16x3", "f16x2", "bf16x2") and (stride == 1 or k == 3):
        fmt = {"bf16x3": 0, "f16x2": 1, "bf16x2": 2}[precision]       # bf16x2: 3x3 layers only (the training data gradients)
        pk = torch.empty(int(lib.himo_conv_packed_weight_bytes(k, cin, cout)), dtype=torch.uint8, device=x.device)
        _lib.check(lib.himo_conv_pack_weights_ex(weight.contiguous().data_ptr(), k, cin, cout, fmt, pk.data_ptr(), _lib.stream_handle()),
                   "himo_conv_pack_weights_ex")
        d.w_packed = pk.data_ptr(); d.packed_format = fmt
    _lib.check(lib.himo_conv2d(ctypes.byref(d), _lib.stream_handle()), "himo_conv2d")
    return y


def upsample2x_nhwc(x: torch.Tensor) -> torch.Tensor:
    """x [H,W,C] -> [2H,2W,C], bilinear, align_corners=True."""
    lib = _lib.load()
    h, w, c = x.shape
    y = torch.empty((2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
    _lib.check(lib.himo_upsample2x(x.data_ptr(), c, h, w, c, y.data_ptr(), c, _lib.stream_handle()), "himo_upsample2x")
    return y
