"""Architecture + parameter table of the scene-flow network this package runs (stage a10).

PARITY UNPINNED.  The reference's network lives in the OpenSceneFlow submodule, which is EMPTY in
/root/reference, so nothing below can cite a reference line.  The only
in-tree facts are: ``model=deflowpp``, ``voxel_size=[0.2, 0.2, 6]``,
``point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3]`` (=> a 512 x 512 x 1 pillar grid),
``num_frames=3`` (assets/slurm/ssl-train-av2.sh:32-33), and the output contract: an (N,3) float32 flow
INCLUDING ego motion, row-aligned with ``pc0`` (save_zip.py:117, tools/test/score.py:583).
Everything else is this build's own specification, written from the public description of the
DeFlow / SeFlow family (pillar embedder -> shared 2-D conv U-Net encoder per frame -> decoder with
skips -> per-point GRU refinement head).  The parity target is this build's own CPU restatement,
oracle/seflow_oracle.py, which consumes the same parameter dict.

Data flow for one sample (pch1 = one history sweep, pc0, pc1; all float32 xyz):
  0. every sweep is brought into pc1's frame: p' = R p + t with inv(pose1) @ pose_k, float32;
     pose_flow = pc0' - pc0.
  1. dynamic pillarisation on the 512 x 512 grid (points outside the range are dropped),
  2. pillar features: per point [xyz, xyz - pillar mean, xyz - pillar centre] (9) -> Linear(9,32, no bias)
     -> BatchNorm(eval) -> ReLU -> mean over the pillar -> 32-channel BEV image (zeros elsewhere),
  3. encoder (shared weights, run on each of the 3 images): three stages of 3x3 conv+BN+GELU blocks,
     stride 2 at the head of each stage: 32->64 (4 convs, 256^2), 64->128 (6 convs, 128^2), 128->256 (6 convs, 64^2),
  4. the three frames' maps are concatenated per scale; decoder: three UpsampleSkip blocks
     (1x1 conv on the coarse map -> bilinear x2 (align_corners) ; 1x1 conv on the skip ; concat ;
     3x3 conv ; 3x3 conv ; no activations) then a final 3x3 conv -> 64 channels at 512^2,
  5. head, for every in-range pc0 point: hidden h = [pc0 image, pc1 image, decoder map] at its pillar (128),
     x = Linear(3,64)(point - pillar centre); 4 iterations of a GRU cell (1x1 convs == per-point
     linears); flow = Linear(192,32) -> GELU -> Linear(32,3) on [h, x],
  6. output: flow_out = pose_flow + network flow for in-range points, pose_flow alone otherwise.

Training mode (BASELINE config 5; the reference's job trains from scratch, assets/slurm/ssl-train-av2.sh:31-34): every
BatchNorm follows torch.nn.BatchNorm -- batch statistics normalise (biased variance), gamma / beta are trainable, the running
mean / variance follow with momentum 0.1 (unbiased variance).  The batch of a layer is what ONE forward call puts through
it: the pillar net is called once per sweep (statistics over that sweep's in-range points; three running-statistics updates
per sample, in the order history, pc0, pc1), the encoder once per sample on its F = 3 images stacked as the batch
(statistics over 3 x H x W positions).  A rank processes its samples one at a time and accumulates their gradients
(himo_amd/seflow/train.py train_batch): per-GPU statistics, un-synchronised across ranks like DDP's default.
"""
from __future__ import annotations

import math

import numpy as np

VOXEL_SIZE = (0.2, 0.2, 6.0)
POINT_CLOUD_RANGE = (-51.2, -51.2, -3.0, 51.2, 51.2, 3.0)
GRID = (512, 512)            # (H = y cells, W = x cells)
NUM_FRAMES = 3               # pch1, pc0, pc1
PILLAR_CH = 32
GRU_ITERS = 4
BN_EPS_PFN = 1e-3
BN_EPS = 1e-5

# encoder: (name, cin, cout, stride)
ENCODER = (
    [("enc1.0", 32, 64, 2)] + [(f"enc1.{i}", 64, 64, 1) for i in range(1, 4)]
    + [("enc2.0", 64, 128, 2)] + [(f"enc2.{i}", 128, 128, 1) for i in range(1, 6)]
    + [("enc3.0", 128, 256, 2)] + [(f"enc3.{i}", 256, 256, 1) for i in range(1, 6)]
)
# decoder blocks: (name, coarse_in, skip_in, latent, out)
DECODER = (
    ("dec1", 256 * NUM_FRAMES, 128 * NUM_FRAMES, 256, 256),   # 64^2 -> 128^2
    ("dec2", 256, 64 * NUM_FRAMES, 128, 128),                 # 128^2 -> 256^2
    ("dec3", 128, PILLAR_CH * NUM_FRAMES, 64, 64),            # 256^2 -> 512^2
)
DEC_OUT = 64
HIDDEN = 2 * PILLAR_CH + DEC_OUT     # 128
XDIM = 64


def param_shapes() -> dict:
    """name -> shape.  Conv weights are stored [kh, kw, cin, cout] (NHWC / channels-last friendly),
    linears [in, out]."""
    s = {"pfn.weight": (9, PILLAR_CH), "pfn.bn.gamma": (PILLAR_CH,), "pfn.bn.beta": (PILLAR_CH,),
         "pfn.bn.mean": (PILLAR_CH,), "pfn.bn.var": (PILLAR_CH,)}
    for name, cin, cout, _ in ENCODER:
        s[f"{name}.weight"] = (3, 3, cin, cout)
        s[f"{name}.bias"] = (cout,)
        for k in ("gamma", "beta", "mean", "var"):
            s[f"{name}.bn.{k}"] = (cout,)
    for name, cin, skip, lat, out in DECODER:
        s[f"{name}.u1.weight"] = (1, 1, cin, lat); s[f"{name}.u1.bias"] = (lat,)
        s[f"{name}.u3.weight"] = (1, 1, skip, lat); s[f"{name}.u3.bias"] = (lat,)
        s[f"{name}.u4.weight"] = (3, 3, 2 * lat, out); s[f"{name}.u4.bias"] = (out,)
        s[f"{name}.u5.weight"] = (3, 3, out, out); s[f"{name}.u5.bias"] = (out,)
    s["dec4.weight"] = (3, 3, DEC_OUT, DEC_OUT); s["dec4.bias"] = (DEC_OUT,)
    s["head.offset.weight"] = (3, XDIM); s["head.offset.bias"] = (XDIM,)
    for g in ("z", "r", "q"):
        s[f"head.gru.{g}.weight"] = (HIDDEN + XDIM, HIDDEN); s[f"head.gru.{g}.bias"] = (HIDDEN,)
    s["head.dec1.weight"] = (HIDDEN + XDIM, 32); s["head.dec1.bias"] = (32,)
    s["head.dec2.weight"] = (32, 3); s["head.dec2.bias"] = (3,)
    return s


def init_params(seed: int = 0, fresh_bn: bool = False) -> dict:
    """Random-init float32 parameters (He-uniform fan-in bounds like torch's defaults; BN with
    non-trivial running statistics so that the BN arithmetic is exercised).  ``fresh_bn``: BatchNorm as a
    from-scratch training job starts it (torch.nn.BatchNorm's reset: gamma 1, beta 0, running mean 0, running var 1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_shapes().items():
        if name.endswith(".weight"):
            fan_in = int(np.prod(shape[:-1]))
            bound = 1.0 / math.sqrt(fan_in)
            # gain sqrt(3): keeps activations O(1) through 20 GELU/conv layers with random weights
            out[name] = rng.uniform(-bound * math.sqrt(3.0), bound * math.sqrt(3.0), shape).astype(np.float32)
        elif name.endswith(".bias"):
            out[name] = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
        elif name.endswith(".gamma"):
            out[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif name.endswith(".beta"):
            out[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif name.endswith(".mean"):
            out[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif name.endswith(".var"):
            out[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(name)
    if fresh_bn:
        for name, a in out.items():
            if ".bn." in name:
                a[...] = 1.0 if name.endswith((".gamma", ".var")) else 0.0
    return out


def conv_flops() -> float:
    """Multiply-add count x2 of all convolutions for ONE sample (3 frames), for the MFMA roofline."""
    H, W = GRID
    total = 0.0
    h, w = H, W
    for name, cin, cout, stride in ENCODER:
        if stride == 2:
            h, w = h // 2, w // 2
        total += NUM_FRAMES * 2.0 * h * w * cin * cout * 9
    size = {"dec1": (H // 8, W // 8), "dec2": (H // 4, W // 4), "dec3": (H // 2, W // 2)}
    for name, cin, skip, lat, out in DECODER:
        ch, cw = size[name]
        total += 2.0 * ch * cw * cin * lat                       # u1 at the coarse size
        total += 2.0 * (2 * ch) * (2 * cw) * skip * lat          # u3
        total += 2.0 * (2 * ch) * (2 * cw) * (2 * lat) * out * 9   # u4
        total += 2.0 * (2 * ch) * (2 * cw) * out * out * 9        # u5
    total += 2.0 * H * W * DEC_OUT * DEC_OUT * 9
    return total


def conv3x3_flops(folded: bool = False) -> float:
    """Flops (2*M*N*K) of the 20 stride-1 3x3 convolutions of one forward: 13 encoder layers x 3 frames, the two
    3x3 convs of each decoder block and the final 3x3 conv -- the launches named conv3x3_mfma_kernel.
    ``folded``: the graph the inference engine executes, where dec1.u5 / dec2.u5 absorb the next block's 1x1 conv and
    produce that block's latent width (half their own)."""
    H, W = GRID
    total = 0.0
    h, w = H, W
    for name, cin, cout, stride in ENCODER:
        if stride == 2:
            h, w = h // 2, w // 2
        else:
            total += NUM_FRAMES * 2.0 * h * w * cin * cout * 9
    size = {"dec1": (H // 4, W // 4), "dec2": (H // 2, W // 2), "dec3": (H, W)}
    for i, (name, cin, skip, lat, out) in enumerate(DECODER):
        oh, ow = size[name]
        u5_out = DECODER[i + 1][3] if folded and i + 1 < len(DECODER) else out
        total += 2.0 * oh * ow * (2 * lat) * out * 9 + 2.0 * oh * ow * out * u5_out * 9
    total += 2.0 * H * W * DEC_OUT * DEC_OUT * 9
    return total


def head_flops_per_point() -> float:
    per_iter = 3 * 2.0 * (HIDDEN + XDIM) * HIDDEN
    return GRU_ITERS * per_iter + 2.0 * 3 * XDIM + 2.0 * (HIDDEN + XDIM) * 32 + 2.0 * 32 * 3


def algorithmic_bytes_per_frame(n_points: int, folded: bool = True) -> dict:
    """HBM bytes one frame (3 sweeps -> flow -> comp_dis) has to move when every operator reads its inputs and writes its
    outputs exactly once, 4 bytes per feature-map value (float32 or the fp16 split pair), weights not counted (L2-resident).
    The whole-step HBM roofline of bench.py (``roofline.step_hbm``) prices the measured step time against this figure.
    ``folded``: the executed inference graph (dec1.u5 / dec2.u5 absorb the next block's 1x1 convolution)."""
    H, W = GRID
    px = lambda d: (H // d) * (W // d) * 4.0
    parts = {"point_io": n_points * (NUM_FRAMES * 16.0 + 12.0),                   # xyzi rows of 3 sweeps in, flow out
             "comp_dis": n_points * 44.0,                                          # xyzi 16 + flow 12 + dt 4 + comp_dis 12
             "pillar_images": NUM_FRAMES * px(1) * PILLAR_CH}
    enc, d = 0.0, 1
    for name, cin, cout, stride in ENCODER:
        enc += px(d) * cin                                                         # input at the input resolution
        if stride == 2:
            d *= 2
        enc += px(d) * cout
    parts["encoder"] = NUM_FRAMES * enc
    dec = 0.0
    res = {"dec1": 8, "dec2": 4, "dec3": 2}
    for i, (name, cin, skip, lat, out) in enumerate(DECODER):
        dc = res[name]                                                              # coarse resolution divisor; output at dc // 2
        if not (folded and i > 0):
            dec += px(dc) * (cin + lat)                                             # u1 (folded into the previous u5 otherwise)
        dec += px(dc) * lat + px(dc // 2) * lat                                     # bilinear x2
        dec += px(dc // 2) * (skip + lat)                                           # u3
        dec += px(dc // 2) * (2 * lat + out)                                        # u4 on the concat
        u5_out = DECODER[i + 1][3] if folded and i + 1 < len(DECODER) else out
        dec += px(dc // 2) * (out + u5_out)                                         # u5
    dec += px(1) * 2 * DEC_OUT                                                      # dec4
    parts["decoder"] = dec
    parts["head_gather"] = n_points * HIDDEN * 4.0
    parts["total"] = sum(parts.values())
    return parts
