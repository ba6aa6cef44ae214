"""GPU experiment: the split activation format (csrc/convsg.hip).  A chain conv(out split) -> conv(in split) must be
bit-identical to the float32-activation f16x2 chain; then per-layer timings of both at the network's 3x3 shapes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc, ACT_SPLIT_IN, ACT_SPLIT_OUT

dev = torch.device("cuda", 0)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
# ---- correctness: two-layer chains ----
for (n, h, w, c0, c1, c2, s0) in [] if "--timing-only" in sys.argv else [(2, 37, 45, 32, 64, 64, 1), (3, 64, 64, 64, 128, 128, 1), (1, 50, 70, 128, 256, 128, 1), (2, 61, 67, 32, 64, 128, 2)]:
    x = torch.randn(n, h, w, c0, device=dev)
    w0 = torch.randn(3, 3, c0, c1, device=dev) * 0.05; b0 = torch.randn(c1, device=dev) * 0.1
    w1 = torch.randn(3, 3, c1, c2, device=dev) * 0.05; b1 = torch.randn(c2, device=dev) * 0.1
    sc = torch.rand(c1, device=dev) + 0.5; sh = torch.randn(c1, device=dev) * 0.1
    ref_mid = conv2d_nhwc(x, w0, b0, stride=s0, epilogue=1, scale=sc, shift=sh, precision="f16x2")
    ref = conv2d_nhwc(ref_mid, w1, b1, precision="f16x2")
    mid = conv2d_nhwc(x, w0, b0, stride=s0, epilogue=1, scale=sc, shift=sh, precision="f16x2", act_layout=ACT_SPLIT_OUT)
    for hint in (0, 0x1004, 0x1002, 0x1001):
        got = conv2d_nhwc(mid, w1, b1, precision="f16x2", act_layout=ACT_SPLIT_IN, tile_hint=hint)
        print(f"chain {n}x{h}x{w} {c0}->{c1}->{c2} s{s0} hint {hint:#x}: equal={torch.equal(got, ref)} maxdiff={float((got-ref).abs().max()):.3e}")
    # split -> split -> float
    mid2 = conv2d_nhwc(mid, w1[:, :, :, :c1] if c2 >= c1 else torch.cat([w1, w1], 3)[:, :, :, :c1], b0, epilogue=1, scale=sc, shift=sh, precision="f16x2", act_layout=ACT_SPLIT_IN | ACT_SPLIT_OUT)
    ref2 = conv2d_nhwc(ref_mid, w1[:, :, :, :c1] if c2 >= c1 else torch.cat([w1, w1], 3)[:, :, :, :c1], b0, epilogue=1, scale=sc, shift=sh, precision="f16x2")
    got3 = conv2d_nhwc(mid2, w1, b1, precision="f16x2", act_layout=ACT_SPLIT_IN)
    ref3 = conv2d_nhwc(ref2, w1, b1, precision="f16x2")
    dec = lambda t: t.view(torch.float16).reshape(*t.shape[:-1], t.shape[-1] // 16, 2, 16).float().sum(-2).reshape(t.shape)
    print(f"   three-layer: equal={torch.equal(got3, ref3)}  mid equal={torch.equal(dec(mid), ref_mid)} mid2 maxdiff={float((dec(mid2) - ref2).abs().max()):.3e} "
          f"nonsplit-out sg vs sp: {torch.equal(conv2d_nhwc(mid, w1, b1, epilogue=1, scale=sc[:1].expand(c2).contiguous(), shift=sh[:1].expand(c2).contiguous(), precision='f16x2', act_layout=ACT_SPLIT_IN), conv2d_nhwc(ref_mid, w1, b1, epilogue=1, scale=sc[:1].expand(c2).contiguous(), shift=sh[:1].expand(c2).contiguous(), precision='f16x2'))}")

# ---- timing ----
shapes = [(3, 256, 256, 64, 64, 1), (3, 128, 128, 128, 128, 1), (3, 64, 64, 256, 256, 1), (1, 128, 128, 256, 256, 0),
          (1, 256, 256, 128, 128, 0), (1, 512, 512, 64, 64, 0)]
for (n, h, w, ci, co, epi) in shapes:
    n *= BATCH
    x = torch.randn(n, h, w, ci, device=dev)
    wt = torch.randn(3, 3, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev); sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
    res = {}
    for tag, lay, hint in (("f32-in 4", 0, 0x1004), ("f32-in 2", 0, 0x1002), ("split-in 4", ACT_SPLIT_IN, 0x1004), ("split-in 2", ACT_SPLIT_IN, 0x1002),
                           ("split-io 4", ACT_SPLIT_IN | ACT_SPLIT_OUT, 0x1004), ("split-io 2", ACT_SPLIT_IN | ACT_SPLIT_OUT, 0x1002), ("split-out 4", ACT_SPLIT_OUT, 0x1004)):
        xin = x if not (lay & ACT_SPLIT_IN) else conv2d_nhwc(x, torch.randn(3, 3, ci, ci, device=dev) * 0.05, torch.zeros(ci, device=dev), precision="f16x2", act_layout=ACT_SPLIT_OUT)
        for _ in range(2): conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=hint, act_layout=lay)
        torch.cuda.synchronize()
        _lib.prof_start(only="conv3x3_f16x2")
        for _ in range(5): conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=hint, act_layout=lay)
        torch.cuda.synchronize()
        p = _lib.prof_stop()
        ms = min(v["min_ms"] for v in p.values())
        res[tag] = ms
    fl = 2.0 * n * h * w * ci * co * 9
    print(f"N{n} {h}x{w} {ci}->{co} epi{epi}: " + "  ".join(f"{k} {v*1e3:7.1f}us ({fl/v/1e9:5.0f}TF)" for k, v in res.items()))
