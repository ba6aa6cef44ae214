"""Per-layer time of the fp16-split network's stride-1 3x3 layer shapes in the split activation format (csrc/convsg.hip), at
the bench's 16 samples per launch: the table behind DESIGN.md's per-layer rates, and the A/B harness for kernel variants
(HIMO_AMD_LIB=build/variants/<name>/libhimo_amd.so python scripts/exp_layers.py [samples] [hint])."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc, ACT_SPLIT_IN, ACT_SPLIT_OUT

dev = torch.device("cuda", 0)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
HINT = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
torch.manual_seed(0)
# (name, images per sample, H, W, Cin, Cout, epilogue, layers of this shape per forward)
shapes = [("enc1.x", 3, 256, 256, 64, 64, 1, 3), ("enc2.x", 3, 128, 128, 128, 128, 1, 5), ("enc3.x", 3, 64, 64, 256, 256, 1, 5),
          ("dec1.u4", 1, 128, 128, 512, 256, 0, 1), ("dec1.u5>dec2.u1", 1, 128, 128, 256, 128, 0, 1), ("dec2.u4", 1, 256, 256, 256, 128, 0, 1),
          ("dec2.u5>dec3.u1", 1, 256, 256, 128, 64, 0, 1), ("dec3.u4", 1, 512, 512, 128, 64, 0, 1), ("dec3.u5|dec4", 1, 512, 512, 64, 64, 0, 2)]
# (the two folded decoder joints write float32 -- they feed the bilinear upsampling; here they are timed with split output like the rest)
total = 0.0
for (name, n, h, w, ci, co, epi, reps) in shapes:
    n *= BATCH
    x = torch.randn(n, h, w, ci, device=dev)
    xin = conv2d_nhwc(x, torch.randn(3, 3, ci, ci, device=dev) * 0.05, torch.zeros(ci, device=dev), precision="f16x2", act_layout=ACT_SPLIT_OUT)
    del x
    wt = torch.randn(3, 3, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev); sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
    lay = ACT_SPLIT_IN | ACT_SPLIT_OUT
    for _ in range(2):
        conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=HINT, act_layout=lay)
    torch.cuda.synchronize()
    _lib.prof_start(only="conv3x3_f16x2")
    for _ in range(6):
        conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=HINT, act_layout=lay)
    torch.cuda.synchronize()
    p = _lib.prof_stop()
    ms = sorted(v["avg_ms"] for v in p.values())[0]
    fl = 2.0 * n * h * w * ci * co * 9
    total += ms * reps
    same = ""
    if HINT:                                   # a pinned variant must reproduce the default variant's bits
        ref = conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=0, act_layout=lay)
        got = conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=HINT, act_layout=lay)
        same = f"  bit-identical to hint 0: {torch.equal(ref, got)}"
        del ref, got
    print(f"{name:14s} N{n:3d} {h}x{w} {ci:3d}->{co:3d} epi{epi}: {ms*1e3:8.1f} us  {fl/ms/1e9:5.0f} TF f32-eq  x{reps}{same}")
    del xin
    torch.cuda.empty_cache()
print(f"sum over the 20 layers of a forward: {total:.3f} ms per {BATCH} samples")
