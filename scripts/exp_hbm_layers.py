"""Per-layer time and HBM rate of the network's bandwidth-side layers in the fp16-split network at the bench's 16 samples per launch:
the three stride-2 3x3 layers and the four 1x1 layers (csrc/convsg.hip).  bytes = input + output, 4 B per value, read / written once."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc, ACT_SPLIT_IN, ACT_SPLIT_OUT

dev = torch.device("cuda", 0)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
# (name, images per sample, H, W, Cin, Cout, ksize, stride, epilogue, split output)
shapes = [("enc1.0 s2", 3, 512, 512, 32, 64, 3, 2, 1, True), ("enc2.0 s2", 3, 256, 256, 64, 128, 3, 2, 1, True), ("enc3.0 s2", 3, 128, 128, 128, 256, 3, 2, 1, True),
          ("dec1.u1 1x1", 1, 64, 64, 768, 256, 1, 1, 0, False), ("dec1.u3 1x1", 1, 128, 128, 384, 256, 1, 1, 0, True),
          ("dec2.u3 1x1", 1, 256, 256, 192, 128, 1, 1, 0, True), ("dec3.u3 1x1", 1, 512, 512, 96, 64, 1, 1, 0, True)]
total = 0.0
for (name, n, h, w, ci, co, k, st, epi, osplit) in shapes:
    n *= BATCH
    x = torch.randn(n, h, w, ci, device=dev)
    # split-format input: any 3x3 convolution that writes the split format (values are irrelevant here)
    xin = conv2d_nhwc(x, torch.randn(3, 3, ci, ci, device=dev) * 0.05, torch.zeros(ci, device=dev), precision="f16x2", act_layout=ACT_SPLIT_OUT)
    del x
    wt = torch.randn(k, k, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev); sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
    lay = ACT_SPLIT_IN | (ACT_SPLIT_OUT if osplit else 0)
    for _ in range(2):
        y = conv2d_nhwc(xin, wt, b, stride=st, epilogue=epi, scale=sc, shift=sh, precision="f16x2", act_layout=lay)
    torch.cuda.synchronize()
    _lib.prof_start()
    for _ in range(6):
        y = conv2d_nhwc(xin, wt, b, stride=st, epilogue=epi, scale=sc, shift=sh, precision="f16x2", act_layout=lay)
    torch.cuda.synchronize()
    p = {kk: v for kk, v in _lib.prof_stop().items() if "conv" in kk}
    kname, v = sorted(p.items(), key=lambda kv: -kv[1]["total_ms"])[0]
    ms = v["avg_ms"]
    byt = 4.0 * (xin.numel() + y.numel())
    fl = 2.0 * y.numel() * ci * k * k
    total += ms
    print(f"{name:12s} N{n:3d} {h}x{w} {ci:3d}->{co:3d}: {ms*1e3:7.1f} us  {byt/1e6:7.0f} MB  {byt/ms/1e9:5.2f} TB/s  {fl/ms/1e9:5.0f} TF f32-eq  ({kname})")
    del xin, y
    torch.cuda.empty_cache()
print(f"sum: {total:.3f} ms per {BATCH} samples")
