"""GPU experiment: time individual conv layer shapes of the network (per-launch us and TFLOP/s)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc

dev = torch.device("cuda", 0)
PREC = sys.argv[1] if len(sys.argv) > 1 else "f32"
HINT = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # samples per launch (images = N * BATCH)
shapes = [  # (N, H, W, Cin, Cout, k, stride, epi)
    (3, 256, 256, 64, 64, 3, 1, 1), (3, 128, 128, 128, 128, 3, 1, 1), (3, 64, 64, 256, 256, 3, 1, 1),
    (1, 128, 128, 512, 256, 3, 1, 0), (1, 128, 128, 256, 256, 3, 1, 0), (1, 256, 256, 256, 128, 3, 1, 0),
    (1, 256, 256, 128, 128, 3, 1, 0), (1, 512, 512, 128, 64, 3, 1, 0), (1, 512, 512, 64, 64, 3, 1, 0),
    (3, 512, 512, 32, 64, 3, 2, 1), (1, 1, 120000, 192, 256, 1, 1, 0), (1, 1, 120000, 192, 128, 1, 1, 0),
    (1, 512, 512, 96, 64, 1, 1, 0),
]
if len(sys.argv) > 4 and sys.argv[4] == "s2":          # the three stride-2 layers
    shapes = [(3, 512, 512, 32, 64, 3, 2, 1), (3, 256, 256, 64, 128, 3, 2, 1), (3, 128, 128, 128, 256, 3, 2, 1)]   # S2SHAPES
for (n, h, w, ci, co, k, s, epi) in shapes:
    if k == 1 and h == 1:
        continue
    n *= BATCH
    x = torch.randn(n, h, w, ci, device=dev)
    wt = torch.randn(k, k, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev); sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
    for _ in range(2): conv2d_nhwc(x, wt, b, stride=s, epilogue=epi, scale=sc, shift=sh, precision=PREC, tile_hint=HINT)
    torch.cuda.synchronize()
    _lib.prof_start()
    for _ in range(5): conv2d_nhwc(x, wt, b, stride=s, epilogue=epi, scale=sc, shift=sh, precision=PREC, tile_hint=HINT)
    torch.cuda.synchronize()
    p = _lib.prof_stop()
    ms = min(v["min_ms"] for v in p.values())
    ho, wo = (h // s, w // s)
    fl = 2.0 * n * ho * wo * ci * co * k * k
    print(f"N{n} {h}x{w} {ci}->{co} k{k} s{s} epi{epi}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s  blocks={n*((ho+7)//8)*((wo+15)//16)*(co//(128 if co%128==0 else 64)) if k==3 else '-'}")
