"""The FastNSF hidden-layer row GEMM ([120k x 128] x [128 x 128], bias + ReLU) per arithmetic and tile hint: is it the
arithmetic or the launch shape that bounds it?  (answer in DESIGN.md section 4, FastNSF)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc

dev = torch.device("cuda", 0)
n = 120_000
x = torch.randn(1, 1, n, 128, device=dev)
w = torch.randn(1, 1, 128, 128, device=dev) * 0.1
b = torch.zeros(128, device=dev)
for prec in ("f32", "bf16x3", "f16x2"):
    for hint in (0, (128 << 4) | 2, (128 << 4) | 1, (64 << 4) | 2, (64 << 4) | 1):
        try:
            for _ in range(3):
                conv2d_nhwc(x, w, b, epilogue=5, precision=prec, tile_hint=hint)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _lib.prof_start()
            for _ in range(10):
                conv2d_nhwc(x, w, b, epilogue=5, precision=prec, tile_hint=hint)
            torch.cuda.synchronize()
            p = _lib.prof_stop()
            k = [(name, v["avg_ms"]) for name, v in p.items() if name.startswith("conv")]
            print(f"{prec:7s} hint {hint:#06x}: {k}  -> {2.0 * n * 128 * 128 / (k[0][1] * 1e-3) / 1e12:.1f} TF, {2 * n * 128 * 4 / (k[0][1] * 1e-3) / 1e9:.0f} GB/s", flush=True)
        except Exception as e:
            print(prec, hex(hint), "failed:", e)
