"""Time of the three x2 bilinear upsamplings of the decoder at the bench's 16 samples per launch (float32 coarse map -> split-format
channel group of the concat buffer), and a plain fill of the same bytes for reference."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import ctypes
import torch
from himo_amd import _lib
from himo_amd.seflow import model  # noqa: F401  (registers the signatures)

dev = torch.device("cuda", 0)
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (h, w, c, pitch) in [(64, 64, 256, 512), (128, 128, 128, 256), (256, 256, 64, 128)]:
    x = torch.randn(B, h, w, c, device=dev)
    y = torch.empty(B, 2 * h, 2 * w, pitch, device=dev)
    call = lambda: _lib.check(lib.himo_upsample2x_batch_ex(B, x.data_ptr(), h * w * c, c, h, w, c, y.data_ptr(), 4 * h * w * pitch, pitch, 1,
                                                           _lib.stream_handle()), "up")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    _lib.prof_start()
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    v = [v for k, v in _lib.prof_stop().items() if "upsample" in k][0]
    wb = B * 4 * h * w * c * 4
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    z = torch.empty(wb // 4, device=dev)
    z.zero_(); torch.cuda.synchronize()
    t0.record()
    for _ in range(10):
        z.zero_()
    t1.record(); torch.cuda.synchronize()
    fill_ms = t0.elapsed_time(t1) / 10
    print(f"{h}x{w}x{c} -> x2: {v['avg_ms'] * 1e3:7.1f} us  written {wb / 1e6:6.0f} MB = {wb / v['avg_ms'] / 1e9:5.2f} TB/s   (contiguous fill of the same bytes: {fill_ms * 1e3:6.1f} us = {wb / fill_ms / 1e9:5.2f} TB/s)")
