"""The linear / 1x1 weight gradient (split-bf16 operands, himo_linear_wgrad_ex flag 2) alone on the training step's shapes: the head's gate
gradients over the four stacked GRU iterations (480k rows x 192 -> 128 | 256) and the decoder's 1x1 layers.
usage (GPU box): python scripts/exp_linear_wgrad.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
import himo_amd.seflow.train  # noqa: F401  (registers the entry points)

dev = torch.device("cuda", 0)
lib = _lib.load()
torch.manual_seed(0)
shapes = [("head q", 480_000, 192, 128), ("head zr", 480_000, 192, 256), ("head out", 120_000, 192, 32), ("dec u3 512^2", 262_144, 64, 64),
          ("dec u3 256^2", 65_536, 128, 128), ("dec u3 128^2", 16_384, 256, 256), ("dec u1", 16_384, 512, 256)]
for name, n, ci, co in shapes:
    x = torch.randn(n, ci, device=dev)
    dz = torch.randn(n, co, device=dev) * 1e-3
    dw = torch.empty(ci, co, device=dev); db = torch.empty(co, device=dev)
    ws = torch.empty(int(lib.himo_wgrad_workspace_bytes_ex(n, ci, co)), dtype=torch.uint8, device=dev)
    call = lambda: _lib.check(lib.himo_linear_wgrad_ex(n, x.data_ptr(), ci, ci, dz.data_ptr(), co, co, dw.data_ptr(), db.data_ptr(), 2,
                                                       ws.data_ptr(), ws.numel(), _lib.stream_handle()))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        call()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 10 * 1e3
    gb = n * (ci + co) * 4 / 1e9
    ref = x.double().T @ dz.double()
    rel = float(((dw.double() - ref).abs().max() / ref.abs().max()).item())
    relb = float(((db.double() - dz.double().sum(0)).abs().max() / dz.double().sum(0).abs().max()).item())
    bits = int(dw.view(torch.int32).to(torch.int64).sum().item()) ^ int(db.view(torch.int32).to(torch.int64).sum().item())
    print(f"{name:14s} n {n:7d} {ci:3d}->{co:3d}: {us:7.1f} us (whole call)  {2.0 * n * ci * co / us / 1e6:6.1f} TF  operands once {gb:.3f} GB = {gb / us * 1e6:6.0f} GB/s  bits {bits & 0xffffffff:08x}  vs float64: dW {rel:.1e} db {relb:.1e}")
