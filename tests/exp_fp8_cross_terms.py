"""EXPERIMENT (CPU, not a test: pytest does not collect it): could the two CROSS terms of the fp16-split product run on the fp8 matrix
instructions?  x w = (xh + xl)(wh + wl) ~ xh wh + xh wl + xl wh: the cross terms are 2^-11 of the product, so they need only ~11 bits LESS
precision than it -- on gfx950 an fp8 matrix instruction does twice the work of an fp16 one, which would turn 3 units of matrix time per
product block into 2.  This emulates the arithmetic inside the CPU restatement's convolutions (oracle/seflow_oracle.py; the head's row
products stay exact) and reports the flow error against the float32 restatement on one synthetic 120k-point sample:
    f16x2      the shipped arithmetic (three fp16 products)
    e4m3 cross cross-term operands rounded to 4 significant bits (fp8 e4m3; unlimited exponent range = ideal per-tensor scaling)
    e5m2 cross ... to 3 significant bits (fp8 e5m2)
    hh only    no cross terms at all (one product): the size of what is being approximated
usage: python tests/exp_fp8_cross_terms.py [n_points]
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import torch.nn.functional as F
from himo_amd.seflow import spec
from himo_amd.synthetic import make_frame
from oracle import seflow_oracle as so

n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
torch.set_num_threads(16)
_conv2d = F.conv2d


def round_bits(t: torch.Tensor, bits: int) -> torch.Tensor:
    """round to `bits` significant bits (round to nearest even), exponent range unlimited"""
    m, e = torch.frexp(t)
    return torch.ldexp(torch.round(m * (1 << bits)) / (1 << bits), e)


def split16(t):
    h = t.to(torch.float16).to(torch.float32)
    l = (t - h).to(torch.float16).to(torch.float32)
    return h, l


def make_conv(mode):
    def conv(x, w, b=None, stride=1, padding=0):
        xh, xl = split16(x)
        wh, wl = split16(w)
        y = _conv2d(xh, wh, None, stride=stride, padding=padding)
        if mode == "f16x2":
            y = y + _conv2d(xh, wl, None, stride=stride, padding=padding) + _conv2d(xl, wh, None, stride=stride, padding=padding)
        elif mode in ("e4m3", "e5m2"):
            q = (lambda t: round_bits(t, 4)) if mode == "e4m3" else (lambda t: round_bits(t, 3))
            y = y + _conv2d(q(xh), q(wl), None, stride=stride, padding=padding) + _conv2d(q(xl), q(wh), None, stride=stride, padding=padding)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return conv


params = spec.init_params(0)
f = [make_frame(i, n_points=n_points) for i in range(3)]
args = (f[0]["pc0"], f[1]["pc0"], f[2]["pc0"], f[0]["pose0"], f[1]["pose0"], f[1]["pose1"])
ref = so.forward(params, *args)
ref = ref.numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
print(f"{n_points} points; |flow| max {np.abs(ref).max():.3f}")
for mode in ("f16x2", "e4m3", "e5m2", "hh"):
    so.F.conv2d = make_conv(mode)
    try:
        out = so.forward(params, *args)
    finally:
        so.F.conv2d = _conv2d
    out = out.numpy() if isinstance(out, torch.Tensor) else np.asarray(out)
    err = np.abs(out - ref)
    print(f"{mode:6s} max abs flow error {err.max():.3e}   mean EPE {np.linalg.norm(out - ref, axis=1).mean():.3e}   (bar: 1e-4)")
