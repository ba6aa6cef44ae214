import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd.seflow.model import conv2d_nhwc, ACT_SPLIT_IN, ACT_SPLIT_OUT
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n, h, w, c0, c1 = 2, 37, 45, 32, 64
x = torch.randn(n, h, w, c0, device=dev)
w0 = torch.randn(3, 3, c0, c1, device=dev) * 0.05; b0 = torch.randn(c1, device=dev) * 0.1
w1 = torch.randn(3, 3, c1, c1, device=dev) * 0.05; b1 = torch.randn(c1, device=dev) * 0.1
sc = torch.rand(c1, device=dev) + 0.5; sh = torch.randn(c1, device=dev) * 0.1
def planes(t):
    v = t.view(torch.float16).reshape(*t.shape[:-1], t.shape[-1] // 16, 2, 16)
    return v[..., 0, :].reshape(t.shape), v[..., 1, :].reshape(t.shape)
def split(t):
    hh = t.half(); return hh, (t - hh.float()).half()
for epi in (0, 1):
    ref_mid = conv2d_nhwc(x, w0, b0, epilogue=1, scale=sc, shift=sh, precision="f16x2")
    mid = conv2d_nhwc(x, w0, b0, epilogue=1, scale=sc, shift=sh, precision="f16x2", act_layout=ACT_SPLIT_OUT)
    hm, lm = planes(mid); hr, lr = split(ref_mid)
    print("layer1 sp split-out: h eq", torch.equal(hm, hr), "l eq", torch.equal(lm, lr))
    for hint in (0x1002, 0x1001):
        ref2 = conv2d_nhwc(ref_mid, w1, b1, epilogue=epi, scale=sc, shift=sh, precision="f16x2")
        f2 = conv2d_nhwc(mid, w1, b1, epilogue=epi, scale=sc, shift=sh, precision="f16x2", act_layout=ACT_SPLIT_IN, tile_hint=hint)
        m2 = conv2d_nhwc(mid, w1, b1, epilogue=epi, scale=sc, shift=sh, precision="f16x2", act_layout=ACT_SPLIT_IN | ACT_SPLIT_OUT, tile_hint=hint)
        h2, l2 = planes(m2); hr2, lr2 = split(ref2)
        bad = (h2 != hr2) | (l2 != lr2)
        print(f"epi {epi} hint {hint:#x}: f32-out eq {torch.equal(f2, ref2)}  split-out h eq {torch.equal(h2, hr2)} l eq {torch.equal(l2, lr2)} bad {int(bad.sum())} of {bad.numel()}")
        if bad.any():
            idx = bad.nonzero()[:8]
            print(idx.tolist())
            i = tuple(idx[0].tolist()); print(float(ref2[i]), float(h2[i]), float(l2[i]), float(hr2[i]), float(lr2[i]))
