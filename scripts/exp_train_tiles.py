"""Would per-shape tile tuning pay in the TRAINING step?  Its convolutions (forward: two-term fp16 split on float32 maps; data
gradient: two-term bf16 split) run the library's heuristic variant (SeFlowTrainer: autotune off).  Per training shape and format:
time of every pinned variant of the weights-from-L2 structure (tile_hint 0x1000 | rows per wave) against hint 0, and whether the
variants return the same bits.  python scripts/exp_train_tiles.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc

dev = torch.device("cuda", 0)
# (name, images per launch, H, W, Cin, Cout (forward roles), stride, layers of this shape per step)
shapes = [("enc1.0", 3, 512, 512, 32, 64, 2, 1), ("enc1.x", 3, 256, 256, 64, 64, 1, 3), ("enc2.0", 3, 256, 256, 64, 128, 2, 1),
          ("enc2.x", 3, 128, 128, 128, 128, 1, 5), ("enc3.0", 3, 128, 128, 128, 256, 2, 1), ("enc3.x", 3, 64, 64, 256, 256, 1, 5),
          ("dec1.u4", 1, 128, 128, 512, 256, 1, 1), ("dec1.u5", 1, 128, 128, 256, 256, 1, 1), ("dec2.u4", 1, 256, 256, 256, 128, 1, 1),
          ("dec2.u5", 1, 256, 256, 128, 128, 1, 1), ("dec3.u4", 1, 512, 512, 128, 64, 1, 1), ("dec3.u5|dec4", 1, 512, 512, 64, 64, 1, 2)]
HINTS = [0, 0x1004, 0x1002, 0x1001, 0x1006, 0x100A, 0x41, 0x42, 0x81, 0x82]


def timed(call, reps=8):
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tot_default, tot_best = 0.0, 0.0
for name, n, h, w, ci, co, st, reps in shapes:
    for role, prec, cin, cout, stride in (("fwd", "f16x2", ci, co, st), ("dgrad", "bf16x2", co, ci, 1)):
        x = torch.randn(n, h, w, cin, device=dev) * (1.0 if role == "fwd" else 1e-3)
        wt = torch.randn(3, 3, cin, cout, device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        ho, wo = (h // 2, w // 2) if stride == 2 else (h, w)
        outs, times = {}, {}
        for hint in HINTS:
            y = torch.empty(n, ho, wo, cout, device=dev)
            try:
                times[hint] = timed(lambda: conv2d_nhwc(x, wt, bias, stride=stride, precision=prec, tile_hint=hint, out=y))
                outs[hint] = y
            except Exception as e:                       # a variant the shape does not admit
                times[hint] = float("inf")
        same = all(torch.equal(outs[0], o) for o in outs.values())
        best = min(times, key=times.get)
        tot_default += times[0] * reps; tot_best += times[best] * reps
        print(f"{name:13s} {role:5s} {prec:6s} N{n} {h}x{w} {cin:3d}->{cout:3d} s{stride}: " +
              "  ".join(f"{hint:#06x} {t:7.1f}" for hint, t in times.items()) + f"   best {best:#06x}  same bits: {same}", flush=True)
        del x, outs
        torch.cuda.empty_cache()
print(f"sum over a step's 3x3 forward + data-gradient convolutions: heuristic {tot_default / 1e3:.3f} ms, best pinned variant per shape {tot_best / 1e3:.3f} ms")
