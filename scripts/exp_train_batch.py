"""What would B samples per launch buy the training step's convolutions?  Per layer shape of the training graph: the forward
convolution (two-term fp16 split on float32 activations), the data gradient (two-term bf16 split) and the 3x3 weight gradient
(split-bf16 operands), timed at one sample's images per launch (what SeFlowTrainer.train_batch does today: samples strictly one
after another) and at B samples' images per launch.  Times are per SAMPLE, whole call (every kernel the call launches), HIP events.
python scripts/exp_train_batch.py [B]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc, ACT_STUFFED_2X
from himo_amd.seflow.train import SeFlowTrainer
import himo_amd.seflow.train  # noqa: F401  (registers the training entry points)

dev = torch.device("cuda", 0)
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# (name, images per sample, H, W, Cin, Cout, stride, layers of this shape per step)
shapes = [("enc1.0", 3, 512, 512, 32, 64, 2, 1), ("enc1.x", 3, 256, 256, 64, 64, 1, 3), ("enc2.0", 3, 256, 256, 64, 128, 2, 1),
          ("enc2.x", 3, 128, 128, 128, 128, 1, 5), ("enc3.0", 3, 128, 128, 128, 256, 2, 1), ("enc3.x", 3, 64, 64, 256, 256, 1, 5),
          ("dec1.u4", 1, 128, 128, 512, 256, 1, 1), ("dec1.u5", 1, 128, 128, 256, 256, 1, 1), ("dec2.u4", 1, 256, 256, 256, 128, 1, 1),
          ("dec2.u5", 1, 256, 256, 128, 128, 1, 1), ("dec3.u4", 1, 512, 512, 128, 64, 1, 1), ("dec3.u5|dec4", 1, 512, 512, 64, 64, 1, 2)]


def timed(call, reps=6):
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = {}
print(f"per-sample microseconds at 1 sample per launch -> at {B} samples per launch")
for name, n1, h, w, ci, co, st, reps in shapes:
    ho, wo = (h // 2, w // 2) if st == 2 else (h, w)
    line = f"{name:13s} {h}x{w} {ci:3d}->{co:3d} s{st}:"
    for b in (1, B):
        n = n1 * b
        x = torch.randn(n, h, w, ci, device=dev)
        wt = torch.randn(3, 3, ci, co, device=dev) * 0.05
        bias = torch.zeros(co, device=dev)
        y = torch.empty(n, ho, wo, co, device=dev)
        def best(call):                                   # the trainer picks the fastest tile variant per shape (train.py _tune_tile)
            ts = []
            for hint in SeFlowTrainer.TILE_HINTS:
                try:
                    ts.append(timed(lambda: call(hint)))
                except Exception:
                    pass                                     # a variant the shape does not admit
            return min(ts)
        fwd = best(lambda hint: conv2d_nhwc(x, wt, bias, stride=st, precision="f16x2", out=y, tile_hint=hint)) / b
        # data gradient: stride 1 = the same convolution with mirrored taps (Cout -> Cin); stride 2 = the same kernel reading the compact
        # dY as its zero-stuffed image at the input's resolution (train.py backward)
        dy = torch.randn(n, h, w, co, device=dev) * 1e-3
        wf = torch.randn(3, 3, co, ci, device=dev) * 0.05
        bz = torch.zeros(ci, device=dev)
        dx = torch.empty(n, h, w, ci, device=dev)
        dys = torch.randn(n, ho, wo, co, device=dev) * 1e-3
        if st == 2:                                       # the compact gradient map read as its zero-stuffed image (HIMO_ACT_STUFFED_2X)
            dgr = best(lambda hint: conv2d_nhwc(dys, wf, bz, precision="bf16x2", out=dx, tile_hint=hint, act_layout=ACT_STUFFED_2X)) / b
        else:
            dgr = best(lambda hint: conv2d_nhwc(dy, wf, bz, precision="bf16x2", out=dx, tile_hint=hint)) / b
        dw = torch.empty(3, 3, ci, co, device=dev)
        ws = torch.empty(int(lib.himo_conv_wgrad_batch_workspace_bytes(n, h, w, ci, co, st)) + 64, dtype=torch.uint8, device=dev)
        wg = timed(lambda: _lib.check(lib.himo_conv3x3_wgrad_batch(n, x.data_ptr(), h * w * ci, ci, h, w, ci, dys.data_ptr(), ho * wo * co, co, co, st,
                                                                   dw.data_ptr(), 2, ws.data_ptr(), ws.numel(), _lib.stream_handle()))) / b
        for k, v in (("fwd", fwd), ("dgrad", dgr), ("wgrad", wg)):
            tot[(k, b)] = tot.get((k, b), 0.0) + v * reps
        line += f"   [{b}] fwd {fwd * 1e3:7.1f} dgrad {dgr * 1e3:7.1f} wgrad {wg * 1e3:7.1f}"
        del x, y, dy, dx, dys, ws
        torch.cuda.empty_cache()
    print(line, flush=True)
for k in ("fwd", "dgrad", "wgrad"):
    print(f"sum over a step's 3x3 layers, {k:5s}: {tot[(k, 1)]:.3f} ms per sample at 1 per launch, {tot[(k, B)]:.3f} at {B} per launch")
s1, sb = (sum(tot[(k, b)] for k in ("fwd", "dgrad", "wgrad")) for b in (1, B))
print(f"all three: {s1:.3f} -> {sb:.3f} ms per sample (the training step is ~8.6 ms per sample)")
