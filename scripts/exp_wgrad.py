"""3x3 weight gradient of the training step's layer shapes: float32 matrix instructions (flag 0) vs split-bf16 operands (flag 2),
one 120k-point sample's images (3 frames for the encoder layers)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
import himo_amd.seflow.train  # noqa: F401  (registers the training entry points)

dev = torch.device("cuda", 0)
lib = _lib.load()
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # samples per pass (SeFlowTrainer(batch=...)): n images x BATCH
shapes = [("enc1.x", 3, 256, 256, 64, 64, 3), ("enc2.x", 3, 128, 128, 128, 128, 5), ("enc3.x", 3, 64, 64, 256, 256, 5),
          ("dec1.u4", 1, 128, 128, 512, 256, 1), ("dec1.u5", 1, 128, 128, 256, 256, 1), ("dec2.u4", 1, 256, 256, 256, 128, 1),
          ("dec2.u5", 1, 256, 256, 128, 128, 1), ("dec3.u4", 1, 512, 512, 128, 64, 1), ("dec3.u5|dec4", 1, 512, 512, 64, 64, 2)]
tot = {0: 0.0, 2: 0.0}
for name, n, h, w, ci, co, reps in shapes:
    n *= BATCH
    x = torch.randn(n, h, w, ci, device=dev)
    dy = torch.randn(n, h, w, co, device=dev) * 1e-3
    dw = torch.empty(3, 3, ci, co, device=dev)
    ws = torch.empty(int(lib.himo_conv_wgrad_batch_workspace_bytes(n, h, w, ci, co, 1)), dtype=torch.uint8, device=dev)
    line = f"{name:14s} N{n} {h}x{w} {ci:3d}->{co:3d}:"
    outs = {}
    for flag in (0, 2):
        call = lambda: _lib.check(lib.himo_conv3x3_wgrad_batch(n, x.data_ptr(), h * w * ci, ci, h, w, ci, dy.data_ptr(), h * w * co, co, co, 1,
                                                              dw.data_ptr(), flag, ws.data_ptr(), ws.numel(), _lib.stream_handle()))
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        _lib.prof_start(only="conv_wgrad_tiled")
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        ms = list(_lib.prof_stop().values())[0]["avg_ms"]
        outs[flag] = dw.clone()
        tot[flag] += ms * reps
        line += f"  flag {flag}: {ms * 1e3:8.1f} us {2.0 * n * h * w * ci * co * 9 / ms / 1e9:6.1f} TF"
    rel = ((outs[2] - outs[0]).abs().max() / outs[0].abs().max()).item()
    print(line + f"  max rel diff {rel:.2e}", flush=True)
print(f"sum over a step's stride-1 layers, {BATCH} sample(s) per pass: float32 {tot[0] / BATCH:.3f} ms, split-bf16 {tot[2] / BATCH:.3f} ms per sample")
