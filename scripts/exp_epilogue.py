"""Where a block of the 3x3 split-activation kernel spends its life, per layer shape of the fp16-split network at the bench's 16
samples per launch: cycles from block start to main loop (index setup, first patch DMA, first weight fragments), in the main loop,
from the loop's end to the last store ISSUED by wave 0 (the matrix pipe draining, the epilogue's GELU / split / LDS transpose) and to
the stores ACKNOWLEDGED.  Needs the instrumented build:
    python scripts/build_epilogue_timing.py && HIMO_AMD_LIB=build/variants/epi/libhimo_amd.so python scripts/exp_epilogue.py
("give the epilogue its own numbers per variant".)  The stamps perturb the kernel a little: the launch times
printed here are this build's, beside the shipped build's from scripts/exp_layers.py."""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc, ACT_SPLIT_IN, ACT_SPLIT_OUT

dev = torch.device("cuda", 0)
lib = _lib.load()
if not hasattr(lib, "himo_exp_epi_read"):
    raise SystemExit("not the instrumented build: HIMO_AMD_LIB=build/variants/epi/libhimo_amd.so (scripts/build_epilogue_timing.py)")
lib.himo_exp_epi_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [("enc1.x", 3, 256, 256, 64, 64, 1), ("enc2.x", 3, 128, 128, 128, 128, 1), ("enc3.x", 3, 64, 64, 256, 256, 1),
          ("dec1.u4", 1, 128, 128, 512, 256, 0), ("dec1.u5>dec2.u1", 1, 128, 128, 256, 128, 0), ("dec2.u4", 1, 256, 256, 256, 128, 0),
          ("dec2.u5>dec3.u1", 1, 256, 256, 128, 64, 0), ("dec3.u4", 1, 512, 512, 128, 64, 0), ("dec3.u5|dec4", 1, 512, 512, 64, 64, 0)]
print("cycles per block (mean over the blocks of one launch): start -> loop | loop | loop end -> stores issued | -> stores acknowledged")
for name, n, h, w, ci, co, epi in shapes:
    n *= BATCH
    x = torch.randn(n, h, w, ci, device=dev)
    xin = conv2d_nhwc(x, torch.randn(3, 3, ci, ci, device=dev) * 0.05, torch.zeros(ci, device=dev), precision="f16x2", act_layout=ACT_SPLIT_OUT)
    del x
    wt = torch.randn(3, 3, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev); sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
    lay = ACT_SPLIT_IN | ACT_SPLIT_OUT
    run = lambda: conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", act_layout=lay)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.himo_exp_epi_reset()
    _lib.prof_start(only="conv3x3_f16x2")
    for _ in range(6):
        run()
    torch.cuda.synchronize()
    p = _lib.prof_stop()
    ms = sorted(v["avg_ms"] for v in p.values())[0]
    import numpy as np
    tab = np.zeros((65536, 4), dtype=np.uint64)
    lib.himo_exp_epi_read(tab.ctypes.data, 65536)
    tab = tab[tab[:, 1] > 0].astype(np.float64)             # the blocks of the last launch that reported
    blocks = max(len(tab), 1)
    pro, loop, issue, ack = (tab[:, k].mean() if len(tab) else 0.0 for k in range(4))
    life = pro + loop + ack
    print(f"{name:16s} {h}x{w} {ci:3d}->{co:3d} epi{epi}: {ms * 1e3:7.1f} us/launch  {blocks:6d} blocks  "
          f"{pro:7.0f} | {loop:8.0f} | {issue:7.0f} | {ack:7.0f}   = {100 * pro / life:4.1f} % | {100 * loop / life:4.1f} % | {100 * ack / life:4.1f} % of a block's life",
          flush=True)
    del xin
    torch.cuda.empty_cache()
