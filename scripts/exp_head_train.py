"""Time of the training head's fused forward (himo_gru_head_train) on a 120k-point sweep; library variants built with
-DHIMO_EXP_SVMASK=<mask> drop groups of saves (results of those are incomplete: timing only)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from himo_amd import _lib
from himo_amd.seflow import spec
from himo_amd.seflow.train import HeadTrainer

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
prec = sys.argv[2] if len(sys.argv) > 2 else "mixed"
rng = np.random.default_rng(0)
H = W = 512
B0 = torch.randn((H * W, 96), device=dev)
DEC = torch.randn((H * W, 64), device=dev)
pid = torch.from_numpy(rng.integers(0, H * W, n).astype(np.int32)).to(dev)
off = torch.randn((n, 3), device=dev) * 0.1
params = spec.init_params(0)
w_off = torch.from_numpy(params["head.offset.weight"]).to(dev)
b_off = torch.from_numpy(params["head.offset.bias"]).to(dev)
ht = HeadTrainer(params, device=dev, precision=prec)
call = lambda: ht.forward_fused(n, pid.data_ptr(), off.data_ptr(), B0.data_ptr() + 128, B0.data_ptr() + 256, 96, DEC.data_ptr(), 64,
                                w_off.data_ptr(), b_off.data_ptr())
for _ in range(3):
    call()
torch.cuda.synchronize()
_lib.prof_start()
for _ in range(10):
    call()
torch.cuda.synchronize()
for k, v in _lib.prof_stop().items():
    if "gru_head" in k:
        print(f"{k}: {v['avg_ms'] * 1e3:.1f} us per {n} points ({prec})")
