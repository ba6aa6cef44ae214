"""FastNSF iteration kernels timed alone (HIP events over 30 iterations): python scripts/exp_nsf.py  [HIMO_AMD_LIB=<variant>]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from himo_amd import _lib
from himo_amd.fastnsf import FastNSF
from himo_amd.synthetic import make_frame

f0, f1 = make_frame(1), make_frame(2)
eng = FastNSF(iters=30)
eng.fit(f0["pc0"], f1["pc0"]); torch.cuda.synchronize()
_lib.prof_start(only="nsf_")
t0 = time.perf_counter(); eng.fit(f0["pc0"], f1["pc0"]); torch.cuda.synchronize(); el = time.perf_counter() - t0
prof = _lib.prof_stop()
print({k: round(v["avg_ms"] * 1e3, 1) for k, v in prof.items()}, "us;  fit of 30 iterations %.1f ms" % (el * 1e3), "loss", eng.loss_history[-1])

