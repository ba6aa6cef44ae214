"""How the CPU restatement of the network (oracle/seflow_oracle.py -- checker / baseline only) scales with torch threads on
this host: picks the thread count bench.py's all-cores CPU leg should use.  python scripts/exp_cpu_threads.py [threads ...]"""
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "oracle"))
import numpy as np
import torch
import seflow_oracle as so
from himo_amd.seflow import spec
from himo_amd.synthetic import make_frame

params = spec.init_params(0)
fr = [make_frame(i, n_points=120_000) for i in range(3)]
print("logical cores", os.cpu_count(), "torch default threads", torch.get_num_threads(), flush=True)
for thr in [int(a) for a in sys.argv[1:]] or [torch.get_num_threads(), 64, 32, 16, 1]:
    torch.set_num_threads(thr)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        so.forward(params, fr[0]["pc0"], fr[1]["pc0"], fr[2]["pc0"], fr[0]["pose0"], fr[1]["pose0"], fr[1]["pose1"])
        ts.append(time.perf_counter() - t0)
    print(f"threads {thr}: {[round(t, 2) for t in ts]} s per frame", flush=True)
