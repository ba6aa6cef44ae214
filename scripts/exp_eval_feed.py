"""Where does a host-fed evaluator batch spend its time?  16 sweeps of 120k points: pack into pinned memory (per array), copies, kernels.
usage: python scripts/exp_eval_feed.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from himo_amd.eval import EvalBatch, InstanceMetrics
from himo_amd.feeder import EvalFeeder
from himo_amd.synthetic import make_frame

B = 16
dev = torch.device("cuda", 0)
frames = [make_frame(500 + i, n_points=120_000, n_instances=30) for i in range(2 * B)]
m = InstanceMetrics("av2")
m.step_frames(frames[:4], res_name="seflowpp_best")
torch.cuda.synchronize()
# (1) the pack alone: every upload() call timed (concatenate into pinned), no device copy
import himo_amd.compdis as cd
sizes = {}
def timing_upload(parts, dtype):
    parts = [np.asarray(p) for p in parts]
    shape = (sum(p.shape[0] for p in parts),) + tuple(parts[0].shape[1:])
    t0 = time.perf_counter()
    pin = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype).pin_memory() if shape not in pins else pins[shape]
    pins[shape] = pin
    t1 = time.perf_counter()
    np.concatenate(parts, 0, pin.numpy(), casting="unsafe")
    t2 = time.perf_counter()
    key = (str(np.dtype(dtype)), shape, str(parts[0].dtype))
    sizes[key] = sizes.get(key, 0.0) + (t2 - t1)
    return pin.to(dev, non_blocking=True)
pins = {}
for rep in range(3):
    sizes.clear()
    t0 = time.perf_counter()
    eb = EvalBatch.from_frames(frames[:B], "seflowpp_best", device=dev, upload=timing_upload)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"from_frames on one thread: {1e3 * (t1 - t0):.2f} ms per batch of {B} (+ {1e3 * (t2 - t1):.2f} ms until the copies landed)")
for k, v in sorted(sizes.items(), key=lambda kv: -kv[1]):
    nbytes = np.prod(k[1]) * np.dtype(k[0]).itemsize
    print(f"   concatenate -> pinned {k}: {1e3 * v:.2f} ms ({nbytes / 1e6:.1f} MB, {nbytes / v / 1e9:.1f} GB/s)")
# (2) the feeder
for n in (4, 12):
    t0 = time.perf_counter()
    for eb in EvalFeeder((frames[(k % 2) * B:(k % 2 + 1) * B] for k in range(n)), res_name="seflowpp_best", device=dev):
        m.step_batch(eb)
    m.flush(); torch.cuda.synchronize()
    print(f"EvalFeeder + step_batch, {n} batches: {1e3 * (time.perf_counter() - t0) / (n * B):.3f} ms per sweep")
for st in (0, 1, 2):
    for n in (4, 12):
        t0 = time.perf_counter()
        for eb in EvalFeeder((frames[(k % 2) * B:(k % 2 + 1) * B] for k in range(n)), res_name="seflowpp_best", device=dev, stage_threads=st):
            m.step_batch(eb)
        m.flush(); torch.cuda.synchronize()
        print(f"EvalFeeder(stage_threads={st}) + step_batch, {n} batches: {1e3 * (time.perf_counter() - t0) / (n * B):.3f} ms per sweep")
t0 = time.perf_counter()
for eb in EvalFeeder((frames[(k % 2) * B:(k % 2 + 1) * B] for k in range(12)), res_name="seflowpp_best", device=dev):
    pass
torch.cuda.synchronize()
print(f"EvalFeeder alone (no scoring), 12 batches: {1e3 * (time.perf_counter() - t0) / (12 * B):.3f} ms per sweep")

# (3) who gets in whose way?  scoring resident batches on the main thread while a second thread (a) only packs into pinned memory,
# (b) only copies a packed batch to the device, (c) does both
import threading
ebs = [EvalBatch.from_frames(frames[k * B:(k + 1) * B], "seflowpp_best", device=dev) for k in range(2)]
def score(n):
    t0 = time.perf_counter()
    for k in range(n):
        m.step_batch(ebs[k % 2])
    m.flush(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * B)
score(4)
print(f"scoring resident batches alone: {1e3 * score(12):.3f} ms per sweep")
stop = [False]
packed = [torch.empty((B * 120_000, 4), dtype=torch.float32).pin_memory() for _ in range(4)]
def pack_only():
    host = [np.ascontiguousarray(f["pc0"]) for f in frames[:B]]
    while not stop[0]:
        for pin in packed:
            np.concatenate(host, 0, pin.numpy())
def copy_only():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        while not stop[0]:
            for pin in packed:
                pin.to(dev, non_blocking=True)
            st.synchronize()
for name, fn in (("packing into pinned memory", pack_only), ("copying pinned batches to the device", copy_only)):
    stop[0] = False
    th = threading.Thread(target=fn); th.start()
    time.sleep(0.05)
    r = score(12)
    stop[0] = True; th.join()
    print(f"scoring resident batches while a second thread is {name}: {1e3 * r:.3f} ms per sweep")
host_big = [np.ascontiguousarray(f["pc0"]) for f in frames[:B]]
pageable = [np.empty((120_000, 4), np.float32) for _ in range(B)]
def copyto_pinned():
    while not stop[0]:
        for pin in packed:
            v = pin.numpy()
            for j, a in enumerate(host_big):
                np.copyto(v[j * 120_000:(j + 1) * 120_000], a)
def copyto_pageable():
    while not stop[0]:
        for dst, a in zip(pageable, host_big):
            np.copyto(dst, a)
def sleeper():
    while not stop[0]:
        time.sleep(0.0005)
def spinner():
    x = 0
    while not stop[0]:
        x += 1
for name, fn in (("np.copyto sweep by sweep into pinned memory", copyto_pinned), ("np.copyto into PAGEABLE memory", copyto_pageable),
                 ("sleeping 0.5 ms at a time", sleeper), ("spinning in Python", spinner)):
    stop[0] = False
    th = threading.Thread(target=fn); th.start()
    time.sleep(0.05)
    r = score(12)
    stop[0] = True; th.join()
    print(f"scoring resident batches while a second thread is {name}: {1e3 * r:.3f} ms per sweep")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
score(12)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
import os
sys.stdout.flush(); os._exit(0)
