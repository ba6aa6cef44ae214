"""Can a reader PROCESS stage straight into memory the GPU copies from?  An anonymous shared mapping made before fork(), registered with
the HIP runtime in the parent (hipHostRegister), written by a forked child, copied to the device by the parent.
usage (GPU box): python scripts/exp_shared_pinned.py"""
import mmap, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
SIZE = 96 << 20
mm = mmap.mmap(-1, SIZE)                                     # MAP_SHARED | MAP_ANONYMOUS
host = torch.frombuffer(mm, dtype=torch.uint8)
host.numpy()[:] = 1                                          # touch every page
rt = torch.cuda.cudart()
t0 = time.perf_counter()
rc = rt.cudaHostRegister(host.data_ptr(), SIZE, 0)
print(f"hipHostRegister of a {SIZE >> 20} MB shared anonymous mapping: rc {rc}, {1e3 * (time.perf_counter() - t0):.1f} ms; is_pinned {host.is_pinned()}")
dst = torch.empty(SIZE, dtype=torch.uint8, device=dev)
for what, src in (("registered shared mapping", host), ("torch pinned allocation", torch.empty(SIZE, dtype=torch.uint8).pin_memory()),
                  ("pageable", torch.empty(SIZE, dtype=torch.uint8))):
    dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dst.copy_(src, non_blocking=True)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 10
    print(f"  copy from {what}: {SIZE / el / 1e9:.1f} GB/s (issue returned after {1e3 * t_issue / 10:.2f} ms per copy)")

r, w = os.pipe()
pid = os.fork()
if pid == 0:                                                 # the child never touches the GPU
    a = np.frombuffer(mm, dtype=np.uint8)
    a[:] = (np.arange(SIZE, dtype=np.uint32) % 251).astype(np.uint8)
    os.write(w, b"x")
    os._exit(0)
os.read(r, 1)
os.waitpid(pid, 0)
dst.copy_(host, non_blocking=True)
torch.cuda.synchronize()
want = (torch.arange(SIZE, dtype=torch.int32, device=dev) % 251).to(torch.uint8)
print("child's bytes arrived on the device:", bool(torch.equal(dst, want)))
print("unregister rc", rt.cudaHostUnregister(host.data_ptr()))
