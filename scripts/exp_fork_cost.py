"""What does fork() cost a process that holds GPU-visible host memory?  Each case in a fresh interpreter: start HIP, set the case up, fork a
child that sleeps, then time fork itself and the first host -> device copy afterwards (the step that took seconds under the evaluator).
usage (GPU box): python scripts/exp_fork_cost.py"""
import ctypes, mmap, os, subprocess, sys, time

CASES = ("nothing else", "2 GB torch pinned", "2 GB torch pinned + MADV_DONTFORK", "2 GB pageable numpy", "slot registered AFTER the fork",
         "2 GB torch pinned, slot registered AFTER the fork")
if len(sys.argv) == 1:
    for c in CASES:
        subprocess.run([sys.executable, __file__, c])
    sys.exit(0)
case = sys.argv[1]
import numpy as np
import torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
libc = ctypes.CDLL(None, use_errno=True)
keep = []
if "torch pinned" in case:
    t = torch.empty(2 << 30, dtype=torch.uint8).pin_memory()
    t.numpy()[::4096] = 1
    t.to(dev, non_blocking=True); torch.cuda.synchronize()
    keep.append(t)
    if "DONTFORK" in case:
        lo = (t.data_ptr() + 4095) & ~4095
        rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t((t.numel() - (lo - t.data_ptr())) & ~4095), 10)
        assert rc == 0, ctypes.get_errno()
if "pageable" in case:
    a = np.ones(2 << 30, dtype=np.uint8)
    keep.append(a)
SIZE = 128 << 20
mm = mmap.mmap(-1, SIZE)
host = torch.frombuffer(mm, dtype=torch.uint8)
host.numpy()[::4096] = 1
rt = torch.cuda.cudart()
after = "AFTER" in case
if not after:
    assert int(rt.cudaHostRegister(host.data_ptr(), SIZE, 0)) == 0
dst = torch.empty(SIZE, dtype=torch.uint8, device=dev)
if not after:
    dst.copy_(host, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
pid = os.fork()
if pid == 0:
    time.sleep(3)
    os._exit(0)
t_fork = time.perf_counter() - t0
t0 = time.perf_counter()
if after:
    assert int(rt.cudaHostRegister(host.data_ptr(), SIZE, 0)) == 0
dst.copy_(host, non_blocking=True); torch.cuda.synchronize()
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
dst.copy_(host, non_blocking=True); torch.cuda.synchronize()
t_second = time.perf_counter() - t0
x = torch.ones(1 << 20, device=dev)
t0 = time.perf_counter()
(x * 2).sum().item()
t_kernel = time.perf_counter() - t0
print(f"{case:52s}: fork {1e3 * t_fork:7.1f} ms; first copy after it {1e3 * t_first:8.1f} ms; second {1e3 * t_second:6.1f} ms; a kernel + read-back {1e3 * t_kernel:6.1f} ms", flush=True)
os.kill(pid, 9); os.waitpid(pid, 0)
