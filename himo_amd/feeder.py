"""Host-fed ends of the pipeline: keep the GPU busy when frames arrive as host arrays.

The reference's drivers are serial loops -- read a frame, compute, write (``save_zip.py:112``, ``eval.py:281``) -- so the
device would idle during every read and every copy.  Here

* ``SampleFeeder``: a background thread pulls frame tuples from any iterable (an h5 / npz dataset walk, a socket, a
  synthetic generator), stages the three sweeps and ``lidar_dt`` in PINNED host buffers and issues the host -> device
  copies on its own HIP stream, ``depth`` batches ahead; the consumer gets ``Sample`` objects whose tensors are already
  ordered after the copy on the consumer's stream (event wait, no host synchronisation);
* ``TrainFeeder``: the training loop's samples -- sweep triplets read on reader threads, staged, copied and LABELLED
  (``ssl_label=seflow_auto``) one to two samples ahead of the optimiser step;
* ``ResultDrain``: device -> pinned-host copies of the per-frame results on the compute stream, handed to a writer
  thread that waits on the copy's event and calls the sink (Feather / npz writer) off the launch thread.

PyTorch supplies the streams, events and pinned allocations; no arithmetic happens here.
"""
from __future__ import annotations

import itertools
import queue
import threading
import time

import numpy as np
import torch

from . import _lib
from .pipeline import Sample


class _PinnedArena:
    """One pinned staging buffer per in-flight batch slot, carved by a bump pointer (pinned allocations cost milliseconds
    each: a batch makes ONE, and only when it outgrows the slot's previous one)."""

    def __init__(self):
        self._buf = torch.empty(0, dtype=torch.uint8)
        self._used = 0
        self._extra, self._extra_used = [], []        # further pinned blocks chained on by take_growing

    def reset(self, need_bytes: int):
        if self._buf.numel() < need_bytes:
            self._buf = _lib.pinned_empty(int(need_bytes * 1.25) + 4096)
        self._used = 0
        self._extra_used = [0] * len(self._extra)

    def take_growing(self, shape, dtype=torch.float32) -> torch.Tensor:
        """``take`` without a size known up front: when the block is exhausted a further pinned block is chained on (kept for
        the following batches, so steady state allocates nothing)."""
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        lo = (self._used + 63) & ~63
        if lo + nbytes <= self._buf.numel():
            self._used = lo + nbytes
            return self._buf[lo:lo + nbytes].view(dtype).view(*shape)
        for i, blk in enumerate(self._extra):
            lo = (self._extra_used[i] + 63) & ~63
            if lo + nbytes <= blk.numel():
                self._extra_used[i] = lo + nbytes
                return blk[lo:lo + nbytes].view(dtype).view(*shape)
        blk = _lib.pinned_empty(max(int(nbytes * 1.25) + 4096, 1 << 24))
        self._extra.append(blk)
        self._extra_used.append(nbytes)
        return blk[:nbytes].view(dtype).view(*shape)

    def take(self, shape, dtype=torch.float32) -> torch.Tensor:
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        lo = (self._used + 63) & ~63
        self._used = lo + nbytes
        return self._buf[lo:lo + nbytes].view(dtype).view(*shape)

    def locate(self, pin: torch.Tensor):
        """(index of the pinned block ``pin`` was carved from, the block, byte offset inside it)"""
        at = pin.data_ptr()
        for i, blk in enumerate([self._buf] + self._extra):
            lo = blk.data_ptr()
            if blk.numel() and lo <= at < lo + blk.numel():
                return i, blk, at - lo
        raise ValueError("not a tensor of this arena")

    def used(self, i: int) -> int:
        return self._used if i == 0 else self._extra_used[i - 1]


# pinned staging arenas outlive the feeder that allocated them: a pinned allocation of a batch's ~100 MB costs tens of
# milliseconds, more than staging the batch itself
_ARENA_POOL: list = []
_ARENA_LOCK = threading.Lock()


def _borrow_arenas(n: int) -> list:
    with _ARENA_LOCK:
        got = [_ARENA_POOL.pop() for _ in range(min(n, len(_ARENA_POOL)))]
    return got + [_PinnedArena() for _ in range(n - len(got))]


def _return_arenas(arenas: list, events: list) -> None:
    for ev in events:
        if ev is not None:
            ev.synchronize()                      # the copies out of these buffers have completed
    with _ARENA_LOCK:
        _ARENA_POOL.extend(arenas)


_COPY_STREAMS = {}
_COPY_STREAMS_LOCK = threading.Lock()


def copy_stream(device):
    """The stream the inference / evaluation feeders issue their host -> device copies on: ONE per device and process.  Programs build a
    feeder per call (per scene list, per bench leg); a stream per feeder would walk through torch's stream pool until one shares a
    hardware queue with the streams the batches are computed on (``seflow.train.side_streams`` has the measurement)."""
    key = (device.type, device.index)
    with _COPY_STREAMS_LOCK:
        if key not in _COPY_STREAMS:
            _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
        return _COPY_STREAMS[key]


class SampleFeeder:
    """Iterate batches ``[(index, f0, Sample), ...]`` of up to ``batch`` frames, prepared ``depth`` batches ahead.

    ``source`` yields ``(index, fh, f0, f1)``: the history / current / next frame dicts of the reference layout (``f1``
    may be None when ``f0`` carries ``pc1`` / ``pose1``), host numpy arrays."""

    _END = object()

    def __init__(self, source, device=None, batch: int = 8, depth: int = 2, stage_threads: int = 3):
        """``stage_threads``: threads that copy a batch's sweeps into the pinned staging buffers (plain memcpys that release
        the GIL; one thread moves ~5 GB/s = ~800 samples/s of 3 x 120k points, short of the pipeline's rate once the source is a
        file mapping instead of arrays already in memory); 0 or 1 = on the feeder thread itself"""
        if batch < 1 or depth < 1:
            raise ValueError("batch and depth must be >= 1")
        self._copiers = None
        if stage_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._copiers = ThreadPoolExecutor(max_workers=stage_threads, thread_name_prefix="himo-stage")
        self.device = device if device is not None else _lib.require_gpu()
        self.batch, self.depth = batch, depth
        self._source = iter(source)
        self._q = queue.Queue(maxsize=depth)
        self._slots = [_PinnedArena() for _ in range(depth + 2)]      # a slot is reused only after its copies completed
        self._slot_done = [None] * (depth + 2)
        self._stream = copy_stream(self.device)
        self._error = None
        self._thread = threading.Thread(target=self._work, name="himo-feeder", daemon=True)
        self._thread.start()

    def _stage_all(self, arena, host):
        """every array of the batch -> its place in the slot's pinned block -> the device as ONE copy of the block's used bytes (on the
        feeder's stream, which the caller has made current); returns (device views per array, the device block they view).  Sixty-four
        copies of 1.4-1.9 MB per 16-sample batch ran far below the link's rate and cost the consumer a ``record_stream`` per tensor."""
        pins = [[arena.take(a.shape) for a in arrs] for arrs in host]
        # pageable (or file mapping) -> pinned with numpy (a plain memcpy that drops the GIL): torch's copy_ spins up its
        # intra-op thread pool on every call from a non-main thread (measured 0.96 ms vs 0.03 ms for a 1.9 MB sweep)
        jobs = [(pin.numpy(), a) for prow, arrs in zip(pins, host) for pin, a in zip(prow, arrs)]
        if self._copiers is not None and len(jobs) > 1:
            list(self._copiers.map(lambda j: np.copyto(j[0], j[1]), jobs))
        else:
            for dst, src in jobs:
                np.copyto(dst, src)
        used = arena.used(0)
        dev_blk = torch.empty(max(used, 1), dtype=torch.uint8, device=self.device)
        if used:
            dev_blk[:used].copy_(arena._buf[:used], non_blocking=True)                            # pinned -> HBM, once
        base = arena._buf.data_ptr()

        def view(pin):
            n = pin.numel() * pin.element_size()
            if n == 0:
                return torch.empty(pin.shape, dtype=pin.dtype, device=self.device)
            off = pin.data_ptr() - base
            return dev_blk[off:off + n].view(pin.dtype).view(pin.shape)
        return [[view(pin) for pin in prow] for prow in pins], dev_blk

    def _work(self):
        try:
            torch.cuda.set_device(self.device)
            slot = 0
            while True:
                items = []
                for item in self._source:
                    items.append(item)
                    if len(items) >= self.batch:
                        break
                if not items:
                    break
                if self._slot_done[slot] is not None:
                    self._slot_done[slot].synchronize()      # the copies that last used this slot's pinned buffers
                arena = self._slots[slot]
                host = []
                for index, fh, f0, f1 in items:
                    pc1 = f1["pc0"] if f1 is not None else f0["pc1"]
                    host.append([np.ascontiguousarray(a, dtype=np.float32) for a in (fh["pc0"], f0["pc0"], pc1, f0["lidar_dt"])])
                arena.reset(sum(a.nbytes + 64 for arrs in host for a in arrs))
                out = []
                with torch.cuda.stream(self._stream):
                    staged, dev_blk = self._stage_all(arena, host)
                    for (index, fh, f0, f1), (dh, d0, d1, dt) in zip(items, staged):
                        s = Sample(dh, d0, d1, np.asarray(fh["pose0"], np.float64), np.asarray(f0["pose0"], np.float64),
                                   np.asarray(f0["pose1"], np.float64), dt, f0.get("scene_id", ""), int(f0.get("timestamp", 0)))
                        out.append((index, f0, s))
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                self._slot_done[slot] = ev
                self._q.put((out, ev, dev_blk))
                slot = (slot + 1) % len(self._slots)
        except BaseException as e:                             # surfaced on the consumer's thread
            self._error = e
        finally:
            if self._copiers is not None:
                self._copiers.shutdown(wait=False)
            self._q.put(self._END)

    def __iter__(self):
        while True:
            got = self._q.get()
            if got is self._END:
                if self._error is not None:
                    raise self._error
                return
            out, ev, dev_blk = got
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                                  # order the consumer's stream after the copy: no host wait
            dev_blk.record_stream(cur)                          # (the batch's tensors view ONE block) allocated on the feeder's stream, used on this one
            yield out


class ResultDrain:
    """``put(key, device_tensor)`` copies to pinned host memory on the current stream and returns at once; a writer thread
    calls ``sink(key, numpy_array)`` when the copy has landed.  ``close()`` waits for everything and re-raises sink errors."""

    _END = object()

    def __init__(self, sink, device=None, depth: int = 32, threads: int = 1, copy: bool = True):
        """``threads``: writer threads.  One (default) calls the sink in ``put`` order -- what a sink that groups results needs
        (``save.H5ResultSink`` flushes a scene when the next one starts); several call it concurrently, in any order -- for sinks
        whose calls are independent (one Feather file per sweep: the encoding and the file write release the GIL).
        ``copy=False`` hands the sink a view of the PINNED buffer, valid only until the sink returns (it is reused)."""
        self.device = device if device is not None else _lib.require_gpu()
        self._sink, self._copy = sink, copy
        self._q = queue.Queue(maxsize=depth)
        self._free = queue.Queue()
        self._error = None
        self._threads = [threading.Thread(target=self._work, name=f"himo-drain-{k}", daemon=True) for k in range(max(1, threads))]
        for t in self._threads:
            t.start()

    def _pinned(self, like: torch.Tensor) -> torch.Tensor:
        """a free pinned buffer that fits (ragged sweeps: capacities are rounded up to a power of two so they are reused)"""
        fit, rest = None, []
        try:
            while fit is None:
                buf = self._free.get_nowait()
                if buf.numel() >= like.numel() and buf.dtype == like.dtype:
                    fit = buf
                else:
                    rest.append(buf)
        except queue.Empty:
            pass
        for buf in rest:
            self._free.put(buf)
        if fit is None:
            fit = _lib.pinned_empty(1 << max(int(like.numel()) - 1, 1).bit_length(), like.dtype)
        return fit

    def put(self, key, t: torch.Tensor) -> None:
        if self._error is not None:
            raise self._error
        buf = self._pinned(t)
        view = buf[:t.numel()].view(t.shape)
        view.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._q.put((key, buf, view, ev))

    def _work(self):
        while True:
            got = self._q.get()
            if got is self._END:
                return
            key, buf, view, ev = got
            try:
                ev.synchronize()
                if self._error is None:
                    self._sink(key, view.numpy().copy() if self._copy else view.numpy())
            except BaseException as e:
                self._error = e
            self._free.put(buf)

    def close(self) -> None:
        for _ in self._threads:
            self._q.put(self._END)
        for t in self._threads:
            t.join()
        if self._error is not None:
            raise self._error


class BatchFeeder:
    """Iterate device-resident batch objects prepared ``depth`` batches ahead of the consumer.

    ``source`` yields items (lists of host frame dicts, ...); ``build(item, upload)`` packs one item into a batch object whose
    device tensors all come from ``upload(parts, dtype)`` and returns ``(object to yield, [its device tensors])``.  A background
    thread runs ``build``: ``upload`` concatenates (and converts) the parts STRAIGHT into pinned staging memory on a small
    thread pool (numpy releases the GIL: the arrays of a batch are staged in parallel) and issues the host -> device copies on
    the feeder's own stream; the consumer's stream is ordered after them by an event, so it never waits for a copy it did not
    need yet."""

    _END = object()

    def __init__(self, source, build, device=None, depth: int = 2, stage_threads: int = 4):
        """``stage_threads``: threads that copy a batch's arrays into the pinned staging memory; 0 = on the feeder thread itself, one
        numpy call per tensor -- slower staging (one core), fewer threads contending with an interpreter-bound consumer (measured for
        the evaluator, scripts/exp_eval_feed.py: no gain -- what had looked like contention there were the interpreter's full
        collections, see eval.InstanceMetrics._accumulate_frame; the option stays for hosts with very few cores)"""
        self.device = device if device is not None else _lib.require_gpu()
        self.depth, self._build = depth, build
        self._source = iter(source)
        self._q = queue.Queue(maxsize=depth)
        self._slots = _borrow_arenas(depth + 2)
        self._slot_done = [None] * (depth + 2)
        self._stream = copy_stream(self.device)
        self._error = None
        self._stop = False
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=stage_threads, thread_name_prefix="himo-stage") if stage_threads > 0 else None
        self._thread = threading.Thread(target=self._work, name="himo-batch-feeder", daemon=True)
        self._thread.start()

    def _work(self):
        try:
            torch.cuda.set_device(self.device)
            slot = 0
            for item in self._source:
                if self._slot_done[slot] is not None:
                    self._slot_done[slot].synchronize()
                arena = self._slots[slot]
                arena.reset(0)
                jobs = []

                def upload(parts, dtype):
                    parts = [np.asarray(p) for p in parts]
                    shape = (sum(p.shape[0] for p in parts),) + tuple(parts[0].shape[1:])
                    tdt = torch.from_numpy(np.empty(0, dtype)).dtype
                    pin = arena.take_growing(shape, tdt)
                    # the device tensor is a view of a device twin of the PINNED BLOCK the staging tensor came from, at the same
                    # offset: a batch goes over as one copy per block (a handful) instead of one per tensor -- small copies run
                    # far below the link's rate (five ~1 MB copies: 3 GB/s; one 6 MB copy: 53 GB/s) -- and the consumer records ONE
                    # storage per block on its stream instead of eleven tensors (0.12 ms of its launch thread each)
                    if pin.numel() == 0:
                        return torch.empty(shape, dtype=tdt, device=self.device)
                    bi, blk, off = arena.locate(pin)
                    if bi not in blocks:
                        blocks[bi] = (blk, torch.empty(blk.numel(), dtype=torch.uint8, device=self.device))
                    nbytes = pin.numel() * pin.element_size()
                    dst = blocks[bi][1][off:off + nbytes].view(tdt).view(shape)
                    # one copy job per PART (a frame's array), not per tensor: a batch's largest tensor (16 sweeps of points:
                    # 30 MB) as one job kept one thread busy for 6 ms while the others idled
                    # (small tensors -- masks, labels, time stamps -- stay ONE job: a future costs ~40 us of interpreter time)
                    host, at = pin.numpy(), 0
                    if self._pool is None:
                        np.concatenate(parts, 0, host, casting="unsafe")
                    elif pin.numel() * pin.element_size() < (4 << 20) or len(parts) == 1:
                        jobs.append((self._pool.submit(np.concatenate, parts, 0, host, casting="unsafe"), None, None))
                    else:
                        for p in parts:
                            n = p.shape[0]
                            jobs.append((self._pool.submit(np.copyto, host[at:at + n], p, casting="unsafe"), None, None))
                            at += n
                    return dst

                with torch.cuda.stream(self._stream):
                    blocks = {}
                    obj, _ = self._build(item, upload)
                    for job, _, _ in jobs:
                        job.result()
                    for bi, (blk, dev_blk) in blocks.items():
                        n = arena.used(bi)
                        dev_blk[:n].copy_(blk[:n], non_blocking=True)
                    tensors = [dev_blk for _, dev_blk in blocks.values()]
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                self._slot_done[slot] = ev
                if not self._offer((obj, tensors, ev)):
                    break
                slot = (slot + 1) % len(self._slots)
        except BaseException as e:
            self._error = e
        finally:
            _return_arenas(self._slots, self._slot_done)
            if self._pool is not None:
                self._pool.shutdown(wait=False)
            self._offer(self._END)

    def _offer(self, item) -> bool:
        while not self._stop:
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def close(self) -> None:
        """Stop early (the consumer gave up): the worker returns its pinned arenas and exits."""
        self._stop = True
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=10)

    def __iter__(self):
        while True:
            got = self._q.get()
            if got is self._END:
                if self._error is not None:
                    raise self._error
                return
            obj, tensors, ev = got
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in tensors:
                if t is not None:
                    t.record_stream(cur)
            yield obj


class _Ref:
    """Where a reader process put one array of a batch: byte offset inside its slot, shape, numpy dtype string."""
    __slots__ = ("off", "shape", "dtype")

    def __init__(self, off, shape, dtype):
        self.off, self.shape, self.dtype = off, tuple(shape), dtype

    def __reduce__(self):
        return _Ref, (self.off, self.shape, self.dtype)


def _swap_refs(obj, view, depth: int = 0):
    """Replace the ``_Ref`` attributes of a batch object (and of the himo_amd objects it holds) by ``view(ref)``."""
    for k, v in list(vars(obj).items()):
        if isinstance(v, _Ref):
            setattr(obj, k, view(v))
        elif depth < 2 and hasattr(v, "__dict__") and type(v).__module__.startswith(__name__.rsplit(".", 1)[0]):
            _swap_refs(v, view, depth + 1)


class ReaderPool:
    """``build(make(k), upload)`` for k = 0 .. n_items-1 on forked reader PROCESSES, results handed back in order.

    Why processes: parsing an HDF5 sweep is ~0.2 ms of interpreter work and packing a batch a dozen numpy calls; on threads that work
    holds the interpreter lock the consumer's launch thread needs (the evaluator launches and scores ~5 k sweeps/s from resident
    batches, 1.4 k fed by threads).  The reference's loops get the same effect from ``DataLoader(num_workers=...)``.

    Each worker writes a batch's arrays end to end into a SLOT -- an anonymous shared mapping made before the fork, so parent and
    children see the same pages (``slot_bytes`` is address space: only pages a batch touched are ever committed) -- and sends back
    the batch object with ``_Ref`` placeholders (a few hundred bytes through a pipe).  A batch larger than a slot makes the pool
    start over from that batch with larger slots.  Workers never touch the GPU; they inherit ``make`` / ``build`` and everything
    those reference (open, memory-mapped scene files) through the fork.

    WHEN to fork: ``start()`` -- called by the constructor of ``ProcessBatchFeeder`` -- should run BEFORE the process has started
    the HIP runtime.  A fork write-protects the parent's private pages for copy-on-write; where those pages are mapped into the
    GPU's address space the driver takes the mappings away and rebuilds them at the next device operation: 0.2 s in a process that
    has only started the runtime, ~3 s in one that has run the evaluator (scripts/exp_fork_cost.py, profiles/r06_exp_fork_cost.txt).
    Programs (``python -m himo_amd.eval``) therefore start their readers first thing.

    WHO forks matters too: ``os.fork()`` deletes, in the child, the interpreter state of every thread but the forking one -- their
    frames and thread-local values are released THERE, and releasing a device or pinned tensor that has streams recorded on it calls
    into a HIP runtime the child must not touch (observed: both children of a second generation segfault inside ``fork()`` when it
    ran on the feeder thread while the main thread's frames held device tensors).  So: ``start()`` on the thread that owns the
    process's device objects (the main thread, as ``DataLoader`` does), and ``restart=False`` wherever the pool iterates on another
    thread of a process with device state -- a batch that does not fit is then an error naming the size to ask for, not a second fork."""

    def __init__(self, n_items: int, make, build, workers: int = 4, n_slots: int | None = None, slot_bytes: int = 512 << 20,
                 on_slots=None, off_slots=None, restart: bool = True):
        import multiprocessing
        self.n, self._make, self._build = int(n_items), make, build
        self.workers = max(1, int(workers))
        self.n_slots = max(int(n_slots) if n_slots else self.workers + 2, 1)
        self.slot_bytes = int(slot_bytes)
        self._on_slots, self._off_slots = on_slots, off_slots
        self._restart = restart
        self._ctx = multiprocessing.get_context("fork")
        self._procs, self._tasks, self._results = [], [], None
        self._maps, self.slots = [], []
        self.restarts = 0
        self._stop = False
        # where the wall time went: slots + fork, shutdown (this process); reading and packing (summed over the workers)
        self.stage_seconds = {"start": 0.0, "halt": 0.0, "read": 0.0, "pack": 0.0, "batches": 0}

    # ---- the child ---------------------------------------------------------------------------------------------------
    def _child(self, w):
        tasks, results, maps = self._tasks[w], self._results, self._maps
        while True:
            t = tasks.get()
            if t is None:
                return
            k, s = t
            try:
                buf = np.frombuffer(maps[s], dtype=np.uint8)
                need = [0]

                def upload(parts, dtype):
                    parts = [np.asarray(p) for p in parts]
                    dt = np.dtype(dtype)
                    shape = (sum(p.shape[0] for p in parts),) + tuple(parts[0].shape[1:])
                    nbytes = int(np.prod(shape)) * dt.itemsize
                    lo = (need[0] + 63) & ~63
                    need[0] = lo + nbytes
                    if nbytes and need[0] <= buf.size:
                        np.concatenate(parts, 0, buf[lo:lo + nbytes].view(dt).reshape(shape), casting="unsafe")
                    return _Ref(lo, shape, dt.str)
                t0 = time.perf_counter()
                item = self._make(k)
                t1 = time.perf_counter()
                obj = self._build(item, upload)
                t2 = time.perf_counter()
                results.put(("grow", k, s, need[0]) if need[0] > buf.size else ("ok", k, s, need[0], obj, t1 - t0, t2 - t1))
            except BaseException as e:
                try:
                    results.put(("error", k, s, e))
                except BaseException:
                    results.put(("error", k, s, RuntimeError(f"{type(e).__name__}: {e}")))

    # ---- the parent --------------------------------------------------------------------------------------------------
    def start(self):
        """make the slots and fork the workers now (idempotent; iteration does it otherwise)"""
        if not self._procs and self.n > 0:
            self._start()

    def _start(self):
        import mmap
        t0 = time.perf_counter()
        flags = mmap.MAP_SHARED | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0)        # address space; pages arrive when touched
        self._maps = [mmap.mmap(-1, self.slot_bytes, flags=flags) for _ in range(self.n_slots)]
        self.slots = [torch.frombuffer(m, dtype=torch.uint8) for m in self._maps]
        self._tasks = [self._ctx.SimpleQueue() for _ in range(self.workers)]
        self._results = self._ctx.SimpleQueue()
        self._procs = [self._ctx.Process(target=self._child, args=(w,), name=f"himo-reader-{w}", daemon=True) for w in range(self.workers)]
        import sys
        for stream in (sys.stdout, sys.stderr):                   # a child inherits what is buffered here and would write it again if it ever flushed
            try:
                stream.flush()
            except Exception:
                pass
        for p in self._procs:
            p.start()
        if self._on_slots is not None:
            self._on_slots(self.slots)
        self.stage_seconds["start"] += time.perf_counter() - t0

    def _halt(self):
        t0 = time.perf_counter()
        for p in self._procs:
            if p.is_alive():
                p.terminate()                      # readers hold nothing that needs an orderly exit
        for p in self._procs:
            p.join(timeout=5)
        self._procs = []
        if self.slots and self._off_slots is not None:
            self._off_slots(self.slots)
        self.slots = []
        maps, self._maps = self._maps, []
        for m in maps:
            try:
                m.close()
            except BufferError:                    # a tensor view is still alive somewhere: the mapping goes with it
                pass
        self.stage_seconds["halt"] += time.perf_counter() - t0

    def close(self):
        self._stop = True

    def _next_message(self):
        while not self._results._reader.poll(0.2):
            if self._stop:
                return None
            dead = [p.name for p in self._procs if not p.is_alive()]
            if dead:
                raise RuntimeError(f"reader process {dead[0]} died")
        return self._results.get()

    def __iter__(self):
        """yields (k, slot index, bytes used, batch object with _Ref placeholders); the consumer calls ``release(slot)`` once it has
        read the slot (copies out of it have completed or are ordered before its next use by ``release``'s caller)."""
        if self.n == 0:
            return
        k_out = k_task = 0
        done, self._free = {}, list(range(self.n_slots))
        self.start()
        try:
            while k_out < self.n and not self._stop:
                while self._free and k_task < self.n:
                    self._tasks[k_task % self.workers].put((k_task, self._free.pop(0)))
                    k_task += 1
                if k_out in done:
                    if done[k_out][0] == "error":          # raised where the serial loop would have met it: after the batches before it
                        raise done[k_out][3]
                    _, k, s, used, obj, t_read, t_pack = done.pop(k_out)
                    st = self.stage_seconds
                    st["read"] += t_read
                    st["pack"] += t_pack
                    st["batches"] += 1
                    yield k, s, used, obj
                    k_out += 1
                    continue
                msg = self._next_message()
                if msg is None:
                    return
                if msg[0] == "grow":
                    if not self._restart:
                        raise RuntimeError(f"batch {msg[1]} needs {msg[3]} bytes, the reader slots hold {self.slot_bytes}: construct the feeder "
                                           f"with slot_bytes >= {int(msg[3] * 1.25)} (address space only: pages are committed as batches touch them)")
                    # start over from the oldest batch not handed out yet, with slots that hold this one
                    self._halt()
                    self.slot_bytes = max(int(msg[3] * 1.25) + 4096, self.slot_bytes)
                    self.restarts += 1
                    done.clear()
                    k_task, self._free = k_out, list(range(self.n_slots))
                    self._start()
                    continue
                done[msg[1]] = msg
        finally:
            self._halt()

    def release(self, slot: int) -> None:
        self._free.append(slot)


class ProcessBatchFeeder:
    """``BatchFeeder`` whose batches are read and packed by forked reader processes (``ReaderPool``): iterate device-resident batch
    objects, ``depth`` ahead of the consumer.  ``make(k)`` reads item k (a list of frame dicts, ...) and ``build(item, upload)`` packs
    it -- both run in a worker; this process's feeder thread issues ONE host -> device copy per batch out of the worker's slot (the
    part of a slot that batches have used is registered with the HIP runtime: the copy is a DMA at the link's rate, asynchronous)
    and turns the placeholders into views of the device block.  The constructor forks the workers BEFORE it touches the device:
    construct it before anything else starts the HIP runtime (see ``ReaderPool``)."""

    _END = object()

    def __init__(self, n_items: int, make, build, device=None, depth: int = 2, workers: int = 4, slot_bytes: int = 2 << 30):
        self._registered = {}                          # slot -> bytes of its head registered with the runtime
        # (no second generation of workers: this pool iterates on the feeder thread of a process with device state -- see ReaderPool;
        # the slots are 2 GB of ADDRESS SPACE each, of which a batch commits and this process registers only what it uses)
        self.pool = ReaderPool(n_items, make, build, workers=workers, n_slots=workers + depth + 1, slot_bytes=slot_bytes,
                               off_slots=self._unregister, restart=False)
        self.forked_before_hip = not torch.cuda.is_initialized()
        self.pool.start()
        self.device = device if device is not None else _lib.require_gpu()
        self.depth = depth
        self._q = queue.Queue(maxsize=depth)
        self._stream = copy_stream(self.device)
        self._error = None
        self._stop = False
        self._thread = threading.Thread(target=self._work, name="himo-process-feeder", daemon=True)
        self._thread.start()

    def _ensure_registered(self, s: int, used: int) -> None:
        """the first ``used`` bytes of slot s are pinned and mapped for the GPU (grown in steps: registering commits the pages)"""
        have = self._registered.get(s, 0)
        if used <= have:
            return
        rt, t = torch.cuda.cudart(), self.pool.slots[s]
        if have:
            rt.cudaHostUnregister(t.data_ptr())
        want = min(t.numel(), ((int(used * 1.25) + (8 << 20)) + 0x1FFFFF) & ~0x1FFFFF)
        rc = rt.cudaHostRegister(t.data_ptr(), want, 0)
        if int(rc) != 0:
            raise RuntimeError(f"hipHostRegister of {want >> 20} MB of a reader slot failed: {rc}")
        self._registered[s] = want

    def _unregister(self, slots):
        torch.cuda.synchronize(self.device)                                   # copies out of the slots have completed
        rt = torch.cuda.cudart()
        for s, t in enumerate(slots):
            if self._registered.pop(s, 0):
                rt.cudaHostUnregister(t.data_ptr())

    def _work(self):
        try:
            torch.cuda.set_device(self.device)
            for k, s, used, obj in self.pool:
                self._ensure_registered(s, used)
                with torch.cuda.stream(self._stream):
                    dev_blk = torch.empty(max(used, 1), dtype=torch.uint8, device=self.device)
                    if used:
                        dev_blk[:used].copy_(self.pool.slots[s][:used], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._stream)

                def view(ref, blk=dev_blk):
                    tdt = torch.from_numpy(np.empty(0, np.dtype(ref.dtype))).dtype
                    n = int(np.prod(ref.shape)) * np.dtype(ref.dtype).itemsize
                    if n == 0:
                        return torch.empty(ref.shape, dtype=tdt, device=self.device)
                    return blk[ref.off:ref.off + n].view(tdt).view(ref.shape)
                _swap_refs(obj, view)
                ev.synchronize()            # the slot goes back to a reader only after the copy out of it (a few ms; this thread idles anyway)
                self.pool.release(s)
                if not self._offer((obj, [dev_blk], ev)):
                    break
        except BaseException as e:
            self._error = e
        finally:
            self.pool.close()
            self._offer(self._END)

    _offer = BatchFeeder._offer
    __iter__ = BatchFeeder.__iter__

    def close(self) -> None:
        self._stop = True
        self.pool.close()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=10)
        if self.pool._procs and not self._thread.is_alive():      # never iterated: the workers of the constructor are still there
            self.pool._halt()


_TRAIN_STREAMS = {}
_TRAIN_STREAMS_LOCK = threading.Lock()


def _train_streams(device, n_label: int, priority: int = 0):
    """(copy stream, label streams) of the training feeders, ONE set per device and process: a run builds a feeder per epoch, and a
    stream per feeder would walk through torch's stream pool until one of them shares a hardware queue with the training step's
    streams (``seflow.train.side_streams`` has the measurement)."""
    key = (device.type, device.index, priority)
    with _TRAIN_STREAMS_LOCK:
        copy, labels = _TRAIN_STREAMS.setdefault(key, (copy_stream(device), []))            # (the process's one copy stream)
        while len(labels) < n_label:
            labels.append(torch.cuda.Stream(device=device, priority=priority))
        return copy, labels[:n_label]


class TrainFeeder:
    """Iterate training samples ``(pch1, pc0, pc1, pose_h1, pose0, pose1, label0, label1, n_labels)`` -- the tuples
    ``seflow.fit.make_sample`` builds on the spot -- prepared AHEAD of the optimiser step, the way the reference's job keeps
    its steps fed (``num_workers=16`` dataloader workers prefetching beside ``train.py``'s step, assets/slurm/ssl-train-av2.sh:31-34;
    the loop shape of save_zip.py:111-113: ``dataset[i]`` -> compute).  Three stages, each on its own thread(s), samples in order:

    * READ: ``workers`` threads call ``seflow.fit.host_sample`` (the frames' sweeps / poses / ground masks or labels, read with
      ``fields=`` from the open scene files) up to ``depth + workers`` triplets ahead and copy the arrays ONCE, from the file
      mapping straight into a pinned arena of their own;
    * UPLOAD: the feeder thread takes the reads in order and issues the host -> device copies on a COPY stream;
    * LABEL: ``label_lanes`` threads, each with its own stream ordered after the sample's copies by an event, generate the cluster
      labels of ``ssl_label=seflow_auto`` (``seflow.ssl_label.auto_labels``: two exact nearest-neighbour passes and two DBSCANs
      per pair, one host wait for the sizes of the compacted sweeps, one for the label count -- through pinned memory, on THAT
      thread).  A pair's chain -- copy, kernels, wait, kernels, wait -- is ~2 ms on an idle device but 5-8 ms beside a training
      step that owns most of the device and of the interpreter lock; with one lane the feeder delivered a sample per chain and
      the step waited for it, with two the chains of successive samples overlap (profiles/r06_fit_stages.txt).

    The consumer's stream is ordered after a sample's last kernel by an event: the training thread never waits on the host for a
    read, a copy or a label, and gets ``n_labels`` as a plain int.  Labels and samples are pure functions of the frames, so a fed
    run ends in the parameter bits of the run that builds every sample inside the step (tests/test_fit_gpu.py).  ``close()``
    stops early."""

    _END = object()

    def __init__(self, dataset, trips, device=None, label_key: str = "seflow_auto", depth: int = 2, workers: int = 1,
                 label_lanes: int = 1, label_priority: int = 0, label_cache: dict | None = None, label_cache_bytes: int = 4 << 30):
        """``label_cache``: a dict the caller keeps between feeders over the SAME dataset (``fit``: one per dataset, for the whole
        run).  Generated labels are a pure function of the sweep pair, and the reference's job reads them from files an offline pass
        wrote once; here the first epoch generates them on the device and leaves a host copy in the dict (up to ``label_cache_bytes``),
        later epochs upload that copy instead of clustering the pair again -- the same labels, bit for bit."""
        from concurrent.futures import ThreadPoolExecutor
        if depth < 1 or workers < 1 or label_lanes < 1:
            raise ValueError("depth, workers and label_lanes must be >= 1")
        self._cache, self._cache_budget = label_cache, int(label_cache_bytes)
        self.device = device if device is not None else _lib.require_gpu()
        self.dataset, self.trips, self.label_key = dataset, list(trips), label_key
        self.depth, self.workers, self.lanes = depth, workers, label_lanes
        self._window = depth + workers + label_lanes           # reads in flight ahead of the copy being issued
        self._free = queue.Queue()                             # (pinned arena, event of the copies that last read it | None)
        self._arenas = _borrow_arenas(self._window + 2)
        for a in self._arenas:
            self._free.put((a, None))
        self._events = []                                      # copy events of arenas handed back (for _return_arenas)
        self._q = queue.Queue(maxsize=depth)
        self._copy, self._label_streams = _train_streams(self.device, label_lanes, label_priority)
        self._lane_ids = itertools.count()
        self._tls = threading.local()                          # per label thread: its stream and its pinned word
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="himo-train-read")
        self._labellers = ThreadPoolExecutor(max_workers=label_lanes, thread_name_prefix="himo-train-label")
        self._error, self._stop = None, False
        self._stat_lock = threading.Lock()
        self.stage_seconds = {"read": 0.0, "upload": 0.0, "labels": 0.0, "samples": 0}      # host time per stage (profiles/r06_fit_stages.txt)
        self._thread = threading.Thread(target=self._work, name="himo-train-feeder", daemon=True)
        self._thread.start()

    # ---- reader threads ------------------------------------------------------------------------------------------
    _F32 = ("pch1", "pc0", "pc1")

    def _read(self, trip):
        import time
        from .seflow.fit import host_sample
        t0 = time.perf_counter()
        h = host_sample(self.dataset, trip, self.label_key)
        cached = self._cache.get(trip[1:]) if self._cache is not None else None
        if cached is not None:                                 # labels generated in an earlier epoch: upload them like stored labels
            h = {k: v for k, v in h.items() if k not in ("gm0", "gm1")}
            h["lab0"], h["lab1"] = cached[0], cached[1]
        arena, ev = self._free.get()
        if ev is not None:
            ev.synchronize()                                   # the copies that last read this arena's pinned memory
        arrs = {k: np.asarray(h[k]) for k in self._F32}
        if "gm0" in h:
            small = {"gm0": (np.asarray(h["gm0"]), torch.uint8), "gm1": (np.asarray(h["gm1"]), torch.uint8)}
        else:
            small = {"lab0": (np.asarray(h["lab0"]), torch.int32), "lab1": (np.asarray(h["lab1"]), torch.int32)}
        arena.reset(sum(a.size * 4 + 64 for a in arrs.values()) + sum(a.size * 4 + 64 for a, _ in small.values()))
        pins = {}
        for k, a in arrs.items():
            pins[k] = arena.take(a.shape, torch.float32)
            np.copyto(pins[k].numpy(), a, casting="unsafe")   # file mapping (or array) -> pinned: the one host copy of a sweep
        n_host = None
        for k, (a, tdt) in small.items():
            pins[k] = arena.take(a.shape, tdt)
            np.copyto(pins[k].numpy(), a, casting="unsafe")
        if cached is not None:
            n_host = cached[2]
        elif "lab0" in h:                                      # stored labels: their count needs no device pass
            n_host = int(max(int(np.max(h["lab0"], initial=0)), int(np.max(h["lab1"], initial=0)))) + 1
        with self._stat_lock:
            self.stage_seconds["read"] += time.perf_counter() - t0
        return arena, pins, (h["pose_h1"], h["pose0"], h["pose1"]), n_host, trip[1:]

    # ---- label threads -------------------------------------------------------------------------------------------
    def _labels(self, dev, copied, poses, n_labels, key=None):
        import time
        t0 = time.perf_counter()
        tls = self._tls
        if getattr(tls, "stream", None) is None:
            torch.cuda.set_device(self.device)
            tls.stream = self._label_streams[next(self._lane_ids) % len(self._label_streams)]
            tls.word = torch.zeros(1, dtype=torch.int32).pin_memory()
        with torch.cuda.stream(tls.stream):
            tls.stream.wait_event(copied)
            for t in dev.values():
                t.record_stream(tls.stream)
            if "gm0" in dev:
                from .seflow.ssl_label import auto_labels
                l0, l1, top = auto_labels(dev["pc0"], dev["pc1"], dev["gm0"], dev["gm1"], poses[1], poses[2], return_top=True)
                tls.word.copy_(top.reshape(1), non_blocking=True)
                keep = self._cache is not None and self._cache_budget > 0
                if keep:                                       # a host copy for the later epochs, behind the same wait as the count
                    if getattr(tls, "host", None) is None or tls.host.numel() < l0.numel() + l1.numel():
                        tls.host = _lib.pinned_empty(int((l0.numel() + l1.numel()) * 1.25) + 16, torch.int32)
                    tls.host[:l0.numel()].copy_(l0, non_blocking=True)
                    tls.host[l0.numel():l0.numel() + l1.numel()].copy_(l1, non_blocking=True)
                counted = torch.cuda.Event()
                counted.record(tls.stream)
                counted.synchronize()                          # this thread's wait, not the training thread's
                n_labels = int(tls.word[0]) + 1
                if keep:
                    both = tls.host[:l0.numel() + l1.numel()].numpy().copy()
                    with self._stat_lock:
                        self._cache_budget -= both.nbytes
                        self._cache[key] = (both[:l0.numel()], both[l0.numel():], n_labels)
            else:
                l0, l1 = dev["lab0"], dev["lab1"]
            done = torch.cuda.Event()
            done.record(tls.stream)
        with self._stat_lock:
            self.stage_seconds["labels"] += time.perf_counter() - t0
            self.stage_seconds["samples"] += 1
        return (dev["pch1"], dev["pc0"], dev["pc1"], poses[0], poses[1], poses[2], l0, l1, n_labels), done

    # ---- feeder thread -------------------------------------------------------------------------------------------
    def _work(self):
        import collections
        import time
        try:
            torch.cuda.set_device(self.device)
            reads, labelled, nxt = collections.deque(), collections.deque(), 0
            while not self._stop:
                while nxt < len(self.trips) and len(reads) < self._window:
                    reads.append(self._pool.submit(self._read, self.trips[nxt]))
                    nxt += 1
                # keep ``lanes`` + 1 samples uploaded and in (or waiting for) the label stage, then hand the oldest on
                while reads and len(labelled) < self.lanes + 1:
                    arena, pins, poses, n_labels, key = reads.popleft().result()
                    t0 = time.perf_counter()
                    with torch.cuda.stream(self._copy):
                        # pinned -> HBM as ONE copy of the arena's used bytes (the sample's five arrays lie in one pinned block: five
                        # ~1 MB copies ran at 3 GB/s, one 6 MB copy at the link's rate), then device views at the same offsets
                        block = arena._buf[:arena._used].to(self.device, non_blocking=True)
                        base = arena._buf.data_ptr()
                        dev = {}
                        for k, v in pins.items():
                            lo = v.data_ptr() - base
                            dev[k] = block[lo:lo + v.numel() * v.element_size()].view(v.dtype).view(v.shape)
                        copied = torch.cuda.Event()
                        copied.record(self._copy)
                    self._free.put((arena, copied))
                    self._events.append(copied)
                    del self._events[:-(self._window + 2)]
                    with self._stat_lock:
                        self.stage_seconds["upload"] += time.perf_counter() - t0
                    labelled.append(self._labellers.submit(self._labels, dev, copied, poses, n_labels, key))
                    dev = None
                if not labelled:
                    break
                if not self._offer(labelled.popleft().result()):
                    break
        except BaseException as e:                             # surfaced on the consumer's thread
            self._error = e
        finally:
            self._pool.shutdown(wait=True, cancel_futures=True)
            self._labellers.shutdown(wait=True, cancel_futures=True)
            _return_arenas(self._arenas, self._events)
            self._offer(self._END)

    def _offer(self, item) -> bool:
        while not self._stop:
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def close(self) -> None:
        """Stop early (the consumer gave up, or ``max_steps`` ended the run): the workers return their pinned arenas and exit."""
        self._stop = True
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=30)

    def __iter__(self):
        while True:
            got = self._q.get()
            if got is self._END:
                if self._error is not None:
                    raise self._error
                return
            sample, done = got
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)                                # after the sample's copies and label kernels: no host wait
            seen = set()
            for t in sample:                                    # allocated on the feeder's streams, used on this one: ONE record per storage
                if isinstance(t, torch.Tensor):                 # (the sweeps -- and uploaded labels -- are views of one block; a record costs
                    key = t.untyped_storage().data_ptr()        #  ~0.1 ms of the launch thread)
                    if key not in seen:
                        seen.add(key)
                        t.record_stream(cur)
            yield sample


class EvalFeeder(BatchFeeder):
    """``eval.EvalBatch`` objects for the evaluator / scorer.  ``source`` yields lists of frame dicts (one list = one batch),
    or ``(frames, comp_dis_list)`` pairs for the zip mode (eval.py:303-304)."""

    def __init__(self, source, res_name: str = "", device=None, depth: int = 2, stage_threads: int = 4):
        from .eval import EvalBatch
        dev = device if device is not None else _lib.require_gpu()

        def build(item, upload):
            frames, comp_dis = item if isinstance(item, tuple) else (item, None)
            eb = EvalBatch.from_frames(list(frames), res_name, comp_dis, device=dev, upload=upload)
            b = eb.batch
            return eb, [b.offsets, b.pose0, b.pose1, b.pc0, b.lidar_dt, b.gm0, b.flow_is_valid, eb.gt, eb.category, eb.instance, eb.est]
        super().__init__(source, build, device=dev, depth=depth, stage_threads=stage_threads)
