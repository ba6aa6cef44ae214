"""The per-frame motion-compensation pipeline of north_star on one MI355X:

    3 sweeps (pch1, pc0, pc1) --voxelise + SeFlow++-style network (a10)--> per-point flow incl. ego motion
                              --ego-motion removal, dt0, flow2compDis (a1-a4)--> comp_dis (N,3) float32

i.e. what the reference does in two programs: ``OpenSceneFlow/save.py`` (README.md:50; absent
submodule) writing ``<res_name>`` into the scene h5, then ``save_zip.py`` (save_zip.py:102-125)
turning it into compensation distances.  Here the flow never leaves HBM: the network head writes
each frame's flow into its rows of one ragged batch buffer and a single fused comp_dis launch
finishes the whole batch.
"""
from __future__ import annotations

from dataclasses import dataclass

import os

import numpy as np
import torch

from . import _lib
from .compdis import CompDisEngine, FrameBatch
from .seflow.model import SeFlowNet


@dataclass
class Sample:
    """One network input resident in HBM.  Sweeps are (N,>=3) float32 rows; poses are host 4x4 float64."""
    pch1: torch.Tensor
    pc0: torch.Tensor
    pc1: torch.Tensor
    pose_h1: np.ndarray
    pose0: np.ndarray
    pose1: np.ndarray
    lidar_dt: torch.Tensor            # (N0,) float32 of pc0
    scene_id: str = ""
    timestamp: int = 0

    @classmethod
    def from_frames(cls, fh: dict, f0: dict, f1: dict | None = None, device=None):
        """Frame dicts (reference layout) -> Sample.  ``f1`` defaults to the ``pc1`` / ``pose1`` stored in ``f0``."""
        dev = device if device is not None else _lib.require_gpu()
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        pc1 = f1["pc0"] if f1 is not None else f0["pc1"]
        return cls(up(fh["pc0"]), up(f0["pc0"]), up(pc1), np.asarray(fh["pose0"], np.float64), np.asarray(f0["pose0"], np.float64),
                   np.asarray(f0["pose1"], np.float64), up(f0["lidar_dt"]), f0.get("scene_id", ""), int(f0.get("timestamp", 0)))


class HiMoPipeline:
    def __init__(self, net: SeFlowNet | None = None, device=None, max_points: int = 140_000, max_batch: int = 8,
                 precision: str = "auto", params: dict | None = None):
        """``net``: a ready network, or None to build one from ``params`` (random-init when None; ``max_batch`` samples per
        backbone launch).  ``precision``: "bf16x3" | "f16x2" | "f32" (see SeFlowNet) or "auto": start in the fast fp16
        split, check every batch's guard words (a non-finite flow value = an activation left fp16's range; a split-output
        layer without a single value above 2^-6 = activations on the split's absolute floor) and, the first time one fires,
        rebuild the network in the bf16 split (float32 range and relative precision), redo that batch and stay there."""
        self.device = device if device is not None else _lib.require_gpu()
        self.auto = net is None and precision == "auto"
        self._net_args = dict(params=params, device=self.device, max_points=max_points, max_batch=max_batch)
        self.net = net if net is not None else SeFlowNet(precision="f16x2" if self.auto else precision, **self._net_args)
        self.compdis = CompDisEngine(device=self.device)
        self._batch = None
        self._bufs, self._turn = {}, 0
        self._finite = None          # device flag of the previous batch (fp16-split precision only)

    def _upload_small(self, host_bytes: np.ndarray) -> torch.Tensor:
        """uint8 array -> device, asynchronously, through a ring of pinned blocks (reused once their copy has completed)"""
        if not hasattr(self, "_pin_ring"):
            self._pin_ring, self._pin_next = [[torch.empty(1 << 16, dtype=torch.uint8, pin_memory=True), None] for _ in range(4)], 0
        n = host_bytes.size
        slot = self._pin_ring[self._pin_next]
        self._pin_next = (self._pin_next + 1) % len(self._pin_ring)
        if slot[0].numel() < n:
            slot[0] = _lib.pinned_empty(n)
        elif slot[1] is not None:
            slot[1].synchronize()
        np.copyto(slot[0].numpy()[:n], host_bytes)
        out = slot[0][:n].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        return out

    def _rows(self, name: str, total: int, width: int | None, dtype=torch.float32) -> torch.Tensor:
        """First ``total`` rows of a grow-only device buffer (reused from batch to batch; never keyed on object identity)."""
        key = (name, width, self._turn)
        buf = self._bufs.get(key)
        if buf is None or buf.shape[0] < total:
            cap = max(total, int(1.25 * buf.shape[0]) if buf is not None else 0)
            buf = torch.empty((cap,) if width is None else (cap, width), dtype=dtype, device=self.device)
            self._bufs[key] = buf
        return buf[:total]

    def _batch_for(self, samples) -> FrameBatch:
        """Ragged batch container over the pc0 sweeps of ``samples``.  Rebuilt on EVERY call -- the sweeps are copied into
        the batch buffers and the offsets / poses uploaded again -- because nothing cheap identifies "the same samples":
        CPython reuses ``id()`` values as soon as an object is freed, so a streaming caller that builds its Samples on the
        fly would otherwise be handed the previous batch's points and poses.  Only the device BUFFERS are reused: two sets
        that alternate, so the tensors ``run`` returned for batch k stay intact while batch k+1 is in flight (a caller
        draining results on a side stream) and are overwritten by batch k+2."""
        self._turn ^= 1
        counts = [int(s.pc0.shape[0]) for s in samples]
        offsets = np.zeros(len(samples) + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        T, F = int(offsets[-1]), len(samples)
        width = int(samples[0].pc0.shape[1])
        # offsets and poses go up in ONE asynchronous copy from a small pinned ring: a pageable upload would block the
        # launch thread until the previous batch's kernels have drained (a bubble per batch)
        meta = np.concatenate([offsets.view(np.uint8), np.stack([s.pose0 for s in samples]).astype(np.float64).reshape(-1).view(np.uint8),
                               np.stack([s.pose1 for s in samples]).astype(np.float64).reshape(-1).view(np.uint8)])
        meta_dev = self._upload_small(meta)
        o_end, p_len = (F + 1) * 8, F * 128
        pc0, dt = self._rows("pc0", T, width), self._rows("lidar_dt", T, None)
        if F:
            torch.cat([s.pc0 for s in samples], dim=0, out=pc0)
            torch.cat([s.lidar_dt for s in samples], dim=0, out=dt)
        self._batch = FrameBatch(
            offsets_host=offsets, offsets=meta_dev[:o_end].view(torch.int64),
            pose0=meta_dev[o_end:o_end + p_len].view(torch.float64).view(F, 4, 4),
            pose1=meta_dev[o_end + p_len:o_end + 2 * p_len].view(torch.float64).view(F, 4, 4),
            pc0=pc0, lidar_dt=dt, flow=self._rows("flow", T, 3),
            meta=[(s.scene_id, s.timestamp) for s in samples])
        return self._batch

    def _forward(self, samples, outs) -> None:
        """Network forward for ``samples`` into the per-sample (N0_k,3) views ``outs``, ``max_batch`` samples per backbone launch."""
        mb = self.net.max_batch
        for lo in range(0, len(samples), mb):
            grp = samples[lo:lo + mb]
            self.net.forward_batch([(s.pch1, s.pc0, s.pc1, s.pose_h1, s.pose0, s.pose1) for s in grp], outs[lo:lo + mb])

    def _fall_back(self):
        """precision="auto": leave the fp16 split for the bf16 split (float32 range) for good"""
        self.net = SeFlowNet(precision="bf16x3", **self._net_args)

    def _guard_begin(self):
        """fp16 split: zero the network's finite-flow word (the fused head ORs 1 into it when it writes a NaN / inf flow value,
        csrc/gruhead.hip) -- stream-ordered, no framework kernel, no host sync"""
        if self.net.precision == "f16x2":
            if not self.net.fused_head:
                raise RuntimeError("precision='f16x2' needs the fused head: its output stage carries the finite-flow guard")
            self.net.clear_nonfinite()

    def _guard_now(self) -> bool:
        """True when the launches since ``_guard_begin`` wrote only finite flow values AND every split-output layer produced
        values above the fp16 split's absolute floor (one small read back: a host sync); the reason otherwise in ``_guard_why``"""
        if self.net.precision != "f16x2":
            return True
        self._guard_why = self.net.guard_verdict(self.net.guard.cpu().tolist())
        return self._guard_why is None

    def _guard_later(self):
        """queue the read-back of the guard words (into pinned memory + an event) for ``sync_check`` one batch later"""
        if self.net.precision != "f16x2":
            return
        if not hasattr(self, "_flag_ring"):
            self._flag_ring, self._flag_next = [torch.zeros(self.net.guard.numel(), dtype=torch.int32).pin_memory() for _ in range(2)], 0
        host = self._flag_ring[self._flag_next]
        self._flag_next ^= 1
        host.copy_(self.net.guard, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._finite = (host, ev, self.net)

    @staticmethod
    def _guard_error(why: str) -> FloatingPointError:
        if why == "overflow":
            return FloatingPointError("non-finite flow: activations left the fp16 range of precision='f16x2'; "
                                      "use precision='bf16x3' (or 'auto') for these weights")
        return FloatingPointError("a layer's activations stayed below 2^-6, on the absolute floor (2^-25) of precision='f16x2''s "
                                  "two-term fp16 split: the flow may miss 1e-4; use precision='bf16x3' (or 'auto') for these weights")

    def flows(self, samples) -> list:
        """Network only, for a list of samples: [(N0_k,3) flow incl. ego motion] -- the h5 ``<res_name>`` payload that
        ``save.run`` writes.  In the fp16 split every call checks its flows for non-finite values BEFORE returning them (one
        host sync per call): ``precision="auto"`` redoes the batch in the bf16 split and stays there, an explicit
        ``"f16x2"`` raises FloatingPointError -- an overflowed activation never reaches a result file."""
        _, outs = self._flows_launch(samples)
        return self._flows_check(samples, outs)

    def _flows_launch(self, samples):
        counts = [int(s.pc0.shape[0]) for s in samples]
        flat = torch.empty((sum(counts), 3), dtype=torch.float32, device=self.device)
        outs = list(torch.split(flat, counts)) if counts else []
        self._guard_begin()
        self._forward(samples, outs)
        return flat, outs

    def _flows_check(self, samples, outs):
        if not self._guard_now():
            if not self.auto:
                raise self._guard_error(self._guard_why)
            self._fall_back()
            self._forward(samples, outs)
        return outs

    def sync_check(self):
        """fp16-split precision only: an activation beyond fp16's range (65504) turns into NaN at the next layer's
        split and reaches the flow, and a layer whose activations all sit below 2^-6 has lost the split's relative
        precision; this raises instead of handing such a batch on.  Checked one batch late by
        ``run`` (so it never stalls the stream) and by the caller after the last batch."""
        if self._finite is not None:
            (host, ev, net), self._finite = self._finite, None
            ev.synchronize()
            why = net.guard_verdict(host.tolist())
            if why is not None:
                raise self._guard_error(why)

    def flow(self, s: Sample, out: torch.Tensor | None = None) -> torch.Tensor:
        """Network only: (N0,3) flow including ego motion (the h5 ``<res_name>`` payload)."""
        return self.net.forward_device(s.pch1, s.pc0, s.pc1, s.pose_h1, s.pose0, s.pose1, out=out)

    def run(self, samples, sensor_dt: float = 0.1, refined: bool = False, copy: bool = False) -> dict:
        """flow + comp_dis for a list of samples; asynchronous on the current stream.
        Returns {"flow", "comp_dis"[, "refined"]} as (T,3) tensors plus "batch" for splitting per frame.

        LIFETIME of the returned tensors: they are views of two alternating grow-only buffer sets, valid until the
        NEXT-BUT-ONE ``run`` overwrites them (batch k's results survive batch k+1 -- a caller draining on a side stream --
        and are gone after batch k+2 is launched).  A caller that collects results over many batches
        (``outs = [pipe.run(b) for b in batches]``) must pass ``copy=True`` (fresh tensors, one device copy per output) or
        clone what it keeps."""
        batch = self._batch_for(samples)
        o = batch.offsets_host
        self.sync_check()                                    # the PREVIOUS batch's flag: no stall on this one

        outs = [batch.flow[int(o[k]):int(o[k + 1])] for k in range(len(samples))]
        self._guard_begin()
        self._forward(samples, outs)
        if not self.auto:
            self._guard_later()                                  # checked one batch late (sync_check)
        elif not self._guard_now():                              # auto: checked now (one host sync per batch)
            self._fall_back()
            self._forward(samples, outs)
        T = batch.total_points
        out = {"comp_dis": self._rows("comp_dis", T, 3)}
        if refined:
            out["refined"] = self._rows("refined", T, 3)
        res = self.compdis.run(batch, sensor_dt=sensor_dt, refined=refined, out=out)
        out = {"flow": batch.flow, "comp_dis": res["comp_dis"], "refined": res.get("refined"), "batch": batch}
        if copy:
            out.update({k: out[k].clone() for k in ("flow", "comp_dis", "refined") if out[k] is not None})
        return out


_BATCH_STREAMS = {}


def batch_streams(device, n: int) -> list:
    """The first ``n`` of the process's streams for batches in flight (``OverlappedPipeline``, ``fastnsf.OverlappedFastNSF``): ONE list per
    device and process.  The HIP runtime multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES: 8) and streams
    that share a queue serialise; a process that builds pipeline after pipeline (bench.py's precision legs, a sweep over checkpoints)
    would otherwise walk through torch's stream pool until unrelated streams collide (``seflow.train.side_streams`` has a measurement).
    Two overlapped objects driven at the same time share these streams -- their batches then take turns instead of overlapping."""
    if os.environ.get("HIMO_SHARED_BATCH_STREAMS", "1") == "0":          # (A/B knob: a fresh set per object, as before round 6)
        return [torch.cuda.Stream(device=device) for _ in range(n)]
    key = (device.type, device.index)
    have = _BATCH_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


class OverlappedPipeline:
    """Several batches in flight (default three; two gave +4.5 %, the third +1 %): ``in_flight`` ``HiMoPipeline``s (own network
    buffers) fed in turn on as many HIP streams, so that one batch's
    latency-bound stages (pillar stage, stride-2 / 1x1 layers, upsampling, the head's gather) run under the other's matrix-bound
    3x3 convolutions -- measured +4 % frames/s in round 3 (profiles/r03_exp_two_streams.txt) and adopted as the default way to
    run a stream of batches in round 4.  Same kernels, same launch order within a batch: results are bit-identical to the
    single-stream pipeline's (tests/test_pipeline_gpu.py).

    ``run`` returns at once; the result dict carries ``"ready"`` (a HIP event recorded on the batch's stream).  Consume results
    on a SIDE stream after ``ready`` (``feeder.ResultDrain`` does) or call ``wait(result)`` to make the current stream wait --
    which also orders every later ``run`` behind this batch, i.e. a caller that consumes each result on the launch stream before
    launching the next gets correct results and no overlap.  Result tensors of a batch stay valid until that inner pipeline's
    next-but-one batch, i.e. for 2 x ``in_flight`` ``run`` calls.  Inputs must be ready on the calling stream (``run`` makes the batch's
    stream wait for the calling stream as of the call)."""

    def __init__(self, params: dict | None = None, device=None, max_points: int = 140_000, max_batch: int = 8, precision: str = "auto",
                 in_flight: int = 3, nets=None):
        self.device = device if device is not None else _lib.require_gpu()
        if nets is not None:
            self.pipes = [HiMoPipeline(net, device=self.device) for net in nets]
        else:
            self.pipes = [HiMoPipeline(None, device=self.device, max_points=max_points, max_batch=max_batch, precision=precision, params=params)
                          for _ in range(in_flight)]
        for p in self.pipes[1:]:                                # one tuning table: a layer shape is timed once, by whoever meets it first
            if hasattr(p.net, "tiles") and hasattr(self.pipes[0].net, "tiles"):
                p.net.tiles = self.pipes[0].net.tiles       # (the tile variants of a layer are bit-identical: tests/test_seflow_gpu.py)
        self.streams = batch_streams(self.device, len(self.pipes))
        self._turn = 0

    @property
    def net(self):
        return self.pipes[0].net

    def _next(self):
        i = self._turn
        self._turn = (self._turn + 1) % len(self.pipes)
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream(self.device))
        return self.pipes[i], st

    def run(self, samples, **kw) -> dict:
        pipe, st = self._next()
        with torch.cuda.stream(st):
            out = pipe.run(samples, **kw)
            ev = torch.cuda.Event()
            ev.record(st)
        # the batch's sweeps are read on ``st`` (packing, pillarisation), not on the stream they were allocated or handed over on: tell
        # the allocator, or a caller that drops its Samples right after this call (a feeder loop) could see their memory handed out
        # again -- and overwritten by the next upload -- before this batch has read it.  One call per distinct storage (a fed batch: one).
        seen = set()
        for smp in samples:
            for t in (smp.pch1, smp.pc0, smp.pc1, smp.lidar_dt):
                if t.is_cuda and t.numel():
                    key = t.untyped_storage().data_ptr()
                    if key not in seen:
                        seen.add(key)
                        t.record_stream(st)
        out["ready"], out["stream"] = ev, st
        return out

    def wait(self, result: dict) -> dict:
        torch.cuda.current_stream(self.device).wait_event(result["ready"])
        return result

    def flows_stream(self, batches):
        """``HiMoPipeline.flows`` over an iterable of sample lists with ``in_flight - 1`` batches of lookahead: yields (samples,
        flows) in order.  Batch k's finite-flow check (a host read-back: a sync of ITS stream) happens while the following batches
        are already queued on the other streams, so the device does not idle through it."""
        pending = []                                            # launched, not yet checked: at most one per inner pipeline
        for samples in batches:
            pipe, st = self._next()                             # (round robin: the batch that used this pipeline last was finished below)
            with torch.cuda.stream(st):
                flat, outs = pipe._flows_launch(samples)
            pending.append((samples, pipe, st, flat, outs))
            if len(pending) == len(self.pipes):
                done = pending.pop(0)
                yield done[0], self._flows_finish(*done)
        for done in pending:
            yield done[0], self._flows_finish(*done)

    def _flows_finish(self, samples, pipe, st, flat, outs):
        with torch.cuda.stream(st):
            outs = pipe._flows_check(samples, outs)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(st)
        flat.record_stream(cur)                                 # allocated on the batch's stream, consumed on the caller's
        return outs

    def sync_check(self):
        for p in self.pipes:
            p.sync_check()
