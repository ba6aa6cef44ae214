"""Drop-in for the reference's ``eval.py``: per-instance motion-compensation metrics (MPE and
Chamfer distance error) for CAR / OTHER_VEHICLES, bucketed by speed and range.

    InstanceMetrics(data_name, sensor_hz)      eval.py:24-48
      .cal_chamfer(pc1, pc2)                   eval.py:50-62    (exact 1-NN on the GPU, nn.hip)
      .step_eval(pc, gt_flow, pc_dt0, gt_category, gt_instance, est_flow=None, est_dis=None)   eval.py:64-149
      .print(res_name, file_name)              eval.py:151-268  (same table, same res-<data>.json)
    main(data_dir, res_name, comp_dis_zip)     eval.py:270-312

What runs where: the per-point chain, the per-instance grouping, both nearest-neighbour searches
and the per-instance means run in HIP kernels (csrc/evalmetrics.hip, csrc/nn.hip) for a whole batch
of sweeps at once (``step_frames``); what stays on the host is the bucket bookkeeping of
eval.py:99-147 over the few dozen per-instance records of each sweep, done in float64 in the
reference's own order so that sharding sweeps over GPUs reproduces the single-GPU numbers exactly.
"""
from __future__ import annotations

import ctypes
import json
import os
import time

import numpy as np
import torch

from . import _lib
from .compdis import CLOSE_DISTANCE_THRESHOLD, CompDisEngine, FrameBatch  # noqa: F401

# Argoverse-2 annotation taxonomy (alphabetical; index 0 = "NONE").  The reference imports these
# tables from its absent submodule (eval.py:21); its in-tree copy is tools/test/score.py:29-94.
ANNOTATION_CATEGORIES = (
    "ANIMAL ARTICULATED_BUS BICYCLE BICYCLIST BOLLARD BOX_TRUCK BUS CONSTRUCTION_BARREL CONSTRUCTION_CONE DOG "
    "LARGE_VEHICLE MESSAGE_BOARD_TRAILER MOBILE_PEDESTRIAN_CROSSING_SIGN MOTORCYCLE MOTORCYCLIST OFFICIAL_SIGNALER "
    "PEDESTRIAN RAILED_VEHICLE REGULAR_VEHICLE SCHOOL_BUS SIGN STOP_SIGN STROLLER TRAFFIC_LIGHT_TRAILER TRUCK "
    "TRUCK_CAB VEHICULAR_TRAILER WHEELCHAIR WHEELED_DEVICE WHEELED_RIDER").split()
CATEGORY_TO_INDEX = {"NONE": 0, **{c: i + 1 for i, c in enumerate(ANNOTATION_CATEGORIES)}}
BUCKETED_METACATAGORIES = {
    "BACKGROUND": ["NONE"],
    "CAR": ["REGULAR_VEHICLE"],
    "PEDESTRIAN": ["PEDESTRIAN", "STROLLER", "WHEELCHAIR", "OFFICIAL_SIGNALER"],
    "WHEELED_VRU": ["BICYCLE", "BICYCLIST", "MOTORCYCLE", "MOTORCYCLIST", "WHEELED_DEVICE", "WHEELED_RIDER"],
    "OTHER_VEHICLES": ["BOX_TRUCK", "LARGE_VEHICLE", "RAILED_VEHICLE", "TRUCK", "TRUCK_CAB", "VEHICULAR_TRAILER",
                       "ARTICULATED_BUS", "BUS", "SCHOOL_BUS"],
}
EVAL_GROUPS = ("CAR", "OTHER_VEHICLES")          # eval.py:75; group ids 1, 2 on the device
RANGES = ("0-10", "10-20", "20-30", "30+")

MODE_FLOW, MODE_COMPDIS, MODE_RAW, MODE_SCORE, MODE_DIRECT = range(5)
DIRECT_EST_IS_DIS = 0x100


class InstanceRecord(ctypes.Structure):
    """include/himo_amd.h: himo_instance_record."""
    _fields_ = [("frame", ctypes.c_int32), ("group", ctypes.c_int32), ("instance", ctypes.c_int64),
                ("num_pts", ctypes.c_int64), ("vel", ctypes.c_double), ("dis", ctypes.c_double),
                ("mpe", ctypes.c_double), ("cham", ctypes.c_double)]


RECORD_DTYPE = np.dtype([("frame", "<i4"), ("group", "<i4"), ("instance", "<i8"), ("num_pts", "<i8"),
                         ("vel", "<f8"), ("dis", "<f8"), ("mpe", "<f8"), ("cham", "<f8")])
assert RECORD_DTYPE.itemsize == ctypes.sizeof(InstanceRecord) == 56


def class_lut() -> np.ndarray:
    lut = np.zeros(256, dtype=np.uint8)
    for gid, name in enumerate(EVAL_GROUPS, start=1):
        for cat in BUCKETED_METACATAGORIES[name]:
            lut[CATEGORY_TO_INDEX[cat]] = gid
    return lut


def range_name_of(value):
    """eval.py:100-108."""
    if 0 < value < 10:
        return "0-10"
    if 10 <= value < 20:
        return "10-20"
    if 20 <= value < 30:
        return "20-30"
    if value >= 30:
        return "30+"
    return None


class InstanceEvaluator:
    """Device side: ragged batch of sweeps -> per-instance records (one numpy structured array)."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        self.device = device if device is not None else _lib.require_gpu()
        self._ws = None
        self._lut = class_lut()

    def _workspace(self, n_frames, total, max_records):
        need = int(self.lib.himo_eval_workspace_bytes(n_frames, total, max_records))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need + 64, dtype=torch.uint8, device=self.device)
        return self._ws

    def launch(self, n_frames, offsets, pc0, gt, est, lidar_dt, category, instance, eval_mask, mode, sensor_dt=0.1,
               pose0=None, pose1=None, flags=0, max_records=None):
        """Enqueue the kernels for one ragged batch and the device -> pinned-host copy of its counts + records; returns a
        ticket for ``collect``.  Nothing here waits for the device, so the host can stage / launch the next batch (or
        digest the previous one's records) while this one runs."""
        total = int(category.shape[0])
        if max_records is None:
            max_records = max(1024, min(total, 64 * n_frames + 4096))
        nbytes = max_records * RECORD_DTYPE.itemsize
        recs = torch.empty(16 + nbytes, dtype=torch.uint8, device=self.device)       # [counts (2 x int64) | records]
        recs[:16].zero_()
        counts = recs[:16].view(torch.int64)
        ws = self._workspace(n_frames, total, max_records)
        lut = self._lut.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        args = (n_frames, total, _lib.ptr(offsets), _lib.ptr(pose0), _lib.ptr(pose1), _lib.ptr(pc0),
                0 if pc0 is None else pc0.shape[1], _lib.ptr(gt), _lib.ptr(est), _lib.ptr(lidar_dt), _lib.ptr(category),
                _lib.ptr(instance), _lib.ptr(eval_mask), lut, float(sensor_dt), int(mode), int(flags))
        st = self.lib.himo_eval_instances(*args, recs[16:].data_ptr(), max_records, counts.data_ptr(), _lib.ptr(ws), ws.numel(),
                                          _lib.stream_handle())
        _lib.check(st, "himo_eval_instances")
        host = self._pinned(16 + nbytes)
        host[:16 + nbytes].copy_(recs, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        keep = (offsets, pc0, gt, est, lidar_dt, category, instance, eval_mask, pose0, pose1, recs)     # alive until collected
        return {"event": ev, "host": host, "max_records": max_records, "args": args, "keep": keep, "workspace": ws}

    def _pinned(self, nbytes: int) -> torch.Tensor:
        if not hasattr(self, "_pin_pool"):
            self._pin_pool = []
        for i, buf in enumerate(self._pin_pool):
            if buf.numel() >= nbytes:
                return self._pin_pool.pop(i)
        return _lib.pinned_empty(nbytes)

    def collect(self, ticket) -> np.ndarray:
        """Wait for a ticket's copy and return its records sorted (frame, group, instance); a batch with more instances
        than the buffer had room for is run again, synchronously, with room for all."""
        ticket["event"].synchronize()
        host = ticket["host"]
        n_sel, n_rec = (int(v) for v in host[:16].view(torch.int64).tolist())
        if n_rec > ticket["max_records"]:
            a = ticket["args"]
            k = ticket["keep"]
            self._pin_pool.append(host)
            again = self.launch(a[0], k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], a[15], sensor_dt=a[14], pose0=k[8], pose1=k[9],
                                flags=a[16], max_records=n_rec)
            return self.collect(again)
        out = host[16:16 + n_rec * RECORD_DTYPE.itemsize].numpy().view(RECORD_DTYPE).copy()
        self._pin_pool.append(host)
        ticket["keep"] = None
        return np.sort(out, order=["frame", "group", "instance"])    # np.unique order inside each class

    def run(self, n_frames, offsets, pc0, gt, est, lidar_dt, category, instance, eval_mask, mode, sensor_dt=0.1,
            pose0=None, pose1=None, flags=0, max_records=None) -> np.ndarray:
        return self.collect(self.launch(n_frames, offsets, pc0, gt, est, lidar_dt, category, instance, eval_mask, mode,
                                        sensor_dt=sensor_dt, pose0=pose0, pose1=pose1, flags=flags, max_records=max_records))


def _dev(x, dtype, dev):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=dev, dtype=dtype).contiguous()


class EvalBatch:
    """Everything the loop body of eval.py:281-310 reads from a list of frames, resident in HBM: the ragged sweep batch
    (points, poses, lidar_dt, gm0 [, flow_is_valid]) plus ground-truth flow, category, instance id and the estimate
    (``est``: the flow stored under <res_name>, a zip's comp_dis, or None for "raw")."""

    def __init__(self, batch: FrameBatch, gt, category, instance, est, mode: int):
        self.batch, self.gt, self.category, self.instance, self.est, self.mode = batch, gt, category, instance, est, mode

    @classmethod
    def from_frames(cls, frames, res_name: str = "", comp_dis=None, device=None, upload=None):
        """Pack host frame dicts.  ``upload(parts, dtype) -> device tensor`` lets a feeder concatenate straight into pinned
        memory and copy on its own stream (feeder.EvalFeeder); the default is a host concatenation + a synchronous copy."""
        from .compdis import host_upload
        dev = device if device is not None else _lib.require_gpu()
        up = upload if upload is not None else host_upload(dev)
        frames = list(frames)
        batch = FrameBatch.from_frames(frames, "raw", device=dev, with_masks=True, upload=up)
        cat = lambda key, dt: up([np.asarray(f[key]) for f in frames], dt)
        gt = cat("flow", np.float32)
        category = cat("flow_category_indices", np.uint8)
        instance = cat("flow_instance_id", np.int64)
        if comp_dis is not None:
            est, mode = up([np.asarray(c) for c in comp_dis], np.float32), MODE_COMPDIS
        elif res_name == "raw":
            est, mode = None, MODE_RAW
        else:
            est, mode = cat(res_name, np.float32), MODE_FLOW          # KeyError like data[res_name]
        return cls(batch, gt, category, instance, est, mode)


class InstanceMetrics:
    def __init__(self, data_name, sensor_hz=10.0):
        self.frame_cnt = 0
        self.sensor_dt = 1.0 / sensor_hz
        self.data_name = data_name
        # eval.py:30-36: Scania labels are noisy for slow objects; elsewhere slow objects are undistorted
        self.min_vel = 1.5 if data_name in ["scania"] else 3.0
        self.evaluate_data = self.init_evaluate_data()
        self._evaluator = None
        self._compdis = None
        self._log = []          # (sweep key, the sweep's records) so that ranks can merge in sweep order
        self._pending = []      # tickets of batches whose records have not been digested yet (step_batch)

    # ---- containers --------------------------------------------------------------------------------
    def init_evaluate_data(self):
        new = lambda: {"num_pts": [], "mpe": [], "cham": [], "std_mpe": [], "std_cham": []}
        return {c: {"vel": {r: new() for r in RANGES}, "dis": {r: new() for r in RANGES}, "mean": new()}
                for c in EVAL_GROUPS}

    @property
    def evaluator(self) -> InstanceEvaluator:
        if self._evaluator is None:
            self._evaluator = InstanceEvaluator()
        return self._evaluator

    # ---- a8 ------------------------------------------------------------------------------------------
    def cal_chamfer(self, pc1, pc2) -> float:
        """(mean NN(pc1->pc2) + mean NN(pc2->pc1)) / 2, Euclidean, k = 1; NaN for an empty set."""
        if len(pc1) == 0 or len(pc2) == 0:
            return float("nan")
        return chamfer_distance(pc1, pc2)

    # ---- a7: the reference's own entry point --------------------------------------------------------
    def step_eval(self, pc, gt_flow, pc_dt0, gt_category, gt_instance, est_flow=None, est_dis=None):
        """Same arguments as eval.py:64 (already masked, ego-motion-free arrays of ONE sweep)."""
        self.flush()                                   # batches still in flight keep their place in the sweep order
        dev = self.evaluator.device
        n = len(pc)
        if est_flow is None and est_dis is None:
            raise UnboundLocalError("local variable 'refine_pc' referenced before assignment")   # eval.py:74
        pc_t = _dev(pc, torch.float32, dev)
        gt_t = _dev(gt_flow, torch.float64, dev)
        dt_t = _dev(pc_dt0, torch.float32, dev)
        cat_t = _dev(gt_category, torch.uint8, dev)
        ins_t = _dev(np.asarray(gt_instance).astype(np.int64), torch.int64, dev)
        if est_flow is not None:
            est_t, flags = _dev(est_flow, torch.float64, dev), 0
        else:
            est_t, flags = _dev(est_dis, torch.float32, dev), DIRECT_EST_IS_DIS
        off = torch.tensor([0, n], dtype=torch.int64, device=dev)
        mask = torch.ones(n, dtype=torch.uint8, device=dev)
        recs = self.evaluator.run(1, off, pc_t, gt_t, est_t, dt_t, cat_t, ins_t, mask, MODE_DIRECT,
                                  sensor_dt=self.sensor_dt, flags=flags) if n else np.empty(0, RECORD_DTYPE)
        self._accumulate_frame(recs, key=self.frame_cnt)

    # ---- batched fast path: everything on the device --------------------------------------------------------------
    def step_batch(self, eb: EvalBatch, keys=None, depth: int = 2):
        """The loop body of eval.py:281-310 for a device-resident batch of sweeps: eval mask (fused comp_dis kernel),
        per-instance grouping, both NN searches and the means are enqueued and the call returns; the few hundred bytes of
        per-instance records come back through pinned memory and are digested -- in sweep order -- ``depth`` batches
        later, so the host-side bucket bookkeeping of batch k overlaps the kernels of batch k + 1.  ``flush()`` digests
        what is still in flight (``print`` / ``summary`` / ``gather`` call it)."""
        ev = self.evaluator
        if self._compdis is None:
            self._compdis = CompDisEngine(device=ev.device)
        b = eb.batch
        mask = self._compdis.run(b, sensor_dt=self.sensor_dt, data_name=self.data_name)["eval_mask"]   # eval.py:288-296
        ticket = ev.launch(b.n_frames, b.offsets, b.pc0, eb.gt, eb.est, b.lidar_dt, eb.category, eb.instance, mask, eb.mode,
                           sensor_dt=self.sensor_dt, pose0=b.pose0, pose1=b.pose1)
        base = self.frame_cnt + sum(t[1] for t in self._pending)
        self._pending.append((ticket, b.n_frames, list(keys) if keys is not None else [base + k for k in range(b.n_frames)]))
        while len(self._pending) > depth:
            self._digest(self._pending.pop(0))

    def _digest(self, item):
        ticket, n_frames, keys = item
        recs = self.evaluator.collect(ticket)
        bounds = np.searchsorted(recs["frame"], np.arange(n_frames + 1))
        for k in range(n_frames):
            self._accumulate_frame(recs[bounds[k]:bounds[k + 1]], key=keys[k])

    def flush(self):
        while self._pending:
            self._digest(self._pending.pop(0))

    def step_frames(self, frames, res_name: str = "", comp_dis=None, keys=None):
        """Host frame dicts in: ``comp_dis`` = optional list of (N,3) float32 arrays read from a zip (EVAL_FLAG 1);
        otherwise the flow stored under ``res_name`` (or zeros for "raw") is compensated on the fly (EVAL_FLAG 2).
        Synchronous convenience form of ``step_batch`` (packs, uploads, digests); streams of frames should go through
        ``feeder.EvalFeeder`` + ``step_batch``."""
        frames = list(frames)
        if not frames:
            return
        self.flush()
        self.step_batch(EvalBatch.from_frames(frames, res_name, comp_dis, device=self.evaluator.device), keys=keys)
        self.flush()

    # ---- host bookkeeping: eval.py:75-147 on the per-instance records of ONE sweep ---------------------------
    # The reference forms a sweep's bucket means with np.average / np.nanmean / np.nanstd on lists of a handful of floats; through
    # numpy each call costs ~8 us of dispatch, ~0.35 ms per sweep -- as much as the sweep's device work, and it holds the interpreter
    # lock the feeder threads need.  The three helpers below return numpy's BITS for the short, NaN-free lists that occur (numpy sums
    # fewer than 8 float64 values left to right; from 8 on it keeps eight partial sums -- those lists go to numpy itself, as does
    # anything holding a NaN): checked against numpy over random lists in tests/test_host_logic.py and, end to end, by the parity tests
    # against the reference's own eval.py output (tests/test_eval_gpu.py).
    @staticmethod
    def _np_sum(xs):
        total = xs[0]
        for v in xs[1:]:
            total += v
        return total

    @classmethod
    def _average(cls, values, weights):
        """np.average(values, weights=weights) for float values and int weights"""
        n = len(values)
        if n == 1:                                                    # (most buckets of a sweep hold one instance)
            v, w = values[0], float(weights[0])
            if v == v and w != 0.0:
                return v * w / w
            return np.average(values, weights=weights)
        if n > 7:
            return np.average(values, weights=weights)
        scl = tot = None
        for v, w in zip(values, weights):
            if v != v:
                return np.average(values, weights=weights)
            w = float(w)
            if scl is None:
                scl, tot = w, v * w
            else:
                scl += w
                tot += v * w
        if scl == 0.0:
            return np.average(values, weights=weights)            # (numpy's ZeroDivisionError)
        return tot / scl

    @classmethod
    def _nanmean_nanstd(cls, values):
        """(float(np.nanmean(values)), float(np.nanstd(values))) for a list of float64 scalars"""
        n = len(values)
        if n == 1:
            v = float(values[0])
            if v == v:
                return v / 1, float(np.sqrt(np.float64((v - v) * (v - v) / 1)))
        if n == 0 or n > 7:
            return float(np.nanmean(values)), float(np.nanstd(values))
        vals = []
        for v in values:
            if v != v:
                return float(np.nanmean(values)), float(np.nanstd(values))
            vals.append(float(v))
        mean = cls._np_sum(vals) / n
        dev = [(v - mean) * (v - mean) for v in vals]
        return mean, float(np.sqrt(np.float64(cls._np_sum(dev) / n)))

    def _accumulate_frame(self, recs, key=0):
        # what is kept per sweep is its RECORDS (one ndarray view, which the cycle collector does not track), not the nested
        # contribution: ~125 long-lived dicts and lists per sweep made the interpreter run a full collection -- ~0.3 s with torch
        # imported -- every ~2000 sweeps, 0.15 ms per sweep of the thread that launches the device work
        self._log.append((key, recs))
        self._accumulate_records(recs)

    def _accumulate_records(self, recs):
        """One sweep's records into the running lists (eval.py:75-147): every qualifying instance into its speed and its distance
        bucket, then the sweep's mean over its speed buckets.  The records come sorted by (group, instance), so appending in this
        order is the reference's per-group loop over np.unique's ascending instance ids followed by its bucket-wise ``+=``."""
        data = self.evaluate_data
        min_vel, n_groups = self.min_vel, len(EVAL_GROUPS)
        by_speed = {}                                                 # (group id, bucket) -> this sweep's (num_pts, mpe, cham) lists
        for gid, num_pts, vel_ins, dis, mpe, cham in zip(recs["group"].tolist(), recs["num_pts"].tolist(), recs["vel"].tolist(),
                                                         recs["dis"].astype(np.float32).tolist(),     # a float32 mean (eval.py:94)
                                                         recs["mpe"].tolist(), recs["cham"].tolist()):
            if not 1 <= gid <= n_groups or num_pts < 10 or vel_ins < min_vel:
                continue
            score = data[EVAL_GROUPS[gid - 1]]
            name = range_name_of(vel_ins)
            if name is None:
                print("--- [ERROR]: range_name is None --- the value is:", vel_ins, " in ", "vel")
            else:
                slot = score["vel"][name]
                slot["num_pts"].append(num_pts); slot["mpe"].append(mpe); slot["cham"].append(cham)
                mine = by_speed.get((gid, name))
                if mine is None:
                    by_speed[(gid, name)] = ([num_pts], [mpe], [cham])
                else:
                    mine[0].append(num_pts); mine[1].append(mpe); mine[2].append(cham)
            name = range_name_of(dis)
            if name is None:
                print("--- [ERROR]: range_name is None --- the value is:", dis, " in ", "dis")
            else:
                slot = score["dis"][name]
                slot["num_pts"].append(num_pts); slot["mpe"].append(mpe); slot["cham"].append(cham)
        if by_speed:
            for gid, cats_name in enumerate(EVAL_GROUPS, start=1):    # per-sweep mean over the speed buckets only
                totals, mpes, chams = [], [], []
                for name in RANGES:
                    got = by_speed.get((gid, name))
                    if got is not None:
                        mpes.append(self._average(got[1], got[0]))
                        chams.append(self._average(got[2], got[0]))
                        totals.append(sum(got[0]))
                if sum(totals) == 0:
                    continue
                mean = data[cats_name]["mean"]
                mean["num_pts"].append(sum(totals))
                (m_mpe, s_mpe), (m_cham, s_cham) = self._nanmean_nanstd(mpes), self._nanmean_nanstd(chams)
                mean["mpe"].append(m_mpe)
                mean["cham"].append(m_cham)
                mean["std_mpe"].append(s_mpe)
                mean["std_cham"].append(s_cham)
        self.frame_cnt += 1

    # ---- multi-GPU: merge the per-sweep contributions of all ranks in sweep order ------------------------------
    def gather(self):
        """Sweeps are sharded over ranks (i % world); the per-sweep nanmean (eval.py:138) makes every sweep's
        contribution independent, so replaying all contributions in sweep-key order on every rank reproduces
        the single-process lists element for element.  One small object all-gather; no per-sweep traffic."""
        import torch.distributed as dist
        self.flush()
        if not (dist.is_available() and dist.is_initialized()):
            return
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, self._log)
        merged = sorted((item for part in gathered for item in part), key=lambda kv: kv[0])
        self.evaluate_data = self.init_evaluate_data()
        self.frame_cnt = 0
        self._log = merged
        for _, recs in merged:
            self._accumulate_records(recs)

    # ---- reporting: eval.py:151-268 ------------------------------------------------------------------------
    def summary(self) -> dict:
        self.flush()
        wavg = lambda v, w: float(np.average(v, weights=w)) if len(v) > 0 and np.sum(w) > 0 else 0.0
        std = lambda v: float(np.std(v)) if len(v) > 0 else 0.0
        out, pooled = {}, {"mpe": [], "cham": [], "num_pts": []}
        for cat in EVAL_GROUPS:
            raw = self.evaluate_data[cat]
            mean = raw["mean"]
            if len(mean["num_pts"]) == 0:
                continue
            entry = {"overall": {"mpe": wavg(mean["mpe"], mean["num_pts"]), "cd": wavg(mean["cham"], mean["num_pts"]),
                                 "std_mpe": std(mean["std_mpe"]), "std_cd": std(mean["std_cham"]),
                                 "num_pts": int(np.sum(mean["num_pts"])), "num_obj": int(len(mean["num_pts"]))},
                     "velocity": {}, "distance": {}}
            for key, metric in (("velocity", "vel"), ("distance", "dis")):
                for name in RANGES:
                    v = raw[metric][name]
                    entry[key][name] = {"mpe": wavg(v["mpe"], v["num_pts"]), "cd": wavg(v["cham"], v["num_pts"]),
                                        "num_pts": int(np.sum(v["num_pts"])), "num_obj": int(len(v["num_pts"]))}
            out[cat] = entry
            for k in pooled:
                pooled[k].extend(mean[k])
        if pooled["num_pts"]:
            out["Total"] = {"mpe": wavg(pooled["mpe"], pooled["num_pts"]), "cd": wavg(pooled["cham"], pooled["num_pts"]),
                            "num_pts": int(np.sum(pooled["num_pts"])), "num_obj": int(len(pooled["num_pts"]))}
        return out

    def print(self, res_name="flow", file_name="result_av2.json"):
        from tabulate import tabulate
        summ = self.summary()
        data = {}
        if os.path.exists(file_name):
            try:
                with open(file_name) as f:
                    data = json.load(f)
            except json.JSONDecodeError:
                data = {}
        slot = data.setdefault(self.data_name, {}).setdefault(res_name, {})
        rows = []
        print(f"\nHiMo refinement metrics for {res_name} in {self.data_name}:")
        from .compdis import CLOSE_DISTANCE_DEFAULT
        if CLOSE_DISTANCE_THRESHOLD != CLOSE_DISTANCE_DEFAULT:       # a non-default evaluation range is part of the result (the
            slot["close_distance_threshold"] = CLOSE_DISTANCE_THRESHOLD     # reference's file layout is untouched otherwise)
            print(f"(evaluation range CLOSE_DISTANCE_THRESHOLD = {CLOSE_DISTANCE_THRESHOLD:g} m, not the default {CLOSE_DISTANCE_DEFAULT:g} m)")
        for cat, shown in (("CAR", "CAR"), ("OTHER_VEHICLES", "OTHERS")):
            if cat not in summ:
                continue
            slot[cat] = summ[cat]
            o = summ[cat]["overall"]
            rows.append([shown, f"{o['cd']:.3f} ± {o['std_cd']:.2f}", f"{o['mpe']:.3f} ± {o['std_mpe']:.2f}",
                         o["num_pts"], o["num_obj"]])
        if rows:
            with open(file_name, "w") as f:
                json.dump(data, f, indent=4)
        if "Total" in summ:
            t = summ["Total"]
            rows.insert(0, ["Total", f"{t['cd']:.3f}", f"{t['mpe']:.3f}", t["num_pts"], t["num_obj"]])
        print(tabulate(rows, headers=["Class", "CDE (Chamfer) ↓", "MPE (Point Err) ↓", "# Points", "# Objs"],
                       tablefmt="fancy_grid", stralign="center"))
        print(f"Total frames processed: {self.frame_cnt}")
        print(f"Results saved to {file_name}\n")


def nearest_neighbor(query, ref, return_index: bool = True):
    """Exact k=1 Euclidean NN of every ``query`` point among ``ref`` on the GPU: (distances, indices).
    float64 inputs are searched in float64 (bit-comparable with cKDTree), everything else in float32."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    f64 = (query.dtype if isinstance(query, torch.Tensor) else np.asarray(query).dtype) in (torch.float64, np.float64)
    dt = torch.float64 if f64 else torch.float32
    q, r = _dev(query, dt, dev), _dev(ref, dt, dev)
    nq, nr = q.shape[0], r.shape[0]
    d2 = torch.empty(nq, dtype=dt, device=dev)
    idx = torch.empty(nq, dtype=torch.int32, device=dev) if return_index else None
    qo = torch.tensor([0, nq], dtype=torch.int64, device=dev)
    ro = torch.tensor([0, nr], dtype=torch.int64, device=dev)
    _lib.check(lib.himo_nn_search(1, _lib.ptr(qo), _lib.ptr(ro), nq, nr, _lib.ptr(q), _lib.ptr(r), 1 if f64 else 0,
                                  _lib.ptr(d2), _lib.ptr(idx), _lib.stream_handle()), "himo_nn_search")
    _lib.check(lib.himo_sqrt_inplace(nq, _lib.ptr(d2), 1 if f64 else 0, _lib.stream_handle()), "himo_sqrt_inplace")
    d = d2
    if isinstance(query, torch.Tensor):
        return (d, idx) if return_index else d
    return (d.cpu().numpy(), idx.cpu().numpy()) if return_index else d.cpu().numpy()


def chamfer_distance(pc1, pc2) -> float:
    """eval.py:50-62 on the GPU; the search runs in float64 as cKDTree does."""
    a = torch.as_tensor(np.asarray(pc1, dtype=np.float64)) if not isinstance(pc1, torch.Tensor) else pc1.double()
    b = torch.as_tensor(np.asarray(pc2, dtype=np.float64)) if not isinstance(pc2, torch.Tensor) else pc2.double()
    d12 = nearest_neighbor(a, b, return_index=False)
    d21 = nearest_neighbor(b, a, return_index=False)
    d12 = d12 if isinstance(d12, np.ndarray) else d12.cpu().numpy()
    d21 = d21 if isinstance(d21, np.ndarray) else d21.cpu().numpy()
    return float((np.nanmean(d12) + np.nanmean(d21)) / 2.0)


# the first batches of a run pay for the runtime's start, the evaluator's workspaces and the staging memory: ``main`` reports the rate
# after them beside the whole loop's
WARM_BATCHES = 8


def stream_batches(metrics: "InstanceMetrics", source, res_name: str = ""):
    """``source`` yields (sweep keys, frame dicts, comp_dis list | None) per batch.  The frames are read, staged in pinned
    memory and copied to the device by a background thread two batches ahead (feeder.EvalFeeder); the launch thread only
    enqueues kernels (``step_batch``) and digests the previous batches' records."""
    from .feeder import EvalFeeder
    keys_q = []

    def gen():
        for keys, frames, cds in source:
            keys_q.append(list(keys))
            yield (frames, cds) if cds is not None else frames
    metrics.warm_mark = None
    for k, eb in enumerate(EvalFeeder(gen(), res_name=res_name, device=metrics.evaluator.device)):
        if k == WARM_BATCHES:
            metrics.warm_mark = (time.perf_counter(), metrics.frame_cnt + sum(t[1] for t in metrics._pending))
        metrics.step_batch(eb, keys=keys_q.pop(0))
    metrics.flush()


def stream_batches_from_processes(metrics: "InstanceMetrics", key_lists, read, res_name: str = "", workers: int = 4, slot_bytes: int | None = None):
    """``stream_batches`` with the reading and packing on forked reader processes (feeder.ProcessBatchFeeder): ``key_lists[k]`` are
    batch k's sweep keys and ``read(k)`` -- run in a worker -- returns its (frame dicts, comp_dis list | None).  The reference's loop
    gets its frames from ``DataLoader`` worker processes the same way; here it frees the launch thread's interpreter from the
    HDF5 parsing and the packing of the batches it scores.  The workers are forked on entry, BEFORE the evaluator starts the HIP
    runtime when nothing else has (a fork from a process with device state is paid for at its next device call)."""
    from .feeder import ProcessBatchFeeder

    def build(item, upload):                                          # in a reader process: no device call in here
        frames, cds = item
        return EvalBatch.from_frames(list(frames), res_name, cds, device="reader process", upload=upload)
    t0 = time.perf_counter()
    feeder = ProcessBatchFeeder(len(key_lists), read, build, workers=workers, **({} if slot_bytes is None else {"slot_bytes": slot_bytes}))
    metrics.warm_mark = None
    try:
        for k, eb in enumerate(feeder):
            if k == WARM_BATCHES:
                metrics.warm_mark = (time.perf_counter(), metrics.frame_cnt + sum(t[1] for t in metrics._pending))
            metrics.step_batch(eb, keys=list(key_lists[k]))
    finally:
        feeder.close()
        metrics.feed_stats = dict(feeder.pool.stage_seconds, restarts=feeder.pool.restarts, workers=workers, slot_bytes=feeder.pool.slot_bytes,
                                  forked_before_hip=feeder.forked_before_hip, seconds=time.perf_counter() - t0)
    metrics.flush()


def reader_processes_default(n_sweeps: int, dataset) -> int:
    """How many reader processes ``main`` starts when the caller does not say: env ``HIMO_EVAL_WORKERS``, else 4 -- for a dataset that
    survives a fork (``HDF5Dataset`` on h5lite), an evaluation long enough to repay their start (~0.3 s against 1.4 k sweeps/s on
    threads) and a process that holds no device memory yet (the program as started from a shell, a rank under torchrun); otherwise 0:
    the reader threads of this process."""
    env = os.environ.get("HIMO_EVAL_WORKERS")
    if env is not None:
        return max(0, int(env))
    if not getattr(dataset, "fork_safe", False) or n_sweeps < 1024:
        return 0
    if torch.cuda.is_initialized() and torch.cuda.memory_reserved() > 0:
        return 0
    return 4


def main(data_dir: str = "/home/kin/data/av2/h5py/sensor/himo", res_name: str = "", comp_dis_zip: str = "",
         batch_frames: int = 16, dataset=None, file_name: str | None = None, allow_dropped_eval: bool | None = None,
         read_threads: int = 4, num_workers: int | None = None):
    """eval.py:270-313.  Under ``torchrun`` sweep i is scored by rank i % world on its own GPU, the per-sweep records
    are all-gathered once at the end and rank 0 alone prints / writes ``res-<data>.json``.
    ``num_workers``: reader PROCESSES (the loop's ``DataLoader`` workers) for a dataset that can be inherited through fork()
    (``HDF5Dataset`` on h5lite); default: ``reader_processes_default``; 0, or any other dataset: ``read_threads`` threads of this
    process."""
    from . import distenv
    from .dataset import open_dataset
    from .save_zip import read_output_zip
    from .utils import check_valid

    data_name, eval_flag = check_valid(data_dir, res_name, comp_dis_zip)
    with distenv.process_group() as (rank, world):
        metrics = InstanceMetrics(data_name=data_name)
        err = None
        try:
            if dataset is None:
                # only what the loop reads (eval.py:282-310): no next-sweep points, no sensor ids; arrays stored as plain runs come
                # back as views of the file mapping and are copied ONCE, into the feeder's pinned staging memory
                from .dataset import EVAL_FIELDS
                want = EVAL_FIELDS + ((res_name,) if (eval_flag == 2 and res_name and res_name != "raw") else ())
                dataset = open_dataset(data_dir, vis_name=res_name if eval_flag == 2 else "", eval=True,
                                       allow_dropped_eval=allow_dropped_eval, fields=want, zero_copy=True)
            mine = list(range(rank, len(dataset), world))

            def batches():
                # the frames of a batch are read by a few threads (file reads and decompression release the interpreter lock), and
                # the NEXT batch's reads are in flight while this one is staged and copied
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=max(1, read_threads), thread_name_prefix="himo-eval-read") as pool:
                    def read(lo):
                        futs = [pool.submit(dataset.__getitem__, i) for i in mine[lo:lo + batch_frames]]
                        zips = [pool.submit(lambda i=i: read_output_zip(comp_dis_zip, tuple(map(str, dataset.index[i]))))
                                for i in mine[lo:lo + batch_frames]] if (eval_flag == 1 and hasattr(dataset, "index")) else None
                        return futs, zips
                    nxt = read(0) if mine else None
                    for lo in range(0, len(mine), batch_frames):
                        futs, zips = nxt
                        nxt = read(lo + batch_frames) if lo + batch_frames < len(mine) else None
                        frames = [f.result() for f in futs]
                        cds = None
                        if eval_flag == 1:
                            cds = ([z.result() for z in zips] if zips is not None else
                                   [read_output_zip(comp_dis_zip, (f["scene_id"], str(f["timestamp"]))) for f in frames])
                        yield mine[lo:lo + batch_frames], frames, cds
            if num_workers is None:
                num_workers = reader_processes_default(len(mine), dataset)
            t_loop = time.perf_counter()
            if num_workers > 0 and getattr(dataset, "fork_safe", False) and mine:
                key_lists = [mine[lo:lo + batch_frames] for lo in range(0, len(mine), batch_frames)]

                def read(k):                                          # in a reader process
                    frames = [dataset[i] for i in key_lists[k]]
                    cds = None
                    if eval_flag == 1:
                        cds = [read_output_zip(comp_dis_zip, tuple(map(str, dataset.index[i])) if hasattr(dataset, "index") else
                                               (f["scene_id"], str(f["timestamp"]))) for i, f in zip(key_lists[k], frames)]
                    return frames, cds
                stream_batches_from_processes(metrics, key_lists, read, res_name, workers=num_workers)
            else:
                num_workers = 0
                stream_batches(metrics, batches(), res_name)
            t_end = time.perf_counter()
            metrics.loop = {"seconds": t_end - t_loop, "sweeps": len(mine), "reader_processes": num_workers,
                            "reader_threads": 0 if num_workers else max(1, read_threads)}
            mark = getattr(metrics, "warm_mark", None)
            if mark is not None and len(mine) > mark[1]:
                metrics.loop["sweeps_per_s_after_warm_up"] = (len(mine) - mark[1]) / max(t_end - mark[0], 1e-9)
        except Exception as e:
            err = e
        distenv.rendezvous(err, "its sweeps, but no result file was written")
        metrics.gather()
        if rank == 0:
            metrics.print(res_name=res_name, file_name=file_name or f"res-{data_name}.json")
    return metrics


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_dir", default="/home/kin/data/av2/h5py/sensor/himo")
    ap.add_argument("--res_name", "--flow_mode", dest="res_name", default="")
    ap.add_argument("--comp_dis_zip", default="")
    ap.add_argument("--allow_dropped_eval", action="store_true", default=None,
                    help="skip index_eval.pkl sweeps that have no successor sweep in their h5 scene instead of failing")
    ap.add_argument("--num_workers", type=int, default=None,
                    help="reader processes (the reference loop's DataLoader workers); default: 4 for long evaluations of h5 scenes, 0 = reader threads")
    ap.add_argument("--batch_frames", type=int, default=16)
    a = ap.parse_args()
    start_time = time.time()
    done = main(a.data_dir, a.res_name, a.comp_dis_zip, batch_frames=a.batch_frames, allow_dropped_eval=a.allow_dropped_eval, num_workers=a.num_workers)
    print(f"Time used: {time.time() - start_time:.2f} s")
    loop = getattr(done, "loop", None)
    if loop is not None:                                 # the scoring loop alone (without interpreter start, imports, the runtime's start)
        fed_by = f"{loop['reader_processes']} reader processes" if loop["reader_processes"] else f"{loop['reader_threads']} reader threads"
        warm = f"; {loop['sweeps_per_s_after_warm_up']:.0f} sweeps/s after the first {WARM_BATCHES} batches" if "sweeps_per_s_after_warm_up" in loop else ""
        print(f"Scoring loop: {loop['sweeps'] / max(loop['seconds'], 1e-9):.0f} sweeps/s ({loop['sweeps']} sweeps in {loop['seconds']:.2f} s, {fed_by}){warm}")
