"""Where does a batch's time go when reader PROCESSES feed the evaluator?  8 h5 scenes x 33 sweeps of 120k points; per batch of 16: read
(h5lite, views of the mapping) and pack (into the shared slot) timed inside the worker; then the pool alone, the pool with its slots
registered with the HIP runtime, and the whole ProcessBatchFeeder without a consumer.
usage: python scripts/exp_reader_pool.py [cpu]"""
import pickle, shutil, sys, tempfile, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from himo_amd import h5lite
from himo_amd.dataset import EVAL_FIELDS, open_dataset
from himo_amd.eval import EvalBatch
from himo_amd.feeder import ProcessBatchFeeder, ReaderPool
from himo_amd.synthetic import make_frame

CPU_ONLY = len(sys.argv) > 1 and sys.argv[1] == "cpu"
N_SCENES = 8 if not CPU_ONLY else 2
REPEAT = 4
root = Path(tempfile.mkdtemp(prefix="himo_reader_pool_"))
try:
    index = []
    for sc in range(N_SCENES):
        tree = {}
        for k in range(33):
            f = make_frame(9000 + 40 * sc + k, n_points=120_000, scene_id=f"eval{sc:02d}")
            tree[str(f["timestamp"])] = {"lidar": f["pc0"], "lidar_dt": f["lidar_dt"], "lidar_id": f["lidar_id"], "pose": f["pose0"], "ground_mask": f["gm0"],
                                         "flow": f["flow"], "flow_is_valid": f["flow_is_valid"], "flow_category_indices": f["flow_category_indices"],
                                         "flow_instance_id": f["flow_instance_id"], "seflowpp_best": f["seflowpp_best"]}
            index.append([f["scene_id"], str(f["timestamp"])])
        h5lite.write_file(root / f"eval{sc:02d}.h5", tree)
    with open(root / "index_total.pkl", "wb") as fh:
        pickle.dump(index, fh)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = open_dataset(str(root), vis_name="seflowpp_best", eval=True, fields=EVAL_FIELDS + ("seflowpp_best",), zero_copy=True)
    B = 16
    key_lists = [list(range(lo, min(lo + B, len(ds)))) for lo in range(0, len(ds), B)]
    dev = torch.device("cpu") if CPU_ONLY else torch.device("cuda", 0)

    def make(k):
        t0 = time.perf_counter()
        frames = [ds[i] for i in key_lists[k]]
        return frames, time.perf_counter() - t0

    def build(item, upload):
        frames, t_read = item
        t0 = time.perf_counter()
        eb = EvalBatch.from_frames(frames, "seflowpp_best", None, device=dev, upload=upload)
        eb.t_read, eb.t_pack = t_read, time.perf_counter() - t0
        return eb

    def run_pool(workers, on=None, off=None):
        pool = ReaderPool(len(key_lists) * REPEAT, lambda k: make(k % len(key_lists)), build, workers=workers, slot_bytes=128 << 20, on_slots=on, off_slots=off)
        start, halt = pool._start, pool._halt

        def timed(fn, what):
            def run():
                t = time.perf_counter()
                fn()
                fixed[what] = fixed.get(what, 0.0) + time.perf_counter() - t
            return run
        fixed = {}
        pool._start, pool._halt = timed(start, "start"), timed(halt, "halt")
        t0 = time.perf_counter()
        reads, packs, n = [], [], 0
        for k, s, used, obj in pool:
            reads.append(obj.t_read); packs.append(obj.t_pack); n += len(key_lists[k % len(key_lists)])
            pool.release(s)
        del obj
        el = time.perf_counter() - t0
        time.sleep(0.05)
        print(f"      (start {1e3 * fixed.get('start', 0):.0f} ms, halt {1e3 * fixed.get('halt', 0):.0f} ms of {1e3 * el:.0f} ms; steady {n / max(el - fixed.get('start', 0), 1e-9):.0f} sweeps/s)")
        return n / el, 1e3 * np.mean(reads), 1e3 * np.mean(packs), used

    for workers in (1, 4, 8):
        rate, r, p, used = run_pool(workers)
        print(f"pool alone, {workers} reader processes: {rate:7.0f} sweeps/s; in the worker per batch of {B}: read {r:.1f} ms, pack {p:.1f} ms ({used / 1e6:.0f} MB)")
    if not CPU_ONLY:
        torch.zeros(1, device=dev)
        for workers in (1, 4, 8):
            rate, r, p, used = run_pool(workers)
            print(f"pool alone after the HIP runtime started, {workers} reader processes: {rate:7.0f} sweeps/s; read {r:.1f} ms, pack {p:.1f} ms")
        fd = ProcessBatchFeeder.__new__(ProcessBatchFeeder)
        fd._slot_done = {}
        for workers in (1, 4, 8):
            rate, r, p, used = run_pool(workers, on=ProcessBatchFeeder._register, off=fd._unregister)
            print(f"pool with registered slots, {workers} reader processes: {rate:7.0f} sweeps/s; read {r:.1f} ms, pack {p:.1f} ms")
        for workers in (1, 4, 8):
            feeder = ProcessBatchFeeder(len(key_lists) * REPEAT, lambda k: make(k % len(key_lists)), build, device=dev, workers=workers)
            t0 = time.perf_counter()
            n = sum(eb.batch.n_frames for eb in feeder)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print(f"ProcessBatchFeeder without a consumer, {workers} reader processes: {n / el:7.0f} sweeps/s")
    ds.close()
finally:
    shutil.rmtree(root, ignore_errors=True)
