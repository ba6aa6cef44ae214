"""Where does `python -m himo_amd.eval` over .h5 scenes spend its wall time?  Reader threads of the process against forked reader
processes (cProfile of the launch thread on the last pass + wall clock).
usage (GPU box): python scripts/exp_eval_main.py [sweeps per scene = 33]     (8 scenes; 33 distinct 120k-point sweeps per scene, repeated under
new time stamps beyond that: 129 -> 1024 scored sweeps, 5.7 GB of scene files)"""
import contextlib, io, pickle, shutil, sys, tempfile, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import eval as ev, h5lite
from himo_amd.synthetic import make_frame

PER_SCENE = int(sys.argv[1]) if len(sys.argv) > 1 else 33
root = Path(tempfile.mkdtemp(prefix="himo_eval_av2_"))
try:
    index = []
    for sc in range(8):
        tree = {}
        made = [make_frame(9000 + 40 * sc + k, n_points=120_000, scene_id=f"eval{sc:02d}") for k in range(min(PER_SCENE, 33))]
        for k in range(PER_SCENE):
            f = dict(made[k % len(made)])
            f["timestamp"] = int(made[0]["timestamp"]) + k * 100_000_000
            tree[str(f["timestamp"])] = {"lidar": f["pc0"], "lidar_dt": f["lidar_dt"], "lidar_id": f["lidar_id"], "pose": f["pose0"], "ground_mask": f["gm0"],
                                         "flow": f["flow"], "flow_is_valid": f["flow_is_valid"], "flow_category_indices": f["flow_category_indices"],
                                         "flow_instance_id": f["flow_instance_id"], "seflowpp_best": f["seflowpp_best"]}
            index.append([f["scene_id"], str(f["timestamp"])])
        h5lite.write_file(root / f"eval{sc:02d}.h5", tree)
    with open(root / "index_total.pkl", "wb") as fh:
        pickle.dump(index, fh)
    # the PROGRAM, each configuration in its own interpreter (what a user starts; reader processes are forked before the runtime starts)
    import subprocess
    results = {}
    for workers in (0, 2, 4, 8):
        for rep in range(2):
            t0 = time.perf_counter()
            out = subprocess.run([sys.executable, "-W", "ignore", "-m", "himo_amd.eval", "--data_dir", str(root), "--res_name", "seflowpp_best",
                                  "--num_workers", str(workers)], cwd=str(root), capture_output=True, text=True, timeout=600,
                                 env=dict(__import__("os").environ, PYTHONPATH=str(Path(__file__).resolve().parents[1])))
            el = time.perf_counter() - t0
            assert out.returncode == 0, out.stderr[-2000:]
            loop = [l for l in out.stdout.splitlines() if l.startswith("Scoring loop")]
            print(f"python -m himo_amd.eval --num_workers {workers}, run {rep}: process wall {el:.2f} s; {loop[-1] if loop else out.stdout[-300:]}")
        results[workers] = (root / "res-av2.json").read_text()
    print("same result file whatever feeds the evaluator:", all(v == results[0] for v in results.values()))
    # ... and inside ONE process that has already used the device (bench.py's legs, the tests): forking readers from it is paid for at
    # the next device call, which is why main() defaults to reader threads there
    sink = io.StringIO()
    import cProfile, pstats
    for workers in (0, 4):
        for rep in range(3):
            pr = cProfile.Profile()
            with warnings.catch_warnings(), contextlib.redirect_stdout(sink):
                warnings.simplefilter("ignore")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if rep == 2:
                    pr.enable()
                m = ev.main(str(root), res_name="seflowpp_best", batch_frames=16, file_name=str(root / "res.json"), num_workers=workers)
                pr.disable()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            what = f"{workers} reader processes" if workers else "4 reader threads of this process"
            print(f"in a process with device state, {what}, pass {rep}: {m.frame_cnt} sweeps in {el:.3f} s = {m.frame_cnt / el:.0f} sweeps/s   {getattr(m, 'feed_stats', '')}")
        pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
finally:
    shutil.rmtree(root, ignore_errors=True)
