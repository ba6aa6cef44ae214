"""The launch thread of `python -m himo_amd.eval` fed by reader processes, under cProfile (a fresh interpreter: the readers are forked before
the HIP runtime starts).  usage (GPU box): python scripts/prof_eval_program.py [sweeps per scene = 129]"""
import os, pickle, shutil, subprocess, sys, tempfile
from pathlib import Path
R = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] == "--inner":
    sys.path.insert(0, str(R))
    import cProfile, pstats, contextlib, io, warnings
    from himo_amd import eval as ev
    root = sys.argv[2]
    pr = cProfile.Profile()
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        pr.enable()
        m = ev.main(root, res_name="seflowpp_best", batch_frames=16, file_name=str(Path(root) / "res.json"), num_workers=4)
        pr.disable()
    print(m.loop)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in m.feed_stats.items()})
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    pstats.Stats(pr).sort_stats("cumulative").print_stats("himo_amd", 18)
    sys.exit(0)
sys.path.insert(0, str(R))
from himo_amd import h5lite
from himo_amd.synthetic import make_frame
PER_SCENE = int(sys.argv[1]) if len(sys.argv) > 1 else 129
root = Path(tempfile.mkdtemp(prefix="himo_eval_av2_prog_"))
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
    for rep in range(2):
        out = subprocess.run([sys.executable, __file__, "--inner", str(root)], capture_output=True, text=True, timeout=600)
    print(out.stdout[-9000:]); print(out.stderr[-1500:])
finally:
    shutil.rmtree(root, ignore_errors=True)
