"""Where does `python -m himo_amd.eval` over .h5 scenes spend its wall time?  (the second of two passes, cProfile of the main thread + wall clock)"""
import contextlib, io, pickle, shutil, sys, tempfile, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import eval as ev, h5lite
from himo_amd.synthetic import make_frame

root = Path(tempfile.mkdtemp(prefix="himo_eval_av2_"))
try:
    index = []
    for sc in range(8):
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
    sink = io.StringIO()
    import cProfile, pstats
    for rep in range(3):
        pr = cProfile.Profile()
        with warnings.catch_warnings(), contextlib.redirect_stdout(sink):
            warnings.simplefilter("ignore")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if rep == 2:
                pr.enable()
            m = ev.main(str(root), res_name="seflowpp_best", batch_frames=16, file_name=str(root / "res.json"))
            pr.disable()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print(f"pass {rep}: {m.frame_cnt} sweeps in {el:.3f} s = {m.frame_cnt / el:.0f} sweeps/s")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
finally:
    shutil.rmtree(root, ignore_errors=True)
