"""Golden vectors for stages a10-a12 (network forward, self-supervised loss, FastNSF) -- GUARDED: does nothing until the
reference's ``OpenSceneFlow`` submodule is populated.

In this mount ``/root/reference/OpenSceneFlow`` is EMPTY (.gitmodules:1-3 names the URL only, no SHA), so the network,
loss and FastNSF of this build are pinned against the build's own CPU restatements (PARITY UNPINNED).  The
day the sources (and a checkpoint) are mounted, this script is the missing half of the pin:

    python tests/golden/make_golden_network.py [--checkpoint /path/to/seflowpp_best.ckpt]

1. imports the reference model (``src.models.deflowpp`` -- the ``model=deflowpp`` of assets/slurm/ssl-train-av2.sh:32) with
   the launcher's arguments (voxel_size [0.2,0.2,6], point_cloud_range [-51.2,-51.2,-3,51.2,51.2,3], num_frames 3),
2. runs it on the seeded frames of ``himo_amd.synthetic.make_frame`` (the same ones the GPU tests use) through the batch
   dict the reference's ``save.py`` builds (pc0, pc1, pch1, pose0, pose1, poseh1),
3. writes inputs' seeds + the reference's per-point flow to ``tests/golden/network_golden.npz`` and the state dict's
   tensor names / shapes to ``tests/golden/network_state_dict.json`` -- the input of
   ``himo_amd.seflow.checkpoint.from_state_dict``'s name map (spec name -> state-dict key), which cannot be written before
   the names are visible.

Nothing here is imported by the product or by any test; a fixture produced by it is data (inputs + expected outputs).
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
SUB = REF / "OpenSceneFlow"
OUT = Path(__file__).resolve().parent
REPO = OUT.parents[1]

SEEDS = [(900, 901, 902), (910, 911, 912)]          # (history, pc0, pc1) frame indices of make_frame
N_POINTS = 30_000


def submodule_present() -> bool:
    return (SUB / "src").is_dir() and any((SUB / "src").rglob("*.py"))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default="")
    a = ap.parse_args(argv)
    if not submodule_present():
        print(f"{SUB} is empty in this mount: nothing to pin against (stages a10-a12 stay PARITY UNPINNED).")
        return 0

    import torch
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(SUB))
    from himo_amd.synthetic import make_frame
    try:
        from src.models import DeFlowPP as Model                   # class name as exported by OpenSceneFlow's src/models
    except ImportError:
        from src.models.deflowpp import DeFlowPP as Model
    model = Model(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3], grid_feature_size=[512, 512],
                  num_frames=3)
    if a.checkpoint:
        ck = torch.load(a.checkpoint, map_location="cpu", weights_only=False)
        sd = ck.get("state_dict", ck)
        model.load_state_dict({k.removeprefix("model."): v for k, v in sd.items()}, strict=False)
    else:
        torch.manual_seed(0)
    model.eval()
    names = {k: list(v.shape) for k, v in model.state_dict().items()}
    (OUT / "network_state_dict.json").write_text(json.dumps(names, indent=1))

    arrays = {"n_points": np.int64(N_POINTS), "seeds": np.asarray(SEEDS, np.int64)}
    for k, (ih, i0, i1) in enumerate(SEEDS):
        fh, f0, f1 = (make_frame(i, n_points=N_POINTS) for i in (ih, i0, i1))
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x))[None]
        batch = {"pc0": t(f0["pc0"][:, :3]), "pc1": t(f1["pc0"][:, :3]), "pch1": t(fh["pc0"][:, :3]),
                 "pose0": t(f0["pose0"]), "pose1": t(f0["pose1"]), "poseh1": t(fh["pose0"])}
        with torch.no_grad():
            res = model(batch)
        flow = res["flow"][0] if isinstance(res, dict) else res[0]
        # the reference's save.py adds the ego-motion flow back before writing <res_name> (save_zip.py:117 subtracts it)
        arrays[f"{k}/flow_network"] = np.asarray(flow, np.float32)
        if isinstance(res, dict) and "pose_flow" in res:
            arrays[f"{k}/pose_flow"] = np.asarray(res["pose_flow"][0], np.float32)
        if isinstance(res, dict) and "pc0_valid_point_idxes" in res:
            arrays[f"{k}/valid_idx"] = np.asarray(res["pc0_valid_point_idxes"][0], np.int64)
    np.savez_compressed(OUT / "network_golden.npz", **arrays)
    print(f"wrote {OUT / 'network_golden.npz'} ({len(SEEDS)} samples) and network_state_dict.json ({len(names)} tensors); "
          "next: fill the name map for himo_amd.seflow.checkpoint.from_state_dict and add tests/test_network_golden.py")
    return 0


if __name__ == "__main__":
    sys.exit(main())
