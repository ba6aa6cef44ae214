"""Shared fixtures.  GPU tests are marked ``@pytest.mark.gpu``; everything else runs on CPU.

The oracle (oracle/himo_oracle.py) is imported here and only here-and-in-tests: it is the
checker, never the product path.
"""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
GOLDEN = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "oracle"))

RES = "seflowpp_best"
FRAME_FIELDS = ("scene_id", "timestamp", "pc0", "pose0", "pose1", "lidar_dt", "lidar_id", "gm0", "flow",
                "flow_is_valid", "flow_category_indices", "flow_instance_id", RES)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    import himo_oracle
    return himo_oracle


@pytest.fixture(scope="session")
def gold():
    with np.load(GOLDEN / "himo_golden.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def eval_gold():
    return json.loads((GOLDEN / "eval_golden.json").read_text())


def golden_frames(gold, data_name):
    """Rebuild the frame dicts the reference was run on (inputs stored in the fixture)."""
    frames = []
    for i in range(int(gold[f"{data_name}/n_frames"])):
        f = {}
        for k in FRAME_FIELDS:
            v = gold[f"{data_name}/{i}/{k}"]
            if k == "scene_id":
                v = str(v)
            elif k == "timestamp":
                v = int(v)
            f[k] = v
        frames.append(f)
    return frames


@pytest.fixture(scope="session")
def frames_av2(gold):
    return golden_frames(gold, "av2")


@pytest.fixture(scope="session")
def frames_scania(gold):
    return golden_frames(gold, "scania")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)
