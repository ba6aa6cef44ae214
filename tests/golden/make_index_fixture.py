"""Copies the frame-index DATA files the reference ships (assets/docs/av2/index_eval.pkl: 70 frames of 13 scenes, the
frame list of BASELINE config 2; assets/docs/av2/index_total.pkl: every frame of those scenes, which is what supplies each
eval frame's successor) into tests/golden/ as JSON -- data, not source: lists of [scene_id, timestamp] string pairs
(the structure tools/pkl_extract.py:5-19 handles).  Run once in the build container; the GPU box never sees /root/reference.

    python tests/golden/make_index_fixture.py
"""
import json
import pickle
from pathlib import Path

REF = Path("/root/reference/assets/docs/av2")
OUT = Path(__file__).resolve().parent


def main():
    out = {}
    for name in ("index_eval", "index_total"):
        with open(REF / f"{name}.pkl", "rb") as f:
            idx = pickle.load(f)
        assert all(isinstance(s, str) and isinstance(t, str) for s, t in idx)
        out[name] = [[s, t] for s, t in idx]
    (OUT / "av2_index.json").write_text(json.dumps(out))
    print({k: len(v) for k, v in out.items()}, "scenes:", len({s for s, _ in out["index_eval"]}))


if __name__ == "__main__":
    main()
