"""Checkpoint container of the network parameters (README.md:50 ``checkpoint=...``; ssl-train-av2.sh:32 ``save_top_model=3``)."""
import numpy as np
import pytest

from himo_amd.seflow import checkpoint as ck
from himo_amd.seflow import spec


def test_save_load_round_trip_and_validation(tmp_path):
    params = spec.init_params(5)
    path = ck.save_params(tmp_path / "seflowpp_best.npz", params, step=12, epoch=3, adam_m=np.arange(4.0), val=0.25)
    got, extra = ck.load_params(path, with_extra=True)
    assert got.keys() == params.keys() and all(np.array_equal(got[k], params[k]) for k in params)
    assert int(extra["step"]) == 12 and int(extra["epoch"]) == 3 and float(extra["val"]) == 0.25 and len(extra["adam_m"]) == 4
    assert not list(tmp_path.glob("*.writing*"))                  # written aside, then renamed
    bad = dict(params)
    bad.pop("dec4.bias")
    with pytest.raises(KeyError, match="dec4.bias"):
        ck.save_params(tmp_path / "x.npz", bad)
    bad = dict(params, **{"enc1.0.weight": params["enc1.0.weight"][..., :32]})
    with pytest.raises(ValueError, match="enc1.0.weight"):
        ck.check_params(bad)
    with pytest.raises(KeyError):
        ck.save_params(tmp_path / "x.npz", params, bogus=1)


def test_state_dict_adapter_converts_torch_layouts():
    params = spec.init_params(6)
    sd = ck.to_state_dict(params)
    assert sd["enc1.0.weight"].shape == (64, 32, 3, 3) and sd["head.dec1.weight"].shape == (32, 192)
    renamed = {f"model.{k}": v for k, v in sd.items()}                      # a checkpoint with its own prefix
    name_map = {k: f"model.{k}" for k in spec.param_shapes()}
    name_map["head.gru.z.bias"] = lambda d: d["model.head.gru.z.bias"]       # callables for fused / split tensors
    back = ck.from_state_dict(renamed, name_map)
    assert all(np.array_equal(back[k], params[k]) for k in params)


def test_top_k_keeps_the_best_three(tmp_path):
    params = spec.init_params(7)
    top = ck.TopK(tmp_path, k=3)
    vals = [0.9, 0.5, 0.7, 0.8, 0.3, 0.95]
    written = [top.offer(v, e, params) for e, v in enumerate(vals)]
    assert written[3] is not None and written[4] is not None and written[5] is None      # 0.8 displaces 0.9; 0.95 never enters
    kept = sorted(float(ck.load_params(p, with_extra=True)[1]["val"]) for p in tmp_path.glob("*.npz"))
    assert kept == [0.3, 0.5, 0.7]
    assert float(ck.load_params(top.best(), with_extra=True)[1]["val"]) == 0.3


def test_top_k_resumes_its_ranking_and_rejects_non_finite_figures(tmp_path):
    params = spec.init_params(8)
    first = ck.TopK(tmp_path, k=2)
    for e, v in enumerate([0.6, 0.4, 0.5]):
        first.offer(v, e, params)
    fresh = ck.TopK(tmp_path, k=2)                                 # a NEW run pointed at a used directory: ADVICE r03 --
    assert fresh.kept == [] and fresh.best() is None               # it neither ranks against nor deletes another run's files
    assert len(list(tmp_path.glob("*.npz"))) == 2
    resumed = ck.TopK(tmp_path, k=2, resume=True)                  # --resume: a fresh object over the same directory
    assert sorted(v for v, _ in resumed.kept) == [0.4, 0.5]
    assert resumed.offer(float("nan"), 3, params) is None and resumed.offer(float("inf"), 3, params) is None
    assert resumed.offer(0.45, 4, params) is not None
    assert sorted(float(ck.load_params(p, with_extra=True)[1]["val"]) for p in tmp_path.glob("*.npz")) == [0.4, 0.45]   # never more than k
