"""a9: Feather-in-zip wire format (save_zip.py:30-100) -- CPU only."""
from io import BytesIO
from zipfile import ZIP_STORED, ZipFile

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, golden_frames
from himo_amd import save_zip


@pytest.mark.parametrize("data_name", ["av2", "scania"])
def test_read_reference_written_zip(gold, data_name):
    for i, f in enumerate(golden_frames(gold, data_name)):
        cd = save_zip.read_output_zip(str(GOLDEN / f"{data_name}_pred.zip"), (f["scene_id"], str(f["timestamp"])))
        assert cd.dtype == np.float32 and cd.shape == (len(f["pc0"]), 3)
        assert np.array_equal(cd, gold[f"{data_name}/{i}/ref_comp_dis"])


def test_missing_sweep_raises_keyerror():
    with pytest.raises(KeyError):
        save_zip.read_output_zip(str(GOLDEN / "av2_pred.zip"), ("no-such-scene", "0"))


def test_write_then_zip_matches_reference_members(tmp_path, gold):
    frames = golden_frames(gold, "av2")
    for i, f in enumerate(frames):
        # float64 in, float32 columns out (the cast of save_zip.py:70-72)
        save_zip.write_output_file(gold[f"av2/{i}/ref_comp_dis_f64"], (f["scene_id"], str(f["timestamp"])), tmp_path)
    out = save_zip.zip_res(tmp_path, output_file=str(tmp_path / "mine.zip"))
    assert not any(p.is_dir() for p in tmp_path.iterdir())           # scene folders removed after zipping
    with ZipFile(out) as mine, ZipFile(GOLDEN / "av2_pred.zip") as ref:
        assert sorted(mine.namelist()) == sorted(ref.namelist())
        for info in mine.infolist():
            assert info.compress_type == ZIP_STORED
            a = pd.read_feather(BytesIO(mine.read(info.filename)))
            b = pd.read_feather(BytesIO(ref.read(info.filename)))
            assert list(a.columns) == list(b.columns) == list(save_zip.COLUMNS)
            assert all(a[c].dtype == np.float32 for c in a.columns)
            assert a.equals(b)


def test_zip_sink_streams_same_members(tmp_path, gold):
    frames = golden_frames(gold, "scania")
    with save_zip.ZipSink(tmp_path / "s.zip") as sink:
        for i, f in enumerate(frames):
            sink.add(gold[f"scania/{i}/ref_comp_dis"], (f["scene_id"], str(f["timestamp"])))
    for i, f in enumerate(frames):
        cd = save_zip.read_output_zip(str(tmp_path / "s.zip"), (f["scene_id"], str(f["timestamp"])))
        assert np.array_equal(cd, gold[f"scania/{i}/ref_comp_dis"])
    with ZipFile(tmp_path / "s.zip") as z:
        assert all(i.compress_type == ZIP_STORED for i in z.infolist())


def test_empty_sweep_roundtrip(tmp_path):
    save_zip.write_output_file(np.zeros((0, 3)), ("s", "1"), tmp_path)
    save_zip.zip_res(tmp_path, str(tmp_path / "e.zip"))
    assert save_zip.read_output_zip(str(tmp_path / "e.zip"), ("s", "1")).shape == (0, 3)
