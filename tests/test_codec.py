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


# ---- the package's own Feather V2 codec (himo_amd/feather.py) against pandas/pyarrow -------------------------------
def test_own_reader_matches_pandas_on_reference_written_members():
    """The golden zips were written by the reference (pandas -> pyarrow, LZ4-frame compressed buffers)."""
    from himo_amd import feather
    for name in ("av2_gt.zip", "av2_pred.zip", "scania_gt.zip", "scania_pred.zip"):
        with ZipFile(GOLDEN / name) as z:
            for member in z.namelist():
                raw = z.read(member)
                mine, ref = feather.read_table(raw), pd.read_feather(BytesIO(raw))
                assert list(mine) == list(ref.columns)
                for c in ref.columns:
                    assert mine[c].dtype == ref[c].values.dtype and np.array_equal(mine[c], ref[c].values), (member, c)


def test_own_writer_is_read_by_pandas_and_pyarrow():
    import pyarrow.ipc as ipc
    from himo_amd import feather
    rng = np.random.default_rng(0)
    cols = {"comp_dis_x_m": rng.normal(size=4097).astype(np.float32), "eval_mask": (rng.random(4097) > 0.5).astype(np.uint8),
            "flow_instance_id": rng.integers(0, 2**32, 4097, dtype=np.uint32), "b": rng.random(4097) > 0.5,
            "f64": rng.normal(size=4097), "i16": rng.integers(-300, 300, 4097).astype(np.int16)}
    raw = feather.write_table(cols)
    df = pd.read_feather(BytesIO(raw))
    tab = ipc.open_file(BytesIO(raw)).read_all()
    assert tab.num_rows == 4097 and tab.column_names == list(cols)
    for c, v in cols.items():
        assert df[c].values.dtype == v.dtype and np.array_equal(df[c].values, v), c
        assert np.array_equal(feather.read_table(raw)[c], v)
    for n in (0, 1, 7):
        raw = feather.write_table({"comp_dis_x_m": np.arange(n, dtype=np.float32)})
        assert len(pd.read_feather(BytesIO(raw))) == n


def test_own_reader_handles_uncompressed_and_multi_batch_files():
    import pyarrow as pa
    import pyarrow.ipc as ipc
    from himo_amd import feather
    a = np.arange(10, dtype=np.float32)
    sink = BytesIO()
    with ipc.new_file(sink, pa.schema([("a", pa.float32()), ("m", pa.uint8())])) as w:
        for lo in (0, 4, 7):
            hi = {0: 4, 4: 7, 7: 10}[lo]
            w.write_batch(pa.record_batch([pa.array(a[lo:hi]), pa.array((a[lo:hi] % 2).astype(np.uint8))], names=["a", "m"]))
    t = feather.read_table(sink.getvalue())
    assert np.array_equal(t["a"], a) and np.array_equal(t["m"], (a % 2).astype(np.uint8))
    with pytest.raises(ValueError):
        feather.read_table(b"FEA1" + b"\\0" * 64)


def test_lz4_decoder_rejects_garbage():
    import ctypes
    from himo_amd import _lib
    lib = _lib.load()
    dst = ctypes.create_string_buffer(64)
    assert lib.himo_lz4_frame_decompress(b"\\x00" * 32, 32, dst, 64) == -1
    bad = b"\\x04\\x22\\x4d\\x18\\x60\\x40\\x82" + b"\\xff\\xff\\xff\\x7f" + b"\\x00" * 8     # block larger than the input
    assert lib.himo_lz4_frame_decompress(bad, len(bad), dst, 64) == -1


def test_write_matrix_is_the_same_file_as_write_table_and_pyarrow_reads_it():
    """the per-sweep encoder of save_zip (cached framing, columns gathered straight into the file image) against the general
    writer and against pyarrow's reader, for row counts around the 8-byte buffer alignment and for a cache hit"""
    import io
    from himo_amd import feather
    from himo_amd.save_zip import COLUMNS
    rng = np.random.default_rng(4)
    for n in (0, 1, 2, 3, 7, 1000, 120_000, 120_000, 119_999):
        cd = rng.normal(size=(n, 3)).astype(np.float32)
        want = feather.write_table({c: np.ascontiguousarray(cd[:, i]) for i, c in enumerate(COLUMNS)})
        got = feather.write_matrix(cd, COLUMNS)
        assert got.dtype == np.uint8 and got.tobytes() == want
        back = feather.read_table(got.tobytes())
        assert all(np.array_equal(back[c], cd[:, i]) for i, c in enumerate(COLUMNS))
    try:
        import pyarrow.feather as paf
    except ImportError:
        return
    t = paf.read_table(io.BytesIO(feather.write_matrix(cd, COLUMNS).tobytes()))
    assert t.column_names == list(COLUMNS) and np.array_equal(t.column(2).to_numpy(), cd[:, 2])
    with pytest.raises(ValueError):
        feather.write_matrix(cd[:, :2], COLUMNS)
