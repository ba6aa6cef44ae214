"""``himo_amd.h5lite`` -- the dependency-free HDF5 reader / fresh-file writer behind the h5 boundary (SURVEY.md 8b items 1-2,
8f-3) -- pinned against the real HDF5 library:

* the committed fixtures under ``tests/golden/h5`` were written by libhdf5 1.10.6 (``tests/golden/make_h5_fixture.py``) with
  the dataset names and dtypes of dataprocess/extract_sca.py:76-93 + ``ground_mask`` (tools/test/repack_h5_scania.py:29) + an
  existing ``seflowpp_best``; h5lite's arrays must equal the arrays the generator wrote, bit for bit (they are re-made here from
  the same ``make_frame`` seeds);
* wherever an HDF5 library can be loaded (``himo_amd.h5c``: the build image and, with the same image, the GPU box), files of
  every encoding the reader claims -- and the files h5lite writes -- go through both and must agree.
"""
import pickle
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from himo_amd import h5c, h5lite
from himo_amd.synthetic import make_frame

H5 = Path(__file__).resolve().parent / "golden" / "h5"
needs_libhdf5 = pytest.mark.skipif(not h5c.available(), reason="no HDF5 C library to compare with")


def fixture_frames():
    """The frames ``make_h5_fixture.py`` wrote (same seeds): 2 scenes x 4 timestamps."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_h5_fixture", H5.parent / "make_h5_fixture.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.frames()


def test_fixture_files_read_bit_for_bit():
    mod, frames = fixture_frames()
    seen = 0
    for f in frames:
        with h5lite.File(H5 / f"{f['scene_id']}.h5") as h:
            g = h[str(f["timestamp"])]
            want = mod.group_arrays(f)
            assert sorted(g.keys()) == sorted(want)
            for name, a in want.items():
                d = g[name]
                a = np.asarray(a)
                assert d.shape == a.shape and d.dtype == a.dtype, (name, d.shape, d.dtype, a.dtype)
                got = d[()] if a.ndim == 0 else d[:]
                assert np.array_equal(got, a), name
                seen += 1
    assert seen == len(frames) * 13
    with open(H5 / "index_total.pkl", "rb") as fh:
        assert pickle.load(fh) == [[f["scene_id"], str(f["timestamp"])] for f in frames]


def test_fixture_listing_matches_h5dump_of_the_generator():
    """``h5dump -H`` of the fixtures as committed next to them: the datatypes the library reports are the ones h5lite maps."""
    text = (H5 / "h5dump_H.txt").read_text()
    for want in ('DATASET "lidar"', "H5T_IEEE_F32LE", "H5T_IEEE_F64LE", "H5T_STD_U8LE", "H5T_STD_U32LE", "H5T_STD_I64LE",
                 'H5T_ENUM', '"FALSE"', '"TRUE"', 'DATASET "ground_mask"', 'DATASET "seflowpp_best"', "SCALAR"):
        assert want in text, want


def test_unsupported_features_are_named_not_misread(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all" * 100)
    with pytest.raises(OSError, match="not an HDF5 file"):
        h5lite.File(p)
    p.write_bytes(b"")
    with pytest.raises(OSError):
        h5lite.File(p)
    h5lite.write_file(p, {"g": {"x": np.arange(3)}})
    raw = bytearray(p.read_bytes())
    raw[8] = 9                                                      # a superblock version from the future
    p.write_bytes(bytes(raw))
    with pytest.raises(h5lite.Unsupported, match="superblock version 9"):
        h5lite.File(p)
    with pytest.raises(TypeError):
        h5lite.write_file(p, {"c": np.zeros(3, np.complex64)})
    h5lite.write_file(p, {"g": {"x": np.arange(3)}})
    with h5lite.File(p) as f:
        with pytest.raises(KeyError, match="nope"):
            f["g"]["nope"]
        assert "g/x" in f and "g/y" not in f and "h" not in f and f["g/x"].shape == (3,)


def _tree(rng, n_groups, n):
    t = {}
    for g in range(n_groups):
        t[str(315965785000000000 + g * 100000000)] = {
            "lidar": rng.normal(size=(n, 4)).astype(np.float32), "lidar_id": rng.integers(1, 7, n).astype(np.uint8),
            "lidar_dt": rng.uniform(size=n).astype(np.float32), "SensorsCenter": rng.normal(size=(6, 3)).astype(np.float32),
            "pose": rng.normal(size=(4, 4)), "timestamp": np.int64(315965785000000000 + g),
            "flow": rng.normal(size=(n, 3)).astype(np.float32), "flow_is_valid": rng.uniform(size=n) > 0.1,
            "flow_category_indices": rng.integers(0, 30, n).astype(np.uint8),
            "flow_instance_id": rng.integers(0, 99, n).astype(np.uint32), "ground_mask": rng.uniform(size=n) > 0.5,
            "ego_motion": rng.normal(size=(4, 4)).astype(np.float32), "seflowpp_best": rng.normal(size=(n, 3)).astype(np.float32),
            "empty": np.zeros((0, 3), np.float32), "i16": rng.integers(-5, 5, n).astype(np.int16),
            "u64": rng.integers(0, 2 ** 62, n).astype(np.uint64)}
    return t


def _same(reader, t):
    assert sorted(reader.keys()) == sorted(t)
    for g, ds in t.items():
        grp = reader[g]
        assert sorted(grp.keys()) == sorted(ds)
        for k, a in ds.items():
            d = grp[k]
            got = d[()] if np.ndim(a) == 0 else d[:]
            assert d.shape == np.shape(a) and d.dtype == np.asarray(a).dtype, (k, d.shape, d.dtype)
            assert np.array_equal(got, a), k


@needs_libhdf5
@pytest.mark.parametrize("n_groups", [1, 9, 300, 1200])
def test_reader_equals_libhdf5_on_default_encoded_files(tmp_path, n_groups):
    """h5py's default encoding at the sizes that matter: 1 SNOD, a split leaf, a full root B-tree node, a two-level tree."""
    t = _tree(np.random.default_rng(n_groups), n_groups, 37)
    with h5c.File(tmp_path / "a.h5", "w") as f:
        for g, ds in t.items():
            grp = f.create_group(g)
            for k, a in ds.items():
                grp.create_dataset(k, data=a)
    with h5lite.File(tmp_path / "a.h5") as f:
        _same(f, t)


@needs_libhdf5
@pytest.mark.parametrize("libver", ["earliest", "latest"])
def test_reader_equals_libhdf5_on_other_layouts(tmp_path, libver):
    """chunked (version-1 B-tree index; with libver="latest": fixed-array / single-chunk / implicit indexes), deflate / shuffle /
    fletcher32, compact -- every one read back equal; the only encoding left out is named when met (a paged fixed array)."""
    a = np.random.default_rng(5).normal(size=(1000, 4)).astype(np.float32)
    cases = [dict(chunks=(128, 4)), dict(compression="gzip"), dict(compression="gzip", shuffle=True, chunks=(300, 3)),
             dict(chunks=(1000, 4)), dict(fletcher32=True, shuffle=True, compression="gzip"), dict(compact=True), dict(chunks=(7, 2))]
    for kw in cases:
        want = a[:100] if kw.get("compact") else a
        with h5c.File(tmp_path / "b.h5", "w", libver=libver) as f:
            f.create_dataset("x", data=want, **kw)
            f.create_dataset("flag", data=want[:, 0] > 0)
        with h5lite.File(tmp_path / "b.h5") as f:
            assert np.array_equal(f["x"][:], want) and np.array_equal(f["flag"][:], want[:, 0] > 0), kw
    with h5c.File(tmp_path / "p.h5", "w", libver="latest") as f:
        f.create_dataset("x", data=a, chunks=(1, 2))                   # 2000 chunks: the fixed array is paged
    with h5lite.File(tmp_path / "p.h5") as f:
        if libver == "latest":
            with pytest.raises(h5lite.Unsupported, match="paged fixed-array"):
                f["x"][:]


@needs_libhdf5
@pytest.mark.parametrize("n_groups,n_datasets", [(1, 3), (1, 20), (300, 13), (1200, 3), (3, 200)])
def test_latest_format_groups(tmp_path, n_groups, n_datasets):
    """``libver="latest"``: version-2 object headers; compact link messages (up to 8 links) and DENSE groups -- link messages in a
    fractal heap (root direct block, indirect blocks), indexed by a version-2 B-tree (one leaf, and internal nodes at 1200 links)."""
    rng = np.random.default_rng(n_groups + n_datasets)
    t = {f"g{g:05d}": {f"d{i:03d}": rng.normal(size=(4, 3)).astype(np.float32) for i in range(n_datasets)} for g in range(n_groups)}
    t[next(iter(t))]["mask"] = rng.uniform(size=7) > 0.5
    with h5c.File(tmp_path / "c.h5", "w", libver="latest") as f:
        for g, ds in t.items():
            grp = f.create_group(g)
            for k, a in ds.items():
                grp.create_dataset(k, data=a)
    with h5lite.File(tmp_path / "c.h5") as f:
        _same(f, t)


@needs_libhdf5
@pytest.mark.parametrize("n_groups", [0, 1, 9, 40, 300, 1100])
def test_written_files_are_real_hdf5(tmp_path, n_groups):
    """``write_file`` output read back by the library (and by h5lite), then MODIFIED by the library -- a dataset added, one
    deleted, a group added -- and read again: the B-tree / heap / symbol nodes written here are ones libhdf5 can extend."""
    t = _tree(np.random.default_rng(100 + n_groups), n_groups, 29)
    p = tmp_path / "w.h5"
    h5lite.write_file(p, t)
    with h5lite.File(p) as f:
        _same(f, t)
    with h5c.File(p, "r") as f:
        _same(f, t)
    if not n_groups:
        return
    g0 = sorted(t)[0]
    with h5c.File(p, "a") as f:
        f[g0].create_dataset("added", data=np.ones((5, 3), np.float32))
        del f[g0]["flow"]
        f.create_group("newgroup").create_dataset("x", data=np.arange(4))
    with h5lite.File(p) as f:
        assert (f[g0]["added"][:] == 1).all() and "flow" not in f[g0] and (f["newgroup/x"][:] == np.arange(4)).all()
        assert len(f) == n_groups + 1


def test_written_files_pass_h5dump(tmp_path):
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not Path(h5dump).exists():
        pytest.skip("no h5dump")
    t = _tree(np.random.default_rng(3), 3, 11)
    h5lite.write_file(tmp_path / "w.h5", t)
    r = subprocess.run([h5dump, str(tmp_path / "w.h5")], capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr[:500]
    assert r.stdout.count("DATASET") == 3 * 16 and "H5T_ENUM" in r.stdout and "TRUE" in r.stdout
