"""The HDF5 C library through ``ctypes`` -- the subset of calls the h5 boundary needs when ``h5py`` is not installed:
open / create a file, walk groups, read a dataset, create / delete / replace a dataset.

Used by ``save.H5ResultSink`` to put ``<timestamp>/<res_name>`` INTO the scene file the way the reference's absent
``save.py`` does (the dataset ``tools/test/repack_h5_scania.py:50`` skips by name and ``save_zip.py:117`` reads back), by
``tests/golden/make_h5_fixture.py`` to write the committed fixtures with the real library, and by the tests that pin
``h5lite`` (this package's dependency-free reader / writer) against it.  Nothing here is on the compute path.

The objects mimic the few ``h5py`` calls the reference makes (dataprocess/extract_sca.py:76-93,
tools/test/repack_h5_scania.py:41-75): ``File(path, mode)`` as a context manager, ``f.create_group``, ``g[name]``,
``name in g``, ``del g[name]``, ``g.keys()``, ``g.create_dataset(name, data=...)``, ``d[:]`` / ``d[()]``, ``d.shape``,
``d.dtype``.  ``bool`` arrays are stored the way h5py stores them: an 8-bit enum {FALSE=0, TRUE=1}.

``available()`` says whether a library could be loaded; ``HIMO_LIBHDF5`` names one explicitly.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import glob
import os

import numpy as np

_CANDIDATES = ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
               "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*")

_lib = None
_why = None

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT, H5T_ENUM = 0, 1, 8
H5T_SGN_NONE = 0
H5F_LIBVER_EARLIEST = 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5Z_FILTER_DEFLATE = 1


def _paths():
    if os.environ.get("HIMO_LIBHDF5"):
        yield os.environ["HIMO_LIBHDF5"]
        return
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in _CANDIDATES:
        for p in sorted(glob.glob(pat)):
            if "_cpp" not in p and "_hl" not in p and "_fortran" not in p:
                yield p


def load():
    """The loaded library (cached); raises ImportError naming what was tried."""
    global _lib, _why
    if _lib is not None:
        return _lib
    if _why is not None:
        raise ImportError(_why)
    tried = []
    for p in _paths():
        try:
            lib = ctypes.CDLL(p)
            lib.H5open()
            maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
            lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
        except (OSError, AttributeError) as e:
            tried.append(f"{p}: {e}")
            continue
        lib.version = (maj.value, mnr.value, rel.value)
        lib.path = p
        try:
            _declare(lib)
        except (AttributeError, ValueError) as e:               # a build of the library without one of the symbols used here:
            tried.append(f"{p}: {e}")                           # not usable -- callers fall back (save.H5ResultSink: a file beside the scene)
            continue
        _lib = lib
        return lib
    _why = "no HDF5 C library could be loaded (set HIMO_LIBHDF5=/path/to/libhdf5.so); tried: " + ("; ".join(tried) or "nothing found")
    raise ImportError(_why)


def available() -> bool:
    try:
        load()
        return True
    except ImportError:
        return False


def _declare(lib):
    hid = ctypes.c_int64 if lib.version >= (1, 10, 0) else ctypes.c_int       # hid_t grew to 64 bits in 1.10
    lib.hid = hid
    c, P = ctypes, ctypes.POINTER
    sig = {
        "H5Fcreate": (hid, [c.c_char_p, c.c_uint, hid, hid]), "H5Fopen": (hid, [c.c_char_p, c.c_uint, hid]),
        "H5Fclose": (c.c_int, [hid]), "H5Fflush": (c.c_int, [hid, c.c_int]),
        "H5Pcreate": (hid, [hid]), "H5Pclose": (c.c_int, [hid]), "H5Pset_libver_bounds": (c.c_int, [hid, c.c_int, c.c_int]),
        "H5Pset_chunk": (c.c_int, [hid, c.c_int, P(c.c_uint64)]), "H5Pset_deflate": (c.c_int, [hid, c.c_uint]),
        "H5Pset_shuffle": (c.c_int, [hid]), "H5Pset_fletcher32": (c.c_int, [hid]), "H5Pset_layout": (c.c_int, [hid, c.c_int]),
        "H5Pset_obj_track_times": (c.c_int, [hid, c.c_int]), "H5Pset_fclose_degree": (c.c_int, [hid, c.c_int]),
        "H5Gcreate2": (hid, [hid, c.c_char_p, hid, hid, hid]), "H5Gopen2": (hid, [hid, c.c_char_p, hid]), "H5Gclose": (c.c_int, [hid]),
        "H5Oopen": (hid, [hid, c.c_char_p, hid]), "H5Oclose": (c.c_int, [hid]), "H5Iget_type": (c.c_int, [hid]),
        "H5Lexists": (c.c_int, [hid, c.c_char_p, hid]), "H5Ldelete": (c.c_int, [hid, c.c_char_p, hid]),
        "H5Lget_name_by_idx": (c.c_ssize_t, [hid, c.c_char_p, c.c_int, c.c_int, c.c_uint64, c.c_char_p, c.c_size_t, hid]),
        "H5Screate_simple": (hid, [c.c_int, P(c.c_uint64), P(c.c_uint64)]), "H5Screate": (hid, [c.c_int]), "H5Sclose": (c.c_int, [hid]),
        "H5Sget_simple_extent_ndims": (c.c_int, [hid]), "H5Sget_simple_extent_dims": (c.c_int, [hid, P(c.c_uint64), P(c.c_uint64)]),
        "H5Dcreate2": (hid, [hid, c.c_char_p, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, c.c_char_p, hid]),
        "H5Dclose": (c.c_int, [hid]), "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]),
        "H5Dwrite": (c.c_int, [hid, hid, hid, hid, hid, c.c_void_p]), "H5Dread": (c.c_int, [hid, hid, hid, hid, hid, c.c_void_p]),
        "H5Tget_class": (c.c_int, [hid]), "H5Tget_size": (c.c_size_t, [hid]), "H5Tget_sign": (c.c_int, [hid]), "H5Tclose": (c.c_int, [hid]),
        "H5Tenum_create": (hid, [hid]), "H5Tenum_insert": (c.c_int, [hid, c.c_char_p, c.c_void_p]), "H5Tget_nmembers": (c.c_int, [hid]),
        "H5Tget_super": (hid, [hid]), "H5Tcopy": (hid, [hid]),
        "H5Eset_auto2": (c.c_int, [hid, c.c_void_p, c.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)                     # failures become Python exceptions below, not stderr dumps

    def g(name):
        return hid.in_dll(lib, name).value
    lib.P_FILE_ACCESS, lib.P_DATASET_CREATE = g("H5P_CLS_FILE_ACCESS_ID_g"), g("H5P_CLS_DATASET_CREATE_ID_g")
    lib.native = {np.dtype(k): g(v) for k, v in {
        "float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int8": "H5T_NATIVE_INT8_g", "uint8": "H5T_NATIVE_UINT8_g",
        "int16": "H5T_NATIVE_INT16_g", "uint16": "H5T_NATIVE_UINT16_g", "int32": "H5T_NATIVE_INT32_g", "uint32": "H5T_NATIVE_UINT32_g",
        "int64": "H5T_NATIVE_INT64_g", "uint64": "H5T_NATIVE_UINT64_g"}.items()}
    lib.filetype = {np.dtype(k): g(v) for k, v in {
        "float32": "H5T_IEEE_F32LE_g", "float64": "H5T_IEEE_F64LE_g", "int8": "H5T_STD_I8LE_g", "uint8": "H5T_STD_U8LE_g",
        "int16": "H5T_STD_I16LE_g", "uint16": "H5T_STD_U16LE_g", "int32": "H5T_STD_I32LE_g", "uint32": "H5T_STD_U32LE_g",
        "int64": "H5T_STD_I64LE_g", "uint64": "H5T_STD_U64LE_g"}.items()}


def _ok(v, what):
    if v < 0:
        raise OSError(f"libhdf5: {what} failed")
    return v


def _bool_type(lib):
    t = _ok(lib.H5Tenum_create(lib.native[np.dtype("int8")]), "H5Tenum_create")
    for name, val in ((b"FALSE", 0), (b"TRUE", 1)):                     # what h5py writes for numpy bool
        v = ctypes.c_int8(val)
        _ok(lib.H5Tenum_insert(t, name, ctypes.byref(v)), "H5Tenum_insert")
    return t


class Dataset:
    _id = 0
    _file = None

    def __init__(self, lib, did, file=None):
        # ``file``: the File this handle lives in.  The file is opened H5F_CLOSE_STRONG (closing it closes every handle open in
        # it, as h5py does), and the library RE-USES identifier values: a child object collected after its file was closed must
        # not close "its" identifier again -- by then it may name another thread's dataset or group
        self._lib, self._id, self._file = lib, did, file
        sp = _ok(lib.H5Dget_space(did), "H5Dget_space")
        nd = lib.H5Sget_simple_extent_ndims(sp)
        dims = (ctypes.c_uint64 * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(sp, dims, None)
        lib.H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        t = _ok(lib.H5Dget_type(did), "H5Dget_type")
        cls, size = lib.H5Tget_class(t), lib.H5Tget_size(t)
        self._as_bool = False
        if cls == H5T_ENUM:
            self._as_bool = lib.H5Tget_nmembers(t) == 2 and size == 1
            self._mem = np.dtype("int8") if size == 1 else np.dtype(f"int{8 * size}")
        elif cls == H5T_FLOAT:
            self._mem = np.dtype(f"float{8 * size}")
        elif cls == H5T_INTEGER:
            self._mem = np.dtype(("uint" if lib.H5Tget_sign(t) == H5T_SGN_NONE else "int") + str(8 * size))
        else:
            lib.H5Tclose(t)
            raise TypeError(f"HDF5 datatype class {cls} is not read by this binding")
        lib.H5Tclose(t)
        self.dtype = np.dtype(bool) if self._as_bool else self._mem

    def read(self):
        out = np.empty(self.shape, self._mem)
        if out.size:
            _ok(self._lib.H5Dread(self._id, self._lib.native[self._mem], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), "H5Dread")
        return out.astype(bool) if self._as_bool else out

    def __getitem__(self, key):
        a = self.read()
        return a[key] if a.ndim or key != () else a[()]

    def close(self):
        if self._id and (self._file is None or self._file._fid):
            self._lib.H5Dclose(self._id)
        self._id = 0

    __del__ = close


class Group:
    _id = 0
    _owned = True
    _file = None

    def __init__(self, lib, gid, owned=True, file=None):
        self._lib, self._id, self._owned, self._file = lib, gid, owned, file

    def __contains__(self, name):
        return self._lib.H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def keys(self):
        # links by index until the library says there is none (H5Gget_num_objs is a deprecated symbol that builds without the
        # 1.6 API lack; H5Gget_info's struct differs between versions)
        out = []
        i = 0
        while True:
            ln = self._lib.H5Lget_name_by_idx(self._id, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
            if ln < 0:
                break
            i += 1
            buf = ctypes.create_string_buffer(ln + 1)
            self._lib.H5Lget_name_by_idx(self._id, b".", H5_INDEX_NAME, H5_ITER_INC, i - 1, buf, ln + 1, H5P_DEFAULT)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __getitem__(self, name):
        if name not in self:
            raise KeyError(name)
        oid = _ok(self._lib.H5Oopen(self._id, name.encode(), H5P_DEFAULT), f"H5Oopen({name})")
        kind = self._lib.H5Iget_type(oid)
        self._lib.H5Oclose(oid)
        home = self._file if self._file is not None else self            # (a File is its own home)
        if kind == 2:                                                    # H5I_GROUP
            return Group(self._lib, _ok(self._lib.H5Gopen2(self._id, name.encode(), H5P_DEFAULT), "H5Gopen2"), file=home)
        return Dataset(self._lib, _ok(self._lib.H5Dopen2(self._id, name.encode(), H5P_DEFAULT), "H5Dopen2"), file=home)

    def __delitem__(self, name):
        _ok(self._lib.H5Ldelete(self._id, name.encode(), H5P_DEFAULT), f"H5Ldelete({name})")

    def create_group(self, name):
        return Group(self._lib, _ok(self._lib.H5Gcreate2(self._id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Gcreate2({name})"),
                     file=self._file if self._file is not None else self)

    def create_dataset(self, name, data, chunks=None, compression=None, shuffle=False, fletcher32=False, compact=False):
        """``h5py``'s ``create_dataset(name, data=...)``: contiguous layout unless ``chunks`` / ``compression="gzip"``."""
        lib = self._lib
        a = np.asarray(data)
        is_bool = a.dtype == np.bool_
        mem = a.astype(np.int8) if is_bool else np.ascontiguousarray(a)
        if not is_bool and mem.dtype not in lib.native:
            raise TypeError(f"dtype {a.dtype} is not written by this binding")
        ftype = _bool_type(lib) if is_bool else lib.filetype[mem.dtype]
        mtype = ftype if is_bool else lib.native[mem.dtype]
        if a.ndim:
            dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
            space = _ok(lib.H5Screate_simple(a.ndim, dims, None), "H5Screate_simple")
        else:
            space = _ok(lib.H5Screate(0), "H5Screate")                   # H5S_SCALAR
        dcpl = _ok(lib.H5Pcreate(lib.P_DATASET_CREATE), "H5Pcreate")
        lib.H5Pset_obj_track_times(dcpl, 0)                              # h5py's default (track_times=False)
        if compression or shuffle or fletcher32:
            chunks = chunks or tuple(min(s, 4096) for s in a.shape)
        if chunks:
            _ok(lib.H5Pset_chunk(dcpl, a.ndim, (ctypes.c_uint64 * a.ndim)(*chunks)), "H5Pset_chunk")
            if shuffle:
                _ok(lib.H5Pset_shuffle(dcpl), "H5Pset_shuffle")
            if compression:
                _ok(lib.H5Pset_deflate(dcpl, 4), "H5Pset_deflate")
            if fletcher32:
                _ok(lib.H5Pset_fletcher32(dcpl), "H5Pset_fletcher32")
        elif compact:
            _ok(lib.H5Pset_layout(dcpl, 0), "H5Pset_layout")
        did = lib.H5Dcreate2(self._id, name.encode(), ftype, space, H5P_DEFAULT, dcpl, H5P_DEFAULT)
        try:
            _ok(did, f"H5Dcreate2({name})")
            if mem.size:
                _ok(lib.H5Dwrite(did, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, mem.ctypes.data), f"H5Dwrite({name})")
        finally:
            if did >= 0:
                lib.H5Dclose(did)
            lib.H5Pclose(dcpl)
            lib.H5Sclose(space)
            if is_bool:
                lib.H5Tclose(ftype)

    def close(self):
        if self._id and self._owned and (self._file is None or self._file._fid):
            self._lib.H5Gclose(self._id)
        self._id = 0

    __del__ = close


class File(Group):
    _fid = 0

    """``File(path, "r" | "a" | "r+" | "w")``; ``libver="latest"`` writes the 1.10 file format (version-2 object headers,
    link messages) instead of the library's -- and h5py's -- default earliest-compatible one."""

    def __init__(self, path, mode="r", libver="earliest"):
        lib = load()
        lib.H5Eset_auto2(0, None, None)                                  # (the library's error printing is per THREAD: writer threads too)
        fapl = _ok(lib.H5Pcreate(lib.P_FILE_ACCESS), "H5Pcreate")
        lib.H5Pset_fclose_degree(fapl, 3)                                # H5F_CLOSE_STRONG, as h5py: closing the file closes what is open in it
        if libver == "latest":
            high = 2 if lib.version >= (1, 10, 2) else 1                 # H5F_LIBVER_LATEST's enum value moved in 1.10.2
            if lib.version >= (1, 12, 0):
                high = 3 if lib.version < (1, 14, 0) else 4
            lib.H5Pset_libver_bounds(fapl, high, high)
        p = os.fspath(path).encode()
        if mode == "w":
            fid = lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, fapl)
        elif mode in ("a", "r+"):
            fid = lib.H5Fopen(p, H5F_ACC_RDWR, fapl) if os.path.exists(path) else lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, fapl)
        elif mode == "r":
            fid = lib.H5Fopen(p, H5F_ACC_RDONLY, fapl)
        else:
            raise ValueError(f"mode {mode!r}")
        lib.H5Pclose(fapl)
        _ok(fid, f"opening {path} ({mode})")
        self._fid = fid
        super().__init__(lib, _ok(lib.H5Gopen2(fid, b"/", H5P_DEFAULT), "H5Gopen2(/)"))

    def close(self):
        super().close()
        if getattr(self, "_fid", 0):
            self._lib.H5Fclose(self._fid)
            self._fid = 0

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
