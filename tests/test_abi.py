"""The C-ABI library loads and exports every symbol include/*.h declares (CPU only, no compute)."""
import ctypes
import re

import pytest

from conftest import REPO


def declared_symbols():
    names = []
    for h in sorted((REPO / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names += re.findall(r"\b(himo_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "himo_compdis_batch" in syms and "himo_compdis_frame" in syms and len(syms) >= 10


def test_library_exports_every_declared_symbol():
    from himo_amd import _lib
    import himo_amd.seflow.model  # noqa: F401  (registers the network entry points' signatures)
    import himo_amd.ssl_loss  # noqa: F401
    import himo_amd.fastnsf  # noqa: F401
    import himo_amd.seflow.train  # noqa: F401
    import himo_amd.seflow.ssl_label  # noqa: F401
    lib = _lib.load()
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} declared in include/ but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in himo_amd/_lib.py"
    assert lib.himo_abi_version() == 1
    assert lib.himo_status_string(0) == b"ok"
    assert b"empty" in lib.himo_status_string(2)
    assert lib.himo_compdis_workspace_bytes(1) >= 4 + 96
    assert lib.himo_compdis_workspace_bytes(256) >= 256 * 100


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from himo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    from himo_amd import utils
    with pytest.raises(RuntimeError, match="no CPU path"):
        utils.flow2compDis(np.zeros((4, 3), np.float32), np.zeros(4, np.float32), 0.1)


def test_status_to_exception_mapping():
    import numpy as np
    from himo_amd import _lib
    with pytest.raises(ValueError, match="empty sequence"):
        _lib.check(_lib.ERR_EMPTY_FRAME)
    with pytest.raises(np.linalg.LinAlgError):
        _lib.check(_lib.ERR_SINGULAR_POSE)
    with pytest.raises(ValueError):
        _lib.check(_lib.ERR_INVALID_ARGUMENT, "x")
    _lib.check(_lib.OK)


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(import|from)\s+\S*(himo_oracle|seflow_oracle|oracle)\b", re.M)
    for p in list((REPO / "himo_amd").rglob("*.py")):
        src = p.read_text()
        assert not pat.search(src), p
        assert "sys.path" not in src or "oracle" not in src, p      # no path games either


def test_ctypes_mirrors_have_the_c_struct_sizes():
    """The structs that cross the boundary by address: the Python mirrors must match the compiled layout."""
    from himo_amd import _lib
    from himo_amd.eval import InstanceRecord
    from himo_amd.seflow.model import ConvDesc, HimoOp, HimoSweep, HimoHeadSample
    import ctypes
    lib = _lib.load()
    for name, mirror in (("himo_conv_desc", ConvDesc), ("himo_op", HimoOp), ("himo_sweep", HimoSweep), ("himo_instance_record", InstanceRecord),
                         ("himo_head_sample", HimoHeadSample)):
        assert lib.himo_abi_sizeof(name.encode()) == ctypes.sizeof(mirror), name
    assert lib.himo_abi_sizeof(b"no_such_struct") == 0
