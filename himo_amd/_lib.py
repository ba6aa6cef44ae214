"""ctypes binding of libhimo_amd.so (declared in include/himo_amd.h).

There is deliberately NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing callers get an ImportError that says how to build it; if no GPU is
visible, the operators raise instead of computing on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint, c_void_p
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("HIMO_AMD_LIB", _PKG / "libhimo_amd.so"))

# name -> (restype, argtypes); mirrors include/himo_amd.h one to one
SIGNATURES = {
    "himo_abi_version": (c_int, []),
    "himo_abi_sizeof": (c_size_t, [ctypes.c_char_p]),
    "himo_status_string": (c_char_p, [c_int]),
    "himo_last_hip_error": (c_char_p, []),
    "himo_prof_enable": (None, [c_int]),
    "himo_prof_filter": (None, [ctypes.c_char_p]),
    "himo_prof_reset": (None, []),
    "himo_prof_summary": (c_size_t, [ctypes.c_char_p, c_size_t]),
    "himo_mfma_sustained_tflops": (c_int, [c_int, c_int, c_double, ctypes.POINTER(c_double), c_void_p]),
    "himo_compdis_workspace_bytes": (c_size_t, [c_int]),
    "himo_compdis_batch": (c_int, [c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                   c_double, c_uint, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   ctypes.POINTER(c_float), c_float, c_void_p, c_size_t, c_void_p]),
    "himo_compdis_frame": (c_int, [c_int64, ctypes.POINTER(c_double), ctypes.POINTER(c_double), c_void_p, c_int,
                                   c_void_p, c_void_p, c_double, c_uint, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "himo_flow2compdis": (c_int, [c_int64, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p]),
    "himo_refine_pts": (c_int, [c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "himo_ego_pts_mask": (c_int, [c_int64, c_void_p, c_int, ctypes.POINTER(c_float), c_void_p, c_void_p]),
    "himo_dt0": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "himo_lz4_frame_decompress": (c_int64, [c_void_p, c_int64, c_void_p, c_int64]),
    "himo_nn_search": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p,
                               c_void_p, c_void_p]),
    "himo_sqrt_inplace": (c_int, [c_int64, c_void_p, c_int, c_void_p]),
    "himo_eval_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "himo_eval_instances": (c_int, [c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(ctypes.c_uint8), c_double,
                                    c_int, c_uint, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
}

FLAG_F32_CHAIN = 0x1
FLAG_RAW = 0x2
FLAG_SCANIA = 0x4
FLAG_POSE_IS_EGO = 0x8

OK, ERR_INVALID_ARGUMENT, ERR_EMPTY_FRAME, ERR_WORKSPACE, ERR_SINGULAR_POSE, ERR_HIP, ERR_UNSUPPORTED = range(7)

_lib = None


def register(signatures: dict) -> None:
    """Let sibling modules declare the entry points of further translation units."""
    SIGNATURES.update(signatures)
    if _lib is not None:
        _bind(_lib, signatures)


def _bind(lib, signatures):
    for name, (restype, argtypes) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - build/ABI mismatch
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild with "
                              f"`make -C {_PKG / 'csrc'}` (or python -c 'import __graft_entry__ as g; g.build()')") from e
        fn.restype = restype
        fn.argtypes = argtypes


def load():
    """Load libhimo_amd.so once; raises ImportError (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"HIP extension {LIB_PATH} is missing -- build it with `make -C {_PKG / 'csrc'}` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch first: libhimo_amd.so must bind to the HIP runtime torch ships (one runtime per process);
        # loading it before torch would pull in /opt/rocm's libamdhip64 as a second, device-less runtime
        import torch  # noqa: F401
        lib = ctypes.CDLL(str(LIB_PATH))
        _bind(lib, SIGNATURES)
        if lib.himo_abi_version() != 1:
            raise ImportError(f"{LIB_PATH}: ABI version {lib.himo_abi_version()} != 1")
        _lib = lib
    return _lib


class HimoError(RuntimeError):
    pass


def check(status: int, what: str = "") -> None:
    """Map a himo_status to the exception the reference would raise at the same point."""
    if status == OK:
        return
    lib = load()
    msg = lib.himo_status_string(status).decode()
    if status == ERR_EMPTY_FRAME:
        raise ValueError("max() arg is an empty sequence")          # save_zip.py:120 on an empty sweep
    if status == ERR_SINGULAR_POSE:
        import numpy as np
        raise np.linalg.LinAlgError("Singular matrix")               # save_zip.py:115
    if status == ERR_HIP:
        raise HimoError(f"{what}: {msg}: {lib.himo_last_hip_error().decode()}")
    if status == ERR_INVALID_ARGUMENT:
        raise ValueError(f"{what}: {msg}")
    raise HimoError(f"{what}: {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("himo_amd needs a HIP device (MI355X / gfx950); none is visible and there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_LIBC = None


def pinned_empty(numel: int, dtype=None):
    """``torch.empty(numel, dtype, pin_memory=True)`` whose pages stay OUT of forked children (madvise MADV_DONTFORK).  The runtime maps
    pinned host memory into the GPU's address space through the pages of an ordinary private mapping; a fork() write-protects such pages
    for copy-on-write, the driver's notifier then takes the mapping away from the GPU, and the first device operation afterwards pays for
    putting it back -- 1.5 s per GB of pinned memory the process holds (scripts/exp_fork_cost.py: 3.06 s with 2 GB pinned, 0.22 s with
    the advice).  Reader processes (feeder.ReaderPool) are forked from processes that hold gigabytes of staging arenas."""
    import torch
    global _LIBC
    t = torch.empty(int(numel), dtype=dtype if dtype is not None else torch.uint8, pin_memory=True)
    lo = (t.data_ptr() + 4095) & ~4095
    n = (t.data_ptr() + t.numel() * t.element_size() - lo) & ~4095
    if n > 0 and os.environ.get("HIMO_PINNED_DONTFORK", "1") != "0":
        if _LIBC is None:
            _LIBC = ctypes.CDLL(None, use_errno=True)
        _LIBC.madvise(c_void_p(lo), c_size_t(n), 10)          # MADV_DONTFORK; advice only: a refusal changes nothing but fork's cost
    return t


def stream_handle() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def prof_start(only: str | None = None):
    """Start per-kernel timing (HIP events around every launch whose ProfScope name contains ``only``; all when None)."""
    lib = load()
    lib.himo_prof_reset()
    lib.himo_prof_filter((only or "").encode())
    lib.himo_prof_enable(1)


def prof_stop() -> dict:
    """{kernel name: {"count", "total_ms", "avg_ms", "min_ms", "max_ms"}} for launches since prof_start()."""
    lib = load()
    lib.himo_prof_enable(0)
    need = lib.himo_prof_summary(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    lib.himo_prof_summary(buf, len(buf))
    lib.himo_prof_reset()
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, tot, mn, mx = line.split()
        out[name] = {"count": int(n), "total_ms": float(tot), "avg_ms": float(tot) / max(int(n), 1),
                     "min_ms": float(mn), "max_ms": float(mx)}
    return out


def mfma_sustained_tflops(kind: str = "f16", zero_operands: bool = False, seconds: float = 0.4) -> float:
    """himo_mfma_sustained_tflops: the rate the chip's matrix pipes sustain on their own (register-resident independent chains
    on every SIMD) for ``kind`` "f16" | "bf16" | "f32", with random or all-zero operand bits.  Synchronous; ~``seconds``."""
    out = ctypes.c_double(0.0)
    check(load().himo_mfma_sustained_tflops({"f16": 0, "bf16": 1, "f32": 2}[kind], int(bool(zero_operands)), float(seconds),
                                           ctypes.byref(out), stream_handle()), "himo_mfma_sustained_tflops")
    return out.value
