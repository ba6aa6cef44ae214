"""The sweep-to-sweep neighbour search (csrc/nngrid.hip) on the uniform bench cloud and on LiDAR-shaped sweeps: time of the build and
of the query kernel per cell size, for the in-tree library and for variants with other block shapes.

  python scripts/exp_nn_grid.py build   -> writes build/variants/nng_<name>/libhimo_amd.so for every variant below (here or on the box;
                                           the constants are replaced in a COPY of the source, the shipped file has no switches)
  python scripts/exp_nn_grid.py         -> the table for the library selected by HIMO_AMD_LIB (default: in-tree)
  python scripts/exp_nn_grid.py all     -> the table for the in-tree library and every built variant (one subprocess each)
"""
import os, subprocess, sys
from pathlib import Path

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
VARIANTS = {                                  # name -> {constant: value}
    "w4": {"kNngWaves": 4}, "w16": {"kNngWaves": 16},
    "seg8": {"kNngSegCells": 8}, "seg32": {"kNngSegCells": 32},
}


def build():
    import re
    src = (R / "himo_amd/csrc/nngrid.hip").read_text()
    subprocess.run(["make", "-C", str(R / "himo_amd/csrc"), "-j16"], check=True, stdout=subprocess.DEVNULL)
    objs = sorted(str(p) for p in (R / "build/csrc").glob("*.o") if p.name != "nngrid.o")
    for name, consts in VARIANTS.items():
        out = R / "build/variants" / f"nng_{name}"
        out.mkdir(parents=True, exist_ok=True)
        s = src
        for k, v in consts.items():
            s, n = re.subn(rf"constexpr int {k} = \d+;", f"constexpr int {k} = {v};", s)
            assert n == 1, k
        s = s.replace('#include "nngrid.h"', f'#include "{R}/himo_amd/csrc/nngrid.h"')
        (out / "nngrid.hip").write_text(s)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-w", f"-I{R}/himo_amd/csrc",
                        "-c", str(out / "nngrid.hip"), "-o", str(out / "nngrid.o")], check=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out / "libhimo_amd.so"), str(out / "nngrid.o")] + objs, check=True)
        print("built", out / "libhimo_amd.so")


def table():
    import numpy as np, torch
    from himo_amd import _lib, ssl_loss            # (ssl_loss registers himo_nn_grid's signature)
    from himo_amd.synthetic import make_frame
    lib = _lib.load()
    dev = _lib.require_gpu()
    print("library:", os.environ.get("HIMO_AMD_LIB", "in-tree"))
    for cloud in ("uniform", "rings"):
        q, r = (torch.from_numpy(make_frame(s, cloud=cloud)["pc0"][:, :3].copy()).to(dev) for s in (0, 1))
        n = len(q)
        d2 = torch.empty(n, dtype=torch.float32, device=dev)
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        ref = None
        for cell, w in [(2.0, 52), (1.0, 104), (0.5, 208), (0.25, 416)]:
            ws = torch.empty(int(lib.himo_nn_grid_workspace_bytes(n, w, w)), dtype=torch.uint8, device=dev)

            def call():
                _lib.check(lib.himo_nn_grid(n, _lib.ptr(q), n, _lib.ptr(r), -52.0, -52.0, cell, w, w, _lib.ptr(d2), _lib.ptr(idx),
                                            _lib.ptr(ws), ws.numel(), _lib.stream_handle()), "himo_nn_grid")
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            _lib.prof_start()
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            p = _lib.prof_stop()
            if ref is None:
                ref = idx.clone()
            same = bool(torch.equal(ref, idx))
            print(f"  {cloud:8s} cell {cell:5.2f} m: build {p['nn_grid_build']['avg_ms'] * 1e3:7.1f} us  query {p['nn_grid_query_kernel']['avg_ms'] * 1e3:7.1f} us "
                  f"(min {p['nn_grid_query_kernel']['min_ms'] * 1e3:7.1f})  same rows as the first cell size: {same}")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    if mode == "build":
        build()
    elif mode == "all":
        for name in ["in-tree"] + sorted(VARIANTS):
            env = dict(os.environ)
            if name != "in-tree":
                lib = R / "build/variants" / f"nng_{name}" / "libhimo_amd.so"
                if not lib.exists():
                    continue
                env["HIMO_AMD_LIB"] = str(lib)
            subprocess.run([sys.executable, __file__], env=env, check=False)
    else:
        table()
