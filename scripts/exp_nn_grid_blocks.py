"""Per-block life of the neighbour-search query kernel (csrc/nngrid.hip) on the uniform / LiDAR-shaped sweeps: s_memtime stamps, rings walked,
segments and candidates per 64-query block, written by an INSTRUMENTED COPY of the kernel (build/variants/nng_stamps/; the shipped source has
no instrumentation).   python scripts/exp_nn_grid_blocks.py build   |   python scripts/exp_nn_grid_blocks.py   (GPU box)"""
import ctypes, os, subprocess, sys
from pathlib import Path

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
OUT = R / "build/variants/nng_stamps"


def ins(text, anchor, new, before=False):
    pos = text.index(anchor)
    return text[:pos] + new + text[pos:] if before else text[:pos + len(anchor)] + new + text[pos + len(anchor):]


def build():
    s = (R / "himo_amd/csrc/nngrid.hip").read_text()
    s = s.replace('#include "nngrid.h"', f'#include "{R}/himo_amd/csrc/nngrid.h"')
    s = ins(s, "namespace himo {\n", "constexpr int kStampBlocks = 8192;\n__device__ unsigned long long g_stamp[kStampBlocks * 6];\n")
    s = ins(s, "    const int base = blockIdx.x * 64;\n    if (base >= Q.n) return;\n",
            "    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();\n    int st_rings = 0, st_segs = 0, st_cand = 0, st_merges = 0;\n")
    s = ins(s, "            bool alive = act;\n", "            ++st_segs;\n")
    s = ins(s, "                const int p1 = n0, p2 = p1 + n1, p3 = p2 + n2, total = p3 + n3;\n", "                ++st_rings; st_cand += total;\n")
    s = ins(s, "                if (fresh) {            // merge", "                ++st_merges;\n", before=True)
    s = ins(s, "    if (wave == 0 && valid) {\n        const int o = __float_as_int(me.w);",
            "    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < kStampBlocks) {\n"
            "        unsigned long long* e = g_stamp + (size_t)blockIdx.x * 6;\n"
            "        e[0] = st_t0; e[1] = __builtin_amdgcn_s_memtime(); e[2] = st_rings; e[3] = st_segs; e[4] = st_cand; e[5] = st_merges;\n    }\n", before=True)
    s += '\nextern "C" int himo_exp_nng_stamps(unsigned long long* out, int n_blocks) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(himo::g_stamp), (size_t)n_blocks * 48); }\n'
    OUT.mkdir(parents=True, exist_ok=True)
    (OUT / "nngrid.hip").write_text(s)
    subprocess.run(["make", "-C", str(R / "himo_amd/csrc"), "-j16"], check=True, stdout=subprocess.DEVNULL)
    objs = sorted(str(p) for p in (R / "build/csrc").glob("*.o") if p.name != "nngrid.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-w", "-c", str(OUT / "nngrid.hip"), "-o", str(OUT / "nngrid.o")], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT / "libhimo_amd.so"), str(OUT / "nngrid.o")] + objs, check=True)
    print("built", OUT / "libhimo_amd.so")


def run():
    os.environ["HIMO_AMD_LIB"] = str(OUT / "libhimo_amd.so")
    import numpy as np, torch
    from himo_amd import _lib, ssl_loss
    from himo_amd.synthetic import make_frame
    lib = _lib.load()
    dev = _lib.require_gpu()
    lib.himo_exp_nng_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for cloud in ("uniform", "rings"):
        q, r = (torch.from_numpy(make_frame(s, cloud=cloud)["pc0"][:, :3].copy()).to(dev) for s in (0, 1))
        n = len(q)
        for cell, w in [(1.0, 104), (0.5, 208)]:
            d2 = torch.empty(n, dtype=torch.float32, device=dev); idx = torch.empty(n, dtype=torch.int32, device=dev)
            ws = torch.empty(int(lib.himo_nn_grid_workspace_bytes(n, w, w)), dtype=torch.uint8, device=dev)
            for _ in range(3):
                _lib.check(lib.himo_nn_grid(n, _lib.ptr(q), n, _lib.ptr(r), -52.0, -52.0, cell, w, w, _lib.ptr(d2), _lib.ptr(idx), _lib.ptr(ws), ws.numel(),
                                            _lib.stream_handle()), "himo_nn_grid")
            torch.cuda.synchronize()
            nb = (n + 63) // 64
            st = np.zeros((nb, 6), dtype=np.uint64)
            assert lib.himo_exp_nng_stamps(st.ctypes.data, nb) == 0
            t0, t1 = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64)
            life = (t1 - t0) / 100.0                    # s_memtime ticks at 100 MHz -> us
            start = (t0 - t0.min()) / 100.0
            end = (t1 - t0.min()) / 100.0
            rings, segs, cand, merges = (st[:, k].astype(np.int64) for k in (2, 3, 4, 5))
            print(f"{cloud} cell {cell}: {nb} blocks, kernel span {end.max():.1f} us; block life us: mean {life.mean():.1f} p50 {np.median(life):.1f} p90 {np.quantile(life, .9):.1f} "
                  f"p99 {np.quantile(life, .99):.1f} max {life.max():.1f}; last block starts at {start.max():.1f} us; blocks started after 10 us: {(start > 10).sum()}")
            print(f"    per block: rings mean {rings.mean():.1f} max {rings.max()}, segments mean {segs.mean():.2f} max {segs.max()}, candidates mean {cand.mean():.0f} max {cand.max()} "
                  f"total {cand.sum():.3g}, merges mean {merges.mean():.1f} max {merges.max()}")
            order = np.argsort(-life)[:8]
            for b in order:
                print(f"      block {b}: start {start[b]:.1f} life {life[b]:.1f} us rings {rings[b]} segs {segs[b]} cand {cand[b]} merges {merges[b]}")
            # a linear model of a block's life
            A = np.stack([np.ones(nb), rings, segs, cand, merges], 1).astype(np.float64)
            coef, *_ = np.linalg.lstsq(A, life, rcond=None)
            print(f"    life ~ {coef[0]:.2f} + {coef[1]:.3f}/ring + {coef[2]:.2f}/segment + {coef[3] * 1e3:.3f}/1000 candidates + {coef[4]:.3f}/merge  (us)")


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
