"""persistent blocks for the 64-channel 3x3 layers -- a block walks several 8-row x 64-channel tiles, the slab pipeline
(patch DMA, weight rotation) running on ACROSS tiles so that only a block's first tile pays a prologue.  The variant lives in
scripts/exp_persistent_tiles.patch (against csrc/convsg.hip: `conv3_presplit_persist_kernel`, tile hint 0x1009), NOT in the shipped source.

  python scripts/exp_persistent_tiles.py build   applies the patch to a COPY of the source, builds build/variants/persist/libhimo_amd.so
  python scripts/exp_persistent_tiles.py         (GPU box) times the 64-channel layer shapes: shipped 8-row tiles (hint 0x1008) against the
                                                 persistent variant (0x1009), same library, bit-identity checked
"""
import os, shutil, subprocess, sys
from pathlib import Path

R = Path(__file__).resolve().parents[1]
OUT = R / "build/variants/persist"
LIB = OUT / "libhimo_amd.so"


def build():
    src = OUT / "src"
    shutil.rmtree(src, ignore_errors=True)
    (src / "himo_amd").mkdir(parents=True)
    shutil.copytree(R / "himo_amd/csrc", src / "himo_amd/csrc")
    shutil.copytree(R / "include", src / "include")
    subprocess.run(["patch", "-p1", "-i", str(R / "scripts/exp_persistent_tiles.patch")], cwd=src, check=True)
    subprocess.run(["make", "-C", str(R / "himo_amd/csrc"), "-j16"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-w", "-ffp-contract=off", "-c",
                    str(src / "himo_amd/csrc/convsg.hip"), "-o", str(OUT / "convsg.o")], check=True)
    objs = sorted(str(p) for p in (R / "build/csrc").glob("*.o") if p.name != "convsg.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), str(OUT / "convsg.o")] + objs, check=True)
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-w", "-ffp-contract=off", "-S", "--cuda-device-only",
                          "-o", "-", str(src / "himo_amd/csrc/convsg.hip")], capture_output=True, text=True).stdout
    for blk in asm.split(".name:")[1:]:
        if "conv3_presplit_persist_kernel" in blk.splitlines()[0]:
            stats = {k: v for k, v in (ln.strip().split(":", 1) for ln in blk.splitlines()[1:12] if ":" in ln)}
            print("persistent kernel", blk.splitlines()[0].strip()[:60], "vgpr", stats.get(".vgpr_count", "?").strip(),
                  "vgpr spills", stats.get(".vgpr_spill_count", "?").strip(), "sgpr spills", stats.get(".sgpr_spill_count", "?").strip())
    print("built", LIB)


def run():
    os.environ["HIMO_AMD_LIB"] = str(LIB)
    sys.path.insert(0, str(R))
    import torch
    from himo_amd import _lib
    from himo_amd.seflow.model import conv2d_nhwc, ACT_SPLIT_IN, ACT_SPLIT_OUT
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for (name, n, h, w, epi) in [("enc1.x (16 samples)", 48, 256, 256, 1), ("dec3.u5 | dec4 (16 samples)", 16, 512, 512, 0),
                                 ("enc1.x (1 sample)", 3, 256, 256, 1), ("dec3.u5 | dec4 (1 sample)", 1, 512, 512, 0)]:
        ci = co = 64
        x = torch.randn(n, h, w, ci, device=dev)
        xin = conv2d_nhwc(x, torch.randn(3, 3, ci, ci, device=dev) * 0.05, torch.zeros(ci, device=dev), precision="f16x2", act_layout=ACT_SPLIT_OUT)
        del x
        wt = torch.randn(3, 3, ci, co, device=dev) * 0.05
        b = torch.randn(co, device=dev) * 0.1; sc = torch.rand(co, device=dev) + 0.5; sh = torch.randn(co, device=dev) * 0.1
        lay = ACT_SPLIT_IN | ACT_SPLIT_OUT
        outs, times = {}, {}
        for rnd in range(3):                               # interleaved rounds: the clock drifts with load
            for hint in (0x1008, 0x1009):
                for _ in range(2):
                    outs[hint] = conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=hint, act_layout=lay)
                torch.cuda.synchronize()
                _lib.prof_start(only="conv3x3_f16x2")
                for _ in range(8):
                    conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", tile_hint=hint, act_layout=lay)
                torch.cuda.synchronize()
                p = _lib.prof_stop()
                ms = min(v["avg_ms"] for v in p.values())
                times[hint] = min(times.get(hint, 1e9), ms)
        fl = 2.0 * n * h * w * ci * co * 9
        same = torch.equal(outs[0x1008], outs[0x1009])
        print(f"{name:30s}: 8-row tiles {times[0x1008] * 1e3:8.1f} us ({fl / times[0x1008] / 1e9:4.0f} TF-eq)   persistent {times[0x1009] * 1e3:8.1f} us "
              f"({fl / times[0x1009] / 1e9:4.0f} TF-eq)   {times[0x1008] / times[0x1009]:.3f}x   same bits: {same}", flush=True)
        del xin, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
