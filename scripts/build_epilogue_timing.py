"""Builds build/variants/epi/libhimo_amd.so: the library with s_memtime stamps in the 3x3 split-activation kernel (csrc/convsg.hip
conv3_presplit_kernel) -- block start, main loop start, main loop end, stores issued, stores acknowledged -- written per block
into a device table that scripts/exp_epilogue.py reads.  The shipped kernel carries no instrumentation: this script inserts it into a
COPY of the source at build time (anchors below; it fails loudly when one no longer matches).
usage (repo root, here or on the GPU box): python scripts/build_epilogue_timing.py"""
import subprocess, sys
from pathlib import Path

R = Path(__file__).resolve().parents[1]
src = (R / "himo_amd/csrc/convsg.hip").read_text()
out = R / "build/variants/epi"
out.mkdir(parents=True, exist_ok=True)


def insert(text, anchor, new, before=True, nth=0):
    pos = -1
    for _ in range(nth + 1):
        pos = text.index(anchor, pos + 1)
    return text[:pos] + new + text[pos:] if before else text[:pos + len(anchor)] + new + text[pos + len(anchor):]


s = src
s = insert(s, "namespace himo {\n", "constexpr int kEpiBlocks = 65536;\n__device__ unsigned long long g_epi[kEpiBlocks * 4];      // per block (of the last launch): prologue, loop, epilogue to issue, to acknowledge -- no atomics: 32768 blocks on five addresses doubled the short kernels' time\n", before=False)
s = insert(s, "    __shared__ __attribute__((aligned(1024))) unsigned char patch[2 * kBuf];\n",
           "    const unsigned long long epi_t0 = __builtin_amdgcn_s_memtime();\n", before=False)            # conv3 kernel (first occurrence)
s = insert(s, "#pragma unroll 1\n    for (int slab = 0; slab < slabs; ++slab) {", "    const unsigned long long epi_t1 = __builtin_amdgcn_s_memtime();\n")
s = insert(s, "    float* __restrict__ yout = a.y + image_offset(img, a.n_inner, a.y_batch_stride, a.y_outer_stride);\n",
           "    const unsigned long long epi_t2 = __builtin_amdgcn_s_memtime();\n")
s = insert(s, "        return;\n",
           "        if (threadIdx.x == 0) {\n"
           "            const unsigned long long t3 = __builtin_amdgcn_s_memtime();\n"
           "            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
           "            const unsigned long long t4 = __builtin_amdgcn_s_memtime();\n"
           "            if (blockIdx.x < kEpiBlocks) {\n"
           "                unsigned long long* e = g_epi + (size_t)blockIdx.x * 4;\n"
           "                e[0] = epi_t1 - epi_t0; e[1] = epi_t2 - epi_t1; e[2] = t3 - epi_t2; e[3] = t4 - epi_t2;\n"
           "            }\n"
           "        }\n")
s += ('\nextern "C" int himo_exp_epi_reset(void) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(himo::g_epi)) != hipSuccess) return 1; return (int)hipMemset(p, 0, sizeof(unsigned long long) * himo::kEpiBlocks * 4); }\n'
      'extern "C" int himo_exp_epi_read(unsigned long long* out, int n_blocks) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(himo::g_epi), (size_t)n_blocks * 32); }\n')
(out / "convsg.hip").write_text(s)
subprocess.check_call(["make", "-C", str(R / "himo_amd/csrc"), "-j8"], stdout=subprocess.DEVNULL)
hipcc = "/opt/rocm/bin/hipcc"
subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-ffp-contract=off",
                       "-I", str(R / "himo_amd/csrc"), "-I", str(R / "include"), "-c", str(out / "convsg.hip"), "-o", str(out / "convsg.o")])
objs = [str(p) for p in sorted((R / "build/csrc").glob("*.o")) if p.name != "convsg.o"] + [str(out / "convsg.o")]
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out / "libhimo_amd.so")] + objs)
print(out / "libhimo_amd.so")
