// mfma_peak.hip -- what the matrix pipes of THIS chip sustain with nothing else going on: back-to-back independent
// v_mfma chains from registers (no memory traffic in the loop), random (non-zero) operand bits, long enough (>= 0.2 s per
// leg) to sit at the power-managed clock.  The denominators for "fraction of what the chip delivers" in DESIGN.md section 4.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, int zero, unsigned mask_a = 0xffffffffu, unsigned mask_b = 0xffffffffu) {
    uint4 a[2], b[2];
    a[0] = seed[threadIdx.x]; a[1] = seed[256 + threadIdx.x]; b[0] = seed[512 + threadIdx.x]; b[1] = seed[768 + threadIdx.x];
    if (zero) { a[0] = a[1] = b[0] = b[1] = make_uint4(0, 0, 0, 0); }
    for (int i = 0; i < 2; ++i) {
        a[i].x &= mask_a; a[i].y &= mask_a; a[i].z &= mask_a; a[i].w &= mask_a;
        b[i].x &= mask_b; b[i].y &= mask_b; b[i].z &= mask_b; b[i].w &= mask_b;
    }
    floatx16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[c & 1]), __builtin_bit_cast(f16x8, b[(c >> 1) & 1]), acc[c], 0, 0, 0);
                if (KIND == 1) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[c & 1]), __builtin_bit_cast(bf16x8, b[(c >> 1) & 1]), acc[c], 0, 0, 0);
                if (KIND == 2) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a[c & 1].x), __builtin_bit_cast(float, b[(c >> 1) & 1].x), acc[c], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int KIND, int CHAINS>
static void leg(const char* name, double flops_per_mfma, const uint4* seed, float* sink, int waves_per_simd, int zero, unsigned mask_a = 0xffffffffu, unsigned mask_b = 0xffffffffu) {
    const int blocks = 256 * waves_per_simd;               // 4 waves per block: one per SIMD
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mfma_loop<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, seed, sink, iters, zero, mask_a, mask_b);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_loop<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, seed, sink, iters, zero, mask_a, mask_b);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)reps * blocks * 4 * iters * 4 * CHAINS;
    const double tf = n_mfma * flops_per_mfma / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD if the pipe were never idle: 1024 SIMDs
    const double mfma_per_simd_per_s = n_mfma / 1024.0 / (ms * 1e-3);
    if (mask_a != 0xffffffffu || mask_b != 0xffffffffu) printf("masks A %08x B %08x  ", mask_a, mask_b);
    printf("%-34s waves/SIMD %d operands %-6s: %8.1f ms  %8.1f TFLOP/s  (%.1f M matrix instr/s per SIMD)\n", name, waves_per_simd,
           zero ? "zero" : "random", ms, tf, mfma_per_simd_per_s / 1e6);
    fflush(stdout);
}

int main() {
    std::vector<unsigned> h(4 * 1024);
    srand(1);
    for (auto& v : h) {            // random fp16 / bf16 bit patterns with moderate exponents (no inf / nan)
        unsigned lo = (rand() & 0x83ff) | (((rand() % 6) + 12) << 10), hi = (rand() & 0x83ff) | (((rand() % 6) + 12) << 10);
        v = lo | (hi << 16);
    }
    uint4* seed; float* sink;
    hipMalloc(&seed, h.size() * 4); hipMalloc(&sink, 4096);
    hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int zero = 0; zero < 2; ++zero) {
        for (int wps : {1, 2}) {
            leg<0, 4>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, seed, sink, wps, zero);
            leg<1, 4>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, seed, sink, wps, zero);
        }
        leg<2, 4>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, seed, sink, 2, zero);
    }
    // how the sustained rate depends on the operands' mantissa width (fp16: 10 mantissa bits per half): low bits cleared
    for (unsigned keep : {8u, 6u, 4u, 2u, 0u}) {
        const unsigned m16 = 0xffffu & ~((1u << (10 - keep)) - 1u), m = m16 | (m16 << 16);
        leg<0, 4>("f16, both operands, mantissa bits kept", 2.0 * 32 * 32 * 16, seed, sink, 2, 0, m, m);
    }
    for (unsigned keep : {6u, 4u, 2u}) {
        const unsigned m16 = 0xffffu & ~((1u << (10 - keep)) - 1u), m = m16 | (m16 << 16);
        leg<0, 4>("f16, A only, mantissa bits kept", 2.0 * 32 * 32 * 16, seed, sink, 2, 0, m, 0xffffffffu);
    }
    // sign-free / exponent-constant variants: what part of the power is the mantissa array
    leg<0, 4>("f16, mantissas zero in both (powers of two)", 2.0 * 32 * 32 * 16, seed, sink, 2, 0, 0xfc00fc00u, 0xfc00fc00u);
    leg<0, 4>("f16, A all zero, B random", 2.0 * 32 * 32 * 16, seed, sink, 2, 0, 0u, 0xffffffffu);
    return 0;
}
