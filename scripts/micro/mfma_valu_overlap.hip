// mfma_valu_overlap.hip -- do the matrix instructions of one wave and the vector instructions of ANOTHER wave on the same SIMD
// overlap?  (The fused head and the 64-channel convolutions behave as if they did not: their time is the SUM of the matrix time and
// of everything else, at three to four waves per SIMD.)  One 512-thread block per CU = two waves per SIMD (waves w and w + 4 share
// a SIMD); every leg runs the same instruction counts per wave-role, only WHO runs WHAT WHEN changes:
//   M      every wave: matrix instructions only            V      every wave: vector instructions only
//   M|V    waves 0-3 matrix only, waves 4-7 vector only    (perfect overlap: max(M, V) at half the waves each)
//   MV     every wave alternates [P matrix][Q vector] phases, all waves in phase (barrier per phase pair)
//   MV~    the same, waves 4-7 start with the vector phase (anti-phase, no barriers)
//   MVi    every wave: the two streams interleaved instruction by instruction by the compiler (one basic block)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo scripts/micro/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int P = 32;      // matrix instructions per phase (4 independent accumulators): 1024 cycles of matrix pipe
constexpr int Q = 256;     // vector instructions per phase (8 independent fma chains)

__device__ inline void mphase(floatx16 (&acc)[4], const f16x8& a, const f16x8& b) {
#pragma unroll
    for (int i = 0; i < P; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
}
__device__ inline void vphase(float (&v)[8], float k) {
#pragma unroll
    for (int i = 0; i < Q; ++i) v[i & 7] = fmaf(v[i & 7], k, 0.5f);
}

template <int MODE>
__global__ __launch_bounds__(512) void overlap_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, long long* cycles) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f16x8 a = __builtin_bit_cast(f16x8, seed[threadIdx.x & 255]), b = __builtin_bit_cast(f16x8, seed[256 + (threadIdx.x & 255)]);
    floatx16 acc[4];
    float v[8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float k = 0.999f + (float)(threadIdx.x & 3) * 1e-4f;
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { mphase(acc, a, b); }
        if (MODE == 1) { vphase(v, k); }
        if (MODE == 2) { if (wave < 4) mphase(acc, a, b); else vphase(v, k); }
        if (MODE == 3) { mphase(acc, a, b); vphase(v, k); __syncthreads(); }
        if (MODE == 4) { if (wave < 4) { mphase(acc, a, b); vphase(v, k); } else { vphase(v, k); mphase(acc, a, b); } }
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < Q / P; ++j) v[(i * (Q / P) + j) & 7] = fmaf(v[(i * (Q / P) + j) & 7], k, 0.5f);
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cycles[wave] = t1 - t0;
}

// one wave per SIMD (256-thread block, one block per CU): R vector instructions after every matrix instruction, in one basic block;
// KIND 0: v_fma_f32 (8 chains), 1: v_exp_f32 (transcendental), 2: ds_read_b32 (LDS), 3: v_cvt_f16_f32 + v_pack style split arithmetic
template <int R, int KIND>
__global__ __launch_bounds__(256) void ratio_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, long long* cycles) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = (float)threadIdx.x; lds[256 + threadIdx.x] = 1.f; lds[512 + threadIdx.x] = 2.f; lds[768 + threadIdx.x] = 3.f;
    __syncthreads();
    const f16x8 a = __builtin_bit_cast(f16x8, seed[threadIdx.x & 255]), b = __builtin_bit_cast(f16x8, seed[256 + (threadIdx.x & 255)]);
    floatx16 acc[4];
    float v[8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float k = 0.999f + (float)(threadIdx.x & 3) * 1e-4f;
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int q = (i * R + j) & 7;
                if (KIND == 0) v[q] = fmaf(v[q], k, 0.5f);
                if (KIND == 1) v[q] = __builtin_amdgcn_exp2f(v[q]);
                if (KIND == 2) v[q] += lds[(threadIdx.x + (int)__builtin_bit_cast(unsigned, v[(q + 1) & 7])) & 1023];
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int R, int KIND>
static void ratio_leg(const char* kind, const uint4* seed, float* sink, long long* d_cycles) {
    const int iters = 4000;
    hipLaunchKernelGGL((ratio_kernel<R, KIND>), dim3(256), dim3(256), 0, 0, seed, sink, iters, d_cycles);
    hipLaunchKernelGGL((ratio_kernel<R, KIND>), dim3(256), dim3(256), 0, 0, seed, sink, iters, d_cycles);
    hipDeviceSynchronize();
    long long cyc; hipMemcpy(&cyc, d_cycles, sizeof(cyc), hipMemcpyDeviceToHost);
    printf("  %-10s %2d per matrix instruction: %6.1f cycles per matrix instruction (+ its %d fillers)\n", kind, R, (double)cyc / iters / 16, R);
}

template <int MODE>
static void leg(const char* name, const uint4* seed, float* sink, long long* d_cycles, int blocks_per_cu) {
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((overlap_kernel<MODE>), dim3(blocks), dim3(512), 0, 0, seed, sink, iters, d_cycles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((overlap_kernel<MODE>), dim3(blocks), dim3(512), 0, 0, seed, sink, iters, d_cycles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc[8]; hipMemcpy(cyc, d_cycles, sizeof(cyc), hipMemcpyDeviceToHost);
    printf("%-6s %d block(s)/CU  %8.3f ms per launch   cycles per iteration: wave0 %7.0f  wave4 %7.0f\n", name, blocks_per_cu, ms / 5,
           (double)cyc[0] / iters, (double)cyc[4] / iters);
}

int main() {
    std::vector<uint4> h(512);
    unsigned x = 12345u;
    for (auto& q : h) { unsigned w[4]; for (int i = 0; i < 4; ++i) { x = x * 1664525u + 1013904223u; w[i] = (x & 0x3fff3fffu) | 0x30003000u; } q = make_uint4(w[0], w[1], w[2], w[3]); }
    uint4* seed; float* sink; long long* cyc;
    hipMalloc(&seed, h.size() * sizeof(uint4)); hipMalloc(&sink, 4096); hipMalloc(&cyc, 64);
    hipMemcpy(seed, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    printf("per wave and iteration: P = %d matrix instructions (%d cycles of matrix pipe), Q = %d vector instructions; two waves per SIMD per block\n", P, P * 32, Q);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        leg<0>("M", seed, sink, cyc, bpc);
        leg<1>("V", seed, sink, cyc, bpc);
        leg<2>("M|V", seed, sink, cyc, bpc);
        leg<3>("MV", seed, sink, cyc, bpc);
        leg<4>("MV~", seed, sink, cyc, bpc);
        leg<5>("MVi", seed, sink, cyc, bpc);
    }
    printf("one wave per SIMD, R independent instructions issued after every v_mfma_f32_32x32x16_f16 (32 cycles alone):\n");
    ratio_leg<0, 0>("v_fma", seed, sink, cyc); ratio_leg<1, 0>("v_fma", seed, sink, cyc); ratio_leg<2, 0>("v_fma", seed, sink, cyc);
    ratio_leg<4, 0>("v_fma", seed, sink, cyc); ratio_leg<6, 0>("v_fma", seed, sink, cyc); ratio_leg<8, 0>("v_fma", seed, sink, cyc);
    ratio_leg<12, 0>("v_fma", seed, sink, cyc); ratio_leg<16, 0>("v_fma", seed, sink, cyc);
    ratio_leg<1, 1>("v_exp", seed, sink, cyc); ratio_leg<2, 1>("v_exp", seed, sink, cyc); ratio_leg<4, 1>("v_exp", seed, sink, cyc); ratio_leg<8, 1>("v_exp", seed, sink, cyc);
    ratio_leg<1, 2>("ds_read+add", seed, sink, cyc); ratio_leg<2, 2>("ds_read+add", seed, sink, cyc); ratio_leg<4, 2>("ds_read+add", seed, sink, cyc);
    return 0;
}
