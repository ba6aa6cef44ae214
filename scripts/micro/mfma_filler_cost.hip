// mfma_filler_cost.hip -- what ONE extra instruction costs when it sits between back-to-back v_mfma_f32_32x32x16_f16 of the same
// wave (one wave per SIMD, 4 independent accumulators, operands in registers): cycles per matrix instruction with R fillers of one
// kind after each, minus the 32 cycles of the matrix instruction alone, per filler.  Inline asm fixes the instruction and its
// placement; every filler is independent of the matrix instructions and of the other fillers (8 rotating destinations).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfc scripts/micro/mfma_filler_cost.hip && /tmp/mfc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

enum { K_NONE, K_VFMA, K_VMOV, K_VCVT, K_VEXP, K_SALU, K_DSREAD128, K_DSWRITE32, K_GLOAD128, K_ACCREAD, K_VPKFMA, K_VADD_I, K_DSWRITE16 };

template <int KIND>
__device__ inline void filler(float (&v)[8], int q, unsigned& sreg, unsigned lds_addr, const uint4* gptr, floatx4 (&ld)[2], floatx16& accx) {
    if (KIND == K_VFMA) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(v[q]) : "v"(v[(q + 1) & 7]));
    if (KIND == K_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[q]) : "v"(v[(q + 1) & 7]));
    if (KIND == K_VCVT) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(v[q]) : "v"(v[(q + 1) & 7]));
    if (KIND == K_VEXP) asm volatile("v_exp_f32 %0, %1" : "=v"(v[q]) : "v"(v[(q + 1) & 7]));
    if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
    if (KIND == K_DSREAD128) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[q & 1]) : "v"(lds_addr));
    if (KIND == K_DSWRITE32) asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr), "v"(v[q]));
    if (KIND == K_DSWRITE16) asm volatile("ds_write_b16 %0, %1" :: "v"(lds_addr), "v"(v[q]));
    if (KIND == K_GLOAD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[q & 1]) : "v"(gptr));
    if (KIND == K_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[q]) : "a"(accx[q]));
    if (KIND == K_VPKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&v[q & 6])) : "v"(*reinterpret_cast<double*>(&v[(q + 2) & 6])));
    if (KIND == K_VADD_I) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[q]) : "v"(v[(q + 1) & 7]));
}

template <int KIND, int R>
__global__ __launch_bounds__(256) void filler_kernel(const uint4* __restrict__ seed, float* __restrict__ sink, int iters, long long* cycles) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (float)i;
    __syncthreads();
    const f16x8 a = __builtin_bit_cast(f16x8, seed[threadIdx.x & 255]), b = __builtin_bit_cast(f16x8, seed[256 + (threadIdx.x & 255)]);
    floatx16 acc[4], accx;
    float v[8];
    floatx4 ld[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned sreg = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accx[r] = (float)r; for (int c = 0; c < 4; ++c) acc[c][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(&lds[0]) + (threadIdx.x & 63) * 16;
    const uint4* gptr = seed + (threadIdx.x & 255);
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < R; ++j) filler<KIND>(v, (i * R + j) & 7, sreg, lds_addr, gptr, ld, accx);
        }
        if (KIND == K_DSREAD128 || KIND == K_DSWRITE32 || KIND == K_DSWRITE16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == K_GLOAD128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const long long t1 = clock64();
    float s = (float)sreg + ld[0][0] + ld[1][1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND, int R>
static double run(const uint4* seed, float* sink, long long* d_cycles) {
    const int iters = 2000;
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((filler_kernel<KIND, R>), dim3(256), dim3(256), 0, 0, seed, sink, iters, d_cycles);
    (void)hipDeviceSynchronize();
    long long cyc; (void)hipMemcpy(&cyc, d_cycles, sizeof(cyc), hipMemcpyDeviceToHost);
    return (double)cyc / iters / 16;
}

template <int KIND>
static void kind(const char* name, const uint4* seed, float* sink, long long* cyc) {
    const double c1 = run<KIND, 1>(seed, sink, cyc), c2 = run<KIND, 2>(seed, sink, cyc), c4 = run<KIND, 4>(seed, sink, cyc), c8 = run<KIND, 8>(seed, sink, cyc);
    printf("  %-22s cycles per matrix instruction with 1 / 2 / 4 / 8 fillers: %6.1f %6.1f %6.1f %6.1f   -> per filler %5.2f %5.2f %5.2f %5.2f\n", name, c1, c2, c4, c8,
           c1 - 32, (c2 - 32) / 2, (c4 - 32) / 4, (c8 - 32) / 8);
}

int main() {
    std::vector<uint4> h(512);
    unsigned x = 12345u;
    for (auto& q : h) { unsigned w[4]; for (int i = 0; i < 4; ++i) { x = x * 1664525u + 1013904223u; w[i] = (x & 0x3fff3fffu) | 0x30003000u; } q = make_uint4(w[0], w[1], w[2], w[3]); }
    uint4* seed; float* sink; long long* cyc;
    (void)hipMalloc(&seed, h.size() * sizeof(uint4)); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 64);
    (void)hipMemcpy(seed, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    printf("matrix instructions alone: %.1f cycles each (one wave per SIMD, 4 accumulators)\n", run<K_NONE, 0>(seed, sink, cyc));
    kind<K_SALU>("s_add_u32", seed, sink, cyc);
    kind<K_VMOV>("v_mov_b32", seed, sink, cyc);
    kind<K_VFMA>("v_fma_f32", seed, sink, cyc);
    kind<K_VADD_I>("v_add_u32", seed, sink, cyc);
    kind<K_VPKFMA>("v_pk_fma_f32", seed, sink, cyc);
    kind<K_VCVT>("v_cvt_f16_f32", seed, sink, cyc);
    kind<K_VEXP>("v_exp_f32", seed, sink, cyc);
    kind<K_ACCREAD>("v_accvgpr_read_b32", seed, sink, cyc);
    kind<K_DSREAD128>("ds_read_b128", seed, sink, cyc);
    kind<K_DSWRITE32>("ds_write_b32", seed, sink, cyc);
    kind<K_DSWRITE16>("ds_write_b16", seed, sink, cyc);
    kind<K_GLOAD128>("global_load_dwordx4", seed, sink, cyc);
    return 0;
}
