// winograd_core.hip -- VERDICT r02 item 4: "settle Winograd with a measurement".  ONLY the transform-domain core of F(2x2, 3x3) for
// enc3's shape (256 -> 256 channels, 64 x 64 pixels, 48 images = 49,152 tiles of 2 x 2 outputs) in the fp16-split arithmetic
// (3 matrix instructions per product block): for every transform point xi (16) a GEMM [tiles x Cin] . [Cin x Cout], with the
// transformed activations V read from LDS and the transformed weights U streamed from L2.  NO input transform, NO output
// transform, NO HBM traffic for V, no stores of results worth mentioning: everything a real kernel would add is left out, so the
// time measured here is a LOWER bound for a Winograd layer.  The direct kernel (csrc/convsg.hip) runs this layer in 558-585 us.
//
// Structure (the only one that fits the register file -- DESIGN.md section 4): the 16 transform-domain accumulators of a
// 32-tile x 32-channel wave tile are 256 registers (AGPRs), so a wave is alone on its SIMD; a block = 4 waves = 4 output-channel
// tiles of the same 32 tiles; per 16-channel slab V is 16 xi x 32 tiles x 64 B = 32 KB of LDS (two buffers are declared, as a
// real kernel would need, so one block per CU), read as ds_read_b128 fragments (conflict-free layout not even attempted: the
// reads are 2 per 3 matrix instructions), and U is 2 global loads of 16 B per lane per (xi, slab).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wino scripts/micro/winograd_core.hip && /tmp/wino
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int XI = 16, CIN = 256, COUT = 256, SLABS = CIN / 16, TILES_PER_BLOCK = 32;

// U layout: [xi][slab][plane][cout][16 halves] (one 32-byte row per output channel and plane: a fragment load is lane (li, lh) ->
// cout li, halves 8 lh .. 8 lh + 7, as in the direct kernels' packed weights)
__global__ __launch_bounds__(256, 1) void winograd_core_kernel(const unsigned short* __restrict__ U, const uint4* __restrict__ vseed,
                                                               float* __restrict__ out, int n_tile_blocks) {
    __shared__ __attribute__((aligned(16))) unsigned char V[2][XI * TILES_PER_BLOCK * 64];      // [buffer][xi][tile][plane h | plane l][16 halves]
    for (int i = threadIdx.x; i < 2 * XI * TILES_PER_BLOCK * 4; i += 256) reinterpret_cast<uint4*>(&V[0][0])[i] = vseed[i & 4095];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int cout0 = (blockIdx.x % (COUT / 128)) * 128 + wave * 32;
    floatx16 acc[XI];
#pragma unroll
    for (int x = 0; x < XI; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    const unsigned char* ub = reinterpret_cast<const unsigned char*>(U);
    const unsigned lane_off = (unsigned)(cout0 + li) * 32u + (unsigned)lh * 16u;
    const unsigned plane = COUT * 32u, block = 2u * plane;                                   // bytes per plane, per (xi, slab)
#pragma unroll 1
    for (int slab = 0; slab < SLABS; ++slab) {
        const unsigned char* vb = &V[slab & 1][0] + li * 64 + lh * 16;
#pragma unroll
        for (int x = 0; x < XI; ++x) {
            const unsigned off = (unsigned)(x * SLABS + slab) * block + lane_off;
            const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ub + off));
            const f16x8 bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ub + off + plane));
            const f16x8 ah = *reinterpret_cast<const f16x8*>(vb + x * TILES_PER_BLOCK * 64);
            const f16x8 al = *reinterpret_cast<const f16x8*>(vb + x * TILES_PER_BLOCK * 64 + 32);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[x], 0, 0, 0);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[x], 0, 0, 0);
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[x], 0, 0, 0);
        }
        __syncthreads();                                     // where a real kernel hands the next slab's V over
    }
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < XI; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[x][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;         // 4 bytes per lane: keeps the work alive, nothing like a real epilogue
}

int main() {
    const size_t u_halves = (size_t)XI * SLABS * 2 * COUT * 16;
    std::vector<unsigned short> hu(u_halves);
    std::vector<uint4> hv(4096);
    unsigned x = 777u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x; };
    for (auto& h : hu) h = (unsigned short)(((rnd() >> 8) & 0x3fffu) | 0x2000u | ((rnd() & 1u) << 15));     // random fp16 of magnitude 2^-7 .. 1
    for (auto& q : hv) { unsigned w[4]; for (int i = 0; i < 4; ++i) w[i] = (rnd() & 0xbfffbfffu) | 0x20002000u; q = make_uint4(w[0], w[1], w[2], w[3]); }
    unsigned short* U; uint4* vseed; float* out;
    const int tiles = 48 * 32 * 32, tile_blocks = tiles / TILES_PER_BLOCK, blocks = tile_blocks * (COUT / 128);
    (void)hipMalloc(&U, u_halves * 2); (void)hipMalloc(&vseed, hv.size() * sizeof(uint4)); (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMemcpy(U, hu.data(), u_halves * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(vseed, hv.data(), hv.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(winograd_core_kernel, dim3(blocks), dim3(256), 0, 0, U, vseed, out, tile_blocks);
    (void)hipDeviceSynchronize();
    const int reps = 20;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(winograd_core_kernel, dim3(blocks), dim3(256), 0, 0, U, vseed, out, tile_blocks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3;
    const double n_mfma = (double)blocks * 4 * SLABS * XI * 3;
    printf("Winograd F(2x2,3x3) transform-domain core, enc3 shape (48 x 64 x 64 px, 256 -> 256), fp16 split, %d blocks of %d tiles x 128 channels:\n", blocks, TILES_PER_BLOCK);
    printf("  %.1f us per layer   (%.0f issued fp16 TFLOP/s = %.1f %% of 2500; U = %.1f MB streamed per block-pass from L2)\n", us,
           n_mfma * 32768.0 / (us * 1e-6) / 1e12, n_mfma * 32768.0 / (us * 1e-6) / 1e12 / 25.0, u_halves * 2 / 1e6);
    printf("  direct 3x3 kernel on the same layer: 558-585 us (profiles/r02_conv3x3_per_layer.txt, r03_exp_conv_epilogue_and_patch_rows.txt)\n");
    printf("  go / no-go bound of VERDICT r02 item 4: the core alone must be <= 330 us (1.75x) before transforms are worth building\n");
    return 0;
}
