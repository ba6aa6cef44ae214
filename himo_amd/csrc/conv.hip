// conv.hip -- stage a10, backbone + head arithmetic: NHWC float32 convolution / row-GEMM on the
// gfx950 matrix cores with fused epilogues, and bilinear x2 upsampling.
//
// No reference source exists for this stage (OpenSceneFlow submodule absent).
// Specification: himo_amd/seflow/spec.py steps 3-5; oracle: oracle/seflow_oracle.py (PyTorch CPU fp32).
//
// Why float32 MFMA: north_star asks for per-point flow within 1e-4 of a float32 CPU path.  bf16
// inputs lose that after ~20 conv layers, so the matrix cores run v_mfma_f32_32x32x2_f32 -- exact
// float32 products and accumulation (a k-ordered fma chain) at the float32 vector peak (157 TF) but
// issued from 64-cycle matrix instructions that leave the VALU free for staging and epilogues.
//
// Kernel: implicit GEMM.  M = output pixels (or point rows), N = output channels, K = taps x Cin.
//   * block = 256 threads = 4 waves (2 along M x 2 along N); block tile 128 pixels x BN channels;
//     for 3x3 convs the 128 pixels are an 8 x 16 spatial tile so that the input halo patch
//     ((8-1)s+3) x ((16-1)s+3) pixels is loaded ONCE per 16-channel slab and reused by all 9 taps;
//   * LDS holds the patch k-major ([16][patch pixels], so an A fragment is one conflict-free
//     ds_read_b32 per lane) and the [16][BN] weight slab of the current tap;
//   * the fragment loop issues 2 A reads + BN/64 B reads per 2*BN/64 MFMAs -- the LDS is ~12% busy;
//   * epilogues are fused: bias, BatchNorm(eval) scale/shift, exact-erf GELU, the GRU gate math.
// Channel concatenation never copies: every tensor is addressed as base + n*batch_stride +
// pixel*pitch + channel, so frames live as channel groups of one wider buffer.
#include "conv_common.h"

namespace himo {

// MI = 32-row MFMA tiles per wave along M: 2 -> 128-pixel block tile (8 x 16), 1 -> 64-pixel tile (4 x 16) for the
// low-resolution layers whose 128-pixel tiling would leave most of the 256 CUs idle
template <int KS, int S, int BN, int EPI, int MI>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    constexpr int BM = 64 * MI, BK = 16, TH = 4 * MI, TW = 16;
    constexpr int PH = KS == 1 ? 1 : (TH - 1) * S + KS;
    constexpr int PW = KS == 1 ? BM : (TW - 1) * S + KS;
    constexpr int PP = PH * PW + 1;                 // +1: break the power-of-two plane stride
    constexpr int WN = BN / 2, NI = WN / 32;
    constexpr int T = KS * KS;
    constexpr bool kPatchDouble = S == 1;           // stride-2 patches (561 px) stay single-buffered
    constexpr int kPatchBufs = kPatchDouble ? 2 : 1;
    constexpr int kPatchItems = PH * PW * (BK / 4);
    constexpr int kPatchPerThread = (kPatchItems + 255) / 256;
    constexpr int kWPerThread = BK * (BN / 4) / 256;
    // one raw block (patch planes, then weight slabs): the row-GEMM epilogue reuses its first 16 KB as the four waves' store
    // staging areas (store_block_vec)
    constexpr int kPatchFloats = kPatchBufs * BK * PP, kWFloats = 2 * BK * BN;
    static_assert(kPatchFloats % 4 == 0, "weight slabs stay 16-byte aligned");
    __shared__ __attribute__((aligned(16))) float smem[kPatchFloats + kWFloats < 4096 ? 4096 : kPatchFloats + kWFloats];
    auto& patchT = *reinterpret_cast<float (*)[kPatchBufs][BK * PP]>(smem);
    auto& wt = *reinterpret_cast<float (*)[2][BK * BN]>(smem + kPatchFloats);

    // ---- block -> (image, spatial tile, channel tile) ----
    const int n_tiles_n = (a.Cout + BN - 1) / BN;
    int bid = xcd_block_id(blockIdx.x, gridDim.x);
    const int tn = bid % n_tiles_n; bid /= n_tiles_n;
    int oy0, ox0, img;
    int64_t row0 = 0;                               // KS == 1: first linear pixel of the tile
    if (KS == 1) {
        const int64_t rows = (int64_t)a.Ho * a.Wo;
        const int tiles = (int)((rows + BM - 1) / BM);
        img = bid / tiles;
        row0 = (int64_t)(bid % tiles) * BM;
        oy0 = ox0 = 0;
    } else {
        const int tx = (a.Wo + TW - 1) / TW, ty = (a.Ho + TH - 1) / TH;
        ox0 = (bid % tx) * TW; bid /= tx;
        oy0 = (bid % ty) * TH; img = bid / ty;
    }
    const int n0 = tn * BN;
    const float* __restrict__ xin = a.x + image_offset(img, a.n_inner, a.x_batch_stride, a.x_outer_stride);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;

    // A-fragment pixel offsets inside the patch for this lane's two 32-row MFMA tiles
    int ppA[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = wm * (32 * MI) + mi * 32 + li;
        ppA[mi] = KS == 1 ? m : ((m / TW) * S) * PW + (m % TW) * S;
    }

    floatx16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int iy0 = oy0 * S - (KS / 2), ix0 = ox0 * S - (KS / 2);
    const int64_t in_rows = (int64_t)a.H * a.W;

    // global -> registers (issued early, consumed after the MFMAs of the current step)
    auto load_patch = [&](int ci0, float4 (&r)[kPatchDouble ? kPatchPerThread : 1]) {
        if constexpr (!kPatchDouble) return;
#pragma unroll
        for (int it = 0; it < kPatchPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < kPatchItems) {
                const int pp = item / (BK / 4), q = item % (BK / 4);
                int64_t pix;
                bool ok;
                if (KS == 1) { pix = row0 + pp; ok = pix < in_rows; }
                else {
                    const int iy = iy0 + pp / PW, ix = ix0 + pp % PW;
                    ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    pix = (int64_t)iy * a.W + ix;
                }
                const int ci = ci0 + q * 4;
                if (ok && ci < a.Cin) v = *reinterpret_cast<const float4*>(xin + pix * a.x_pitch + ci);
            }
            r[it] = v;
        }
    };
    // registers -> LDS, transposed to k-major so that an A fragment is one conflict-free ds_read_b32
    auto store_patch = [&](int buf, const float4 (&r)[kPatchDouble ? kPatchPerThread : 1]) {
        if constexpr (!kPatchDouble) return;
#pragma unroll
        for (int it = 0; it < kPatchPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            if (item < kPatchItems) {
                const int pp = item / (BK / 4), q = item % (BK / 4);
                float* dst = &patchT[buf][(q * 4) * PP + pp];
                dst[0] = r[it].x; dst[PP] = r[it].y; dst[2 * PP] = r[it].z; dst[3 * PP] = r[it].w;
            }
        }
    };
    auto load_w = [&](int tap, int ci0, float4 (&r)[kWPerThread]) {
#pragma unroll
        for (int it = 0; it < kWPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            const int k = item / (BN / 4), q = item % (BN / 4);
            const int ci = ci0 + k, co = n0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < a.Cin && co < a.Cout)
                v = *reinterpret_cast<const float4*>(a.w + ((int64_t)tap * a.Cin + ci) * a.Cout + co);
            r[it] = v;
        }
    };
    auto store_w = [&](int buf, const float4 (&r)[kWPerThread]) {
#pragma unroll
        for (int it = 0; it < kWPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            *reinterpret_cast<float4*>(&wt[buf][(item / (BN / 4)) * BN + (item % (BN / 4)) * 4]) = r[it];
        }
    };

    // direct (unpipelined) patch staging, one float4 in flight per thread: used for the prologue-free
    // stride-2 variant whose 561-pixel patch would need 36 prefetch registers per lane
    auto stage_patch_direct = [&](int ci0) {
        for (int item = threadIdx.x; item < kPatchItems; item += 256) {
            const int pp = item / (BK / 4), q = item % (BK / 4);
            const int iy = iy0 + pp / PW, ix = ix0 + pp % PW;
            const bool ok = KS == 1 ? (row0 + pp < in_rows) : (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W);
            const int64_t pix = KS == 1 ? row0 + pp : (int64_t)iy * a.W + ix;
            const int ci = ci0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && ci < a.Cin) v = *reinterpret_cast<const float4*>(xin + pix * a.x_pitch + ci);
            float* dst = &patchT[0][(q * 4) * PP + pp];
            dst[0] = v.x; dst[PP] = v.y; dst[2 * PP] = v.z; dst[3 * PP] = v.w;
        }
    };

    float4 pr[kPatchDouble ? kPatchPerThread : 1], wr[kWPerThread];
    load_w(0, 0, wr);
    if constexpr (kPatchDouble) { load_patch(0, pr); store_patch(0, pr); }
    else stage_patch_direct(0);
    store_w(0, wr);
    __syncthreads();

    int pbuf = 0, wbuf = 0;
#pragma unroll 1
    for (int ci0 = 0; ci0 < a.Cin; ci0 += BK) {
#pragma unroll 1
        for (int tap = 0; tap < T; ++tap) {
            const bool last_tap = tap == T - 1;
            const bool has_next = !(last_tap && ci0 + BK >= a.Cin);
            const int ntap = last_tap ? 0 : tap + 1, nci0 = last_tap ? ci0 + BK : ci0;
            if (has_next) {                                  // software pipeline: next slab's loads fly under the MFMAs
                load_w(ntap, nci0, wr);
                if (last_tap && kPatchDouble) load_patch(nci0, pr);
            }
            const int tapoff = KS == 1 ? 0 : (tap / KS) * PW + (tap % KS);
            const float* __restrict__ pa = patchT[kPatchDouble ? pbuf : 0];
            const float* __restrict__ pw = wt[wbuf];
#pragma unroll
            for (int t = 0; t < BK / 2; ++t) {
                const int k = 2 * t + lh;
                float af[MI], bf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi] = pa[k * PP + ppA[mi] + tapoff];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bf[ni] = pw[k * BN + wn * WN + ni * 32 + li];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
            }
            if (has_next) {
                store_w(wbuf ^ 1, wr);
                if (last_tap && kPatchDouble) store_patch(pbuf ^ 1, pr);
            }
            __syncthreads();
            wbuf ^= 1;
            if (last_tap && kPatchDouble) pbuf ^= 1;
            if (last_tap && !kPatchDouble && has_next) {     // single-buffered patch: refill after everyone is done with it
                stage_patch_direct(nci0);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31] ----
    float* __restrict__ yout = a.y + image_offset(img, a.n_inner, a.y_batch_stride, a.y_outer_stride);
    const int64_t out_rows = (int64_t)a.Ho * a.Wo;
    if constexpr (KS == 1 && EPI != kEpiGruZR && EPI != kEpiGruQ) {
        // row GEMMs (the FastNSF MLP, 1x1 layers): an accumulator block's 32 rows are consecutive output rows, so it leaves as
        // 16-byte stores through the wave's LDS staging area -- and the ReLU mask of the input-gradient GEMMs is read with the
        // same 16-byte pattern instead of one 4-byte load per element (FastNSF fit: its row GEMMs 130 -> 94 ms per 100 iterations)
        if (a.act_flags & kActVecStore) {
            __syncthreads();                       // every wave is done with the patch / weight slabs
            unsigned char* stg = reinterpret_cast<unsigned char*>(smem) + wave * 4096;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ch0 = n0 + wn * WN + ni * 32;
                const int co = ch0 + li < a.Cout ? ch0 + li : a.Cout - 1;
                const float b = a.bias ? a.bias[co] : 0.f;
                float sc = 1.f, sh = 0.f;
                if (EPI == kEpiBiasBnGelu) { sc = a.scale[co]; sh = a.shift[co]; }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    unsigned word[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) word[r] = __builtin_bit_cast(unsigned, epilogue_value<EPI>(acc[mi][ni][r] + b, sc, sh));
                    const int64_t pix0 = row0 + wm * (32 * MI) + mi * 32;
                    const int64_t left = out_rows - pix0;
                    store_block_vec<false, EPI == kEpiReluMask>(a, yout, stg, word, lane, pix0, left < 0 ? 0 : left > 32 ? 32 : (int)left, ch0);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = n0 + wn * WN + ni * 32 + li;
        if (co >= a.Cout) continue;
        const float b = a.bias ? a.bias[co] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (EPI == kEpiBiasBnGelu) { sc = a.scale[co]; sh = a.shift[co]; }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int64_t pix;
                bool ok;
                if (KS == 1) { pix = row0 + m; ok = pix < out_rows; }
                else {
                    const int oy = oy0 + m / TW, ox = ox0 + m % TW;
                    ok = oy < a.Ho && ox < a.Wo;
                    pix = (int64_t)oy * a.Wo + ox;
                }
                if (!ok) continue;
                epilogue_store<EPI>(a, yout, pix, co, acc[mi][ni][r] + b, sc, sh);
            }
        }
    }
}

// bilinear x2, align_corners=True (torch's area_pixel_compute_source_index / lambda formulation)
struct UpArgs {
    const float* x; int x_pitch; int H, W, C;
    float* y; int y_pitch;
    int64_t x_bs, y_bs;   // image blockIdx.y of a batch
    float ry, rx;   // (H-1)/(2H-1), (W-1)/(2W-1)
    int out_split;  // y in the split activation format (convsg.hip)
};

// SPLIT = false: float32 output, 4 channels per thread.  SPLIT = true: output in the split activation format, 8 channels per
// thread -- one 16-byte store of high parts and one of low parts into the pixel's [16 high | 16 low] record.
template <bool SPLIT>
__global__ __launch_bounds__(256) void upsample2x_kernel(UpArgs a) {
    constexpr int CPT = SPLIT ? 8 : 4;
    a.x += (int64_t)blockIdx.y * a.x_bs;
    a.y += (int64_t)blockIdx.y * a.y_bs;
    const int cq = a.C / CPT;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)(2 * a.H) * (2 * a.W) * cq;
    if (item >= total) return;
    const int q = (int)(item % cq);
    const int64_t pix = item / cq;
    const int ox = (int)(pix % (2 * a.W)), oy = (int)(pix / (2 * a.W));
    const float sy = a.ry * (float)oy, sx = a.rx * (float)ox;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    float o[CPT];
#pragma unroll
    for (int part = 0; part < CPT / 4; ++part) {
        auto ld = [&](int yy, int xx) { return *reinterpret_cast<const float4*>(a.x + ((int64_t)yy * a.W + xx) * a.x_pitch + q * CPT + part * 4); };
        const float4 v00 = ld(y0, x0), v01 = ld(y0, x1), v10 = ld(y1, x0), v11 = ld(y1, x1);
        o[part * 4 + 0] = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        o[part * 4 + 1] = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        o[part * 4 + 2] = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        o[part * 4 + 3] = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    }
    if constexpr (SPLIT) {
        unsigned h[8], l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) split2_rounded(o[k], h[k], l[k]);
        unsigned char* rec = reinterpret_cast<unsigned char*>(a.y + pix * a.y_pitch + (q >> 1) * 16) + (q & 1) * 16;
        *reinterpret_cast<uint4*>(rec) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        *reinterpret_cast<uint4*>(rec + 32) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    } else {
        *reinterpret_cast<float4*>(a.y + pix * a.y_pitch + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Split output, round 3: the same 8 channels per thread and the same arithmetic (identical bits) as upsample2x_kernel<true>, with the
// index arithmetic taken out of the lanes: a block works inside ONE output row (blockIdx.y; the row's source rows and weights are
// scalars) and a thread's pixel / channel group come from one 32-bit division instead of three 64-bit divisions and remainders.
// Measured (scripts/exp_upsample.py): the 8-channel kernel ran at 2.4-2.6 TB/s of output where its stores alone sustain 5.8 -- neither
// the bytes nor the load pattern (a 4-channel variant with fully contiguous loads and DPP-paired stores was 30 % SLOWER: twice the
// threads, twice the index arithmetic) but the integer work per lane.
__global__ __launch_bounds__(256) void upsample2x_split_row_kernel(UpArgs a) {
    a.x += (int64_t)blockIdx.z * a.x_bs;
    a.y += (int64_t)blockIdx.z * a.y_bs;
    const unsigned cq = (unsigned)a.C >> 3;
    const unsigned item = blockIdx.x * 256u + threadIdx.x;           // within the row: pixel * cq + group
    const unsigned ox = item / cq, q = item - ox * cq;
    if (ox >= 2u * (unsigned)a.W) return;
    const int oy = (int)blockIdx.y;
    const float sy = a.ry * (float)oy, sx = a.rx * (float)ox;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* row0 = a.x + (int64_t)y0 * a.W * a.x_pitch + q * 8;
    const float* row1 = a.x + (int64_t)y1 * a.W * a.x_pitch + q * 8;
    const unsigned c0 = (unsigned)x0 * (unsigned)a.x_pitch, c1 = (unsigned)x1 * (unsigned)a.x_pitch;
    float o[8];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const float4 v00 = *reinterpret_cast<const float4*>(row0 + c0 + part * 4), v01 = *reinterpret_cast<const float4*>(row0 + c1 + part * 4);
        const float4 v10 = *reinterpret_cast<const float4*>(row1 + c0 + part * 4), v11 = *reinterpret_cast<const float4*>(row1 + c1 + part * 4);
        o[part * 4 + 0] = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        o[part * 4 + 1] = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        o[part * 4 + 2] = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        o[part * 4 + 3] = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    }
    unsigned h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split2_rounded(o[k], h[k], l[k]);
    unsigned char* rec = reinterpret_cast<unsigned char*>(a.y + ((int64_t)oy * (2 * a.W) + ox) * a.y_pitch + (q >> 1) * 16) + (q & 1) * 16;
    *reinterpret_cast<uint4*>(rec) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    *reinterpret_cast<uint4*>(rec + 32) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// Split output, the decoder's channel counts (C = 8 CQ, CQ = 8 | 16 | 32): the source pixels of a tile through LDS.  Every output
// value needs four source values; read from global memory that is four 16-byte load instructions per 4 channels, and the kernels above
// are bound by exactly that count (a wave's dwordx4 load occupies the CU's vector-memory path for ~40-60 cycles: 8 of them per 2 KB of
// output = 2.4-2.6 TB/s, while the same stores without the loads sustain 5.8).  Here a block owns kUpRows output rows x 256 / CQ output
// columns: the (kUpRows / 2 + 2) x (PX / 2 + 2) source pixels they can touch are loaded ONCE, coalesced, into LDS (1.1 global loads per
// 8 output channels instead of 8) and the four corners come from ds_read_b128.  Same arithmetic per value: identical bits.
constexpr int kUpRows = 8;
template <int CQ>
__global__ __launch_bounds__(256) void upsample2x_split_lds_kernel(UpArgs a) {
    constexpr int PX = 256 / CQ, NC = PX / 2 + 2, NR = kUpRows / 2 + 2, C = CQ * 8, C4 = C / 4;
    constexpr int CP = C + 4;        // pixel pitch in LDS: + 16 bytes, so the two source pixels of a 16-lane read phase interleave in the banks
    __shared__ __attribute__((aligned(16))) float tile[NR * NC * CP];
    a.x += (int64_t)blockIdx.z * a.x_bs;
    a.y += (int64_t)blockIdx.z * a.y_bs;
    const int ox0 = (int)blockIdx.x * PX, oy0 = (int)blockIdx.y * kUpRows;
    const int cy0 = (int)(a.ry * (float)oy0), cx0 = (int)(a.rx * (float)ox0);       // first source row / column of the tile
    for (int e = threadIdx.x; e < NR * NC * C4; e += 256) {
        const int px = e / C4, j = e - px * C4;
        const int r = px / NC, c = px - r * NC;
        const int yy = min(cy0 + r, a.H - 1), xx = min(cx0 + c, a.W - 1);          // (clamped duplicates past the border are never selected)
        *reinterpret_cast<float4*>(&tile[px * CP + j * 4]) = *reinterpret_cast<const float4*>(a.x + ((int64_t)yy * a.W + xx) * a.x_pitch + j * 4);
    }
    __syncthreads();
    const int q = threadIdx.x % CQ, p = threadIdx.x / CQ;
    const int ox = ox0 + p;
    if (ox >= 2 * a.W) return;
    const float sx = a.rx * (float)ox;
    const int x0 = (int)sx;
    const int x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
    const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
    const int t0 = (x0 - cx0) * CP + q * 8, t1 = (x1 - cx0) * CP + q * 8;
#pragma unroll
    for (int r = 0; r < kUpRows; ++r) {
        const int oy = oy0 + r;
        if (oy >= 2 * a.H) break;
        const float sy = a.ry * (float)oy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        const float* r0 = &tile[(y0 - cy0) * NC * CP], * r1 = &tile[(y1 - cy0) * NC * CP];
        float o[8];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float4 v00 = *reinterpret_cast<const float4*>(r0 + t0 + part * 4), v01 = *reinterpret_cast<const float4*>(r0 + t1 + part * 4);
            const float4 v10 = *reinterpret_cast<const float4*>(r1 + t0 + part * 4), v11 = *reinterpret_cast<const float4*>(r1 + t1 + part * 4);
            o[part * 4 + 0] = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
            o[part * 4 + 1] = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
            o[part * 4 + 2] = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
            o[part * 4 + 3] = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        }
        unsigned h[8], l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) split2_rounded(o[k], h[k], l[k]);
        unsigned char* rec = reinterpret_cast<unsigned char*>(a.y + ((int64_t)oy * (2 * a.W) + ox) * a.y_pitch + (q >> 1) * 16) + (q & 1) * 16;
        *reinterpret_cast<uint4*>(rec) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        *reinterpret_cast<uint4*>(rec + 32) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    }
}

template <int KS, int S, int BN, int MI>
static void launch_epi(const ConvArgs& a_in, int epi, dim3 grid, hipStream_t s) {
    ConvArgs a = a_in;
    // row GEMMs: 16-byte epilogue stores when the output (and, for the ReLU-mask epilogue, the mask source) admits them
    if (KS == 1 && vec_store_ok(a) && !(a.Cout & 31) &&
        (epi != kEpiReluMask || (!(reinterpret_cast<uintptr_t>(a.aux_in) & 15u) && !(a.aux_in_pitch & 3))))
        a.act_flags |= kActVecStore;
    switch (epi) {
        case kEpiBias: hipLaunchKernelGGL((conv_mfma_kernel<KS, S, BN, kEpiBias, MI>), grid, dim3(256), 0, s, a); break;
        case kEpiBiasBnGelu: hipLaunchKernelGGL((conv_mfma_kernel<KS, S, BN, kEpiBiasBnGelu, MI>), grid, dim3(256), 0, s, a); break;
        case kEpiBiasGelu: hipLaunchKernelGGL((conv_mfma_kernel<KS, S, BN, kEpiBiasGelu, MI>), grid, dim3(256), 0, s, a); break;
        case kEpiGruZR: hipLaunchKernelGGL((conv_mfma_kernel<KS, S, BN, kEpiGruZR, MI>), grid, dim3(256), 0, s, a); break;
        case kEpiBiasRelu: hipLaunchKernelGGL((conv_mfma_kernel<KS, S, BN, kEpiBiasRelu, MI>), grid, dim3(256), 0, s, a); break;
        case kEpiReluMask: hipLaunchKernelGGL((conv_mfma_kernel<KS, S, BN, kEpiReluMask, MI>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((conv_mfma_kernel<KS, S, BN, kEpiGruQ, MI>), grid, dim3(256), 0, s, a); break;
    }
}

template <int KS, int S>
static void launch_tile(const ConvArgs& a, int epi, int bn, int mi, dim3 grid, hipStream_t s) {
    if (bn == 128) { if (mi == 2) launch_epi<KS, S, 128, 2>(a, epi, grid, s); else launch_epi<KS, S, 128, 1>(a, epi, grid, s); }
    else { if (mi == 2) launch_epi<KS, S, 64, 2>(a, epi, grid, s); else launch_epi<KS, S, 64, 1>(a, epi, grid, s); }
}

}  // namespace himo

using namespace himo;

extern "C" int himo_conv2d(const himo_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->y) return HIMO_ERR_INVALID_ARGUMENT;
    if (d->n < 1 || d->h < 1 || d->w_in < 1 || d->cin < 1 || d->cout < 1) return HIMO_ERR_INVALID_ARGUMENT;
    if (!(d->ksize == 1 || d->ksize == 3) || !(d->stride == 1 || d->stride == 2)) return HIMO_ERR_UNSUPPORTED;
    if (d->ksize == 1 && d->stride != 1) return HIMO_ERR_UNSUPPORTED;
    if (d->epilogue < 0 || d->epilogue > kEpiReluMask) return HIMO_ERR_INVALID_ARGUMENT;
    if (d->epilogue == kEpiBiasBnGelu && (!d->scale || !d->shift)) return HIMO_ERR_INVALID_ARGUMENT;
    if ((d->epilogue == kEpiGruZR || d->epilogue == kEpiGruQ) && (!d->aux_in || !d->aux_out)) return HIMO_ERR_INVALID_ARGUMENT;
    if (d->epilogue == kEpiReluMask && !d->aux_in) return HIMO_ERR_INVALID_ARGUMENT;
    // 16-byte vector loads: channel counts / pitches / bases must be multiples of 4 floats
    if ((d->cin & 3) || (d->cout & 3) || (d->x_pitch & 3) || (d->x_batch_stride & 3) || !aligned16(d->x) || !aligned16(d->w))
        return HIMO_ERR_UNSUPPORTED;
    ConvArgs a{};
    a.x = d->x; a.x_batch_stride = d->x_batch_stride; a.x_pitch = d->x_pitch;
    a.w = d->w; a.bias = d->bias; a.scale = d->scale; a.shift = d->shift;
    a.y = d->y; a.y_batch_stride = d->y_batch_stride; a.y_pitch = d->y_pitch;
    const int n_outer = d->n_outer > 1 ? d->n_outer : 1;
    if (n_outer > 1 && ((d->x_outer_stride & 3) || (d->y_outer_stride & 3))) return HIMO_ERR_INVALID_ARGUMENT;
    a.N = d->n * n_outer; a.n_inner = d->n; a.x_outer_stride = d->x_outer_stride; a.y_outer_stride = d->y_outer_stride;
    a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Cout = d->cout;
    a.Ho = d->stride == 2 ? (d->h + 1) / 2 : d->h;      // 3x3, pad 1: ceil(H / stride)
    a.Wo = d->stride == 2 ? (d->w_in + 1) / 2 : d->w_in;
    a.aux_in = d->aux_in; a.aux_in_pitch = d->aux_in_pitch; a.aux_out = d->aux_out; a.aux_out_pitch = d->aux_out_pitch;
    a.act_flags = d->act_layout;
    a.range_seen = (d->act_layout & 2) ? d->d_range_seen : nullptr;
    if (d->act_layout & (8 | 16)) {      // HIMO_ACT_ACCUMULATE (y += result) / HIMO_ACT_STUFFED_2X (compact input read zero-stuffed):
        // the two-term bf16 3x3 stride-1 kernel with the bias epilogue only
        // ... or, HIMO_ACT_ACCUMULATE alone, a row GEMM (ksize 1) of either bf16 split with the bias epilogue (csrc/convbf.hip)
        const bool gemm_acc = d->act_layout == 8 && d->w_packed && d->ksize == 1 && d->epilogue == kEpiBias && (d->packed_format == 0 || d->packed_format == 2);
        if (!gemm_acc &&
            ((d->act_layout & ~(8 | 16)) || !d->w_packed || d->packed_format != 2 || d->ksize != 3 || d->stride != 1 || d->epilogue != kEpiBias))
            return HIMO_ERR_UNSUPPORTED;
        if ((d->act_layout & 16) && ((d->h & 1) || (d->w_in & 1) || (int64_t)(d->h / 2) * (d->w_in / 2) * d->x_pitch * 4 >= ((int64_t)1 << 31)))
            return HIMO_ERR_UNSUPPORTED;
    } else if (d->act_layout) {   // split activation format: fp16-split 3x3 layers only, whole 16-channel groups
        if ((d->act_layout & ~3) || !d->w_packed || d->packed_format != 1) return HIMO_ERR_UNSUPPORTED;
        if (d->ksize == 1 && !(d->act_layout & 1)) return HIMO_ERR_UNSUPPORTED;      // 1x1: split output only with split input
        if (d->epilogue != kEpiBias && d->epilogue != kEpiBiasBnGelu) return HIMO_ERR_UNSUPPORTED;
        if (((d->act_layout & 1) && ((d->cin & 15) || (d->x_pitch & 15))) ||
            ((d->act_layout & 2) && ((d->cout & 15) || (d->y_pitch & 15))))
            return HIMO_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    // split precision: every stride-1 layer, and the 3x3 stride-2 layers unless the caller pins a float32 tile
    if (d->w_packed && (d->stride == 1 || (d->ksize == 3 && d->epilogue != kEpiGruZR && d->epilogue != kEpiGruQ &&
                                          (d->tile_hint == 0 || (d->tile_hint & 0x1000)))))
        return launch_conv_bf16x3(a, d->ksize, d->epilogue, d->w_packed, d->tile_hint, d->packed_format, d->stride, s);
    // tile choice: 128 x 128 when that still gives >= 2 blocks per CU, else shrink M then N so the chip is filled
    auto blocks_for = [&](int bn, int mi) -> int64_t {
        const int bm = 64 * mi, th = 4 * mi;
        const int64_t tm = d->ksize == 1 ? (int64_t)a.N * (((int64_t)a.Ho * a.Wo + bm - 1) / bm)
                                         : (int64_t)a.N * ((a.Ho + th - 1) / th) * ((a.Wo + 15) / 16);
        return tm * ((d->cout + bn - 1) / bn);
    };
    const bool can128 = d->cout >= 128 && (d->cout % 128) == 0;
    const bool gru = d->epilogue == kEpiGruZR;          // the z|r split needs the 128-wide channel tile
    int bn = can128 ? 128 : 64, mi = 2;
    const int64_t want = 512;
    if (blocks_for(bn, mi) < want) mi = 1;
    if (blocks_for(bn, mi) < want && bn == 128 && !gru) bn = 64;
    if (d->tile_hint) {
        const int hb = d->tile_hint >> 4, hm = d->tile_hint & 15;
        if ((hb == 64 || (hb == 128 && can128)) && (hm == 1 || hm == 2)) { bn = hb; mi = hm; }
    }
    const dim3 grid((unsigned)blocks_for(bn, mi));
    const char* name = d->ksize == 1 ? "conv1x1_mfma_kernel" : (d->stride == 2 ? "conv3x3s2_mfma_kernel" : "conv3x3_mfma_kernel");
    {
        ProfScope ps(name, s);
        if (d->ksize == 1) launch_tile<1, 1>(a, d->epilogue, bn, mi, grid, s);
        else if (d->stride == 1) launch_tile<3, 1>(a, d->epilogue, bn, mi, grid, s);
        else launch_tile<3, 2>(a, d->epilogue, bn, mi, grid, s);
    }
    HIMO_LAUNCH_CHECK("conv_mfma_kernel");
    return HIMO_OK;
}

extern "C" int himo_upsample2x(const float* d_x, int x_pitch, int h, int w, int c, float* d_y, int y_pitch, void* stream) {
    return himo_upsample2x_batch(1, d_x, 0, x_pitch, h, w, c, d_y, 0, y_pitch, stream);
}

extern "C" int himo_upsample2x_batch(int n, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int c, float* d_y,
                                     int64_t y_batch_stride, int y_pitch, void* stream) {
    return himo_upsample2x_batch_ex(n, d_x, x_batch_stride, x_pitch, h, w, c, d_y, y_batch_stride, y_pitch, 0, stream);
}

extern "C" int himo_upsample2x_batch_ex(int n, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int c, float* d_y,
                                        int64_t y_batch_stride, int y_pitch, int out_split, void* stream) {
    if (out_split && ((c & 15) || (y_pitch & 15) || (y_batch_stride & 15) || (reinterpret_cast<uintptr_t>(d_y) & 63))) return HIMO_ERR_INVALID_ARGUMENT;
    if (n < 1 || !d_x || !d_y || h < 1 || w < 1 || c < 4 || (c & 3) || (x_pitch & 3) || (y_pitch & 3) || (x_batch_stride & 3) ||
        (y_batch_stride & 3))
        return HIMO_ERR_INVALID_ARGUMENT;
    UpArgs a{};
    a.x = d_x; a.x_pitch = x_pitch; a.H = h; a.W = w; a.C = c; a.y = d_y; a.y_pitch = y_pitch;
    a.x_bs = x_batch_stride; a.y_bs = y_batch_stride; a.out_split = out_split ? 1 : 0;
    a.ry = h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    a.rx = w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    ProfScope ps("upsample2x_kernel", (hipStream_t)stream);
    if (out_split && (c == 64 || c == 128 || c == 256) && (2 * h + kUpRows - 1) / kUpRows <= 65535 && n <= 65535) {
        const int cq = c / 8, px = 256 / cq;
        const dim3 grid((2 * w + px - 1) / px, (2 * h + kUpRows - 1) / kUpRows, n);
        if (cq == 8) hipLaunchKernelGGL(upsample2x_split_lds_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else if (cq == 16) hipLaunchKernelGGL(upsample2x_split_lds_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(upsample2x_split_lds_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, a);
        HIMO_LAUNCH_CHECK("upsample2x_split_lds_kernel");
        return HIMO_OK;
    }
    if (out_split && 2 * h <= 65535 && n <= 65535 && (int64_t)w * x_pitch < ((int64_t)1 << 31)) {
        const unsigned row_items = (unsigned)(2 * w) * (unsigned)(c / 8);
        hipLaunchKernelGGL(upsample2x_split_row_kernel, dim3((row_items + 255) / 256, 2 * h, n), dim3(256), 0, (hipStream_t)stream, a);
        HIMO_LAUNCH_CHECK("upsample2x_split_row_kernel");
        return HIMO_OK;
    }
    const int64_t total = (int64_t)(2 * h) * (2 * w) * (c / (out_split ? 8 : 4));
    const dim3 grid((unsigned)((total + 255) / 256), n);
    if (out_split) hipLaunchKernelGGL(upsample2x_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(upsample2x_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    HIMO_LAUNCH_CHECK("upsample2x_kernel");
    return HIMO_OK;
}
