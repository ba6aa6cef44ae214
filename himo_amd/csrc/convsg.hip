// convsg.hip -- the convolutions of the fp16-split network (3x3 stride 1, 3x3 stride 2, 1x1) whose INPUT is already split
// in HBM ("split activation format"), staged global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
// conversion work and no ds_write in the consumer.
//
// Split activation format: same addressing as float32 NHWC (pixel pitch `x_pitch` floats), but every 16-channel group
// of a pixel -- 64 bytes -- holds [h0..h15 | l0..l15] (fp16 high parts, fp16 low parts; x = h + l, bf16x3.h) instead of
// sixteen floats.  The producer writes it -- the epilogues here (OSPLIT) and in convsp.hip, the pillar feature kernel, the
// upsampling -- so a value is split ONCE instead of once per consuming block and halo overlap; the numbers that reach the
// matrix instructions are bit-identical to the float32-activation kernels' (convsp.hip / convbf.hip).
//
// LDS image of a slab's halo patch: (TH + 2) rows x 40 pixel slots (34 used) x 64 bytes; the four 16-byte pieces of a pixel
// (plane s, k-half lh) sit at slot (2 s + lh) ^ ((px >> 2) & 3), px = the pixel's column in its row.  A row is 2560 bytes = ten
// 256-byte bank rows, so the bank of a piece depends on its column alone: a fragment read is `ds_read_b128 v[kx][s] offset:row`
// with six per-lane offsets computed once, and every 16-lane group of a ds_read_b128 covers the 256-byte bank row exactly once
// ({0,12,20,24} and {4,8,16,28} give four different (px >> 2) & 3).  LDS-DMA writes lane-linearly (M0 base + lane * 16) in
// units of 16 pixels that run linearly over the patch (a unit may straddle two rows); the same XOR is applied to each lane's
// SOURCE address: a permutation inside the pixel's 64 bytes -- coalescing is unchanged.  Pixels outside the image (and the 6
// unused slots of a row) read a 64-byte zero page.
//
// Everything else as convsp.hip: a wave owns MI rows x 32 pixels x NT column tiles of 32 output channels (NT = 2: every
// activation fragment meets two weight fragments -- the 8-row tiles of the 64-channel layers and the 4-row x 64-channel
// tiles of the wide decoder layers, both autotune candidates), weight fragments straight from
// L2 two taps ahead in three statically rotated register sets, patch double-buffered, ONE barrier per 16-channel slab.
// The DMA of slab + 1 is issued before the first tap of slab; the in-order vmcnt of the weight loads behind it has
// retired it by tap 2, the explicit wait before the barrier only documents that.  Weight fragments are addressed as
// uniform base + 32-bit lane offset (saddr loads), and each tap's loads and fragment reads are pinned ahead of its matrix
// instructions (sched_group_barrier): DESIGN.md section 4 has the measurements and the variants that did not pay.
// Epilogue: an accumulator block's 32 pixels x 128 bytes are transposed through a wave-private 4 KB corner of the (then free)
// patch memory and leave as 16-byte stores, eight full 128-byte lines per instruction (store_block_vec, conv_common.h);
// blocks walk the grid in an XCD-aware order (xcd_block_id).
// Specification / oracle as conv.hip (reference network absent: PARITY UNPINNED).
#include "conv_common.h"
#include "bf16x3.h"

namespace himo {

__device__ __attribute__((aligned(64))) unsigned char g_zero_page[64];


__device__ inline void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// S = 2 (the stride-2 layers): a patch row holds the 65 input columns DE-INTERLEAVED -- even columns 0, 2, .. 64 in slots
// 0 .. 32 (padded to 48), odd columns 1, 3, .. 63 in slots 48 .. 79 -- so that the 32 output pixels of a fragment (input
// columns 2 li + kx) are again 32 consecutive slots for every tap (kx = 0: li, 1: 48 + li, 2: li + 1).  The gather costs
// nothing: every DMA lane has its own source address anyway.
template <int EPI, int PH, int MI, bool OSPLIT, int S, int NT = 1>
__global__ __launch_bounds__(256, NT == 2 ? (MI == 2 ? 3 : 2) : (S == 2 || MI == 4) ? 3 : 4)
void conv3_presplit_kernel(ConvArgs a, const unsigned short* __restrict__ wpk) {
    // S = 1: patch rows of 40 pixel slots (34 used; 2560 bytes = ten 256-byte bank rows, so the bank of a piece depends on its
    // column alone and the swizzle term can be taken from the column whatever the row), DMA units of 16 pixels run LINEARLY
    // over the patch and may straddle rows.  (48-slot rows until round 3: 61 KB for the 8-row tiles = two blocks per CU;
    // 51 KB = three.)  S = 2: 80-slot rows, five whole units per row.
    constexpr int TW = 32, TH = MI * PH, PHt = S == 1 ? TH + 2 : 2 * TH + 1, PWP = S == 1 ? 40 : 80, UPR = PWP / 16;
    constexpr int kRow = PWP * 64, kBuf = (PHt * kRow + 1023) / 1024 * 1024;
    constexpr int kUnits = S == 1 ? (PHt * PWP + 15) / 16 : PHt * UPR, kUPW = (kUnits + 3) / 4;    // DMA units of 16 pixels; units per wave
    constexpr int BN = (4 / PH) * 32 * NT;                       // NT column tiles of 32 channels per wave
    __shared__ __attribute__((aligned(1024))) unsigned char patch[2 * kBuf];

    const int n_tiles_n = (a.Cout + BN - 1) / BN;
    int bid = xcd_block_id(blockIdx.x, gridDim.x);
    const int tn = bid % n_tiles_n; bid /= n_tiles_n;
    const int tx = (a.Wo + TW - 1) / TW, ty = (a.Ho + TH - 1) / TH;
    const int ox0 = (bid % tx) * TW; bid /= tx;
    const int oy0 = (bid % ty) * TH;
    const int img = bid / ty;
    const unsigned char* __restrict__ xin = reinterpret_cast<const unsigned char*>(a.x + image_offset(img, a.n_inner, a.x_batch_stride, a.x_outer_stride));
    const int slabs = a.Cin >> 4;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wp = wave % PH, wc = wave / PH;
    const int li = lane & 31, lh = lane >> 5;
    const int co = tn * BN + wc * NT * 32 + li;                  // this lane's channel in column tile 0 (+ 32 per tile)
    const bool co_ok = co < a.Cout;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(&patch[0]);

    // this lane's DMA sources: unit u = wave + 4 k covers patch row u / UPR, slots 16 (u % UPR) .. + 15; a byte offset
    // into the image (an image is < 2 GB), or -1 for the zero page
    int src[kUPW];
#pragma unroll
    for (int k = 0; k < kUPW; ++k) {
        const int u = wave + 4 * k;
        const int q = u * 16 + (lane >> 2);                                  // S = 1: linear pixel slot of the patch
        const int row = S == 1 ? q / PWP : u / UPR, px = S == 1 ? q - row * PWP : (u % UPR) * 16 + (lane >> 2);   // slot within the patch row
        const int col = S == 1 ? px : px < 48 ? 2 * px : 2 * (px - 48) + 1;   // input column it holds
        const int iy = iy0 + row, ix = ix0 + col;
        const bool ok = u < kUnits && row < PHt && (S == 1 ? px < TW + 2 : px <= 32 || px >= 48) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const int sl = (lane & 3) ^ ((px >> 2) & 3);
        src[k] = ok ? (int)((((int64_t)iy * a.W + ix) * a.x_pitch) * 4 + sl * 16) : -1;
    }
    const unsigned char* zero = g_zero_page + (lane & 3) * 16;
    auto stage = [&](int slab, int buf) {
#pragma unroll
        for (int k = 0; k < kUPW; ++k) {
            const int u = wave + 4 * k;
            if (u < kUnits)
                glds16(src[k] >= 0 ? xin + (unsigned)(src[k] + slab * 64) : zero,
                       lds_base + buf * kBuf + (S == 1 ? u * 1024 : ((u / UPR) * PWP + (u % UPR) * 16) * 64));
        }
    };

    floatx16 acc[MI][NT];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nt][r] = 0.f;

    // weight fragments: uniform base + 32-bit offsets (the packed weights are a few MB): one scalar multiply-add for the
    // (tap, slab) block and one vector add per load
    const unsigned char* __restrict__ wb = reinterpret_cast<const unsigned char*>(wpk);
    unsigned b_lane[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = co + nt * 32 < a.Cout ? co + nt * 32 : a.Cout - 1;
        b_lane[nt] = (unsigned)c * 32u + (unsigned)lh * 16u;
    }
    const unsigned b_plane = (unsigned)a.Cout * 32u, b_block = 2u * b_plane;          // bytes per plane, per (tap, slab)
    auto load_b = [&](int tap, int slab, uint4 (&b)[NT][2]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned off = (unsigned)(tap * slabs + slab) * b_block + b_lane[nt];
            b[nt][0] = *reinterpret_cast<const uint4*>(wb + off);
            b[nt][1] = *reinterpret_cast<const uint4*>(wb + (off + b_plane));
        }
    };

    // fragment read offsets (buffer 0, kernel row 0, this wave's first image row): [kx][plane]
    int rd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int p = S == 1 ? li + kx : (kx & 1) * 48 + li + (kx >> 1), f = (p >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) rd[kx][s] = wp * MI * S * kRow + p * 64 + (((2 * s + lh) ^ f) << 4);
    }

    uint4 bq[3][NT][2];
    stage(0, 0);
    load_b(0, 0, bq[0]);
    load_b(1, 0, bq[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll 1
    for (int slab = 0; slab < slabs; ++slab) {
        const int buf = slab & 1;
        const bool more = slab + 1 < slabs;
        const int nslab = more ? slab + 1 : slab;
        if (more) stage(slab + 1, buf ^ 1);                  // that buffer was last read before the previous barrier
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const int rowoff = buf * kBuf + ky * kRow;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tap = ky * 3 + kx;
                const int t2 = tap + 2;
                load_b(t2 < 9 ? t2 : t2 - 9, t2 < 9 ? slab : nslab, bq[(kx + 2) % 3]);
                f16x8 af[MI][2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        af[mi][s] = *reinterpret_cast<const f16x8*>(&patch[rd[kx][s] + rowoff + mi * S * kRow]);
                const uint4 (&bcur)[NT][2] = bq[kx];
#define HIMO_TERM16(SA, SB)                                                                                        \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)              \
        acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][SA], __builtin_bit_cast(f16x8, bcur[nt][SB]), acc[mi][nt], 0, 0, 0);
                { HIMO_TERM16(1, 0) HIMO_TERM16(0, 1) HIMO_TERM16(0, 0) }
#undef HIMO_TERM16
                // this tap's weight prefetch and ALL its activation-fragment reads before its matrix instructions (the
                // compiler otherwise feeds each MFMA pair from a just-issued ds_read and sinks the prefetch next to its use)
                __builtin_amdgcn_sched_group_barrier(0x020, 2 * NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, MI * 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MI * NT * 3, 0);
            }
        }
        if (more) {
            if (NT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // at most the next slab's first two weight fragments stay in flight
            __syncthreads();
        }
    }

    float* __restrict__ yout = a.y + image_offset(img, a.n_inner, a.y_batch_stride, a.y_outer_stride);
    float eA[NT], eB[NT];                            // value = fma(acc, A, B): packing scale, bias and folded BatchNorm (conv_common.h)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = co + nt * 32 < a.Cout ? co + nt * 32 : a.Cout - 1;
        epi_affine<EPI>(kF16AccScale, a.bias ? a.bias[c] : 0.f, EPI == kEpiBiasBnGelu ? a.scale[c] : 1.f, EPI == kEpiBiasBnGelu ? a.shift[c] : 0.f,
                        eA[nt], eB[nt]);
    }
    if (NT > 1 || (a.act_flags & kActVecStore)) {   // 16-byte stores through a wave-private LDS transpose (store_block_vec)
        __syncthreads();                         // every wave has read its last patch rows: the patch memory is free
        unsigned char* stg = patch + wave * 4096;
        const int n_px = a.Wo - ox0 < 32 ? a.Wo - ox0 : 32;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int oy = oy0 + wp * MI + mi;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                unsigned word[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = epi_activate<EPI>(acc[mi][nt][r], eA[nt], eB[nt]);
                    word[r] = OSPLIT ? split_word(v, li & 1) : __builtin_bit_cast(unsigned, v);
                    if (OSPLIT && mi == 0 && nt == 0 && r == 0)          // (a lane's sample counts only where it is a pixel and a channel of the image)
                        note_range(a, (oy < a.Ho && 4 * lh < n_px && tn * BN + wc * NT * 32 + li < a.Cout) ? v : 0.f);
                }
                store_block_vec<OSPLIT>(a, yout, stg, word, lane, (int64_t)oy * a.Wo + ox0, oy < a.Ho ? n_px : 0, tn * BN + (wc * NT + nt) * 32);
            }
        }
        return;
    }
    if (!co_ok) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int oy = oy0 + wp * MI + mi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (oy < a.Ho && ox < a.Wo) {                       // (scalar-store fallback: outputs that do not admit 16-byte stores)
                const float v = epi_activate<EPI>(acc[mi][0][r], eA[0], eB[0]);
                if (OSPLIT) split_store<kEpiBias, true>(a, yout, (int64_t)oy * a.Wo + ox, co, v, 1.f, 0.f);
                else yout[((int64_t)oy * a.Wo + ox) * a.y_pitch + co] = v;
            }
        }
    }
}

// 1x1 layers (row GEMMs over the pixels of an image) on the same structure: a block owns TH = MI * PH segments of 32
// consecutive pixels, two 16-channel slabs are staged per barrier step, weight fragments run one slab ahead in two
// register sets.  Same summation order as convbf.hip's row GEMM (slab by slab, terms l*h, h*l, h*h): identical bits.
template <int EPI, int PH, int MI, bool OSPLIT>
__global__ __launch_bounds__(256, 4)
void conv1_presplit_kernel(ConvArgs a, const unsigned short* __restrict__ wpk) {
    constexpr int G = 2, TH = MI * PH, BM = TH * 32;
    constexpr int kSeg = 32 * 64, kSlab = TH * kSeg, kBuf = G * kSlab;
    constexpr int kUnits = G * TH * 2, kUPW = (kUnits + 3) / 4;
    constexpr int BN = (4 / PH) * 32;
    __shared__ __attribute__((aligned(1024))) unsigned char patch[2 * kBuf < 16384 ? 16384 : 2 * kBuf];   // >= the epilogue's 4 x 4 KB

    const int n_tiles_n = (a.Cout + BN - 1) / BN;
    int bid = xcd_block_id(blockIdx.x, gridDim.x);
    const int tn = bid % n_tiles_n; bid /= n_tiles_n;
    const int64_t rows = (int64_t)a.Ho * a.Wo;
    const int tiles = (int)((rows + BM - 1) / BM);
    const int img = bid / tiles;
    const int64_t row0 = (int64_t)(bid % tiles) * BM;
    const unsigned char* __restrict__ xin = reinterpret_cast<const unsigned char*>(a.x + image_offset(img, a.n_inner, a.x_batch_stride, a.x_outer_stride));
    const int slabs = a.Cin >> 4, steps = (slabs + G - 1) / G;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wp = wave % PH, wc = wave / PH;
    const int li = lane & 31, lh = lane >> 5;
    const int co = tn * BN + wc * 32 + li;
    const bool co_ok = co < a.Cout;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(&patch[0]);

    // DMA unit u = wave + 4 k: slab g = u / (2 TH) of the step, segment (u / 2) % TH, pixels 16 (u % 2) .. + 15
    int src[kUPW];
#pragma unroll
    for (int k = 0; k < kUPW; ++k) {
        const int u = wave + 4 * k;
        const int g = u / (2 * TH), seg = (u >> 1) % TH, px = (u & 1) * 16 + (lane >> 2);
        const int64_t pix = row0 + seg * 32 + px;
        const int sl = (lane & 3) ^ ((px >> 2) & 3);
        src[k] = (u < kUnits && pix < rows) ? (int)(pix * a.x_pitch * 4 + sl * 16 + g * 64) : -1;
    }
    const unsigned char* zero = g_zero_page + (lane & 3) * 16;
    auto stage = [&](int step, int buf) {
#pragma unroll
        for (int k = 0; k < kUPW; ++k) {
            const int u = wave + 4 * k;
            if (u < kUnits) {
                const int g = u / (2 * TH);
                const bool ok = src[k] >= 0 && step * G + g < slabs;
                glds16(ok ? xin + (unsigned)(src[k] + step * (G * 64)) : zero, lds_base + buf * kBuf + g * kSlab + ((u >> 1) % TH) * kSeg + (u & 1) * 1024);
            }
        }
    };

    floatx16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

    const int co_ld = co_ok ? co : a.Cout - 1;
    const unsigned char* __restrict__ wb = reinterpret_cast<const unsigned char*>(wpk);
    const unsigned b_lane = (unsigned)co_ld * 32u + (unsigned)lh * 16u, b_plane = (unsigned)a.Cout * 32u;
    auto load_b = [&](int slab, uint4 (&b)[2]) {
        const unsigned off = (unsigned)slab * (2u * b_plane) + b_lane;
        b[0] = *reinterpret_cast<const uint4*>(wb + off);
        b[1] = *reinterpret_cast<const uint4*>(wb + (off + b_plane));
    };
    int rd[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) rd[s] = wp * MI * kSeg + li * 64 + (((2 * s + lh) ^ ((li >> 2) & 3)) << 4);

    uint4 bq[2][2];
    stage(0, 0);
    load_b(0, bq[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll 1
    for (int step = 0; step < steps; ++step) {
        const int buf = step & 1;
        const bool more = step + 1 < steps;
        if (more) stage(step + 1, buf ^ 1);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int nslab = step * G + g + 1;
            load_b(nslab < slabs ? nslab : slabs - 1, bq[(g + 1) & 1]);         // clamped: a slab past the end meets zero activations
            f16x8 af[MI][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi][s] = *reinterpret_cast<const f16x8*>(&patch[rd[s] + buf * kBuf + g * kSlab + mi * kSeg]);
            const uint4 (&bcur)[2] = bq[g & 1];
#define HIMO_TERM16(SA, SB)                                                                                        \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                \
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][SA], __builtin_bit_cast(f16x8, bcur[SB]), acc[mi], 0, 0, 0);
            HIMO_TERM16(1, 0) HIMO_TERM16(0, 1) HIMO_TERM16(0, 0)
#undef HIMO_TERM16
        }
        if (more) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // only the next slab's weight fragments stay in flight
            __syncthreads();
        }
    }

    float* __restrict__ yout = a.y + image_offset(img, a.n_inner, a.y_batch_stride, a.y_outer_stride);
    float eA, eB;
    epi_affine<EPI>(kF16AccScale, a.bias ? a.bias[co_ld] : 0.f, EPI == kEpiBiasBnGelu ? a.scale[co_ld] : 1.f, EPI == kEpiBiasBnGelu ? a.shift[co_ld] : 0.f, eA, eB);
    if (a.act_flags & kActVecStore) {            // 16-byte stores through a wave-private LDS transpose (store_block_vec)
        __syncthreads();
        unsigned char* stg = patch + wave * 4096;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t pix0 = row0 + (wp * MI + mi) * 32;
            const int64_t left = rows - pix0;
            unsigned word[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = epi_activate<EPI>(acc[mi][r], eA, eB);
                word[r] = OSPLIT ? split_word(v, li & 1) : __builtin_bit_cast(unsigned, v);
                if (OSPLIT && mi == 0 && r == 0) note_range(a, (4 * lh < left && tn * BN + wc * 32 + li < a.Cout) ? v : 0.f);
            }
            store_block_vec<OSPLIT>(a, yout, stg, word, lane, pix0, left < 0 ? 0 : left > 32 ? 32 : (int)left, tn * BN + wc * 32);
        }
        return;
    }
    if (!co_ok) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t pix = row0 + (wp * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (pix < rows) {
                const float v = epi_activate<EPI>(acc[mi][r], eA, eB);
                if (OSPLIT) split_store<kEpiBias, true>(a, yout, pix, co, v, 1.f, 0.f);
                else yout[pix * a.y_pitch + co] = v;
            }
        }
    }
}

template <int PH, int MI>
static void launch_sg1(const ConvArgs& a_in, int epi, bool osplit, const unsigned short* w, dim3 grid, hipStream_t s) {
    ConvArgs a = a_in;
    if (vec_store_ok(a)) a.act_flags |= kActVecStore;
#define HIMO_SG(E, O) hipLaunchKernelGGL((conv1_presplit_kernel<E, PH, MI, O>), grid, dim3(256), 0, s, a, w)
    if (epi == kEpiBias) { if (osplit) HIMO_SG(kEpiBias, true); else HIMO_SG(kEpiBias, false); }
    else { if (osplit) HIMO_SG(kEpiBiasBnGelu, true); else HIMO_SG(kEpiBiasBnGelu, false); }
#undef HIMO_SG
}

// 1x1 layers whose input is in the split activation format.  rows_hint: 32-pixel segments per wave (4 | 2 | 1).
bool launch_conv1_presplit(const ConvArgs& a, int epilogue, const void* w_packed, int rows_hint, bool out_split, hipStream_t s) {
    if ((epilogue != kEpiBias && epilogue != kEpiBiasBnGelu) || (a.Cin & 15) || (out_split && (a.Cout & 15))) return false;
    if ((int64_t)a.Ho * a.Wo * a.x_pitch * 4 >= (int64_t)1 << 31) return false;       // 32-bit DMA source offsets
    const bool wide = a.Cout > 64;
    const int bn = wide ? 128 : 64, ph = wide ? 1 : 2;
    auto blocks_for = [&](int mi) -> int64_t {
        return (int64_t)a.N * (((int64_t)a.Ho * a.Wo + mi * ph * 32 - 1) / (mi * ph * 32)) * ((a.Cout + bn - 1) / bn);
    };
    int mi = wide ? 4 : 2;
    while (mi > 1 && blocks_for(mi) < 2048) mi >>= 1;
    if (rows_hint == 4 || rows_hint == 2 || rows_hint == 1) mi = rows_hint;
    if (!wide && mi == 4) mi = 2;
    const dim3 grid((unsigned)blocks_for(mi));
    const unsigned short* w = (const unsigned short*)w_packed;
    ProfScope ps("conv1x1_f16x2_kernel", s);
    if (wide) {
        if (mi == 4) launch_sg1<1, 4>(a, epilogue, out_split, w, grid, s);
        else if (mi == 2) launch_sg1<1, 2>(a, epilogue, out_split, w, grid, s);
        else launch_sg1<1, 1>(a, epilogue, out_split, w, grid, s);
    } else {
        if (mi == 2) launch_sg1<2, 2>(a, epilogue, out_split, w, grid, s);
        else launch_sg1<2, 1>(a, epilogue, out_split, w, grid, s);
    }
    return true;
}

template <int PH, int MI, int S = 1, int NT = 1>
static void launch_sg(const ConvArgs& a_in, int epi, bool osplit, const unsigned short* w, dim3 grid, hipStream_t s) {
    ConvArgs a = a_in;
    if (vec_store_ok(a)) a.act_flags |= kActVecStore;
#define HIMO_SG(E, O) hipLaunchKernelGGL((conv3_presplit_kernel<E, PH, MI, O, S, NT>), grid, dim3(256), 0, s, a, w)
    if (epi == kEpiBias) { if (osplit) HIMO_SG(kEpiBias, true); else HIMO_SG(kEpiBias, false); }
    else { if (osplit) HIMO_SG(kEpiBiasBnGelu, true); else HIMO_SG(kEpiBiasBnGelu, false); }
#undef HIMO_SG
}

// 3x3 layers (stride 1 | 2) whose input is in the split activation format (fp16-split weights, Cin a multiple of 16,
// bias or bias + BN + GELU epilogue).  rows_hint: image rows per wave (4 | 2 | 1), 0 = heuristic.  false = not applicable.
bool launch_conv3_presplit(const ConvArgs& a, int epilogue, const void* w_packed, int rows_hint, bool out_split, int stride, hipStream_t s) {
    if ((epilogue != kEpiBias && epilogue != kEpiBiasBnGelu) || (a.Cin & 15) || (out_split && (a.Cout & 15))) return false;
    if ((int64_t)a.H * a.W * a.x_pitch * 4 >= (int64_t)1 << 31) return false;       // 32-bit DMA source offsets
    const bool wide = a.Cout > 64;
    const int bn = wide ? 128 : 64, ph = wide ? 1 : 2;
    if (stride == 2) {                                  // two output rows per block: a 5-row x 80-slot patch, double-buffered
        const int64_t blocks = (int64_t)a.N * ((a.Ho + 1) / 2) * ((a.Wo + 31) / 32) * ((a.Cout + bn - 1) / bn);
        ProfScope ps("conv3x3s2_f16x2_kernel", s);
        if (wide) launch_sg<1, 2, 2>(a, epilogue, out_split, (const unsigned short*)w_packed, dim3((unsigned)blocks), s);
        else launch_sg<2, 1, 2>(a, epilogue, out_split, (const unsigned short*)w_packed, dim3((unsigned)blocks), s);
        return true;
    }
    // 64-channel layers, rows_hint 8: a wave owns 2 rows x 32 pixels x BOTH 32-channel column tiles (every activation fragment
    // meets two weight fragments), four waves = an 8-row tile on a 10-row patch (1.25x halo instead of 1.5x; 51 KB: three blocks per CU)
    if (!wide && rows_hint == 8 && vec_store_ok(a)) {
        const int64_t blocks = (int64_t)a.N * ((a.Ho + 7) / 8) * ((a.Wo + 31) / 32) * ((a.Cout + 63) / 64);
        ProfScope ps("conv3x3_f16x2_kernel", s);
        launch_sg<4, 2, 1, 2>(a, epilogue, out_split, (const unsigned short*)w_packed, dim3((unsigned)blocks), s);
        return true;
    }
    // wide layers, rows_hint 12: 4 rows x 64 channels per wave (128 accumulator registers, two waves per SIMD): half the
    // fragment reads AND half the weight loads per matrix instruction; 3-4 % on the 256-channel decoder layers, slower on
    // the 128-channel encoder ones (the autotune decides per layer)
    if (wide && rows_hint == 12 && vec_store_ok(a)) {
        ProfScope ps("conv3x3_f16x2_kernel", s);
        if (a.Cout % 256 == 0) {
            const int64_t blocks = (int64_t)a.N * ((a.Ho + 3) / 4) * ((a.Wo + 31) / 32) * (a.Cout / 256);
            launch_sg<1, 4, 1, 2>(a, epilogue, out_split, (const unsigned short*)w_packed, dim3((unsigned)blocks), s);
        } else {
            const int64_t blocks = (int64_t)a.N * ((a.Ho + 7) / 8) * ((a.Wo + 31) / 32) * ((a.Cout + 127) / 128);
            launch_sg<2, 4, 1, 2>(a, epilogue, out_split, (const unsigned short*)w_packed, dim3((unsigned)blocks), s);
        }
        return true;
    }
    auto blocks_for = [&](int mi) -> int64_t {
        const int th = mi * ph;
        return (int64_t)a.N * ((a.Ho + th - 1) / th) * ((a.Wo + 31) / 32) * ((a.Cout + bn - 1) / bn);
    };
    int mi = blocks_for(4) >= 1024 ? 4 : 2;
    if (rows_hint == 4 || rows_hint == 2 || rows_hint == 1) mi = rows_hint;
    if (!wide && mi == 4) mi = 2;                       // 64-channel blocks: 8-row patches would not leave three blocks per CU
    const dim3 grid((unsigned)blocks_for(mi));
    const unsigned short* w = (const unsigned short*)w_packed;
    ProfScope ps("conv3x3_f16x2_kernel", s);
    if (wide) {
        if (mi == 4) launch_sg<1, 4>(a, epilogue, out_split, w, grid, s);
        else if (mi == 2) launch_sg<1, 2>(a, epilogue, out_split, w, grid, s);
        else launch_sg<1, 1>(a, epilogue, out_split, w, grid, s);
    } else {
        if (mi == 2) launch_sg<2, 2>(a, epilogue, out_split, w, grid, s);
        else launch_sg<2, 1>(a, epilogue, out_split, w, grid, s);
    }
    return true;
}

}  // namespace himo
