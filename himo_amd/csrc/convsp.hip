// convsp.hip -- 3x3 stride-1 NHWC convolution on split-precision matrix instructions, second structure:
// WEIGHT FRAGMENTS STRAIGHT FROM L2, ONE BARRIER PER 16-CHANNEL SLAB.
//
// Same arithmetic as convbf.hip (FMT = 3: three bf16 planes, six products; FMT = 2: two fp16 planes, x = h + l,
// three products) and the same packed-weight layout [tap][slab][FMT][cout][16]; what changes is who reads
// what.  PMC on convbf.hip's kernel showed the matrix pipe 30 % busy with waves parked 46 % of the time at the
// per-tap barriers that hand the weight tiles through LDS.  Here a wave owns 128 pixels (four image rows x 32 columns)
// x 32 output channels: no two waves of a pixel group need the same weights, so a weight fragment is one 16-byte
// global load per lane (32 channels x 32 bytes = 1 KB contiguous in the packed layout) with a one-tap register
// prefetch, and LDS holds only the activations: the halo patch of a slab, split into planes while it is staged,
// double-buffered -- ONE barrier per slab (9 taps = 108 / 216 matrix instructions per wave).
//   PH = 1: block = 128 pixels x 128 channels (waves = 4 channel tiles);
//   PH = 2: block = 256 pixels x  64 channels (waves = 2 pixel groups x 2 channel tiles), for the 64-channel layers.
//   PH = 4: block = 4 pixel groups x ONE 32-channel tile (stride 1, the two-term formats; rows_hint 9 | 10): every wave of the block reads
//           the SAME weight fragments, so they come from L2 once per block and from the CU's vector cache for the other three waves.
//           For the low-resolution wide layers of a one-sample training step: a 64 x 64 x 256 -> 256 layer's 768 blocks of PH = 1,
//           one image row each, pull 1.2 MB of weight fragments apiece through L2 -- 0.9 GB per layer, which is what its 58 us were.
// With float32 activations in HBM this is the kernel of every 3x3 layer (bf16 split: always; fp16 split: training, and
// `SeFlowNet.split_acts = False`); the fp16-split inference network stores its maps already split and runs convsg.hip,
// which replaces the register staging below by LDS-DMA.  This kernel's epilogue can WRITE that format (kActSplitOut).
// Specification / oracle as conv.hip (reference network absent: PARITY UNPINNED).
#include "conv_common.h"
#include "bf16x3.h"

namespace himo {

// LDS layout of one patch pixel (16 channels x 2 bytes per plane), two ways to make the ds_read_b128 fragment reads --
// 32 CONSECUTIVE pixels from an arbitrary start (the tap offset) -- conflict-free:
//   padded  : 48-byte pitch (16 bytes of padding), no address arithmetic;
//   swizzled: 32-byte pitch, the two 16-byte halves swapped for pixels with bit 3 set.  ds_read_b128's lane groups
//             ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) pair lanes whose pixels are 8 or 24 apart, so with the swap
//             every group covers the 16 slots of the 256-byte bank row exactly once wherever the run starts.
// The swizzle costs three integer instructions per fragment read but a third less LDS: it is used where that buys
// occupancy or double buffering (the 256-pixel blocks of the 64-channel layers, the stride-2 patches), the padding
// where it does not (measured: 388 vs 362 TFLOP/s float32-equivalent on the 128-pixel blocks).
template <bool SWZ> struct PatchLayout;
template <> struct PatchLayout<false> {
    static constexpr int kPitch = 48;
    __device__ static inline int slot(int pix, int half) { return pix * 48 + (half << 4); }
};
template <> struct PatchLayout<true> {
    static constexpr int kPitch = 32;
    __device__ static inline int slot(int pix, int half) { return pix * 32 + ((half ^ ((pix >> 3) & 1)) << 4); }
};

// S = 2 (the three stride-2 layers): the halo patch is (2 TH + 1) x 65 input pixels, stored with even and odd
// columns de-interleaved ([33 even | 32 odd] per row) so that the 32 output pixels of a fragment -- input columns
// 2 li + kx -- are again 32 CONSECUTIVE patch pixels for every tap.
// FMT = 4: the two-term bf16 split x = h + m (16 significant bits, float32 range; three products h*h + h*m + m*h) -- the data-gradient
// convolutions of the mixed-precision training step (himo_conv_pack_weights_ex format 2); NP = planes of the format.
// STUF (FMT = 4, S = 1 only): the zero-stuffed input of kActStuffedIn, as its own instantiation (the plain kernels sit at their
// register limit: the second path spilled in them).
template <int EPI, int PH, int FMT, int MI, int S, bool STUF = false>
__global__ __launch_bounds__(256, (FMT != 3 && S == 1) ? 3 : 2)
void conv3_split_kernel(ConvArgs a, const unsigned short* __restrict__ wpk) {
    constexpr int NP = FMT == 3 ? 3 : 2;
    constexpr int TW = 32, TH = MI * PH;
    constexpr int PW = S == 1 ? TW + 2 : 2 * TW + 1, PHt = S == 1 ? TH + 2 : 2 * TH + 1, NPIX = PHt * PW;
    constexpr int BN = (4 / PH) * 32;
    constexpr int kPatchItems = NPIX * 4;
    constexpr int kPatchPerThread = (kPatchItems + 255) / 256;
    // double-buffered while two blocks still fit a CU's LDS; else single-buffered with a second barrier per slab
    using PL = PatchLayout<(PH >= 2 || S == 2)>;
    constexpr int NB = 2 * NP * NPIX * PL::kPitch <= 66 * 1024 ? 2 : 1;
    // raw bytes, at least the 4 x 4 KB the vectorised epilogue stages through (store_block_vec, conv_common.h)
    constexpr int kPatchBytes = NB * NP * NPIX * PL::kPitch;
    __shared__ __attribute__((aligned(16))) unsigned char patch_raw[kPatchBytes < 16384 ? 16384 : kPatchBytes];
    auto& patch = *reinterpret_cast<unsigned char (*)[NB][NP][NPIX * PL::kPitch]>(patch_raw);

    const int n_tiles_n = (a.Cout + BN - 1) / BN;
    int bid = xcd_block_id(blockIdx.x, gridDim.x);
    const int tn = bid % n_tiles_n; bid /= n_tiles_n;
    const int tx = (a.Wo + TW - 1) / TW, ty = (a.Ho + TH - 1) / TH;
    const int ox0 = (bid % tx) * TW; bid /= tx;
    const int oy0 = (bid % ty) * TH;
    const int img = bid / ty;
    const float* __restrict__ xin = a.x + image_offset(img, a.n_inner, a.x_batch_stride, a.x_outer_stride);
    const int slabs = (a.Cin + 15) / 16;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wp = wave % PH, wc = wave / PH;
    const int wpu = __builtin_amdgcn_readfirstlane(wp);          // (scalar copy: the zero-row test of the stuffed input)
    const int li = lane & 31, lh = lane >> 5;
    const int co = tn * BN + wc * 32 + li;
    const bool co_ok = co < a.Cout;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

    floatx16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

    // The halo patch comes through a BUFFER resource over this image: an item's byte offset (pixel, channel quad) is computed
    // ONCE, out-of-image items get an offset beyond the resource (the load returns zeros without a branch), and a slab is the
    // scalar offset of the instruction -- a slab's loads cost no vector arithmetic at all.  (Per-slab 64-bit per-lane address
    // arithmetic with an exec-mask branch around every load was ~25 vector instructions per item and slab.)
    // kActStuffedIn (FMT = 4, S = 1: the data gradient of a stride-2 layer): x is the COMPACT [H / 2][W / 2] gradient map and the
    // image convolved is its zero-stuffed x2 version -- pixel (iy, ix) exists where both are even, at compact (iy / 2, ix / 2); every
    // other item lies "outside" and loads zeros.  Rows with odd iy are zero as a whole: their matrix instructions are skipped below.
    constexpr bool stuffed = STUF;
    const int xh = stuffed ? a.H >> 1 : a.H, xw = stuffed ? a.W >> 1 : a.W;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xin), 0, (int)min((int64_t)xh * xw * a.x_pitch * 4, (int64_t)0x7fffffff), 0x00020000);
    constexpr unsigned kOutside = 0x80000000u;
    unsigned poff[kPatchPerThread];
#pragma unroll
    for (int it = 0; it < kPatchPerThread; ++it) {
        const int item = it * 256 + threadIdx.x;
        const int pp = item >> 2, q = item & 3;
        const int pc = pp % PW;
        const int iy = iy0 + pp / PW, ix = ix0 + (S == 1 ? pc : pc < 33 ? 2 * pc : 2 * (pc - 33) + 1);
        bool ok = item < kPatchItems && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        if (stuffed) {
            ok = ok && !((iy | ix) & 1);
            poff[it] = ok ? (unsigned)((((iy >> 1) * xw + (ix >> 1)) * a.x_pitch + q * 4) * 4) : kOutside;
        } else {
            poff[it] = ok ? (unsigned)(((iy * a.W + ix) * a.x_pitch + q * 4) * 4) : kOutside;
        }
    }
    const bool ragged_cin = (a.Cin & 15) != 0;          // the last slab of a Cin that is no multiple of 16: per-quad test
    auto load_patch = [&](int slab, float4 (&r)[kPatchPerThread]) {
#pragma unroll
        for (int it = 0; it < kPatchPerThread; ++it) {
            unsigned off = poff[it];
            if (ragged_cin && slab * 16 + (int)((it * 256 + threadIdx.x) & 3) * 4 >= a.Cin) off = kOutside;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, slab * 64, 0);
            r[it] = __builtin_bit_cast(float4, v);      // (whole-vector cast: element extraction from this builtin's result miscompiles to one dword)
        }
    };
    // part 0..2: a third of the items each (the staging of the NEXT slab is spread over the three kernel rows of the
    // current one, so its conversion work runs in the shadow of the matrix instructions); part < 0: everything
    auto store_patch = [&](int buf, const float4 (&r)[kPatchPerThread], int part) {
#pragma unroll
        for (int it = 0; it < kPatchPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            const int mine = it * 3 / kPatchPerThread;          // compile-time per unrolled iteration
            if (item < kPatchItems && (part < 0 || part == mine)) {
                const int pp = item >> 2, q = item & 3;
                unsigned h[4], m[4], l[4];
                const int off = PL::slot(pp, q >> 1) + (q & 1) * 8;
                if (FMT == 3) {
                    split3(r[it].x, h[0], m[0], l[0]); split3(r[it].y, h[1], m[1], l[1]);
                    split3(r[it].z, h[2], m[2], l[2]); split3(r[it].w, h[3], m[3], l[3]);
                    *reinterpret_cast<uint2*>(&patch[buf][1][off]) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
                } else if (FMT == 4) {
                    const float xv[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = bf16_rne_bits(xv[e]); l[e] = bf16_rne_bits(xv[e] - bf16_bits_to_float(h[e])); }
                } else {
                    split2(r[it].x, h[0], l[0]); split2(r[it].y, h[1], l[1]);
                    split2(r[it].z, h[2], l[2]); split2(r[it].w, h[3], l[3]);
                }
                *reinterpret_cast<uint2*>(&patch[buf][0][off]) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(&patch[buf][NP - 1][off]) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
        }
    };
    // this wave's weight fragments of one (tap, slab): FMT x 16 bytes per lane
    const int co_ld = co_ok ? co : a.Cout - 1;        // out-of-range lanes read a valid column; their results are never stored
    auto load_b = [&](int tap, int slab, uint4 (&b)[NP]) {
        const unsigned short* base = wpk + (((int64_t)tap * slabs + slab) * NP) * a.Cout * 16 + (int64_t)co_ld * 16 + lh * 8;
#pragma unroll
        for (int s = 0; s < NP; ++s) b[s] = *reinterpret_cast<const uint4*>(base + (int64_t)s * a.Cout * 16);
    };

    // Weight fragments run TWO taps ahead of their use (an L2 round trip outlasts one tap's matrix work) in three
    // register sets whose roles rotate statically -- tap t reads set t % 3 and loads tap t + 2 into set (t + 2) % 3 --
    // so no copy ever forces a wait on a load issued in the same tap; 9 taps = 3 kernel rows keep the rotation aligned
    // across slabs.  The loads are unconditional (clamped indices) to keep the loop body free of branches.
    float4 pr[kPatchPerThread];
    uint4 bq[3][NP];
    load_patch(0, pr);
    load_b(0, 0, bq[0]);
    load_b(1, 0, bq[1]);
    store_patch(0, pr, -1);
    __syncthreads();

#pragma unroll 1
    for (int slab = 0; slab < slabs; ++slab) {
        const int buf = NB == 2 ? (slab & 1) : 0;
        const bool more = slab + 1 < slabs;
        const int nslab = more ? slab + 1 : slab;            // clamped: the last slab re-loads its own (unused) weights
        if (more) load_patch(slab + 1, pr);
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tap = ky * 3 + kx;
                {   // prefetch tap + 2 (of this slab, or taps 0 / 1 of the next)
                    const int t2 = tap + 2;
                    load_b(t2 < 9 ? t2 : t2 - 9, t2 < 9 ? slab : nslab, bq[(kx + 2) % 3]);
                }
                const int tapoff = S == 1 ? ky * PW + kx : ky * PW + (kx & 1) * 33 + (kx >> 1);
                const uint4 (&bcur)[NP] = bq[kx];
                if constexpr (STUF) {
                    {                            // output row oy reads input row oy + ky - 1: zero as a whole when that is odd (scalar test)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            if ((oy0 + wpu * MI + mi + ky - 1) & 1) continue;
                            bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&patch[buf][0][PL::slot((wp * MI + mi) * PW + li + tapoff, lh)]);
                            bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&patch[buf][1][PL::slot((wp * MI + mi) * PW + li + tapoff, lh)]);
                            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, bcur[0]), acc[mi], 0, 0, 0);
                            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, bcur[1]), acc[mi], 0, 0, 0);
                            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, bcur[0]), acc[mi], 0, 0, 0);
                        }
                        continue;
                    }
                }
                bf16x8 af[MI][NP];
#pragma unroll
                for (int s = 0; s < NP; ++s)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        af[mi][s] = *reinterpret_cast<const bf16x8*>(&patch[buf][s][PL::slot((wp * MI + mi) * S * PW + li + tapoff, lh)]);
#define HIMO_TERM(SA, SB)                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                \
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][SA], __builtin_bit_cast(bf16x8, bcur[SB]), acc[mi], 0, 0, 0);
#define HIMO_TERM16(ACC, SA, SB)                                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                \
        ACC[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[mi][SA]),                      \
                                                        __builtin_bit_cast(f16x8, bcur[SB]), ACC[mi], 0, 0, 0);
                if constexpr (FMT == 3) {
                    HIMO_TERM(2, 0) HIMO_TERM(0, 2) HIMO_TERM(1, 1) HIMO_TERM(1, 0) HIMO_TERM(0, 1) HIMO_TERM(0, 0)
                } else if constexpr (FMT == 4) {
                    HIMO_TERM(1, 0) HIMO_TERM(0, 1) HIMO_TERM(0, 0)
                } else {
                    HIMO_TERM16(acc, 1, 0) HIMO_TERM16(acc, 0, 1) HIMO_TERM16(acc, 0, 0)
                }
#undef HIMO_TERM16
#undef HIMO_TERM
                // scheduling: this tap's weight prefetch and ALL its activation-fragment reads are issued before its
                // matrix instructions (the compiler otherwise feeds each MFMA pair from a just-issued ds_read and
                // exposes the LDS latency four to six times per tap)
                // (measured: +20 % for the bf16 split; the fp16 split has half the matrix work per read and no registers to
                // spare for it -- its schedule is left to the compiler)
                if (FMT == 3) {
                    __builtin_amdgcn_sched_group_barrier(0x020, FMT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, MI * FMT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MI * 6, 0);
                }
            }
            if (NB == 2 && more) store_patch(buf ^ 1, pr, ky);   // the other buffer was last read before the previous barrier
        }
        if (more) {
            if (NB == 1) {
                __syncthreads();               // single buffer: every wave is done reading this slab's patch
                store_patch(0, pr, -1);
            }
            __syncthreads();
        }
    }

    float* __restrict__ yout = a.y + image_offset(img, a.n_inner, a.y_batch_stride, a.y_outer_stride);
    const bool osplit = (a.act_flags & kActSplitOut) != 0;       // output in the split activation format (convsg.hip)
    if (EPI != kEpiReluMask && (a.act_flags & kActVecStore)) {   // 16-byte stores through a wave-private LDS transpose
        const int cl = co_ok ? co : a.Cout - 1;
        const float bv = a.bias ? a.bias[cl] : 0.f;
        float scv = 1.f, shv = 0.f;
        if (EPI == kEpiBiasBnGelu) { scv = a.scale[cl]; shv = a.shift[cl]; }
        float eA = 1.f, eB = 0.f;                                 // fused bias / BatchNorm affine of the backbone's epilogues (conv_common.h)
        if (kEpiAffine<EPI>) epi_affine<EPI>(FMT == 2 ? kF16AccScale : 1.f, bv, scv, shv, eA, eB);
        __syncthreads();                                          // every wave has read its last patch rows
        unsigned char* stg = patch_raw + wave * 4096;
        const int n_px = a.Wo - ox0 < 32 ? a.Wo - ox0 : 32;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int oy = oy0 + wp * MI + mi;
            unsigned word[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mi][r];
                if (kEpiAffine<EPI>) {
                    v = epi_activate<EPI>(v, eA, eB);
                } else {
                    if (FMT == 2) v *= kF16AccScale;
                    v = epilogue_value<EPI>(v + bv, scv, shv);
                }
                word[r] = (FMT == 2 && osplit) ? split_word(v, li & 1) : __builtin_bit_cast(unsigned, v);
                if (FMT == 2 && mi == 0 && r == 0 && osplit) note_range(a, (oy < a.Ho && 4 * lh < n_px && tn * BN + wc * 32 + li < a.Cout) ? v : 0.f);
            }
            if (FMT == 2 && osplit) store_block_vec<true>(a, yout, stg, word, lane, (int64_t)oy * a.Wo + ox0, oy < a.Ho ? n_px : 0, tn * BN + wc * 32);
            else if (FMT == 4 && (a.act_flags & kActAccumulate))
                store_block_vec<false, false, true>(a, yout, stg, word, lane, (int64_t)oy * a.Wo + ox0, oy < a.Ho ? n_px : 0, tn * BN + wc * 32);
            else store_block_vec<false>(a, yout, stg, word, lane, (int64_t)oy * a.Wo + ox0, oy < a.Ho ? n_px : 0, tn * BN + wc * 32);
        }
        return;
    }
    if (!co_ok) return;
    const float b = a.bias ? a.bias[co] : 0.f;
    float sc = 1.f, sh = 0.f;
    if (EPI == kEpiBiasBnGelu) { sc = a.scale[co]; sh = a.shift[co]; }
    float eA = 1.f, eB = 0.f;
    if (kEpiAffine<EPI>) epi_affine<EPI>(FMT == 2 ? kF16AccScale : 1.f, b, sc, sh, eA, eB);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int oy = oy0 + wp * MI + mi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = acc[mi][r];
            if (oy < a.Ho && ox < a.Wo) {
                const int64_t pix = (int64_t)oy * a.Wo + ox;
                if (kEpiAffine<EPI>) {
                    v = epi_activate<EPI>(v, eA, eB);
                    if (FMT == 2 && osplit) split_store<kEpiBias, true>(a, yout, pix, co, v, 1.f, 0.f);
                    else yout[pix * a.y_pitch + co] = v;
                } else {
                    if (FMT == 2) v *= kF16AccScale;
                    if (FMT == 2 && osplit) split_store<EPI, true>(a, yout, pix, co, v + b, sc, sh);
                    else epilogue_store<EPI>(a, yout, pix, co, v + b, sc, sh);
                }
            }
        }
    }
}

template <int PH, int FMT, int MI, int S>
static void launch_sp_epi(const ConvArgs& a_in, int epi, const unsigned short* w, dim3 grid, hipStream_t s) {
    ConvArgs a = a_in;
    if (vec_store_ok(a) && (!(a.act_flags & kActSplitOut) || !(a.Cout & 15))) a.act_flags |= kActVecStore;
    if constexpr (FMT == 4) {          // the data-gradient format: bias epilogue only (launch_conv3_split has checked)
        if constexpr (S == 1) {
            if (a.act_flags & kActStuffedIn) { hipLaunchKernelGGL((conv3_split_kernel<kEpiBias, PH, FMT, MI, S, true>), grid, dim3(256), 0, s, a, w); return; }
        }
        hipLaunchKernelGGL((conv3_split_kernel<kEpiBias, PH, FMT, MI, S>), grid, dim3(256), 0, s, a, w);
        return;
    }
    switch (epi) {
        case kEpiBias: hipLaunchKernelGGL((conv3_split_kernel<kEpiBias, PH, FMT, MI, S>), grid, dim3(256), 0, s, a, w); break;
        case kEpiBiasBnGelu: hipLaunchKernelGGL((conv3_split_kernel<kEpiBiasBnGelu, PH, FMT, MI, S>), grid, dim3(256), 0, s, a, w); break;
        case kEpiBiasGelu: hipLaunchKernelGGL((conv3_split_kernel<kEpiBiasGelu, PH, FMT, MI, S>), grid, dim3(256), 0, s, a, w); break;
        case kEpiBiasRelu: hipLaunchKernelGGL((conv3_split_kernel<kEpiBiasRelu, PH, FMT, MI, S>), grid, dim3(256), 0, s, a, w); break;
        default: hipLaunchKernelGGL((conv3_split_kernel<kEpiReluMask, PH, FMT, MI, S>), grid, dim3(256), 0, s, a, w); break;
    }
}

template <int PH, int FMT>
static void launch_sp_mi(const ConvArgs& a, int epi, int mi, int stride, const unsigned short* w, dim3 grid, hipStream_t s) {
    if (stride == 2) {                                                         // stride 2: two output rows per wave, or one
        if (mi == 1) launch_sp_epi<PH, FMT, 1, 2>(a, epi, w, grid, s);        // (64-channel blocks: keeps the 5 x 65 patch double-buffered)
        else launch_sp_epi<PH, FMT, 2, 2>(a, epi, w, grid, s);
    }
    else if (mi == 4) launch_sp_epi<PH, FMT, 4, 1>(a, epi, w, grid, s);
    else if (mi == 1) launch_sp_epi<PH, FMT, 1, 1>(a, epi, w, grid, s);         // small images: more, smaller blocks
    else launch_sp_epi<PH, FMT, 2, 1>(a, epi, w, grid, s);
}

// 3x3 layers (stride 1 | 2) with a plain epilogue; returns false when this structure does not apply (GRU epilogues).
// rows_hint: 0 = heuristic, else image rows per wave (4 | 2 | 1; stride 2 always uses 2).
bool launch_conv3_split(const ConvArgs& a, int epilogue, const void* w_packed, int format, int rows_hint, int stride, hipStream_t s) {
    if (epilogue == kEpiGruZR || epilogue == kEpiGruQ) return false;
    if (format == 2 && (epilogue != kEpiBias || (a.act_flags & ~(kActAccumulate | kActStuffedIn)))) return false;   // two-term bf16: float32 maps, bias epilogue
    if ((a.act_flags & kActAccumulate) && (format != 2 || !vec_store_ok(a))) return false;              // y += result: that kernel's 16-byte store path only
    if ((a.act_flags & kActStuffedIn) && (format != 2 || stride != 1)) return false;                    // zero-stuffed input: that kernel, stride 1
    if ((int64_t)a.H * a.W * a.x_pitch * 4 >= ((int64_t)1 << 31)) return false;                         // 32-bit byte offsets into an image (buffer resource)
    // rows_hint 5 | 6: PH = 2 forced (wide layers too), 1 | 2 rows per wave; 9 | 10: PH = 4, 1 | 2 rows per wave -- stride 1, two-term formats
    const int ph_hint = (stride == 1 && format != 0 && (rows_hint == 5 || rows_hint == 6)) ? 2
                      : (stride == 1 && format != 0 && (rows_hint == 9 || rows_hint == 10)) ? 4 : 0;
    if (!ph_hint && rows_hint > 4) return false;
    const bool wide = a.Cout > 64 && ph_hint == 0;     // PH = 1: 128-channel tiles; PH = 2: 64-channel tiles
    const int ph = ph_hint ? ph_hint : (wide ? 1 : 2), bn = (4 / ph) * 32;
    auto blocks_for = [&](int mi) -> int64_t {
        const int th = mi * ph;
        return (int64_t)a.N * ((a.Ho + th - 1) / th) * ((a.Wo + 31) / 32) * ((a.Cout + bn - 1) / bn);
    };
    int mi = blocks_for(4) >= 1024 ? 4 : 2;            // two blocks per CU, at least two rounds of them
    if (rows_hint == 4 || rows_hint == 2 || rows_hint == 1) mi = rows_hint;
    if (ph_hint) mi = rows_hint & 3;
    if (stride == 2) mi = (rows_hint == 1 || rows_hint == 2) ? rows_hint : (wide ? 2 : 1);
    const dim3 grid((unsigned)blocks_for(mi));
    const unsigned short* w = (const unsigned short*)w_packed;
    const char* name = stride == 2 ? (format == 1 ? "conv3x3s2_f16x2_kernel" : format == 2 ? "conv3x3s2_bf16x2_kernel" : "conv3x3s2_bf16x3_kernel")
                                   : (format == 1 ? "conv3x3_f16x2_kernel" : format == 2 ? "conv3x3_bf16x2_kernel" : "conv3x3_bf16x3_kernel");
    ProfScope ps(name, s);
    if (ph == 4) {                                      // stride 1, rows per wave 1 | 2 only (fewer instantiations)
        if (format == 2) { if (mi == 1) launch_sp_epi<4, 4, 1, 1>(a, epilogue, w, grid, s); else launch_sp_epi<4, 4, 2, 1>(a, epilogue, w, grid, s); }
        else { if (mi == 1) launch_sp_epi<4, 2, 1, 1>(a, epilogue, w, grid, s); else launch_sp_epi<4, 2, 2, 1>(a, epilogue, w, grid, s); }
        return true;
    }
    if (format == 2) { if (wide) launch_sp_mi<1, 4>(a, epilogue, mi, stride, w, grid, s); else launch_sp_mi<2, 4>(a, epilogue, mi, stride, w, grid, s); }
    else if (format == 1) { if (wide) launch_sp_mi<1, 2>(a, epilogue, mi, stride, w, grid, s); else launch_sp_mi<2, 2>(a, epilogue, mi, stride, w, grid, s); }
    else { if (wide) launch_sp_mi<1, 3>(a, epilogue, mi, stride, w, grid, s); else launch_sp_mi<2, 3>(a, epilogue, mi, stride, w, grid, s); }
    return true;
}

}  // namespace himo
