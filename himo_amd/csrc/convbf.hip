// convbf.hip -- stride-1 NHWC convolution / row GEMM with float32-class accuracy on the bf16 matrix
// instructions of gfx950: every float32 operand is split into three bf16 terms (x = h + m + l, 8 + 8 + 8
// mantissa bits) and the product is formed from the six significant cross terms
//     h*h + h*m + m*h + m*m + h*l + l*h            (dropped: m*l, l*m, l*l  <= 2^-24 relative)
// accumulated in float32 by v_mfma_f32_32x32x16_bf16.  Six 32-cycle bf16 MFMAs replace eight 64-cycle
// float32 MFMAs per 32x32x16 block: 2.67x the float32-MFMA rate at the same 1e-4 parity budget
// (north_star's bar against a float32 CPU path rules out plain bf16).
//
// Specification and oracle: as conv.hip (himo_amd/seflow/spec.py, oracle/seflow_oracle.py; reference source absent).
//
// Structure (block = 4 waves, 64*MI pixels x BN channels; the pixel tile is 2*MI rows x 32 columns so that one
// MFMA tile is 32 consecutive pixels of one image row):
//   * activations are split while they are staged: the input halo patch of a 16-channel slab is held in LDS as
//     three bf16 planes [pixel][16 + 8 pad] -- the 48-byte pixel pitch makes every ds_read_b128 fragment read
//     conflict-free -- and reused by all nine taps;
//   * weights are split ONCE at load time (himo_conv_pack_weights) into [tap][slab][term][cout][16] so a tap's
//     slab is three contiguous runs that are copied, double-buffered, into [cout][16 + 8 pad] LDS planes;
//   * per tap a wave reads 3*MI A fragments + 3*NI B fragments (16 bytes per lane each) and issues 6*MI*NI MFMAs,
//     rotating over the MI*NI accumulators so that consecutive MFMAs never share one;
//   * the next tap's weights (and the next slab's patch) are prefetched into registers under the MFMAs.
//
// FMT = 2 runs the same structure on the two-term fp16 split (bf16x3.h: x = h + l, three products per block);
// its two planes leave LDS room to stage a whole kernel row of weights per barrier.  The 3x3 layers normally run the
// second structure in convsp.hip (weight fragments straight from L2, one barrier per slab); this file keeps the row
// GEMMs (1x1 layers, GRU epilogues) and remains selectable per layer through himo_conv_desc.tile_hint.
#include "conv_common.h"
#include "bf16x3.h"
#include <algorithm>

namespace himo {

constexpr int kRowBytes = 48;      // LDS pitch of one pixel / one output channel: 16 bf16 + 8 bf16 of padding

// weights [T][Cin][Cout] float32 -> [T][slabs][FMT][Cout][16] (FMT = 3: bf16 h, m, l; FMT = 2: fp16 h, l')
template <int FMT>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int T, int Cin, int Cout,
                                                           unsigned short* __restrict__ out) {
    const int slabs = (Cin + 15) / 16;
    const int64_t total = (int64_t)T * slabs * Cout * 16;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= total) return;
    const int k = (int)(item % 16);
    const int co = (int)((item / 16) % Cout);
    const int slab = (int)((item / (16 * (int64_t)Cout)) % slabs);
    const int tap = (int)(item / (16 * (int64_t)Cout * slabs));
    const int ci = slab * 16 + k;
    const float x = ci < Cin ? w[((int64_t)tap * Cin + ci) * Cout + co] : 0.f;
    unsigned h, m = 0, l;
    constexpr int NP = FMT == 3 ? 3 : 2;               // FMT = 4: two bf16 planes (h, m), see convsp.hip
    if (FMT == 3) split3(x, h, m, l);
    else if (FMT == 4) { h = bf16_rne_bits(x); l = bf16_rne_bits(x - bf16_bits_to_float(h)); }
    else split2(x * kF16WeightScale, h, l);
    const int64_t base = (((int64_t)tap * slabs + slab) * NP) * Cout * 16 + (int64_t)co * 16 + k;
    out[base] = (unsigned short)h;
    if (FMT == 3) {
        out[base + (int64_t)Cout * 16] = (unsigned short)m;
        out[base + 2 * (int64_t)Cout * 16] = (unsigned short)l;
    } else {
        out[base + (int64_t)Cout * 16] = (unsigned short)l;
    }
}

// FMT = 4: the two-term bf16 split (planes h, m; products h*h + h*m + m*h): float32 range at 16 significant bits -- the data-gradient
// GEMMs of the mixed-precision training step (himo_conv_pack_weights_ex format 2; see convsp.hip).  NP = planes of the format.
template <int KS, int BN, int EPI, int MI, int FMT>
__global__ __launch_bounds__(256) void conv_bf16x3_kernel(ConvArgs a, const unsigned short* __restrict__ wpk) {
    constexpr int TW = 32, TH = 2 * MI, BM = 64 * MI;
    constexpr int PH = KS == 1 ? 1 : TH + 2;
    constexpr int PW = KS == 1 ? BM : TW + 2;
    constexpr int NPIX = PH * PW;
    constexpr int T = KS * KS;
    constexpr int WN = BN / 2, NI = WN / 32;
    constexpr int kPatchItems = NPIX * 4;                       // float4 (4 channels) per item
    constexpr int kPatchPerThread = (kPatchItems + 255) / 256;
    // taps whose weights are staged together (one barrier per group): the whole kernel row for the fp16 split, whose
    // two planes leave the LDS room; one tap for the three-plane bf16 split
    constexpr int G = (KS == 3 && FMT == 2) ? 3 : 1;
    constexpr int NP = FMT == 3 ? 3 : 2;
    constexpr int kWItemsTap = NP * BN * 2;                     // 16-byte half rows of one tap
    constexpr int kWItems = G * kWItemsTap;
    constexpr int kWPerThread = (kWItems + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char patch[NP][NPIX * kRowBytes];
    constexpr int WB = G == 1 ? 2 : 1;      // a staged kernel row is single-buffered (the register prefetch hides the loads)
    __shared__ __attribute__((aligned(16))) unsigned char wts[WB][G][NP][BN * kRowBytes];

    const int n_tiles_n = (a.Cout + BN - 1) / BN;
    int bid = xcd_block_id(blockIdx.x, gridDim.x);
    const int tn = bid % n_tiles_n; bid /= n_tiles_n;
    int oy0 = 0, ox0 = 0, img;
    int64_t row0 = 0;
    if (KS == 1) {
        const int64_t rows = (int64_t)a.Ho * a.Wo;
        const int tiles = (int)((rows + BM - 1) / BM);
        img = bid / tiles;
        row0 = (int64_t)(bid % tiles) * BM;
    } else {
        const int tx = (a.Wo + TW - 1) / TW, ty = (a.Ho + TH - 1) / TH;
        ox0 = (bid % tx) * TW; bid /= tx;
        oy0 = (bid % ty) * TH; img = bid / ty;
    }
    const int n0 = tn * BN;
    const float* __restrict__ xin = a.x + image_offset(img, a.n_inner, a.x_batch_stride, a.x_outer_stride);
    const int slabs = (a.Cin + 15) / 16;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;

    int ppA[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) ppA[mi] = KS == 1 ? (wm * MI + mi) * 32 + li : (wm * MI + mi) * PW + li;

    floatx16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int iy0 = oy0 - (KS / 2), ix0 = ox0 - (KS / 2);
    const int64_t in_rows = (int64_t)a.H * a.W;

    // the patch comes through a buffer resource over this image (as convsp.hip): an item's byte offset is computed once, items
    // outside the image lie beyond the resource (zeros, no branch), the slab is the instruction's scalar offset
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xin), 0, (int)std::min<int64_t>(in_rows * a.x_pitch * 4, (int64_t)0x7fffffff), 0x00020000);
    constexpr unsigned kOutside = 0x80000000u;
    unsigned poff[kPatchPerThread];
#pragma unroll
    for (int it = 0; it < kPatchPerThread; ++it) {
        const int item = it * 256 + threadIdx.x;
        const int pp = item >> 2, q = item & 3;
        int64_t pix;
        bool ok;
        if (KS == 1) { pix = row0 + pp; ok = pix < in_rows; }
        else {
            const int iy = iy0 + pp / PW, ix = ix0 + pp % PW;
            ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            pix = (int64_t)iy * a.W + ix;
        }
        poff[it] = (ok && item < kPatchItems) ? (unsigned)(((int)pix * a.x_pitch + q * 4) * 4) : kOutside;
    }
    const bool ragged_cin = (a.Cin & 15) != 0;          // the last slab of a Cin that is no multiple of 16: per-quad test
    auto load_patch = [&](int slab, float4 (&r)[kPatchPerThread]) {
#pragma unroll
        for (int it = 0; it < kPatchPerThread; ++it) {
            unsigned off = poff[it];
            if (ragged_cin && slab * 16 + (int)((it * 256 + threadIdx.x) & 3) * 4 >= a.Cin) off = kOutside;
            // (whole-vector cast: element extraction from this builtin's result miscompiles to one dword)
            r[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, slab * 64, 0));
        }
    };
    auto store_patch = [&](const float4 (&r)[kPatchPerThread]) {       // split into the three bf16 planes
#pragma unroll
        for (int it = 0; it < kPatchPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            if (item < kPatchItems) {
                const int pp = item >> 2, q = item & 3;
                unsigned h[4], m[4], l[4];
                const int off = pp * kRowBytes + q * 8;
                if (FMT == 3) {
                    split3(r[it].x, h[0], m[0], l[0]); split3(r[it].y, h[1], m[1], l[1]);
                    split3(r[it].z, h[2], m[2], l[2]); split3(r[it].w, h[3], m[3], l[3]);
                    *reinterpret_cast<uint2*>(&patch[1][off]) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
                } else if (FMT == 4) {
                    const float xv[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = bf16_rne_bits(xv[e]); l[e] = bf16_rne_bits(xv[e] - bf16_bits_to_float(h[e])); }
                } else {
                    split2(r[it].x, h[0], l[0]); split2(r[it].y, h[1], l[1]);
                    split2(r[it].z, h[2], l[2]); split2(r[it].w, h[3], l[3]);
                }
                *reinterpret_cast<uint2*>(&patch[0][off]) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(&patch[NP - 1][off]) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
        }
    };
    auto load_w = [&](int tap0, int slab, uint4 (&r)[kWPerThread]) {       // taps tap0 .. tap0 + G - 1 of one slab
#pragma unroll
        for (int it = 0; it < kWPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (item < kWItems) {
                const int t = item / kWItemsTap, rest = item % kWItemsTap;
                const int s = rest / (BN * 2), rem = rest % (BN * 2);
                const int co = n0 + (rem >> 1), half = rem & 1;
                const unsigned short* base = wpk + (((int64_t)(tap0 + t) * slabs + slab) * NP) * a.Cout * 16;
                if (co < a.Cout) v = *reinterpret_cast<const uint4*>(base + ((int64_t)s * a.Cout + co) * 16 + half * 8);
            }
            r[it] = v;
        }
    };
    auto store_w = [&](int buf, const uint4 (&r)[kWPerThread]) {
#pragma unroll
        for (int it = 0; it < kWPerThread; ++it) {
            const int item = it * 256 + threadIdx.x;
            if (item < kWItems) {
                const int t = item / kWItemsTap, rest = item % kWItemsTap;
                const int s = rest / (BN * 2), rem = rest % (BN * 2);
                *reinterpret_cast<uint4*>(&wts[buf][t][s][(rem >> 1) * kRowBytes + (rem & 1) * 16]) = r[it];
            }
        }
    };

    float4 pr[kPatchPerThread];
    uint4 wr[kWPerThread];
    load_patch(0, pr);
    load_w(0, 0, wr);
    store_patch(pr);
    store_w(0, wr);
    __syncthreads();

    int wbuf = 0;
#pragma unroll 1
    for (int slab = 0; slab < slabs; ++slab) {
#pragma unroll 1
        for (int tap0 = 0; tap0 < T; tap0 += G) {
            const bool last_grp = tap0 + G >= T;
            const bool has_next = !(last_grp && slab + 1 >= slabs);
            const int ntap = last_grp ? 0 : tap0 + G, nslab = last_grp ? slab + 1 : slab;
            if (has_next) {
                load_w(ntap, nslab, wr);
                if (last_grp) load_patch(nslab, pr);
            }
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int tap = tap0 + t;
                const int tapoff = KS == 1 ? 0 : (tap / KS) * PW + (tap % KS);
                bf16x8 af[MI][NP], bf[NI][NP];
#pragma unroll
                for (int s = 0; s < NP; ++s) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        af[mi][s] = *reinterpret_cast<const bf16x8*>(&patch[s][(ppA[mi] + tapoff) * kRowBytes + lh * 16]);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bf[ni][s] = *reinterpret_cast<const bf16x8*>(&wts[wbuf][t][s][(wn * WN + ni * 32 + li) * kRowBytes + lh * 16]);
                }
                // cross terms, smallest first; accumulators rotate so consecutive MFMAs are independent
#define HIMO_TERM(SA, SB)                                                                                         \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)            \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][SA], bf[ni][SB], acc[mi][ni], 0, 0, 0);
#define HIMO_TERM16(ACC, SA, SB)                                                                                  \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)            \
        ACC[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[mi][SA]),                 \
                                                            __builtin_bit_cast(f16x8, bf[ni][SB]), ACC[mi][ni], 0, 0, 0);
                if constexpr (FMT == 3) {
                    HIMO_TERM(2, 0) HIMO_TERM(0, 2) HIMO_TERM(1, 1) HIMO_TERM(1, 0) HIMO_TERM(0, 1) HIMO_TERM(0, 0)
                } else if constexpr (FMT == 4) {
                    HIMO_TERM(1, 0) HIMO_TERM(0, 1) HIMO_TERM(0, 0)
                } else {
                    HIMO_TERM16(acc, 1, 0) HIMO_TERM16(acc, 0, 1) HIMO_TERM16(acc, 0, 0)
                }
#undef HIMO_TERM16
#undef HIMO_TERM
            }
            if (WB == 2) {
                if (has_next) store_w(wbuf ^ 1, wr);
                if (last_grp && has_next) {
                    __syncthreads();                            // every wave is done with this slab's patch
                    store_patch(pr);
                }
                __syncthreads();
                wbuf ^= 1;
            } else if (has_next) {
                __syncthreads();                                // every wave is done with these weights (and the patch)
                store_w(0, wr);
                if (last_grp) store_patch(pr);
                __syncthreads();
            }
        }
    }

    float* __restrict__ yout = a.y + image_offset(img, a.n_inner, a.y_batch_stride, a.y_outer_stride);
    const int64_t out_rows = (int64_t)a.Ho * a.Wo;
    // Row GEMMs with an element-wise epilogue: 32-bit element offsets from the (uniform) image base -- a tile's first row times the
    // pitch once per 32-row block, the 16 rows of a lane as scalar multiples of the pitch -- and no per-store bound test in whole
    // tiles.  (A 64-bit multiply-add and an exec-mask branch per store made the epilogue of a 64-channel 1x1 layer, 64 stores per
    // lane, longer than its 48 matrix instructions.)
    constexpr bool kFastEpi = KS == 1 && (kEpiAffine<EPI> || EPI == kEpiBiasRelu || EPI == kEpiReluMask);
    if (kFastEpi && out_rows * a.y_pitch < (1ll << 30) && (EPI != kEpiReluMask || out_rows * a.aux_in_pitch < (1ll << 30))) {
        const int wmu = __builtin_amdgcn_readfirstlane(wm);
        const float* __restrict__ aux = a.aux_in;                  // (the mask is addressed by the row, as in epilogue_store)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int co = n0 + wn * WN + ni * 32 + li;
            if (co >= a.Cout) continue;
            const float b = a.bias ? a.bias[co] : 0.f;
            float sc = 1.f, sh = 0.f;
            if (EPI == kEpiBiasBnGelu) { sc = a.scale[co]; sh = a.shift[co]; }
            float eA = 1.f, eB = 0.f;
            if (kEpiAffine<EPI>) epi_affine<EPI>(FMT == 2 ? kF16AccScale : 1.f, b, sc, sh, eA, eB);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int64_t p0 = row0 + (wmu * MI + mi) * 32;                       // scalar: this 32-row block's first row
                const int n_valid = (int)(out_rows - p0 < 32 ? out_rows - p0 : 32);  // scalar
                const unsigned off = ((unsigned)p0 + 4u * lh) * (unsigned)a.y_pitch + (unsigned)co;
                const unsigned aoff = EPI == kEpiReluMask ? ((unsigned)p0 + 4u * lh) * (unsigned)a.aux_in_pitch + (unsigned)co : 0u;
                auto value = [&](int r, unsigned col) -> float {
                    float v = acc[mi][ni][r];
                    if (kEpiAffine<EPI>) return epi_activate<EPI>(v, eA, eB);
                    if (FMT == 2) v *= kF16AccScale;
                    v += b;
                    if (EPI == kEpiBiasRelu) return fmaxf(v, 0.f);
                    return aux[aoff + col * (unsigned)a.aux_in_pitch] > 0.f ? v : 0.f;      // kEpiReluMask
                };
                // HIMO_ACT_ACCUMULATE (row GEMMs with the bias epilogue): y += result -- the fan-in of a gradient with two producers in
                // the second producer's own epilogue instead of a scratch map and an add pass (the decoder's skip gradient into the
                // pillar-image gradient the head's scatter wrote: 3.2 GB of traffic -> 1.6 GB per 8-sample pass)
                const bool accum = EPI == kEpiBias && (a.act_flags & kActAccumulate);
                if (n_valid == 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned col = (r & 3) + 8 * (r >> 2);
                        float v = value(r, col);
                        if (accum) v += yout[off + col * (unsigned)a.y_pitch];
                        yout[off + col * (unsigned)a.y_pitch] = v;
                    }
                } else if (n_valid > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned col = (r & 3) + 8 * (r >> 2);
                        if ((int)(col + 4 * lh) < n_valid) {
                            float v = value(r, col);
                            if (accum) v += yout[off + col * (unsigned)a.y_pitch];
                            yout[off + col * (unsigned)a.y_pitch] = v;
                        }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = n0 + wn * WN + ni * 32 + li;
        if (co >= a.Cout) continue;
        const float b = a.bias ? a.bias[co] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (EPI == kEpiBiasBnGelu) { sc = a.scale[co]; sh = a.shift[co]; }
        float eA = 1.f, eB = 0.f;                                 // fused bias / BatchNorm affine (conv_common.h): same bits as convsg / convsp
        if (kEpiAffine<EPI>) epi_affine<EPI>(FMT == 2 ? kF16AccScale : 1.f, b, sc, sh, eA, eB);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (r & 3) + 8 * (r >> 2) + 4 * lh;           // pixel within the 32-pixel MFMA tile
                int64_t pix;
                bool ok;
                if (KS == 1) { pix = row0 + (wm * MI + mi) * 32 + col; ok = pix < out_rows; }
                else {
                    const int oy = oy0 + wm * MI + mi, ox = ox0 + col;
                    ok = oy < a.Ho && ox < a.Wo;
                    pix = (int64_t)oy * a.Wo + ox;
                }
                float v = acc[mi][ni][r];
                if (kEpiAffine<EPI>) {
                    if (ok) yout[pix * a.y_pitch + co] = epi_activate<EPI>(v, eA, eB);
                } else {
                    if (FMT == 2) v *= kF16AccScale;
                    if (ok) epilogue_store<EPI>(a, yout, pix, co, v + b, sc, sh);
                }
            }
        }
    }
}

template <int KS, int BN, int MI, int FMT>
static void launch_bf_epi(const ConvArgs& a, int epi, const unsigned short* w, dim3 grid, hipStream_t s) {
    if constexpr (FMT == 4) {              // the data-gradient format: bias, or bias + ReLU mask of the layer's input (launch_conv_bf16x3 has checked)
        if (epi == kEpiReluMask) hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiReluMask, MI, FMT>), grid, dim3(256), 0, s, a, w);
        else hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiBias, MI, FMT>), grid, dim3(256), 0, s, a, w);
        return;
    }
    switch (epi) {
        case kEpiBias: hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiBias, MI, FMT>), grid, dim3(256), 0, s, a, w); break;
        case kEpiBiasBnGelu: hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiBiasBnGelu, MI, FMT>), grid, dim3(256), 0, s, a, w); break;
        case kEpiBiasGelu: hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiBiasGelu, MI, FMT>), grid, dim3(256), 0, s, a, w); break;
        case kEpiGruZR: hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiGruZR, MI, FMT>), grid, dim3(256), 0, s, a, w); break;
        case kEpiBiasRelu: hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiBiasRelu, MI, FMT>), grid, dim3(256), 0, s, a, w); break;
        case kEpiReluMask: hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiReluMask, MI, FMT>), grid, dim3(256), 0, s, a, w); break;
        default: hipLaunchKernelGGL((conv_bf16x3_kernel<KS, BN, kEpiGruQ, MI, FMT>), grid, dim3(256), 0, s, a, w); break;
    }
}

template <int KS, int FMT>
static void launch_bf_tile(const ConvArgs& a, int epi, int bn, int mi, const unsigned short* w, dim3 grid, hipStream_t s) {
    if (bn == 128) { if (mi == 2) launch_bf_epi<KS, 128, 2, FMT>(a, epi, w, grid, s); else launch_bf_epi<KS, 128, 1, FMT>(a, epi, w, grid, s); }
    else { if (mi == 2) launch_bf_epi<KS, 64, 2, FMT>(a, epi, w, grid, s); else launch_bf_epi<KS, 64, 1, FMT>(a, epi, w, grid, s); }
}

bool launch_conv3_split(const ConvArgs& a, int epilogue, const void* w_packed, int format, int rows_hint, int stride, hipStream_t s);   // convsp.hip

bool launch_conv3_presplit(const ConvArgs& a, int epilogue, const void* w_packed, int rows_hint, bool out_split, int stride, hipStream_t s);
bool launch_conv1_presplit(const ConvArgs& a, int epilogue, const void* w_packed, int rows_hint, bool out_split, hipStream_t s);   // convsg.hip

int launch_conv_bf16x3(const ConvArgs& a, int ksize, int epilogue, const void* w_packed, int tile_hint, int format, int stride, hipStream_t s) {
    if ((a.act_flags & kActSplitIn) && ksize == 1) {
        if (!launch_conv1_presplit(a, epilogue, w_packed, tile_hint & 15, (a.act_flags & kActSplitOut) != 0, s)) return HIMO_ERR_UNSUPPORTED;
        HIMO_LAUNCH_CHECK("conv1_presplit_kernel");
        return HIMO_OK;
    }
    if (a.act_flags & kActSplitIn) {            // input already split in HBM: the LDS-DMA kernel
        if (!launch_conv3_presplit(a, epilogue, w_packed, tile_hint & 15, (a.act_flags & kActSplitOut) != 0, stride, s)) return HIMO_ERR_UNSUPPORTED;
        HIMO_LAUNCH_CHECK("conv3_presplit_kernel");
        return HIMO_OK;
    }
    const bool gemm_accumulate = a.act_flags == kActAccumulate && ksize == 1 && epilogue == kEpiBias &&
                                 (int64_t)a.Ho * a.Wo * a.y_pitch < (1ll << 30);      // (the row GEMMs' 32-bit-offset epilogue: the one that implements it)
    if (a.act_flags && !gemm_accumulate && !(ksize == 3 && (!tile_hint || (tile_hint & 0x1000) || stride == 2))) return HIMO_ERR_UNSUPPORTED;
    // 3x3 layers: the weights-from-L2 structure (convsp.hip; tile_hint 0x1000 | rows-per-wave pins its variant) unless the
    // caller pins a tile of this file's kernel
    if (ksize == 3 && (!tile_hint || (tile_hint & 0x1000) || stride == 2) &&
        launch_conv3_split(a, epilogue, w_packed, format, tile_hint & 15, stride, s)) {
        HIMO_LAUNCH_CHECK("conv3_split_kernel");
        return HIMO_OK;
    }
    if (ksize == 3 && (tile_hint & 0x1000) && (tile_hint & 15) > 4) return HIMO_ERR_UNSUPPORTED;     // a pinned variant this layer does not admit
    if (stride != 1) return HIMO_ERR_UNSUPPORTED;
    if ((int64_t)a.H * a.W * a.x_pitch * 4 >= ((int64_t)1 << 31)) return HIMO_ERR_UNSUPPORTED;          // 32-bit byte offsets into an image (buffer resource)
    if (format == 2 && (ksize != 1 || (epilogue != kEpiBias && epilogue != kEpiReluMask) || (a.act_flags && !gemm_accumulate))) return HIMO_ERR_UNSUPPORTED;     // two-term bf16: 3x3 in convsp.hip, row GEMMs here
    auto blocks_for = [&](int bn, int mi) -> int64_t {
        const int bm = 64 * mi, th = 2 * mi;
        const int64_t tm = ksize == 1 ? (int64_t)a.N * (((int64_t)a.Ho * a.Wo + bm - 1) / bm)
                                      : (int64_t)a.N * ((a.Ho + th - 1) / th) * ((a.Wo + 31) / 32);
        return tm * ((a.Cout + bn - 1) / bn);
    };
    const bool can128 = a.Cout >= 128 && (a.Cout % 128) == 0;
    int bn = can128 ? 128 : 64, mi = 2;
    const int64_t want = 512;
    if (blocks_for(bn, mi) < want) mi = 1;
    if (blocks_for(bn, mi) < want && bn == 128) bn = 64;
    if (tile_hint) {                                            // caller-tuned tile: (bn << 4) | mi
        const int hb = tile_hint >> 4, hm = tile_hint & 15;
        if ((hb == 64 || (hb == 128 && can128)) && (hm == 1 || hm == 2)) { bn = hb; mi = hm; }
    }
    const dim3 grid((unsigned)blocks_for(bn, mi));
    const bool f16 = format == 1;
    const char* name = ksize == 1 ? (f16 ? "conv1x1_f16x2_kernel" : format == 2 ? "conv1x1_bf16x2_kernel" : "conv1x1_bf16x3_kernel")
                                  : (f16 ? "conv3x3_f16x2_kernel" : "conv3x3_bf16x3_kernel");
    {
        ProfScope ps(name, s);
        const unsigned short* w = (const unsigned short*)w_packed;
        if (ksize == 1) {
            if (f16) launch_bf_tile<1, 2>(a, epilogue, bn, mi, w, grid, s);
            else if (format == 2) launch_bf_tile<1, 4>(a, epilogue, bn, mi, w, grid, s);
            else launch_bf_tile<1, 3>(a, epilogue, bn, mi, w, grid, s);
        }
        else { if (f16) launch_bf_tile<3, 2>(a, epilogue, bn, mi, w, grid, s); else launch_bf_tile<3, 3>(a, epilogue, bn, mi, w, grid, s); }
    }
    HIMO_LAUNCH_CHECK("conv_bf16x3_kernel");
    return HIMO_OK;
}

}  // namespace himo

using namespace himo;

extern "C" size_t himo_conv_packed_weight_bytes(int ksize, int cin, int cout) {
    return (size_t)ksize * ksize * ((cin + 15) / 16) * 3 * (size_t)cout * 16 * 2;
}

extern "C" int himo_conv_pack_weights_ex(const float* d_w, int ksize, int cin, int cout, int format, void* d_packed, void* stream) {
    if (!d_w || !d_packed || cin < 1 || cout < 1 || !(ksize == 1 || ksize == 3) || format < 0 || format > 2) return HIMO_ERR_INVALID_ARGUMENT;
    const int T = ksize * ksize;
    const int64_t total = (int64_t)T * ((cin + 15) / 16) * cout * 16;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (format == 1) hipLaunchKernelGGL(pack_weights_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, d_w, T, cin, cout, (unsigned short*)d_packed);
    else if (format == 2) hipLaunchKernelGGL(pack_weights_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, d_w, T, cin, cout, (unsigned short*)d_packed);
    else hipLaunchKernelGGL(pack_weights_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, d_w, T, cin, cout, (unsigned short*)d_packed);
    HIMO_LAUNCH_CHECK("pack_weights_kernel");
    return HIMO_OK;
}

// All layers of a small MLP in ONE launch (FastNSF re-packs after every optimiser step): per layer W [cin][cout] float32 ->
// its fp16-split copy for the forward product (format 1) and the two-term bf16 copy of W^T for the input-gradient product
// (format 2).  blockIdx.y = layer; a thread packs one forward item and one backward item.
namespace himo {
constexpr int kMlpMaxLayers = 16;
struct MlpPackBatch {
    int n;
    const float* w[kMlpMaxLayers]; int cin[kMlpMaxLayers], cout[kMlpMaxLayers];
    unsigned short* fwd[kMlpMaxLayers]; unsigned short* bwd[kMlpMaxLayers];
};
__global__ __launch_bounds__(256) void mlp_repack_kernel(MlpPackBatch b) {
    const int L = blockIdx.y;
    const float* __restrict__ w = b.w[L];
    const int cin = b.cin[L], cout = b.cout[L];
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    {   // forward: rows = cin (slabs of 16), columns = cout; fp16 split with the weight packing scale
        const int slabs = (cin + 15) / 16;
        if (b.fwd[L] && item < (int64_t)slabs * cout * 16) {
            const int k = (int)(item % 16), co = (int)((item / 16) % cout), slab = (int)(item / (16 * (int64_t)cout));
            const int ci = slab * 16 + k;
            unsigned h, l;
            split2((ci < cin ? w[(int64_t)ci * cout + co] : 0.f) * kF16WeightScale, h, l);
            const int64_t base = ((int64_t)slab * 2) * cout * 16 + (int64_t)co * 16 + k;
            b.fwd[L][base] = (unsigned short)h; b.fwd[L][base + (int64_t)cout * 16] = (unsigned short)l;
        }
    }
    {   // backward: W^T [cout][cin]: rows = cout (slabs), columns = cin; two-term bf16
        const int slabs = (cout + 15) / 16;
        if (b.bwd[L] && item < (int64_t)slabs * cin * 16) {
            const int k = (int)(item % 16), ci = (int)((item / 16) % cin), slab = (int)(item / (16 * (int64_t)cin));
            const int co = slab * 16 + k;
            const float x = co < cout ? w[(int64_t)ci * cout + co] : 0.f;
            const unsigned h = bf16_rne_bits(x), l = bf16_rne_bits(x - bf16_bits_to_float(h));
            const int64_t base = ((int64_t)slab * 2) * cin * 16 + (int64_t)ci * 16 + k;
            b.bwd[L][base] = (unsigned short)h; b.bwd[L][base + (int64_t)cin * 16] = (unsigned short)l;
        }
    }
}
}  // namespace himo

extern "C" int himo_mlp_repack(int n_layers, const float* const* h_w, const int* h_cin, const int* h_cout, void* const* h_fwd_packed,
                               void* const* h_bwd_packed, void* stream) {
    if (n_layers < 1 || n_layers > kMlpMaxLayers || !h_w || !h_cin || !h_cout || !h_fwd_packed || !h_bwd_packed) return HIMO_ERR_INVALID_ARGUMENT;
    MlpPackBatch b{};
    b.n = n_layers;
    int64_t most = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (!h_w[i] || h_cin[i] < 1 || h_cout[i] < 1) return HIMO_ERR_INVALID_ARGUMENT;
        b.w[i] = h_w[i]; b.cin[i] = h_cin[i]; b.cout[i] = h_cout[i];
        b.fwd[i] = (unsigned short*)h_fwd_packed[i]; b.bwd[i] = (unsigned short*)h_bwd_packed[i];
        const int64_t f = (int64_t)((h_cin[i] + 15) / 16) * h_cout[i] * 16, r = (int64_t)((h_cout[i] + 15) / 16) * h_cin[i] * 16;
        most = std::max(most, std::max(f, r));
    }
    hipLaunchKernelGGL(mlp_repack_kernel, dim3((unsigned)((most + 255) / 256), n_layers), dim3(256), 0, (hipStream_t)stream, b);
    HIMO_LAUNCH_CHECK("mlp_repack_kernel");
    return HIMO_OK;
}

// Every packed weight copy a training step needs, in ONE launch (himo_weight_prepare_batch): a device-resident table of jobs, each a
// himo_conv_pack_weights_ex of one tensor, optionally of its data-gradient form (taps mirrored, cin <-> cout: what weight_flip_kernel
// + pack_weights_kernel produced in two launches per layer -- ~100 launches of 4-6 us per step before).  A block finds its job by
// bisection over the jobs' first-block numbers.
namespace himo {
__device__ __forceinline__ void pack_store(float x, int format, unsigned short* __restrict__ out, int64_t base, int64_t plane) {
    unsigned h, m = 0, l;
    if (format == 0) { split3(x, h, m, l); out[base] = (unsigned short)h; out[base + plane] = (unsigned short)m; out[base + 2 * plane] = (unsigned short)l; }
    else if (format == 2) { h = bf16_rne_bits(x); l = bf16_rne_bits(x - bf16_bits_to_float(h)); out[base] = (unsigned short)h; out[base + plane] = (unsigned short)l; }
    else { split2(x * kF16WeightScale, h, l); out[base] = (unsigned short)h; out[base + plane] = (unsigned short)l; }
}
__global__ __launch_bounds__(256) void weight_prepare_kernel(const himo_weight_job* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {                                   // last job whose first block is <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const himo_weight_job j = jobs[lo];
    const int T = j.ksize * j.ksize;
    const int Cin = j.flip ? j.cout : j.cin, Cout = j.flip ? j.cin : j.cout;       // of the packed (logical) tensor
    const int slabs = (Cin + 15) / 16;
    const int64_t item = (int64_t)((int)blockIdx.x - j.first_block) * 256 + threadIdx.x;
    if (item >= (int64_t)T * slabs * Cout * 16) return;
    const int k = (int)(item % 16);
    const int co = (int)((item / 16) % Cout);
    const int slab = (int)((item / (16 * (int64_t)Cout)) % slabs);
    const int tap = (int)(item / (16 * (int64_t)Cout * slabs));
    const int ci = slab * 16 + k;
    float x = 0.f;
    if (ci < Cin) x = j.flip ? j.w[((int64_t)(T - 1 - tap) * j.cin + co) * j.cout + ci]          // wf[tap][ci][co] = w[T-1-tap][co][ci]
                             : j.w[((int64_t)tap * Cin + ci) * Cout + co];
    const int NP = j.format == 0 ? 3 : 2;
    pack_store(x, j.format, (unsigned short*)j.packed, (((int64_t)tap * slabs + slab) * NP) * Cout * 16 + (int64_t)co * 16 + k, (int64_t)Cout * 16);
}
}  // namespace himo

extern "C" int himo_weight_job_blocks(int ksize, int cin, int cout, int flip) {
    if (cin < 1 || cout < 1 || !(ksize == 1 || ksize == 3)) return -1;
    const int64_t items = (int64_t)ksize * ksize * (((flip ? cout : cin) + 15) / 16) * (flip ? cin : cout) * 16;
    return (int)((items + 255) / 256);
}

extern "C" int himo_weight_prepare_batch(const himo_weight_job* d_jobs, int n_jobs, int total_blocks, void* stream) {
    if (!d_jobs || n_jobs < 1 || total_blocks < 1) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(weight_prepare_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, d_jobs, n_jobs);
    HIMO_LAUNCH_CHECK("weight_prepare_kernel");
    return HIMO_OK;
}

extern "C" int himo_conv_pack_weights(const float* d_w, int ksize, int cin, int cout, void* d_packed, void* stream) {
    return himo_conv_pack_weights_ex(d_w, ksize, cin, cout, 0, d_packed, stream);
}
