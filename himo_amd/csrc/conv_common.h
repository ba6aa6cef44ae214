// conv_common.h -- argument block, epilogue kinds and the per-element fused epilogue shared by the float32-MFMA
// convolution (conv.hip) and the split-bf16 convolution (convbf.hip).
#pragma once
#include "himo_common.h"
#include "bf16x3.h"
#include <math.h>

namespace himo {

typedef float floatx16 __attribute__((ext_vector_type(16)));

enum Epilogue {
    kEpiBias = 0,         // y = acc + bias
    kEpiBiasBnGelu = 1,   // y = gelu((acc + bias) * scale + shift)
    kEpiBiasGelu = 2,     // y = gelu(acc + bias)
    kEpiGruZR = 3,        // cols [0,C/2): z = sigmoid(.) -> out ; cols [C/2,C): r = sigmoid(.), aux_out = r * h
    kEpiGruQ = 4,         // q = tanh(.) ; h = (1 - z) * h + z * q  (in place in aux_out)
    kEpiBiasRelu = 5,     // y = max(acc + bias, 0)                      (coordinate-MLP forward, FastNSF)
    kEpiReluMask = 6      // y = aux_in > 0 ? acc : 0                    (its backward: dX = (dZ W^T) * relu'(X))
};

struct ConvArgs {
    const float* x; int64_t x_batch_stride; int x_pitch;      // input  [n][H][W] pixels, `x_pitch` floats apart
    const float* w;                                           // [KS][KS][Cin][Cout]
    const float* bias; const float* scale; const float* shift;
    float* y; int64_t y_batch_stride; int y_pitch;            // output
    int N, H, W, Cin, Cout;                                   // N = n_inner * n_outer images; input spatial size (KS=1 rows: H = 1, W = rows)
    int n_inner; int64_t x_outer_stride, y_outer_stride;      // image i sits at (i % n_inner) * batch_stride + (i / n_inner) * outer_stride
    int Ho, Wo;
    // GRU epilogues
    const float* aux_in; int aux_in_pitch;                    // z (kEpiGruQ) / h (kEpiGruZR), [rows][pitch]
    float* aux_out; int aux_out_pitch;                        // r*h (kEpiGruZR) / h in place (kEpiGruQ)
    int act_flags;                                            // kActSplitIn | kActSplitOut: split activation format (convsg.hip)
    unsigned* range_seen;                                     // himo_conv_desc.d_range_seen (split outputs; may be null)
};
enum ActFlags { kActSplitIn = 1, kActSplitOut = 2, kActVecStore = 4, kActAccumulate = 8, kActStuffedIn = 16 };
// kActStuffedIn (HIMO_ACT_STUFFED_2X): x is a compact [H / 2][W / 2] map read as its zero-stuffed x2 image (two-term bf16 3x3 kernel)
// kActVecStore: set by the launchers (vec_store_ok).  kActAccumulate (HIMO_ACT_ACCUMULATE): y += result -- float32 output of the
// two-term bf16 3x3 kernels only (the training step's stride-2 data gradients add into the decoder's skip gradient in place)

// GELU (erf form).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below float32 resolution of the 1 + erf sum)
// with the hardware exp2 / rcp: a dozen instructions where the library erff takes three times that -- the epilogue of
// a 64-channel 3x3 layer is otherwise as long as a third of its matrix work.
// two-level image addressing: n_inner frames of a sample (channel groups or planes), n_outer samples of a batch
__device__ inline int64_t image_offset(int img, int n_inner, int64_t batch_stride, int64_t outer_stride) {
    return (int64_t)(img % n_inner) * batch_stride + (int64_t)(img / n_inner) * outer_stride;
}

// XCD-aware block order.  The dispatcher is observed to place block b on XCD b % 8 (each XCD has its own 4 MB L2), so
// consecutive block ids -- vertically / horizontally adjacent tiles that share halo rows, and the channel tiles of one
// pixel tile that share the whole patch -- land on eight different L2s and every one of them fetches the shared rows from
// HBM again.  This maps the hardware id to a logical id such that XCD x works through the CONTIGUOUS range
// [x * n / 8, (x + 1) * n / 8) of the logical (image, tile row, tile column, channel tile) order: neighbours in that order
// run on the same XCD close in time and meet in its L2.  Bijective for any grid size; a different placement would change
// speed only.
__device__ inline int xcd_block_id(int bid, int n_blocks) {
    const int per = n_blocks >> 3, rem = n_blocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// GELU through erf(|v| / sqrt 2) = 1 - p(t) t exp(-v^2 / 2), t = 1 / (1 + 0.3275911 |v| / sqrt 2) (Abramowitz & Stegun 7.1.26, 1.5e-7
// absolute).  Arranged for instruction count -- the convolutions' epilogues run beside other waves' matrix instructions, where
// a SIMD issues only ~5 vector instructions per matrix instruction (scripts/micro/mfma_filler_cost.hip), so every one counts:
// 11 vector + 2 transcendental instructions (17 + 2 in the textbook arrangement): the constants of |v| / sqrt 2 are folded into
// the polynomial argument and the exponent, and v / 2 (1 + sign(v) erf) = v / 2 + |v / 2| erf is ONE fused multiply-add.
__device__ inline float gelu_exact(float v) {
    const float hv = 0.5f * v;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(v), 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float ex = __builtin_amdgcn_exp2f((v * v) * (-0.5f * 1.44269504088896340736f));      // exp(-v^2 / 2)
    const float e = fmaf(-(p * t), ex, 1.0f);                 // erf(|v| / sqrt 2)
    return fmaf(fabsf(hv), e, hv);
}
// d gelu / dv = Phi(v) + v phi(v) in the same arrangement: the erf polynomial and ONE exponential (exp(-v^2 / 2) is both the tail of
// the erf form and the density) -- 16 vector + 2 transcendental instructions where erff + expf from the library take ~80; the
// BatchNorm backward pass evaluates it once per activation and was bound by exactly that, not by its three HBM streams
__device__ inline float gelu_grad_exact(float v) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(v), 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float ex = __builtin_amdgcn_exp2f((v * v) * (-0.5f * 1.44269504088896340736f));      // exp(-v^2 / 2)
    const float e = fmaf(-(p * t), ex, 1.0f);                 // erf(|v| / sqrt 2)
    return fmaf(v * 0.39894228040143267794f, ex, 0.5f + copysignf(0.5f * e, v));
}
// GRU gates: hardware exp2 / rcp (~1 ulp each); the gate outputs are O(1) and feed a 1e-4 abs budget
__device__ inline float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ inline float tanh_f(float v) {
    const float e = __expf(-2.0f * fabsf(v));            // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, v);
}


// ---- the two epilogues of the backbone (bias; bias + BatchNorm + GELU) with the fewest vector instructions ------------------------
// value = fma(acc, A, B) with per-channel constants A = k (* scale), B = bias (* scale + shift): the accumulator's packing scale k
// (a power of two, so A is exact), the bias and the folded BatchNorm in ONE instruction instead of mul + add + mul + add.
template <int EPI> constexpr bool kEpiAffine = EPI == kEpiBias || EPI == kEpiBiasBnGelu || EPI == kEpiBiasGelu;
template <int EPI>
__device__ inline void epi_affine(float k, float bias, float sc, float sh, float& A, float& B) {
    if (EPI == kEpiBiasBnGelu) { A = k * sc; B = fmaf(bias, sc, sh); } else { A = k; B = bias; }
}
template <int EPI>
__device__ inline float epi_activate(float acc, float A, float B) {
    const float v = fmaf(acc, A, B);
    return EPI == kEpiBias ? v : gelu_exact(v);
}

// one output element: bias -> (BatchNorm scale/shift) -> activation / GRU gate math -> store
template <int EPI>
__device__ inline void epilogue_store(const ConvArgs& a, float* __restrict__ yout, int64_t pix, int co, float v, float sc, float sh) {
    if (EPI == kEpiBias) {
        yout[pix * a.y_pitch + co] = v;
    } else if (EPI == kEpiBiasBnGelu) {
        v = v * sc + sh;
        yout[pix * a.y_pitch + co] = gelu_exact(v);
    } else if (EPI == kEpiBiasGelu) {
        yout[pix * a.y_pitch + co] = gelu_exact(v);
    } else if (EPI == kEpiGruZR) {
        const int half = a.Cout / 2;
        const float g = sigmoid_f(v);
        if (co < half) yout[pix * a.y_pitch + co] = g;                               // z
        else a.aux_out[pix * a.aux_out_pitch + (co - half)] = g * a.aux_in[pix * a.aux_in_pitch + (co - half)];   // r * h
    } else if (EPI == kEpiBiasRelu) {
        yout[pix * a.y_pitch + co] = fmaxf(v, 0.f);
    } else if (EPI == kEpiReluMask) {
        yout[pix * a.y_pitch + co] = a.aux_in[pix * a.aux_in_pitch + co] > 0.f ? v : 0.f;
    } else if (EPI == kEpiGruQ) {
        const float q = tanh_f(v);
        const float z = a.aux_in[pix * a.aux_in_pitch + co];
        const float h = a.aux_out[pix * a.aux_out_pitch + co];
        a.aux_out[pix * a.aux_out_pitch + co] = (1.0f - z) * h + z * q;
    }
}

// one output element in the split activation format (convsg.hip): activation, then x = h + l as two fp16 into the
// pixel's 64-byte record of the 16-channel group [h0..h15 | l0..l15].
// PAIRED (accumulator layouts where lane parity = channel parity and both lanes of a pair hold the same pixel): the even
// lane takes its neighbour's high part, the odd lane its neighbour's low part, and each stores ONE 32-bit word -- a
// wave's 32 channels of a pixel leave as one full 128-byte line per store instruction, as the float32 epilogue does.
__device__ inline unsigned split_word(float v, bool odd);
template <int EPI, bool PAIRED>
__device__ inline void split_store(const ConvArgs& a, float* __restrict__ yout, int64_t pix, int co, float v, float sc, float sh) {
    if (EPI == kEpiBiasBnGelu) v = gelu_exact(v * sc + sh);
    else if (EPI == kEpiBiasGelu) v = gelu_exact(v);
    else if (EPI == kEpiBiasRelu) v = fmaxf(v, 0.f);
    if (PAIRED) {
        const bool odd = co & 1;
        unsigned* rec = reinterpret_cast<unsigned*>(yout + pix * a.y_pitch + (co & ~15));
        rec[(odd ? 8 : 0) + ((co & 15) >> 1)] = split_word(v, odd);
        return;
    }
    unsigned h, l;
    split2_rounded(v, h, l);
    {
        unsigned short* rec = reinterpret_cast<unsigned short*>(yout + pix * a.y_pitch + (co & ~15));
        rec[co & 15] = (unsigned short)h;
        rec[16 + (co & 15)] = (unsigned short)l;
    }
}

// Low-side guard of the two-term fp16 split (bf16x3.h: below |x| = 1/4 the low part is an fp16 subnormal, a value keeps 2^-25
// ABSOLUTE): a split-output epilogue hands ONE of its values per lane here; the layer's word is set as soon as one of them
// reaches 2^-6 (where the split still carries 19 bits).  A layer whose word stays 0 lives wholly on the absolute floor -- e.g.
// a BatchNorm gamma of 1e-3 compensated by large weights one layer on multiplies that floor straight into the flow (measured:
// 1.3e-4 at gain 1e-3, tests/test_parity_hardening_gpu.py).  Cost: a compare, a branch and -- until the word is set -- a store.
constexpr float kSplitRangeGuard = 0.015625f;
__device__ inline void note_range(const ConvArgs& a, float v) {
    if (a.range_seen && __ballot(fabsf(v) >= kSplitRangeGuard) != 0ull && (threadIdx.x & 63) == 0 &&
        __hip_atomic_load(a.range_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        __hip_atomic_store(a.range_seen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the output admits 16-byte stores: base, pitches and strides multiples of four floats, whole 4-channel groups
inline bool vec_store_ok(const ConvArgs& a) {
    return (reinterpret_cast<uintptr_t>(a.y) & 15u) == 0 && !(a.y_pitch & 3) && !(a.y_batch_stride & 3) && !(a.y_outer_stride & 3) && !(a.Cout & 3) &&
           (int64_t)a.Ho * a.Wo * a.y_pitch < ((int64_t)1 << 29);            // 32-bit byte offsets inside an output image
}

// Epilogue of one accumulator block (v_mfma 32x32 layout: lane (li, lh) holds output channel ch0 + li of the 16 pixels
// (r & 3) + 8 (r >> 2) + 4 lh, r = 0..15, of a 32-pixel row segment) as SIXTEEN-BYTE stores.  Written straight from the
// accumulator layout a wave's tile leaves as 16 dword store instructions per block (64 for a 4-block tile), each covering
// two 128-byte lines: measured, those stores -- not the activation / split arithmetic -- were the exposed part of the
// epilogue (kernels 13 % faster with the stores removed, 3 % with the arithmetic removed).  Here the block's 32 pixels x
// 128 bytes are transposed through a wave-private 4 KB LDS area -- 16 ds_write_b32, 4 ds_read_b128 -- and leave as 4
// dwordx4 stores per lane, every instruction writing eight full 128-byte lines.  Pixel p sits in slot p ^ ((p >> 2) & 1):
// the two half-waves write pixels 4 apart (512 bytes: the same banks) in the same instruction, the swap moves one of them
// 128 bytes on.  DS operations of a wave execute in order, so no barrier is needed between its writes and reads.
// word = this lane's 32-bit output for pixel r (float bits, or the paired split word of split_word()).
// MASK (float32 outputs only): the kEpiReluMask epilogue -- an output is kept where aux_in (same pixel, same channel) is positive,
// read with the same 16-byte pattern as the store.
template <bool OSPLIT, bool MASK = false, bool ACC = false>
__device__ inline void store_block_vec(const ConvArgs& a, float* __restrict__ yout, unsigned char* stg, const unsigned (&word)[16],
                                       int lane, int64_t pix0, int n_valid_px, int ch0) {
    const int li = lane & 31, lh = lane >> 5;
    const int widx = OSPLIT ? (li >> 4) * 16 + ((li & 1) ? 8 : 0) + ((li & 15) >> 1) : li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int px = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int slot = px ^ ((px >> 2) & 1);
        *reinterpret_cast<unsigned*>(stg + slot * 128 + widx * 4) = word[r];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = t * 64 + lane;
        const int slot = q >> 3, piece = q & 7;
        const int px = slot ^ ((slot >> 2) & 1);
        uint4 d = *reinterpret_cast<const uint4*>(stg + q * 16);
        const int ch = ch0 + (OSPLIT ? (piece >> 2) * 16 : piece * 4);
        if (px < n_valid_px && ch < a.Cout) {
            if (MASK) {
                const float4 m = *reinterpret_cast<const float4*>(a.aux_in + (pix0 + px) * (int64_t)a.aux_in_pitch + ch0 + piece * 4);
                d.x = m.x > 0.f ? d.x : 0u; d.y = m.y > 0.f ? d.y : 0u; d.z = m.z > 0.f ? d.z : 0u; d.w = m.w > 0.f ? d.w : 0u;
            }
            // 32-bit offset from the (uniform) image base: vec_store_ok() admits only images below 2 GB, and a 64-bit multiply-add per
            // store was ~8 vector instructions of the block's ~60
            float* dst = yout + (((unsigned)pix0 + (unsigned)px) * (unsigned)a.y_pitch + (unsigned)(ch0 + piece * 4));
            if (ACC) {                                     // y += result (float32 words)
                const float4 o = *reinterpret_cast<const float4*>(dst);
                d.x = __builtin_bit_cast(unsigned, o.x + __builtin_bit_cast(float, d.x)); d.y = __builtin_bit_cast(unsigned, o.y + __builtin_bit_cast(float, d.y));
                d.z = __builtin_bit_cast(unsigned, o.z + __builtin_bit_cast(float, d.z)); d.w = __builtin_bit_cast(unsigned, o.w + __builtin_bit_cast(float, d.w));
            }
            *reinterpret_cast<uint4*>(dst) = d;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// activation of one epilogue value (the two epilogues the split-input kernels support)
template <int EPI>
__device__ inline float epilogue_value(float v, float sc, float sh) {
    if (EPI == kEpiBiasBnGelu) return gelu_exact(v * sc + sh);
    if (EPI == kEpiBiasGelu) return gelu_exact(v);
    if (EPI == kEpiBiasRelu) return fmaxf(v, 0.f);
    return v;
}
// split x = h + l and pair up with the neighbouring lane (channel parity = lane parity, same pixel): the even lane ends up
// with (h_even, h_odd), the odd lane with (l_even, l_odd) -- the 32-bit words of the split record (see split_store).
// Both lanes form p = h | l << 16, fetch the neighbour's p by DPP and pick their two half-words with one byte permute
// (selector per lane parity): pack + mov_dpp + perm instead of two selects, a shift-or pair and the DPP move.
__device__ inline unsigned split_word(float v, bool odd) {
    const unsigned p = split2_packed(v);
    const unsigned q = (unsigned)__builtin_amdgcn_mov_dpp((int)p, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    // v_perm_b32: selector bytes 0-3 pick from the SECOND operand (p), 4-7 from the first (q)
    return __builtin_amdgcn_perm(q, p, odd ? 0x03020706u : 0x05040100u);
}

// implemented in convbf.hip: stride-1 convolutions / row GEMMs on split-bf16 matrix instructions
int launch_conv_bf16x3(const ConvArgs& a, int ksize, int epilogue, const void* w_packed, int tile_hint, int format, int stride, hipStream_t s);

}  // namespace himo
