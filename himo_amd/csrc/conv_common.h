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
};
enum ActFlags { kActSplitIn = 1, kActSplitOut = 2 };

// GELU (erf form).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below float32 resolution of the 1 + erf sum)
// with the hardware exp2 / rcp: a dozen instructions where the library erff takes three times that -- the epilogue of
// a 64-channel 3x3 layer is otherwise as long as a third of its matrix work.
// two-level image addressing: n_inner frames of a sample (channel groups or planes), n_outer samples of a batch
__device__ inline int64_t image_offset(int img, int n_inner, int64_t batch_stride, int64_t outer_stride) {
    return (int64_t)(img % n_inner) * batch_stride + (int64_t)(img / n_inner) * outer_stride;
}

__device__ inline float gelu_exact(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-x * x);            // erf(|v| / sqrt 2)
    return v * 0.5f * (1.0f + copysignf(e, v));
}
// GRU gates: hardware exp2 / rcp (~1 ulp each); the gate outputs are O(1) and feed a 1e-4 abs budget
__device__ inline float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ inline float tanh_f(float v) {
    const float e = __expf(-2.0f * fabsf(v));            // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, v);
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
template <int EPI, bool PAIRED>
__device__ inline void split_store(const ConvArgs& a, float* __restrict__ yout, int64_t pix, int co, float v, float sc, float sh) {
    if (EPI == kEpiBiasBnGelu) v = gelu_exact(v * sc + sh);
    else if (EPI == kEpiBiasGelu) v = gelu_exact(v);
    else if (EPI == kEpiBiasRelu) v = fmaxf(v, 0.f);
    unsigned h, l;
    split2_rounded(v, h, l);
    if (PAIRED) {
        const bool odd = co & 1;
        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? h : l), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        unsigned* rec = reinterpret_cast<unsigned*>(yout + pix * a.y_pitch + (co & ~15));
        rec[(odd ? 8 : 0) + ((co & 15) >> 1)] = odd ? (recv | (l << 16)) : (h | (recv << 16));
    } else {
        unsigned short* rec = reinterpret_cast<unsigned short*>(yout + pix * a.y_pitch + (co & ~15));
        rec[co & 15] = (unsigned short)h;
        rec[16 + (co & 15)] = (unsigned short)l;
    }
}

// implemented in convbf.hip: stride-1 convolutions / row GEMMs on split-bf16 matrix instructions
int launch_conv_bf16x3(const ConvArgs& a, int ksize, int epilogue, const void* w_packed, int tile_hint, int format, int stride, hipStream_t s);

}  // namespace himo
