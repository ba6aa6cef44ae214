// bf16x3.h -- the three-term bf16 split of a float32 (x = h + m + l, 8 + 8 + 8 mantissa bits) used by the split-bf16
// matrix kernels (convbf.hip, gruhead.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace himo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned bf16_rne_bits(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;        // round to nearest even (inputs are finite)
}
__device__ inline float bf16_bits_to_float(unsigned b) { return __builtin_bit_cast(float, b << 16); }

// x -> (h, m, l) with x == h + m + l up to 2^-24 |x|
__device__ inline void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rne_bits(x);
    const float r1 = x - bf16_bits_to_float(h);
    m = bf16_rne_bits(r1);
    const float r2 = r1 - bf16_bits_to_float(m);
    l = bf16_rne_bits(r2);
}


// ---- two-term fp16 split:  x = h + l,  h = fp16(x),  l = fp16(x - h) -----------------------------------------------
// The product keeps three terms, ha*hb + ha*lb + la*hb (dropped: la*lb, 2^-22 relative): THREE fp16 matrix instructions
// per float32 product block instead of six bf16 ones -- all into ONE accumulator, because the low part is stored
// UNSCALED.  That is only sound because gfx950 keeps fp16 subnormals exactly, in v_cvt_f16_f32 and in
// v_mfma_f32_32x32x16_f16 (measured: a 2^-20 operand comes through the matrix instruction exactly): l is subnormal for
// |x| < 1/4, so a value is represented to max(2^-25 absolute, 2^-23 relative).  For the WEIGHTS (typically ~0.05, whose
// low parts would sit deep in the subnormal range) that absolute floor would cost precision, so they are packed
// multiplied by 2^6 -- exact, |w| < 1023 -- and the accumulator is multiplied by 2^-6 in the epilogue (measured flow error
// without the weight scale 5.3e-5, with it 3.1e-5 = the float32-MFMA kernels' own 3.05e-5).
// One accumulator instead of two (the form of rounds 1-2: l' = 2^11 (x - h), always a normal number, cross terms in a second
// accumulator; removed in round 5) halves the accumulator registers: the 4-row convolution tiles and the fused head run
// three waves per SIMD instead of two (+13 % on the 128/256-channel layers).
// Range: |x| must stay below 65504 (fp16 max) -- true for normalised activations; the bf16 split has float32's range -- and
// a layer whose values ALL sit below ~2^-6 has lost the relative precision (the absolute floor above): both ends are watched
// at run time (the head's finite-flow word, the split-output epilogues' range word: conv_common.h note_range) and
// pipeline.HiMoPipeline(precision="auto") leaves for the bf16 split when either fires.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kF16WeightScale = 64.0f, kF16AccScale = 1.0f / 64.0f;

__device__ inline void split2(float x, unsigned& h, unsigned& l) {
    const _Float16 hh = (_Float16)x;                            // round to nearest even
    const _Float16 ll = (_Float16)(x - (float)hh);
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
}

// split of a float32 value that was just COMPUTED (an epilogue result): the empty asm keeps its float32 rounding.  Without it
// the compiler folds the producing multiply / add into v_fma_mixlo_f16 -- ONE rounding to fp16 -- and ties then split
// differently from the same float32 value split by a consumer that loaded it from memory.
__device__ inline void split2_rounded(float x, unsigned& h, unsigned& l) {
    asm("" : "+v"(x));
    split2(x, h, l);
}

// the same split as ONE 32-bit word h | l << 16 (epilogues): v_cvt_f16_f32 for h; x - h is exact in float32, so the low part
// is a single rounding however the compiler forms it (cvt / sub / cvt, or one v_fma_mix); the empty asm pins x's own rounding
typedef _Float16 himo_half2 __attribute__((ext_vector_type(2)));
__device__ inline unsigned split2_packed(float x) {
    asm("" : "+v"(x));
    const _Float16 hh = (_Float16)x;
    const _Float16 ll = (_Float16)(x - (float)hh);
    return __builtin_bit_cast(unsigned, himo_half2{hh, ll});
}

}  // namespace himo
