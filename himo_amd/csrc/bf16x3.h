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


// ---- two-term fp16 split with a scaled low part:  x = h + 2^-11 * l', h and l' both NORMAL fp16 numbers -------------
// 11 + 11 mantissa bits (relative representation error 2^-22); the product keeps three terms,
//     acc0 += ha*hb ;  acc1 += ha*lb' + la'*hb ;  result = acc0 + 2^-11 * acc1       (dropped: 2^-22 la' lb')
// i.e. THREE fp16 matrix instructions per float32 product block instead of six bf16 ones.  The 2^11 scale keeps the low
// parts out of the fp16 subnormal range (weights ~0.05 would otherwise lose most of their low part's bits).  An h that
// is itself subnormal (|x| < 2^-14) needs no special case: v_cvt_f16_f32 and v_mfma_f32_32x32x16_f16 both keep
// subnormals on gfx950 (measured: a 2^-20 input comes through the matrix instruction exactly).  Range: |x| must stay
// below 65504 (fp16 max) -- true for normalised activations; the bf16 split has float32's range and remains available.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// kF16Scaled = false (default): the low part is stored UNSCALED (l = x - h).  It is then an fp16 subnormal for |x| < 1/4
// -- kept exactly by the conversion and by the matrix instruction on gfx950 -- so a value is represented to
// max(2^-25 absolute, 2^-23 relative) and all three products share one scale and ONE accumulator: half the accumulator
// registers (one more wave per SIMD for the 4-row tiles) and no epilogue combine.  -DHIMO_F16_SCALED restores the 2^11
// scaling with its separate cross-term accumulator (22-bit relative precision at every magnitude).
// The WEIGHTS (typically ~0.05, whose low parts would be deep in the subnormal range) are packed multiplied by 2^6 --
// exact, |w| < 1023 -- and the accumulator is multiplied by 2^-6 in the epilogue: their low parts become normal fp16
// numbers again, which is where the unscaled form lost its precision (measured flow error 5.3e-5 -> back to ~3e-5).
#ifdef HIMO_F16_SCALED
constexpr bool kF16Scaled = true;
constexpr float kF16LowScale = 2048.0f, kF16LowInv = 1.0f / 2048.0f;
constexpr float kF16WeightScale = 1.0f, kF16AccScale = 1.0f;
#else
constexpr bool kF16Scaled = false;
constexpr float kF16LowScale = 1.0f, kF16LowInv = 1.0f;
constexpr float kF16WeightScale = 64.0f, kF16AccScale = 1.0f / 64.0f;
#endif

__device__ inline void split2(float x, unsigned& h, unsigned& l) {
    const _Float16 hh = (_Float16)x;                            // round to nearest even
    const _Float16 ll = (_Float16)((x - (float)hh) * kF16LowScale);
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
}

}  // namespace himo
