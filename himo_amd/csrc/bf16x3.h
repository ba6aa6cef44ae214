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

}  // namespace himo
