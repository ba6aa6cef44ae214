// lanetranspose.h -- an 8 lanes x 8 values block of 16-bit words transposed across the lanes of a wave, in registers.
//
// Accumulator-layout registers hold one COLUMN per lane; a row-major matrix operand in LDS wants a row's eight consecutive columns
// as one 16-byte word.  Written element by element that is eight ds_write_b16 per lane -- and an LDS store costs its instruction,
// not its bytes (csrc/nsffused.hip backward: 64 of them per lane and step were 100 of 358 us).  Transposed first, lane q of each
// group of eight holds row q's eight columns: ONE ds_write_b128.  xor-1 at half-word granularity (DPP quad_perm + v_perm), xor-2
// and xor-4 at dword granularity (DPP / ds_swizzle + selects).
#pragma once
#include <hip/hip_runtime.h>

namespace himo {

__device__ __forceinline__ unsigned lane_sel(bool c, unsigned a, unsigned b) { return c ? a : b; }
__device__ __forceinline__ uint4 lane_transpose8(const uint4& v, int lane) {
    const bool e = lane & 1, f = lane & 2, g = lane & 4;
    unsigned d[4] = {v.x, v.y, v.z, v.w};
    const unsigned selA = e ? 0x03020706u : 0x05040100u;        // even: (own.lo, partner.lo); odd: (partner.hi, own.hi)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const unsigned pd = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d[m], 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]: lane ^ 1
        d[m] = __builtin_amdgcn_perm(pd, d[m], selA);
    }
#pragma unroll
    for (int m = 0; m < 4; m += 2) {                            // lane ^ 2: 2 x 2 blocks of the dword matrix
        const unsigned send = lane_sel(f, d[m], d[m + 1]);
        const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        d[m] = lane_sel(f, recv, d[m]);
        d[m + 1] = lane_sel(f, d[m + 1], recv);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {                               // lane ^ 4
        const unsigned send = lane_sel(g, d[m], d[m + 2]);
        const unsigned recv = (unsigned)__builtin_amdgcn_ds_swizzle((int)send, 0x101F);                     // bit mode: and 0x1f, or 0, xor 4
        d[m] = lane_sel(g, recv, d[m]);
        d[m + 2] = lane_sel(g, d[m + 2], recv);
    }
    return uint4{d[0], d[1], d[2], d[3]};
}

}  // namespace himo
