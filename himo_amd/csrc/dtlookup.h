// dtlookup.h -- the lookup rule of the distance-transform objective (stage a12, csrc/dtloss.hip builds the volume), shared by the
// stand-alone loss kernel and the fused FastNSF forward (csrc/nsffused.hip).  PARITY UNPINNED: this build's own specification
// (himo_amd/fastnsf.py); oracle: oracle/fastnsf_oracle.py dt_lookup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace himo {

constexpr unsigned short kDtInf = 0xFFFFu;

struct DtGrid {
    float ox, oy, oz, cell;
    int nx, ny, nz, window;
};

__device__ inline float dt_value(const unsigned short* __restrict__ vol, const DtGrid& g, int ix, int iy, int iz) {
    const unsigned short v = vol[((int64_t)iz * g.ny + iy) * g.nx + ix];
    const float d = v == kDtInf ? (float)g.window : sqrtf((float)v);
    return fminf(d, (float)g.window) * g.cell;
}

// Trilinear interpolation of the volume at p.  Returns whether p is IN THE VOLUME (cell-centre coordinate strictly inside
// (0, n - 1) on every axis; NaN is outside); D and gr = d D / d p (per metre; zero unless in the volume and D <= trunc) are
// only meaningful then.
__device__ inline bool dt_lookup(const float (&p)[3], const DtGrid& g, const unsigned short* __restrict__ vol, float trunc, float& D,
                                 float (&gr)[3]) {
    const int dims[3] = {g.nx, g.ny, g.nz};
    const float org[3] = {g.ox, g.oy, g.oz};
    int i0[3]; float f[3];
    bool inside = true;
    gr[0] = gr[1] = gr[2] = 0.f;
    D = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float u = (p[c] - org[c]) / g.cell - 0.5f;
        const float hi = (float)(dims[c] - 1);
        inside = inside && (u > 0.f && u < hi);
        int b = (int)floorf(u);
        if (b > dims[c] - 2) b = dims[c] - 2;
        if (b < 0) b = 0;
        i0[c] = b; f[c] = u - (float)b;
    }
    if (!inside) return false;
    const int x0 = i0[0], y0 = i0[1], z0 = i0[2], x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;      // in the volume: all eight cells exist
    const float d000 = dt_value(vol, g, x0, y0, z0), d100 = dt_value(vol, g, x1, y0, z0);
    const float d010 = dt_value(vol, g, x0, y1, z0), d110 = dt_value(vol, g, x1, y1, z0);
    const float d001 = dt_value(vol, g, x0, y0, z1), d101 = dt_value(vol, g, x1, y0, z1);
    const float d011 = dt_value(vol, g, x0, y1, z1), d111 = dt_value(vol, g, x1, y1, z1);
    const float fx = f[0], fy = f[1], fz = f[2], gx = 1.f - fx, gy = 1.f - fy, gz = 1.f - fz;
    const float c00 = d000 * gx + d100 * fx, c10 = d010 * gx + d110 * fx, c01 = d001 * gx + d101 * fx, c11 = d011 * gx + d111 * fx;
    const float c0 = c00 * gy + c10 * fy, c1 = c01 * gy + c11 * fy;
    D = c0 * gz + c1 * fz;
    if (D <= trunc) {
        const float s = 1.0f / g.cell;
        gr[0] = (((d100 - d000) * gy + (d110 - d010) * fy) * gz + ((d101 - d001) * gy + (d111 - d011) * fy) * fz) * s;
        gr[1] = ((c10 - c00) * gz + (c11 - c01) * fz) * s;
        gr[2] = (c1 - c0) * s;
    }
    return true;
}

}  // namespace himo
