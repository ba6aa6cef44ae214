// dtloss.hip -- stage a12, the distance-transform objective that makes FastNSF "fast" (BASELINE config 4, README.md:53
// `model=fastnsf`): instead of an exact nearest-neighbour search per optimiser iteration, the target sweep pc1 is turned ONCE per
// pair into a volume of distances to its nearest occupied voxel, and an iteration's loss is a trilinear lookup into that volume.
//
// PARITY UNPINNED (the reference's implementation is in the absent OpenSceneFlow submodule; only the method's name is in the tree).
// This build's specification (himo_amd/fastnsf.py, oracle/fastnsf_oracle.py dt_*):
//   volume    nx x ny x nz cells of edge `cell` from `origin`, x fastest; a cell is occupied when a pc1 point falls into it;
//             G[c] = squared distance IN CELLS from cell c to the nearest occupied cell (uint16; 0xFFFF = further than the window
//             W in some axis), D[c] = min(sqrt(G[c]), W) * cell -- exact Euclidean distance transform up to W cells;
//   lookup    u = (p - origin) / cell - 1/2 (cell-centre coordinates), clamped to [0, n - 1] per axis; D(p) = trilinear interpolation
//             of the eight surrounding D values; loss = (1 / n) sum_i [D(p_i) <= trunc] D(p_i); gradient = its exact derivative
//             (zero along an axis where the point is clamped).
// The transform is separable: min over (dx, dy, dz) of dx^2 + dy^2 + dz^2 = min_dz (dz^2 + min_dy (dy^2 + min_dx dx^2)), each
// pass a windowed min-plus through an LDS tile (the volume is 113 M cells for the 106 x 106 x 10 m box at 0.1 m: three passes of
// ~0.25 GB read + written each, milliseconds once per pair against ~0.3 ms of NN searches EVERY iteration).
// HBM-bound streams; integer arithmetic; bit-deterministic.
#include "himo_common.h"
#include <math.h>

namespace himo {

constexpr unsigned short kDtInf = 0xFFFFu;
constexpr int kDtMaxWindow = 40;

struct DtGrid {
    float ox, oy, oz, cell;
    int nx, ny, nz, window;
};

__device__ inline unsigned dt_add(unsigned short g, int d2) {           // saturating: INF stays INF
    return g == kDtInf ? 0xFFFFFFFFu : (unsigned)g + (unsigned)d2;
}

__global__ __launch_bounds__(256) void dt_mark_kernel(int n, const float* __restrict__ pts, DtGrid g, unsigned short* __restrict__ vol) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float fx = floorf((pts[i * 3] - g.ox) / g.cell), fy = floorf((pts[i * 3 + 1] - g.oy) / g.cell), fz = floorf((pts[i * 3 + 2] - g.oz) / g.cell);
    if (!(fx >= 0.f && fx < (float)g.nx && fy >= 0.f && fy < (float)g.ny && fz >= 0.f && fz < (float)g.nz)) return;     // also drops NaN
    vol[((int64_t)(int)fz * g.ny + (int)fy) * g.nx + (int)fx] = 0;
}

// pass along x: out[x] = min over |dx| <= W of (in[x + dx] == 0 ? dx^2 : INF).  One block = 256 consecutive cells of one row.
__global__ __launch_bounds__(256) void dt_pass_x_kernel(DtGrid g, const unsigned short* __restrict__ in, unsigned short* __restrict__ out) {
    __shared__ unsigned short tile[256 + 2 * kDtMaxWindow];
    const int W = g.window;
    const int x_blocks = (g.nx + 255) / 256;
    const int64_t row = blockIdx.x / x_blocks;                         // z * ny + y
    const int x0 = (int)(blockIdx.x % x_blocks) * 256;
    const unsigned short* src = in + row * g.nx;
    for (int t = threadIdx.x; t < 256 + 2 * W; t += 256) {
        const int x = x0 - W + t;
        tile[t] = (x >= 0 && x < g.nx) ? src[x] : kDtInf;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= g.nx) return;
    unsigned best = 0xFFFFFFFFu;
    for (int d = -W; d <= W; ++d)
        if (tile[threadIdx.x + W + d] == 0) { const unsigned v = (unsigned)(d * d); best = v < best ? v : best; }
    out[row * g.nx + x] = best >= kDtInf ? kDtInf : (unsigned short)best;
}

// pass along an axis of stride `stride` cells and length `len` (y: stride nx; z: stride nx * ny): out[j] = min over |d| <= W of
// in[j + d] + d^2.  One block = 64 consecutive x (the lanes: coalesced) x 64 positions along the axis; blockIdx.y = the line bundle.
__global__ __launch_bounds__(256) void dt_pass_axis_kernel(DtGrid g, const unsigned short* __restrict__ in, unsigned short* __restrict__ out,
                                                           int64_t stride, int len, int64_t outer_stride, int x_tiles) {
    __shared__ unsigned short tile[64 + 2 * kDtMaxWindow][64];
    const int W = g.window;
    const int lx = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int xt = blockIdx.x % x_tiles, jt = blockIdx.x / x_tiles;
    const int x = xt * 64 + lx, j0 = jt * 64;
    const int64_t base = (int64_t)blockIdx.y * outer_stride + x;       // blockIdx.y: the index of the remaining (outer) dimension
    const bool xin = x < g.nx;
    for (int t = grp; t < 64 + 2 * W; t += 4) {
        const int j = j0 - W + t;
        tile[t][lx] = (xin && j >= 0 && j < len) ? in[base + (int64_t)j * stride] : kDtInf;
    }
    __syncthreads();
    if (!xin) return;
    for (int k = grp; k < 64; k += 4) {
        const int j = j0 + k;
        if (j >= len) break;
        unsigned best = 0xFFFFFFFFu;
        for (int d = -W; d <= W; ++d) {
            const unsigned v = dt_add(tile[k + W + d][lx], d * d);
            best = v < best ? v : best;
        }
        out[base + (int64_t)j * stride] = best >= kDtInf ? kDtInf : (unsigned short)best;
    }
}

__device__ inline float dt_value(const unsigned short* __restrict__ vol, const DtGrid& g, int ix, int iy, int iz) {
    const unsigned short v = vol[((int64_t)iz * g.ny + iy) * g.nx + ix];
    const float d = v == kDtInf ? (float)g.window : sqrtf((float)v);
    return fminf(d, (float)g.window) * g.cell;
}

// block_sum of doubles in a fixed order (deterministic)
__device__ inline void dt_block_sum(double t, double* out) {
    __shared__ double sh[256];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

__global__ __launch_bounds__(256) void dt_loss_kernel(int n, const float* __restrict__ moved, DtGrid g, const unsigned short* __restrict__ vol,
                                                      float trunc, float* __restrict__ grad, double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double t = 0.0;
    if (i < n) {
        const int dims[3] = {g.nx, g.ny, g.nz};
        const float org[3] = {g.ox, g.oy, g.oz};
        int i0[3]; float f[3]; bool inside[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float u = (moved[i * 3 + c] - org[c]) / g.cell - 0.5f;
            const float hi = (float)(dims[c] - 1);
            inside[c] = u > 0.f && u < hi;                              // NaN -> clamped to 0, no gradient
            const float uc = u > 0.f ? (u < hi ? u : hi) : 0.f;
            int b = (int)floorf(uc);
            if (b > dims[c] - 2) b = dims[c] - 2;
            if (b < 0) b = 0;                                           // a one-cell axis
            i0[c] = b; f[c] = uc - (float)b;
        }
        const int x1 = i0[0] + 1 < g.nx ? i0[0] + 1 : i0[0], y1 = i0[1] + 1 < g.ny ? i0[1] + 1 : i0[1], z1 = i0[2] + 1 < g.nz ? i0[2] + 1 : i0[2];
        const float d000 = dt_value(vol, g, i0[0], i0[1], i0[2]), d100 = dt_value(vol, g, x1, i0[1], i0[2]);
        const float d010 = dt_value(vol, g, i0[0], y1, i0[2]), d110 = dt_value(vol, g, x1, y1, i0[2]);
        const float d001 = dt_value(vol, g, i0[0], i0[1], z1), d101 = dt_value(vol, g, x1, i0[1], z1);
        const float d011 = dt_value(vol, g, i0[0], y1, z1), d111 = dt_value(vol, g, x1, y1, z1);
        const float fx = f[0], fy = f[1], fz = f[2], gx = 1.f - fx, gy = 1.f - fy, gz = 1.f - fz;
        const float c00 = d000 * gx + d100 * fx, c10 = d010 * gx + d110 * fx, c01 = d001 * gx + d101 * fx, c11 = d011 * gx + d111 * fx;
        const float c0 = c00 * gy + c10 * fy, c1 = c01 * gy + c11 * fy;
        const float D = c0 * gz + c1 * fz;
        float gr[3] = {0.f, 0.f, 0.f};
        if (D <= trunc) {
            t = (double)D / (double)n;
            const float s = 1.0f / ((float)n * g.cell);
            const float dDx = ((d100 - d000) * gy + (d110 - d010) * fy) * gz + ((d101 - d001) * gy + (d111 - d011) * fy) * fz;
            const float dDy = (c10 - c00) * gz + (c11 - c01) * fz;
            const float dDz = c1 - c0;
            gr[0] = inside[0] ? dDx * s : 0.f; gr[1] = inside[1] ? dDy * s : 0.f; gr[2] = inside[2] ? dDz * s : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) grad[i * 3 + c] = gr[c];
    }
    dt_block_sum(t, partial + blockIdx.x);
}

__global__ __launch_bounds__(256) void dt_sum_partials_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
    double t = 0.0;
    for (int b = threadIdx.x; b < n; b += 256) t += partial[b];
    __shared__ double res;
    dt_block_sum(t, &res);
    if (threadIdx.x == 0) *out = res;
}

static bool dt_grid_ok(const float* h_origin, float cell, const int* h_dims, int window, DtGrid& g) {
    if (!h_origin || !h_dims || !(cell > 0.f) || window < 1 || window > kDtMaxWindow) return false;
    if (h_dims[0] < 1 || h_dims[1] < 1 || h_dims[2] < 1 || (int64_t)h_dims[0] * h_dims[1] * h_dims[2] > ((int64_t)1 << 31)) return false;
    if (3 * window * window >= kDtInf) return false;
    g = DtGrid{h_origin[0], h_origin[1], h_origin[2], cell, h_dims[0], h_dims[1], h_dims[2], window};
    return true;
}

}  // namespace himo

using namespace himo;

// two uint16 volumes (the passes ping-pong); the finished transform is in the FIRST
extern "C" size_t himo_dt_volume_bytes(const int* h_dims) {
    if (!h_dims) return 0;
    return round_up((size_t)h_dims[0] * h_dims[1] * h_dims[2] * 2, 256) * 2;
}

extern "C" int himo_dt_build(int n1, const float* d_pc1, const float* h_origin, float cell, const int* h_dims, int window,
                             void* d_volume, size_t volume_bytes, void* stream) {
    DtGrid g;
    if (n1 < 0 || (n1 > 0 && !d_pc1) || !d_volume || !dt_grid_ok(h_origin, cell, h_dims, window, g)) return HIMO_ERR_INVALID_ARGUMENT;
    if (volume_bytes < himo_dt_volume_bytes(h_dims) || !aligned16(d_volume)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t one = round_up((size_t)g.nx * g.ny * g.nz * 2, 256);
    unsigned short* A = reinterpret_cast<unsigned short*>(d_volume);
    unsigned short* B = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(d_volume) + one);
    ProfScope ps("dt_build_kernels", s);
    HIMO_HIP(hipMemsetAsync(B, 0xFF, one, s));
    if (n1 > 0) hipLaunchKernelGGL(dt_mark_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, n1, d_pc1, g, B);
    // x: B -> A; y: A -> B; z: B -> A
    hipLaunchKernelGGL(dt_pass_x_kernel, dim3((unsigned)(((int64_t)(g.nx + 255) / 256) * g.ny * g.nz)), dim3(256), 0, s, g, B, A);
    const int x_tiles = (g.nx + 63) / 64;
    hipLaunchKernelGGL(dt_pass_axis_kernel, dim3(x_tiles * ((g.ny + 63) / 64), g.nz), dim3(256), 0, s, g, A, B, (int64_t)g.nx, g.ny,
                       (int64_t)g.nx * g.ny, x_tiles);
    hipLaunchKernelGGL(dt_pass_axis_kernel, dim3(x_tiles * ((g.nz + 63) / 64), g.ny), dim3(256), 0, s, g, B, A, (int64_t)g.nx * g.ny, g.nz,
                       (int64_t)g.nx, x_tiles);
    HIMO_LAUNCH_CHECK("dt_build kernels");
    return HIMO_OK;
}

extern "C" size_t himo_dt_loss_workspace_bytes(int n) { return ((size_t)(n + 255) / 256 + 2) * 8 + 64; }

extern "C" int himo_dt_loss(int n, const float* d_moved, const float* h_origin, float cell, const int* h_dims, int window,
                            const void* d_volume, float trunc_dist, double* d_loss, float* d_grad_moved, void* d_workspace,
                            size_t workspace_bytes, void* stream) {
    DtGrid g;
    if (n < 0 || !d_volume || !d_loss || !d_workspace || !dt_grid_ok(h_origin, cell, h_dims, window, g)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!d_moved || !d_grad_moved)) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_dt_loss_workspace_bytes(n)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(d_workspace);
    const int nb = (n + 255) / 256;
    ProfScope ps("dt_loss_kernel", s);
    if (nb) hipLaunchKernelGGL(dt_loss_kernel, dim3(nb), dim3(256), 0, s, n, d_moved, g, reinterpret_cast<const unsigned short*>(d_volume),
                               trunc_dist, d_grad_moved, partial);
    hipLaunchKernelGGL(dt_sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, nb, d_loss);
    HIMO_LAUNCH_CHECK("dt_loss kernels");
    return HIMO_OK;
}
