// dtloss.hip -- stage a12, the distance-transform objective that makes FastNSF "fast" (BASELINE config 4, README.md:53
// `model=fastnsf`): instead of an exact nearest-neighbour search per optimiser iteration, the target sweep pc1 is turned ONCE per
// pair into a volume of distances to its nearest occupied voxel, and an iteration's loss is a trilinear lookup into that volume.
//
// PARITY UNPINNED (the reference's implementation is in the absent OpenSceneFlow submodule; only the method's name is in the tree).
// This build's specification (himo_amd/fastnsf.py, oracle/fastnsf_oracle.py dt_*):
//   volume    nx x ny x nz cells of edge `cell` from `origin`, x fastest; a cell is occupied when a pc1 point falls into it;
//             G[c] = squared distance IN CELLS from cell c to the nearest occupied cell (uint16; 0xFFFF = further than the window
//             W in some axis), D[c] = min(sqrt(G[c]), W) * cell -- exact Euclidean distance transform up to W cells;
//   lookup    u = (p - origin) / cell - 1/2 (cell-centre coordinates); a point is IN THE VOLUME when 0 < u < n - 1 on every axis;
//             D(p) = trilinear interpolation of the eight surrounding D values;
//             loss = (1 / m) sum_{i in volume} [D(p_i) <= trunc] D(p_i) with m = the number of points in the volume (0 -> loss 0);
//             gradient = its exact derivative.  A point outside the volume (AV2 sweeps reach ~200 m, the volume is the network
//             range + trunc) has no defined distance: it adds neither loss nor gradient and is not counted (csrc/dtlookup.h).
// The transform is separable: min over (dx, dy, dz) of dx^2 + dy^2 + dz^2 = min_dz (dz^2 + min_dy (dy^2 + min_dx dx^2)), each
// pass a windowed min-plus through an LDS tile (the volume is 113 M cells for the 106 x 106 x 10 m box at 0.1 m: three passes of
// ~0.25 GB read + written each, milliseconds once per pair against ~0.3 ms of NN searches EVERY iteration).
// HBM-bound streams; integer arithmetic; bit-deterministic.
#include "himo_common.h"
#include "dtlookup.h"
#include <math.h>

namespace himo {

constexpr int kDtMaxWindow = 40;

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
    // most 256-cell segments of a sweep's occupancy volume hold no occupied cell within the window: those write INF and are done
    const int any = __syncthreads_or(tile[threadIdx.x] == 0 || (threadIdx.x < 2 * W && tile[256 + threadIdx.x] == 0));
    const int x = x0 + threadIdx.x;
    if (x >= g.nx) return;
    unsigned best = 0xFFFFFFFFu;
    if (any)
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

// per point: unnormalised gradient d D / d p (zero outside the volume or beyond trunc); per block: sum of the counted distances and
// the number of points in the volume
__global__ __launch_bounds__(256) void dt_loss_kernel(int n, const float* __restrict__ moved, DtGrid g, const unsigned short* __restrict__ vol,
                                                      float trunc, float* __restrict__ grad, double* __restrict__ partial,
                                                      int* __restrict__ partial_count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double t = 0.0;
    int inside = 0;
    if (i < n) {
        float D, gr[3];
        const float p[3] = {moved[i * 3], moved[i * 3 + 1], moved[i * 3 + 2]};
        inside = dt_lookup(p, g, vol, trunc, D, gr) ? 1 : 0;
        if (inside && D <= trunc) t = (double)D;
#pragma unroll
        for (int c = 0; c < 3; ++c) grad[i * 3 + c] = gr[c];
    }
    dt_block_sum(t, partial + blockIdx.x);
    __shared__ int cnt[256];
    cnt[threadIdx.x] = inside;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) cnt[threadIdx.x] += cnt[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial_count[blockIdx.x] = cnt[0];
}

// every block: m = sum of the count partials (same order everywhere), its 256 points' gradients *= 1 / m; block 0 also writes the loss
__global__ __launch_bounds__(256) void dt_finish_kernel(int n, int nb, const double* __restrict__ partial, const int* __restrict__ partial_count,
                                                        float* __restrict__ grad, double* __restrict__ out) {
    __shared__ int cnt[256];
    int c = 0;
    for (int b = threadIdx.x; b < nb; b += 256) c += partial_count[b];
    cnt[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) cnt[threadIdx.x] += cnt[threadIdx.x + s];
        __syncthreads();
    }
    const int m = cnt[0];
    const float inv = m > 0 ? 1.0f / (float)m : 0.f;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
#pragma unroll
        for (int k = 0; k < 3; ++k) grad[i * 3 + k] *= inv;
    }
    if (blockIdx.x == 0) {
        double t = 0.0;
        for (int b = threadIdx.x; b < nb; b += 256) t += partial[b];
        __shared__ double res;
        dt_block_sum(t, &res);
        if (threadIdx.x == 0) *out = m > 0 ? res / (double)m : 0.0;
    }
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

extern "C" size_t himo_dt_loss_workspace_bytes(int n) { return ((size_t)(n + 255) / 256 + 2) * 12 + 64; }

extern "C" int himo_dt_loss(int n, const float* d_moved, const float* h_origin, float cell, const int* h_dims, int window,
                            const void* d_volume, float trunc_dist, double* d_loss, float* d_grad_moved, void* d_workspace,
                            size_t workspace_bytes, void* stream) {
    DtGrid g;
    if (n < 0 || !d_volume || !d_loss || !d_workspace || !dt_grid_ok(h_origin, cell, h_dims, window, g)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!d_moved || !d_grad_moved)) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_dt_loss_workspace_bytes(n)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (n + 255) / 256;
    double* partial = reinterpret_cast<double*>(d_workspace);
    int* partial_count = reinterpret_cast<int*>(partial + nb + 1);
    ProfScope ps("dt_loss_kernel", s);
    if (nb) hipLaunchKernelGGL(dt_loss_kernel, dim3(nb), dim3(256), 0, s, n, d_moved, g, reinterpret_cast<const unsigned short*>(d_volume),
                               trunc_dist, d_grad_moved, partial, partial_count);
    hipLaunchKernelGGL(dt_finish_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, s, n, nb, partial, partial_count, d_grad_moved, d_loss);
    HIMO_LAUNCH_CHECK("dt_loss kernels");
    return HIMO_OK;
}
