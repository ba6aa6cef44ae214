// nngrid.hip -- exact k=1 nearest neighbour between two full LiDAR sweeps (~120k x 120k points) through
// a uniform BEV cell grid: the correspondence search of the self-supervised Chamfer losses (stage a11).
//
// The reference's loss lives in the absent OpenSceneFlow submodule (in-tree fact:
// `loss_fn=seflowppLoss` with `chamfer_dis`, `dynamic_chamfer_dis`, `cluster_based_pc0pc1` terms,
// assets/slurm/ssl-train-av2.sh:33).  This is this build's own design; the oracle is scipy's cKDTree.
//
// Why a grid here and brute force in nn.hip: per-instance sets (10..5000 points) are best swept
// exhaustively, but sweep-to-sweep search is 1.4e10 pairs (~2 ms) five times per training sample.
// A 1 m x 1 m BEV grid holds ~10 points per cell, so a query inspects ~10^2 candidates instead of 10^5.
//   build:  cell histogram (integer atomics) -> two-level exclusive scan -> counting-sort scatter of the
//           reference rows (xyz + original index packed as float4, so a candidate is ONE 16-byte load)
//   query:  one lane per query walks Chebyshev rings of cells around its own cell and stops when the best
//           distance so far is <= the distance to the nearest unvisited ring; the result is the exact NN.
//           (Tried: walking the queries in cell order after a second counting sort -- the query kernel gains 9 us of
//           52, the extra sort costs 26; not kept.)
// Ties keep the lowest reference index (the rule of nn.hip), so the two kernels are interchangeable.
#include "himo_common.h"
#include <math.h>

namespace himo {

struct NnGrid {
    float x0, y0, inv_cell, cell;
    int gw, gh;
};

constexpr int kNngUnroll = 8;        // candidate loads in flight per lane

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ inline void cell_of(const NnGrid& g, float x, float y, int& cx, int& cy) {
    // points outside the grid are binned into the border cells (the ring bound below stays valid
    // because it is computed from the clamped cell geometry and the true query position)
    cx = clampi((int)floorf((x - g.x0) * g.inv_cell), 0, g.gw - 1);
    cy = clampi((int)floorf((y - g.y0) * g.inv_cell), 0, g.gh - 1);
}

__global__ __launch_bounds__(256) void nng_count_kernel(int64_t n, const float* __restrict__ r, NnGrid g, int* __restrict__ count,
                                                        int* __restrict__ cell_id) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cx, cy;
    cell_of(g, r[i * 3], r[i * 3 + 1], cx, cy);
    const int c = cy * g.gw + cx;
    cell_id[i] = c;
    atomicAdd(&count[c], 1);
}

constexpr int kNngScanBlock = 1024;
__global__ __launch_bounds__(256) void nng_scan_local_kernel(int* v, int n, int* block_sum) {
    __shared__ int wsum[4];
    const int base = blockIdx.x * kNngScanBlock + threadIdx.x * 4;
    int x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = base + k < n ? v[base + k] : 0;
    const int mine = x[0] + x[1] + x[2] + x[3];
    int incl = mine;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) v[base + k] = run; run += x[k]; }
    if (threadIdx.x == 255) block_sum[blockIdx.x] = run;
}

__global__ __launch_bounds__(1024) void nng_scan_top_kernel(int* block_sum, int nblk, int* v, int n) {
    __shared__ int part[1024];
    const int x = (int)threadIdx.x < nblk ? block_sum[threadIdx.x] : 0;
    part[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int y = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += y;
        __syncthreads();
    }
    if ((int)threadIdx.x < nblk) block_sum[threadIdx.x] = part[threadIdx.x] - x;
    if (threadIdx.x == 1023) v[n] = part[1023];     // grand total, so offset(n_cells) is defined
}

// make the per-cell offsets global (one add per cell) so the query kernel needs a single load per cell
__global__ __launch_bounds__(256) void nng_scan_add_kernel(int* v, int n, const int* __restrict__ block_sum) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] += block_sum[i / kNngScanBlock];
}

__global__ __launch_bounds__(256) void nng_fill_kernel(int64_t n, const float* __restrict__ r, const int* __restrict__ cell_id,
                                                       const int* __restrict__ offset, int* __restrict__ cursor,
                                                       float4* __restrict__ sorted) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = cell_id[i];
    const int slot = atomicAdd(&cursor[c], 1);
    sorted[offset[c] + slot] = make_float4(r[i * 3], r[i * 3 + 1], r[i * 3 + 2], __int_as_float((int)i));
}

__global__ __launch_bounds__(256) void nng_query_kernel(int64_t nq, const float* __restrict__ q, NnGrid g,
                                                        const int* __restrict__ offset, const float4* __restrict__ sorted,
                                                        float* __restrict__ dist2, int* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const float qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
    int cx, cy;
    cell_of(g, qx, qy, cx, cy);
    // distance from the query to the border of its (clamped) cell: everything outside ring r is at least
    // r * cell + margin away in the BEV plane (margin may be negative for queries outside the grid -> clamp to 0)
    const float lx = g.x0 + (float)cx * g.cell, ly = g.y0 + (float)cy * g.cell;
    const float margin = fmaxf(0.f, fminf(fminf(qx - lx, lx + g.cell - qx), fminf(qy - ly, ly + g.cell - qy)));
    float best = INFINITY;
    int bi = -1;
    const int rmax = max(max(cx, g.gw - 1 - cx), max(cy, g.gh - 1 - cy));
    // candidates [b, e) of the sorted reference array, kNngUnroll 16-byte loads in flight per step: the walk is a chain of dependent
    // loads (one lane, ~100 candidates, each compare waiting for its load), so its time is the chain length times the memory
    // latency (measured 120k x 120k: 240 -> 52 us).  Loads past the end are clamped to the last candidate (re-evaluating it
    // changes nothing: d == best, same index).
    auto scan = [&](int b, int e) {
        for (int k = b; k < e; k += kNngUnroll) {
            float4 p[kNngUnroll];
#pragma unroll
            for (int j = 0; j < kNngUnroll; ++j) p[j] = sorted[min(k + j, e - 1)];
#pragma unroll
            for (int j = 0; j < kNngUnroll; ++j) {
                const float dx = qx - p[j].x, dy = qy - p[j].y, dz = qz - p[j].z;
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                const int pi = __float_as_int(p[j].w);
                if (d < best || (d == best && pi < bi)) { best = d; bi = pi; }
            }
        }
    };
    for (int ring = 0; ring <= rmax; ++ring) {
        const int ylo = cy - ring, yhi = cy + ring;
        const int xlo = max(cx - ring, 0), xhi = min(cx + ring, g.gw - 1);
        for (int yy = max(ylo, 0); yy <= min(yhi, g.gh - 1); ++yy) {
            if (yy == ylo || yy == yhi) {
                // an edge row of the ring: its cells are consecutive in the sorted array -> ONE run of candidates
                scan(offset[yy * g.gw + xlo], offset[yy * g.gw + xhi + 1]);
            } else {                                            // inner rows (ring >= 1): the two end cells of the ring
                if (cx - ring >= 0) { const int c = yy * g.gw + cx - ring; scan(offset[c], offset[c + 1]); }
                if (cx + ring < g.gw) { const int c = yy * g.gw + cx + ring; scan(offset[c], offset[c + 1]); }
            }
        }
        const float reach = (float)ring * g.cell + margin;     // nearest possible unvisited point (BEV distance)
        if (bi >= 0 && best <= reach * reach) break;
    }
    dist2[i] = best;
    if (idx) idx[i] = bi;
}

static size_t nng_cells_bytes(int cells) { return round_up(((size_t)cells + 1) * 4, 16); }
static size_t nng_blocks_bytes(int cells) { return round_up(((size_t)(cells + kNngScanBlock - 1) / kNngScanBlock + 1) * 4, 16); }

}  // namespace himo

using namespace himo;

extern "C" size_t himo_nn_grid_workspace_bytes(int64_t n_ref, int grid_w, int grid_h) {
    const int cells = grid_w * grid_h;
    const size_t n = (size_t)(n_ref > 0 ? n_ref : 1);
    return 2 * nng_cells_bytes(cells) + nng_blocks_bytes(cells) + round_up(n * 4, 16) + round_up(n * 16, 16) + 64;
}

extern "C" int himo_nn_grid(int64_t nq, const float* d_q, int64_t nr, const float* d_r, float x0, float y0, float cell,
                            int grid_w, int grid_h, float* d_dist2, int32_t* d_idx, void* d_workspace,
                            size_t workspace_bytes, void* stream) {
    if (nq < 0 || nr < 0 || grid_w < 1 || grid_h < 1 || !(cell > 0.f)) return HIMO_ERR_INVALID_ARGUMENT;
    if (nr > 0x7fffffff || (int64_t)grid_w * grid_h > 1024 * kNngScanBlock) return HIMO_ERR_UNSUPPORTED;
    if (nq == 0) return HIMO_OK;
    if (!d_q || !d_dist2 || !d_workspace || (nr > 0 && !d_r)) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_nn_grid_workspace_bytes(nr, grid_w, grid_h) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int cells = grid_w * grid_h;
    char* ws = reinterpret_cast<char*>(d_workspace);
    int* offset = reinterpret_cast<int*>(ws);
    int* cursor = reinterpret_cast<int*>(ws + nng_cells_bytes(cells));
    int* block_sum = reinterpret_cast<int*>(ws + 2 * nng_cells_bytes(cells));
    int* cell_id = reinterpret_cast<int*>(ws + 2 * nng_cells_bytes(cells) + nng_blocks_bytes(cells));
    float4* sorted = reinterpret_cast<float4*>(ws + 2 * nng_cells_bytes(cells) + nng_blocks_bytes(cells) +
                                               round_up((size_t)(nr > 0 ? nr : 1) * 4, 16));
    NnGrid g{x0, y0, 1.0f / cell, cell, grid_w, grid_h};
    HIMO_HIP(hipMemsetAsync(offset, 0, 2 * nng_cells_bytes(cells), s));
    const int nblk = (cells + kNngScanBlock - 1) / kNngScanBlock;
    {
        ProfScope ps("nn_grid_build", s);
        if (nr > 0) hipLaunchKernelGGL(nng_count_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, nr, d_r, g, offset, cell_id);
        hipLaunchKernelGGL(nng_scan_local_kernel, dim3(nblk), dim3(256), 0, s, offset, cells, block_sum);
        hipLaunchKernelGGL(nng_scan_top_kernel, dim3(1), dim3(1024), 0, s, block_sum, nblk, offset, cells);
        hipLaunchKernelGGL(nng_scan_add_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, offset, cells, block_sum);
        if (nr > 0) hipLaunchKernelGGL(nng_fill_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, nr, d_r, cell_id, offset, cursor, sorted);
    }
    HIMO_LAUNCH_CHECK("nn_grid_build");
    {
        ProfScope ps("nn_grid_query_kernel", s);
        hipLaunchKernelGGL(nng_query_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, nq, d_q, g, offset, sorted, d_dist2, d_idx);
    }
    HIMO_LAUNCH_CHECK("nn_grid_query_kernel");
    return HIMO_OK;
}
