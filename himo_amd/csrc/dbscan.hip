// dbscan.hip -- stage a11's missing input: the self-supervised cluster labels (`+ssl_label=seflow_auto`,
// assets/slurm/ssl-train-av2.sh:32) generated on the GPU instead of read from files the reference tree does not hold.
//
// PARITY UNPINNED: the reference's label generator is in the absent OpenSceneFlow submodule; this is this
// build's own specification (himo_amd/seflow/ssl_label.py), its oracle is sklearn.cluster.DBSCAN (oracle/dbscan_oracle.py).
//
// DBSCAN(eps, min_pts) over 3-D points, Euclidean, min_pts counting the point itself (sklearn's convention):
//   core      a point with >= min_pts points within eps;
//   clusters  connected components of the core points under "within eps" -- unique, whatever the processing order;
//   border    a non-core point with a core point within eps joins a cluster; DBSCAN leaves WHICH one to the processing order, this
//             build fixes it: the cluster whose lowest-index member is lowest among the clusters of its core neighbours;
//   noise     everything else: label 0.  Points flagged in `skip` (ground, static) take no part and get label 0.
// Labels are 1 .. K in the order of each cluster's lowest point index: the result is a pure function of the input.
//
// Structure: the points are counting-sorted into BEV cells of edge >= eps (HBM-bound integer work: histogram with integer atomics,
// one-block exclusive scan, scatter of xyz + original index as one 16-byte row), so a point's neighbours are in its 3 x 3 cells;
// core test = one pass over those cells with early exit; components = lock-free union-find on the original indices (hook the
// larger root under the smaller with atomicMin: a component's root ends as its lowest index, deterministically; finds halve the
// paths they walk; one wave per point shares the walk over its neighbours); labels = find.
#include "himo_common.h"
#include <math.h>

namespace himo {

struct DbGrid {
    float x0, y0, inv_cell;
    int gw, gh;
};

__device__ inline int db_clamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void db_count_kernel(int n, const float* __restrict__ xyz, int pitch, const unsigned char* __restrict__ skip,
                                                       DbGrid g, int* __restrict__ count, int* __restrict__ cell_id) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = -1;
    const float x = xyz[(int64_t)i * pitch], y = xyz[(int64_t)i * pitch + 1], z = xyz[(int64_t)i * pitch + 2];
    if (!(skip && skip[i]) && x == x && y == y && z == z) {                 // (NaN rows take no part)
        const int cx = db_clamp((int)floorf((x - g.x0) * g.inv_cell), g.gw - 1), cy = db_clamp((int)floorf((y - g.y0) * g.inv_cell), g.gh - 1);
        c = cy * g.gw + cx;
        atomicAdd(&count[c], 1);
    }
    cell_id[i] = c;
}

// exclusive scan of v[0..n) in place, v[n] = the total, as TWO launches over tiles of 4096 values (four NEIGHBOURING values per thread:
// a wave reads 1 KB runs; wave shuffles inside a wave, 16 wave totals through LDS): (1) every block scans its own tile and leaves the
// tile's total in tile_sum[block]; (2) every block adds the sum of the tiles before it (a handful of values: one wave sums them).
// History: round 5 gave every thread of ONE block a contiguous chunk (stride-n/1024 accesses: 131 us for 120k values); ONE block walking
// the tiles coalesced took 48 us (a load latency per tile, 30 tiles); this takes ~2 x 4.
constexpr int kDbScanTile = 4096;
__global__ __launch_bounds__(1024) void db_scan_tiles_kernel(int* __restrict__ v, int n, int* __restrict__ tile_sum) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int at = blockIdx.x * kDbScanTile + threadIdx.x * 4;
    int x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = at + k < n ? v[at + k] : 0;
    const int mine = x[0] + x[1] + x[2] + x[3];
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int run = incl - mine, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int t = wsum[k]; total += t; if (k < w) run += t; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (at + k < n) v[at + k] = run; run += x[k]; }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void db_scan_offsets_kernel(int* __restrict__ v, int n, const int* __restrict__ tile_sum) {
    __shared__ int s_before, s_all;
    if (threadIdx.x < 64) {                                     // one wave: the tiles before this block's, and all of them (last block)
        int before = 0, all = 0;
        for (int t = threadIdx.x; t < (int)gridDim.x; t += 64) {
            const int x = tile_sum[t];
            all += x;
            if (t < (int)blockIdx.x) before += x;
        }
        for (int off = 32; off; off >>= 1) { before += __shfl_xor(before, off, 64); all += __shfl_xor(all, off, 64); }
        if (threadIdx.x == 0) { s_before = before; s_all = all; }
    }
    __syncthreads();
    const int at = blockIdx.x * kDbScanTile + threadIdx.x * 4, add = s_before;
    if (add) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (at + k < n) v[at + k] += add;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) v[n] = s_all;
}
static void db_scan(int* v, int n, int* tile_sum, hipStream_t s) {
    const int tiles = (n + kDbScanTile - 1) / kDbScanTile;
    hipLaunchKernelGGL(db_scan_tiles_kernel, dim3(tiles), dim3(1024), 0, s, v, n, tile_sum);
    hipLaunchKernelGGL(db_scan_offsets_kernel, dim3(tiles), dim3(1024), 0, s, v, n, tile_sum);
}

__global__ __launch_bounds__(256) void db_scatter_kernel(int n, const float* __restrict__ xyz, int pitch, const int* __restrict__ cell_id,
                                                         const int* __restrict__ offset, int* __restrict__ cursor, float4* __restrict__ rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = cell_id[i];
    if (c < 0) return;
    const int slot = offset[c] + atomicAdd(&cursor[c], 1);
    rows[slot] = float4{xyz[(int64_t)i * pitch], xyz[(int64_t)i * pitch + 1], xyz[(int64_t)i * pitch + 2], __int_as_float(i)};
}

// visit every row of the 3 x 3 cells around (cx, cy): f(row) returns false to stop
template <typename F>
__device__ inline void db_neighbours(const DbGrid& g, int cx, int cy, const int* __restrict__ offset, const float4* __restrict__ rows, F f) {
    for (int dy = -1; dy <= 1; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= g.gh) continue;
        const int x_lo = cx > 0 ? cx - 1 : 0, x_hi = cx + 1 < g.gw ? cx + 1 : g.gw - 1;
        const int lo = offset[y * g.gw + x_lo], hi = offset[y * g.gw + x_hi + 1];          // the three cells of a row are contiguous
        for (int s = lo; s < hi; ++s)
            if (!f(rows[s])) return;
    }
}

// one WAVE per sorted row (grid-stride over the rows that take part; the 64 lanes share the walk over the 3 x 3 cells, one row of three
// cells at a time, and stop as soon as min_pts are counted): core flag of its point (indexed by original index)
__global__ __launch_bounds__(256) void db_core_kernel(const float4* __restrict__ rows, DbGrid g, const int* __restrict__ offset, float eps2,
                                                      int min_pts, unsigned char* __restrict__ core) {
    const int total = offset[g.gw * g.gh];                      // the number of points that take part (the scan's total)
    const int lane = threadIdx.x & 63, n_waves = gridDim.x * 4;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < total; s += n_waves) {
        const float4 p = rows[s];
        const int cx = db_clamp((int)floorf((p.x - g.x0) * g.inv_cell), g.gw - 1), cy = db_clamp((int)floorf((p.y - g.y0) * g.inv_cell), g.gh - 1);
        int cnt = 0;
        for (int dy = -1; dy <= 1 && cnt < min_pts; ++dy) {
            const int y = cy + dy;
            if (y < 0 || y >= g.gh) continue;
            const int x_lo = cx > 0 ? cx - 1 : 0, x_hi = cx + 1 < g.gw ? cx + 1 : g.gw - 1;
            const int lo = offset[y * g.gw + x_lo], hi = offset[y * g.gw + x_hi + 1];      // the three cells of a row are contiguous
            for (int t0 = lo; t0 < hi && cnt < min_pts; t0 += 64) {
                bool in = false;
                if (t0 + lane < hi) {
                    const float4 q = rows[t0 + lane];
                    const float dx = q.x - p.x, dy2 = q.y - p.y, dz = q.z - p.z;
                    in = dx * dx + dy2 * dy2 + dz * dz <= eps2;
                }
                cnt += __popcll(__ballot(in));
            }
        }
        if (lane == 0) core[__float_as_int(p.w)] = cnt >= min_pts ? 1 : 0;
    }
}

__device__ inline int db_find(const int* __restrict__ parent, int x) {
    int p = parent[x];
    while (p != x) { x = p; p = parent[x]; }
    return x;
}
// find with path halving: every visited node is re-pointed at its grandparent.  Safe without locks beside the atomicMin hooks below:
// a store only ever targets a NON-root (its parent differs from itself, and a hooked node never becomes a root again) and writes an
// ancestor of that node -- a lower index of the same component -- so no link a hook relies on is lost (a hook onto a non-root
// re-joins that node's old parent itself, see db_union) and no cycle can form.  Components, and their lowest index, are unchanged;
// the trees become flat, which is what the second and later unions of a dense object's points wait for
__device__ inline int db_find_halve(int* __restrict__ parent, int x) {
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p; p = gp;
    }
    return x;
}
__device__ inline void db_union(int* __restrict__ parent, int a, int b) {
    while (true) {
        a = db_find_halve(parent, a); b = db_find_halve(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }            // hook the larger root a under the smaller b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;                                   // a was still a root: hooked
        a = old;                                                // somebody hooked a meanwhile: carry on from there
    }
}

// one WAVE per sorted row (grid-stride over the rows that take part): a core point is joined with every core neighbour of lower
// original index, the 64 lanes taking the rows of its 3 x 3 cells in turn.  (Round 5: one thread per row -- the participating rows
// are a few per cent of a sweep, sorted to the front: a dozen blocks did all the work, each thread walking ~1000 neighbours of a
// dense object and chasing un-compressed parent chains for each: 355 us per call.)
__global__ __launch_bounds__(256) void db_union_kernel(const float4* __restrict__ rows, DbGrid g, const int* __restrict__ offset, float eps2,
                                                       const unsigned char* __restrict__ core, int* __restrict__ parent) {
    const int total = offset[g.gw * g.gh];                      // the number of points that take part (the scan's total)
    const int lane = threadIdx.x & 63, n_waves = gridDim.x * 4;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < total; s += n_waves) {
        const float4 p = rows[s];
        const int i = __float_as_int(p.w);
        if (!core[i]) continue;
        const int cx = db_clamp((int)floorf((p.x - g.x0) * g.inv_cell), g.gw - 1), cy = db_clamp((int)floorf((p.y - g.y0) * g.inv_cell), g.gh - 1);
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = cy + dy;
            if (y < 0 || y >= g.gh) continue;
            const int x_lo = cx > 0 ? cx - 1 : 0, x_hi = cx + 1 < g.gw ? cx + 1 : g.gw - 1;
            const int lo = offset[y * g.gw + x_lo], hi = offset[y * g.gw + x_hi + 1];      // the three cells of a row are contiguous
            for (int t = lo + lane; t < hi; t += 64) {
                const float4 q = rows[t];
                const int j = __float_as_int(q.w);
                if (j < i && core[j]) {
                    const float dx = q.x - p.x, dy2 = q.y - p.y, dz = q.z - p.z;
                    if (dx * dx + dy2 * dy2 + dz * dz <= eps2) db_union(parent, i, j);
                }
            }
        }
    }
}

// root[i] = lowest index of i's cluster, or -1 (noise / skipped); is_root[i] = 1 for the cluster's lowest index.  One WAVE per sorted
// row: a core point is one find; a border candidate's lanes share the walk over its 3 x 3 cells and the lowest root among its core
// neighbours is a wave minimum
__global__ __launch_bounds__(256) void db_root_kernel(const float4* __restrict__ rows, DbGrid g, const int* __restrict__ offset, float eps2,
                                                      const unsigned char* __restrict__ core, const int* __restrict__ parent, int* __restrict__ root,
                                                      int* __restrict__ is_root) {
    const int total = offset[g.gw * g.gh];
    const int lane = threadIdx.x & 63, n_waves = gridDim.x * 4;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < total; s += n_waves) {
        const float4 p = rows[s];
        const int i = __float_as_int(p.w);
        int r = 0x7fffffff;
        if (core[i]) {
            r = db_find(parent, i);
        } else {
            const int cx = db_clamp((int)floorf((p.x - g.x0) * g.inv_cell), g.gw - 1), cy = db_clamp((int)floorf((p.y - g.y0) * g.inv_cell), g.gh - 1);
            for (int dy = -1; dy <= 1; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= g.gh) continue;
                const int x_lo = cx > 0 ? cx - 1 : 0, x_hi = cx + 1 < g.gw ? cx + 1 : g.gw - 1;
                const int lo = offset[y * g.gw + x_lo], hi = offset[y * g.gw + x_hi + 1];
                for (int t = lo + lane; t < hi; t += 64) {
                    const float4 q = rows[t];
                    const int j = __float_as_int(q.w);
                    if (core[j]) {
                        const float dx = q.x - p.x, dy2 = q.y - p.y, dz = q.z - p.z;
                        if (dx * dx + dy2 * dy2 + dz * dz <= eps2) {
                            const int rj = db_find(parent, j);
                            r = rj < r ? rj : r;
                        }
                    }
                }
            }
            for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(r, off, 64); r = o < r ? o : r; }
        }
        if (lane == 0) {
            root[i] = r == 0x7fffffff ? -1 : r;
            if (r == i) is_root[i] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void db_init_kernel(int n, int* __restrict__ parent, int* __restrict__ root, int* __restrict__ is_root,
                                                      unsigned char* __restrict__ core) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    parent[i] = i; root[i] = -1; is_root[i] = 0; core[i] = 0;
}

__global__ __launch_bounds__(256) void db_label_kernel(int n, const int* __restrict__ root, const int* __restrict__ rank, int* __restrict__ labels,
                                                       int* __restrict__ n_clusters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && n_clusters) *n_clusters = rank[n];
    if (i >= n) return;
    const int r = root[i];
    labels[i] = r < 0 ? 0 : rank[r] + 1;
}

struct DbLayout {
    size_t count, cursor, cell_id, rows, parent, root, is_root, core, tile_sum, total;
};
static DbLayout db_layout(int n, int gw, int gh) {
    DbLayout L{};
    const size_t cells = (size_t)gw * gh;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += round_up(bytes, 256); return at; };
    L.count = take((cells + 1) * 4); L.cursor = take(cells * 4); L.cell_id = take((size_t)n * 4); L.rows = take((size_t)n * 16);
    L.parent = take((size_t)n * 4); L.root = take((size_t)n * 4); L.is_root = take(((size_t)n + 1) * 4); L.core = take((size_t)n);
    L.tile_sum = take((((size_t)(n > (int)cells ? n : (int)cells) + kDbScanTile - 1) / kDbScanTile + 1) * 4);
    L.total = o;
    return L;
}

}  // namespace himo

using namespace himo;

extern "C" size_t himo_dbscan_workspace_bytes(int n, int grid_w, int grid_h) {
    if (n < 0 || grid_w < 1 || grid_h < 1) return 0;
    return db_layout(n, grid_w, grid_h).total + 256;
}

// d_xyz [n][pitch >= 3] float32; d_skip [n] bytes or NULL (non-zero = the point takes no part, label 0); BEV grid of `cell` >= eps
// metre cells from (x0, y0), grid_w x grid_h of them (points beyond it are binned into the border cells: still exact, only slower
// there); d_labels [n] int32: 0 noise / skipped, 1 .. K clusters ordered by their lowest point index; d_n_clusters: K (or NULL).
extern "C" int himo_dbscan(int n, const float* d_xyz, int pitch, const unsigned char* d_skip, float eps, int min_pts, float x0, float y0,
                           float cell, int grid_w, int grid_h, int32_t* d_labels, int32_t* d_n_clusters, void* d_workspace,
                           size_t workspace_bytes, void* stream) {
    if (n < 0 || pitch < 3 || !(eps > 0.f) || min_pts < 1 || !(cell >= eps) || grid_w < 1 || grid_h < 1 || (int64_t)grid_w * grid_h > (1 << 26))
        return HIMO_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!d_xyz || !d_labels)) return HIMO_ERR_INVALID_ARGUMENT;
    if (!d_workspace || workspace_bytes < himo_dbscan_workspace_bytes(n, grid_w, grid_h) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (d_n_clusters) HIMO_HIP(hipMemsetAsync(d_n_clusters, 0, 4, s));
        return HIMO_OK;
    }
    const DbLayout L = db_layout(n, grid_w, grid_h);
    char* w = reinterpret_cast<char*>(d_workspace);
    int* count = reinterpret_cast<int*>(w + L.count); int* cursor = reinterpret_cast<int*>(w + L.cursor);
    int* cell_id = reinterpret_cast<int*>(w + L.cell_id); float4* rows = reinterpret_cast<float4*>(w + L.rows);
    int* parent = reinterpret_cast<int*>(w + L.parent); int* root = reinterpret_cast<int*>(w + L.root);
    int* is_root = reinterpret_cast<int*>(w + L.is_root); unsigned char* core = reinterpret_cast<unsigned char*>(w + L.core);
    int* tile_sum = reinterpret_cast<int*>(w + L.tile_sum);
    const int cells = grid_w * grid_h, nb = (n + 255) / 256;
    const int nw = nb < 4096 ? nb : 4096;                       // blocks of the wave-per-row kernels (grid-stride over the participating rows)
    const DbGrid g{x0, y0, 1.0f / cell, grid_w, grid_h};
    const float eps2 = eps * eps;
    ProfScope ps("dbscan_kernels", s);
    HIMO_HIP(hipMemsetAsync(count, 0, ((size_t)cells + 1) * 4, s));
    HIMO_HIP(hipMemsetAsync(cursor, 0, (size_t)cells * 4, s));
    hipLaunchKernelGGL(db_init_kernel, dim3(nb), dim3(256), 0, s, n, parent, root, is_root, core);
    hipLaunchKernelGGL(db_count_kernel, dim3(nb), dim3(256), 0, s, n, d_xyz, pitch, d_skip, g, count, cell_id);
    db_scan(count, cells, tile_sum, s);                                                       // count -> offsets; count[cells] = points taking part
    hipLaunchKernelGGL(db_scatter_kernel, dim3(nb), dim3(256), 0, s, n, d_xyz, pitch, cell_id, count, cursor, rows);
    // (the sorted-row kernels are launched over n slots and stop at the number of participating points, which only the device knows)
    hipLaunchKernelGGL(db_core_kernel, dim3(nw), dim3(256), 0, s, rows, g, count, eps2, min_pts, core);
    hipLaunchKernelGGL(db_union_kernel, dim3(nw), dim3(256), 0, s, rows, g, count, eps2, core, parent);
    hipLaunchKernelGGL(db_root_kernel, dim3(nw), dim3(256), 0, s, rows, g, count, eps2, core, parent, root, is_root);
    db_scan(is_root, n, tile_sum, s);                                                         // is_root -> rank of each cluster's lowest index
    hipLaunchKernelGGL(db_label_kernel, dim3(nb), dim3(256), 0, s, n, root, is_root, d_labels, d_n_clusters);
    HIMO_LAUNCH_CHECK("dbscan kernels");
    return HIMO_OK;
}
