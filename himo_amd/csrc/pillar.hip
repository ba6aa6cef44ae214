// pillar.hip -- stage a10, front end: ego transform + dynamic pillarisation + pillar feature net,
// N x 3 LiDAR points -> 32-channel BEV pseudo-image (NHWC float32) for gfx950.
//
// No reference source exists for this stage (OpenSceneFlow submodule absent);
// the specification is himo_amd/seflow/spec.py (steps 0-2) and the oracle is
// oracle/seflow_oracle.py::pillar_image.  In-tree facts honoured: voxel_size = [0.2, 0.2, 6],
// point_cloud_range = [-51.2, -51.2, -3, 51.2, 51.2, 3] -> 512 x 512 x 1 pillars
// (assets/slurm/ssl-train-av2.sh:32).
//
// Kernels (one sweep per call, all on one stream):
//   pillar_assign_kernel    coalesced read of the raw point rows, float32 rigid transform into the
//                           target frame, cell index, integer histogram of points per cell
//   cell_scan_*_kernel      two-level exclusive scan of the 262,144 cell counts
//   pillar_fill_kernel      counting-sort scatter of point indices into per-cell lists
//   pillar_feature_kernel   half a wavefront (32 lanes = 32 channels) per cell: the cell's point
//                           list is held in registers (through global memory for cells with more
//                           than 32 points) and put in ascending point order with wavefront shuffles
//                           (so sums are order-deterministic and match a sequential CPU scatter), then
//                           mean -> 9 features -> Linear(9,32) -> BN -> ReLU -> mean over the cell,
//                           one 128-byte NHWC store per cell; empty cells are written as zeros
//                           (so no separate memset of the 32 MB image).
// Only integer atomics are used; the output is bit-deterministic.
#include "himo_common.h"
#include "bf16x3.h"
#include <math.h>
#include <algorithm>

namespace himo {

struct GridSpec {
    float xmin, ymin, zmin;
    float vx, vy, vz;
    float cx0, cy0, cz0;   // cell-centre offsets: v/2 + min, rounded to float32
    int W, H;
};

struct PillarArgs {
    int64_t n;
    const float* pts; int stride;
    float R[9], t[3];
    GridSpec g;
    const float* pfn_w;       // [9][32]
    const float* pfn_scale;   // [32] gamma / sqrt(var + eps)
    const float* pfn_shift;   // [32] beta - mean * scale
    float* xyz_t;             // [n][3]
    int* pid;                 // [n]  cell id (iy * W + ix) or -1
    float* offsets;           // [n][3] point - cell centre (zeros for dropped points)
    float* image; int image_pitch;   // [H*W][pitch], 32 channels written per cell
    int image_split;                 // 1: the 32 channels as two records of the split activation format (convsg.hip)
    int* cell_count;          // [H*W] -> block-local exclusive offsets after the scan
    int* block_sum;           // [H*W/1024 + 1] exclusive offsets of the 1024-cell blocks
    int* cell_cursor;         // [H*W]
    int* order2;              // [n] ascending order for cells too crowded for the LDS stage
    float4* cell_rec;         // [n] (x, y, z, point index as bits) of the transformed points grouped by cell (scatter order): ONE
                              //        16-byte scattered store per point in the fill kernel (four 4-byte ones cost 8.5x their bytes
                              //        in HBM sector writes), one 16-byte load per point in the feature kernel
    unsigned long long* occ;  // incremental images (or NULL): bit c of word w = cell 64 w + c was non-empty the LAST time this
                              //        workspace's image was written -- an empty cell that was empty then is already zero
};

// up to twelve sweeps per launch (blockIdx.y selects the sweep): the stage's kernels are latency chains on small grids,
// so the three sweeps of a sample -- and of up to four samples of a batch -- share each launch instead of queueing
// behind one another (12 argument blocks = 2.8 KB of the 4 KB kernel-argument segment)
constexpr int kMaxSweeps = 12;
struct PillarBatch { PillarArgs s[kMaxSweeps]; };

// cell_count of every sweep of the launch group cleared in ONE launch (16 bytes per lane): as per-sweep
// hipMemsetAsync calls they were 12 serialized ~5 us fill kernels per group, a quarter millisecond per 16-sample step
__global__ __launch_bounds__(256) void pillar_clear_kernel(PillarBatch m, int n_vec4) {
    int4* p = reinterpret_cast<int4*>(m.s[blockIdx.y].cell_count);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_vec4; i += gridDim.x * 256) p[i] = make_int4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void pillar_assign_kernel(PillarBatch m) {
    const PillarArgs& a = m.s[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const float* p = a.pts + i * a.stride;
    const float x = p[0], y = p[1], z = p[2];
    // p' = R p + t with every product and sum rounded separately (this file is built with
    // -ffp-contract=off): the cell a point falls into is a discrete decision, so the transform is
    // specified down to the rounding (spec.py step 0) and the oracle evaluates the identical sequence
    const float tx = ((x * a.R[0] + y * a.R[1]) + z * a.R[2]) + a.t[0];
    const float ty = ((x * a.R[3] + y * a.R[4]) + z * a.R[5]) + a.t[1];
    const float tz = ((x * a.R[6] + y * a.R[7]) + z * a.R[8]) + a.t[2];
    a.xyz_t[i * 3] = tx; a.xyz_t[i * 3 + 1] = ty; a.xyz_t[i * 3 + 2] = tz;
    const float fx = floorf((tx - a.g.xmin) / a.g.vx);
    const float fy = floorf((ty - a.g.ymin) / a.g.vy);
    const float fz = floorf((tz - a.g.zmin) / a.g.vz);
    const bool ok = fx >= 0.f && fx < (float)a.g.W && fy >= 0.f && fy < (float)a.g.H && fz >= 0.f && fz < 1.f;
    int cell = -1;
    if (ok) {
        cell = (int)fy * a.g.W + (int)fx;
        // the histogram's returned count IS the point's slot in its cell's list: the scatter below needs no second round of atomics
        // (kept in order2 until the feature kernel, which runs after the scatter, writes the ascending order there)
        a.order2[i] = atomicAdd(&a.cell_count[cell], 1);
    } else {
        a.offsets[i * 3] = 0.f; a.offsets[i * 3 + 1] = 0.f; a.offsets[i * 3 + 2] = 0.f;
    }
    a.pid[i] = cell;
}

// exclusive scan of the cell counts in two levels: every 1024-cell block scans itself in place and
// leaves its total in block_sum; a single small block then scans the (<= 1024) block totals.
// Consumers read cell_offset(cell) = v[cell] + block_sum[cell / 1024].
constexpr int kScanBlock = 1024;

__global__ __launch_bounds__(256) void cell_scan_local_kernel(PillarBatch m) {
    int* v = m.s[blockIdx.y].cell_count;
    int* block_sum = m.s[blockIdx.y].block_sum;
    const int n = m.s[blockIdx.y].g.W * m.s[blockIdx.y].g.H;
    __shared__ int wsum[4];
    const int base = blockIdx.x * kScanBlock + threadIdx.x * 4;
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

__global__ __launch_bounds__(1024) void cell_scan_top_kernel(PillarBatch m, int nblk) {
    int* block_sum = m.s[blockIdx.y].block_sum;
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
    if (threadIdx.x == 1023) block_sum[nblk] = part[1023];
}

__device__ inline int cell_offset(const PillarArgs& a, int cell);

__device__ inline int cell_offset(const PillarArgs& a, int cell) {
    const int n_cells = a.g.W * a.g.H;
    return cell >= n_cells ? a.block_sum[(n_cells + kScanBlock - 1) / kScanBlock]
                           : a.cell_count[cell] + a.block_sum[cell / kScanBlock];
}

__global__ __launch_bounds__(256) void pillar_fill_kernel(PillarBatch m) {
    const PillarArgs& a = m.s[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int cell = a.pid[i];
    if (cell < 0) return;
    const int at = cell_offset(a, cell) + a.order2[i];
    a.cell_rec[at] = make_float4(a.xyz_t[i * 3], a.xyz_t[i * 3 + 1], a.xyz_t[i * 3 + 2], __int_as_float((int)i));
}

constexpr int kCellsPerBlock = 8;     // 8 cells x 32 lanes = 256 threads

constexpr int kFeatCells = 64;        // cells per block of the feature kernel

// A block owns 64 consecutive cells.  Most cells of a sweep are empty (120k points over 262k cells): their 128-byte
// rows are zero-filled cooperatively, and the non-empty ones are compacted into a short list that the block's eight
// half-waves (32 lanes = 32 channels) work through INDEPENDENTLY: a cell's points live in the half-wave's registers
// (lane j = point j), are ranked and permuted into ascending point order with wavefront shuffles, and the sequential
// (order-deterministic) sums broadcast one point at a time -- no LDS, no barrier after the prologue.  Cells with more
// than 32 points rank in 32-point chunks and go through the order2 array in global memory.
__device__ inline float shfl32(float v, int src) { return __shfl(v, src, 32); }
__device__ inline int shfl32(int v, int src) { return __shfl(v, src, 32); }

__global__ __launch_bounds__(256) void pillar_feature_kernel(PillarBatch m) {
    const PillarArgs& a = m.s[blockIdx.y];
    __shared__ int s_beg[kFeatCells + 1];
    __shared__ int s_list[kFeatCells];
    __shared__ int s_nlist;
    const int sub = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int n_cells = a.g.W * a.g.H;
    const int cell0 = blockIdx.x * kFeatCells;
    static_assert(kFeatCells == 64, "one 64-bit occupancy word per block");
    // incremental images: the cells of this block that held data after the previous pass over this image (all of them the
    // first time); read by every thread BEFORE the barrier, replaced by thread 0 after it
    const unsigned long long was = a.occ ? a.occ[blockIdx.x] : ~0ull;
    if (threadIdx.x <= kFeatCells) s_beg[threadIdx.x] = cell_offset(a, min(cell0 + (int)threadIdx.x, n_cells));
    __syncthreads();
    if (threadIdx.x < 64) {
        // wave 0 compacts the non-empty cells: the single-point ones first, then the others.  The two half-waves of a wave work
        // on neighbouring list entries, and a wave runs the (shuffle-heavy) multi-point path whenever EITHER half needs it: with
        // the kinds grouped a wave meets it for 22 % of its cell pairs instead of 39 % (uniform 120k-point sweep)
        const int cnt = cell0 + (int)threadIdx.x < n_cells ? s_beg[threadIdx.x + 1] - s_beg[threadIdx.x] : 0;
        const unsigned long long one = __ballot(cnt == 1), many = __ballot(cnt > 1);
        const unsigned long long below = (1ull << threadIdx.x) - 1ull;
        if (cnt == 1) s_list[__popcll(one & below)] = threadIdx.x;
        if (cnt > 1) s_list[__popcll(one) + __popcll(many & below)] = threadIdx.x;
        if (threadIdx.x == 0) { s_nlist = __popcll(one | many); if (a.occ) a.occ[blockIdx.x] = one | many; }
    }
    // zero rows of the empty cells (those that are not zero already): 16 bytes per lane, 8 lanes per 128-byte row, 32 rows per pass
#pragma unroll
    for (int i = 0; i < kFeatCells / 32; ++i) {
        const int lc = i * 32 + (threadIdx.x >> 3);
        if (cell0 + lc < n_cells && s_beg[lc + 1] == s_beg[lc] && ((was >> lc) & 1ull))
            *reinterpret_cast<float4*>(a.image + (int64_t)(cell0 + lc) * a.image_pitch + (threadIdx.x & 7) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int nlist = s_nlist;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.pfn_w[k * 32 + c];
    const float scale = a.pfn_scale[c], shift = a.pfn_shift[c];

    for (int k = sub; k < nlist; k += kCellsPerBlock) {
        const int lc = s_list[k];
        const int cell = cell0 + lc;
        const int beg = s_beg[lc];
        const int cnt = s_beg[lc + 1] - beg;
        const int iy = cell / a.g.W, ix = cell - iy * a.g.W;
        const float ccx = (float)ix * a.g.vx + a.g.cx0, ccy = (float)iy * a.g.vy + a.g.cy0, ccz = 0.f * a.g.vz + a.g.cz0;
        const float fc = (float)cnt;
        float acc = 0.f;
        if (cnt == 1) {
            // a single point (the common case on sparse sweeps): no ordering, the mean IS the point, so the three
            // offset-to-mean features are exactly zero and their products drop out of the sum (fma(0, w, v) == v)
            const float4 rc = a.cell_rec[beg];
            const int pj = __float_as_int(rc.w);
            const float x = rc.x, y = rc.y, z = rc.z;
            if (c == 0) a.order2[beg] = pj;
            const float f6[3] = {x - ccx, y - ccy, z - ccz};
            float v = x * w[0];
            v = fmaf(y, w[1], v); v = fmaf(z, w[2], v);
            v = fmaf(x - x, w[3], v); v = fmaf(y - y, w[4], v); v = fmaf(z - z, w[5], v);
            v = fmaf(f6[0], w[6], v); v = fmaf(f6[1], w[7], v); v = fmaf(f6[2], w[8], v);
            v = v * scale + shift;
            acc = 0.f + fmaxf(v, 0.f);
            if (c < 3) a.offsets[(int64_t)pj * 3 + c] = f6[c];
        } else if (cnt <= 32) {
            // lane j holds point j of the cell's (scatter-ordered) list; rank by point index, permute into ascending order
            const bool have = c < cnt;
            const float4 rc = a.cell_rec[beg + (have ? c : 0)];
            const int idx = have ? __float_as_int(rc.w) : 0x7fffffff;
            const float ux = rc.x, uy = rc.y, uz = rc.z;
            int rank = 0;
            for (int j = 0; j < cnt; ++j) rank += shfl32(idx, j) < idx;
            int src = 0;
            for (int j = 0; j < cnt; ++j)
                if (shfl32(rank, j) == c) src = j;
            const int sidx = shfl32(idx, src);
            const float px = shfl32(ux, src), py = shfl32(uy, src), pz = shfl32(uz, src);
            if (have) a.order2[beg + c] = sidx;               // kept for the training backward pass
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int j = 0; j < cnt; ++j) { sx += shfl32(px, j); sy += shfl32(py, j); sz += shfl32(pz, j); }
            const float mx = sx / fc, my = sy / fc, mz = sz / fc;
            for (int j = 0; j < cnt; ++j) {
                const float x = shfl32(px, j), y = shfl32(py, j), z = shfl32(pz, j);
                const int pj = shfl32(sidx, j);
                const float f[9] = {x, y, z, x - mx, y - my, z - mz, x - ccx, y - ccy, z - ccz};
                float v = f[0] * w[0];
#pragma unroll
                for (int q = 1; q < 9; ++q) v = fmaf(f[q], w[q], v);
                v = v * scale + shift;
                acc += fmaxf(v, 0.f);
                if (c < 3) a.offsets[(int64_t)pj * 3 + c] = f[6 + c];
            }
        } else {
            // crowded cell: rank every point against the list in 32-point chunks held in registers, ascending order
            // through order2 in global memory
            for (int j0 = 0; j0 < cnt; j0 += 32) {
                const bool have = j0 + c < cnt;
                const int idx = have ? __float_as_int(a.cell_rec[beg + j0 + c].w) : 0x7fffffff;
                int rank = 0;
                for (int k0 = 0; k0 < cnt; k0 += 32) {
                    const int other = k0 + c < cnt ? __float_as_int(a.cell_rec[beg + k0 + c].w) : 0x7fffffff;
                    const int lim = min(32, cnt - k0);
                    for (int j = 0; j < lim; ++j) rank += shfl32(other, j) < idx;
                }
                if (have) a.order2[beg + rank] = idx;
            }
            __threadfence_block();
            // the ascending list is walked in 32-point chunks: lane c fetches point j0 + c (one gather per chunk instead of a chain
            // of dependent loads per point), the sequential -- order-deterministic -- sums then take the points by shuffle
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int j0 = 0; j0 < cnt; j0 += 32) {
                const bool have = j0 + c < cnt;
                const float* p = a.xyz_t + (int64_t)(have ? a.order2[beg + j0 + c] : 0) * 3;
                const float ux = have ? p[0] : 0.f, uy = have ? p[1] : 0.f, uz = have ? p[2] : 0.f;
                const int lim = min(32, cnt - j0);
                for (int j = 0; j < lim; ++j) { sx += shfl32(ux, j); sy += shfl32(uy, j); sz += shfl32(uz, j); }
            }
            const float mx = sx / fc, my = sy / fc, mz = sz / fc;
            for (int j0 = 0; j0 < cnt; j0 += 32) {
                const bool have = j0 + c < cnt;
                const int pidx = have ? a.order2[beg + j0 + c] : 0;
                const float* p = a.xyz_t + (int64_t)pidx * 3;
                const float ux = have ? p[0] : 0.f, uy = have ? p[1] : 0.f, uz = have ? p[2] : 0.f;
                const int lim = min(32, cnt - j0);
                for (int j = 0; j < lim; ++j) {
                    const float x = shfl32(ux, j), y = shfl32(uy, j), z = shfl32(uz, j);
                    const int pj = shfl32(pidx, j);
                    const float f[9] = {x, y, z, x - mx, y - my, z - mz, x - ccx, y - ccy, z - ccz};
                    float v = f[0] * w[0];
#pragma unroll
                    for (int q = 1; q < 9; ++q) v = fmaf(f[q], w[q], v);
                    v = v * scale + shift;
                    acc += fmaxf(v, 0.f);
                    if (c < 3) a.offsets[(int64_t)pj * 3 + c] = f[6 + c];
                }
            }
        }
        const float feat = acc / fc;
        if (a.image_split) {         // [16 fp16 high | 16 fp16 low] per 16 channels; lane pairs exchange halves: one 32-bit store each
            unsigned h, l;
            split2_rounded(feat, h, l);
            const bool odd = c & 1;
            const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? h : l), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            unsigned* rec = reinterpret_cast<unsigned*>(a.image + (int64_t)cell * a.image_pitch + (c & ~15));
            rec[(odd ? 8 : 0) + ((c & 15) >> 1)] = odd ? (recv | (l << 16)) : (h | (recv << 16));
        } else {
            a.image[(int64_t)cell * a.image_pitch + c] = feat;
        }
    }
}

// ---- training backward (stage a11) ------------------------------------------------------------------------------
// Both kernels walk the per-cell point lists the forward pass left in the workspace (cell offsets + order2, the
// ascending order), so every sum has the forward's fixed order and no float atomics are needed.
struct PillarBwdArgs {
    GridSpec g;
    const int* cell_count; const int* block_sum; const int* order2;
    const float4* cell_rec;                     // (x, y, z, index) grouped by cell, scatter order (PillarArgs::cell_rec)
    const float* xyz_t;
    const float* pfn_w; const float* pfn_scale; const float* pfn_shift;
    const float* d_image; int image_pitch;      // gradient w.r.t. this sweep's 32 image channels
    float* partial;                             // [gridDim.x][9][32]
    // scatter of the head's per-point gradient
    const float* dhx; int dhx_pitch;            // [n][>=128]: d/d [img0 | img1 | dec] rows
    float* d_b0; int b0_pitch; int group0, group1, n_groups;
    float* d_dec; int dec_pitch;
};

__device__ inline int cell_offset_b(const PillarBwdArgs& a, int cell) {
    const int n_cells = a.g.W * a.g.H;
    return cell >= n_cells ? a.block_sum[(n_cells + kScanBlock - 1) / kScanBlock]
                           : a.cell_count[cell] + a.block_sum[cell / kScanBlock];
}

// one block per feature k: 32 groups x 32 channels, each group sums every 32nd partial, fixed-order combine
__global__ __launch_bounds__(1024) void pfn_backward_reduce_kernel(const float* __restrict__ partial, int n_blocks, float* __restrict__ dw,
                                                                   int accumulate) {
    __shared__ float sh[32][32];
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5, k = blockIdx.x;
    {   // 32 groups x 32 channels, four independent chains per thread (the serial chain of 128 dependent loads made this 33 us)
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        int b = grp;
        for (; b + 96 < n_blocks; b += 128) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] += partial[(int64_t)(b + 32 * q) * 288 + k * 32 + c];
        }
        for (; b < n_blocks; b += 32) t[0] += partial[(int64_t)b * 288 + k * 32 + c];
        sh[grp][c] = (t[0] + t[1]) + (t[2] + t[3]);
    }
    __syncthreads();
    if (grp == 0) {
        float r = sh[0][c];
#pragma unroll
        for (int g = 1; g < 32; ++g) r += sh[g][c];
        dw[k * 32 + c] = accumulate ? dw[k * 32 + c] + r : r;
    }
}

// ---- BatchNorm of the pillar feature net in TRAINING mode (BASELINE config 5; semantics of torch.nn.BatchNorm1d over the
// in-range points of ONE sweep -- each sweep is one call of the embedder, oracle/seflow_oracle.py pillar_image(training=True)) ----
// y = feats W for every point of every cell (the forward's fma chain, the forward's ascending order); per-channel float64
// sum / sum of squares as block partials [2][32]
struct PfnBnArgs {
    PillarBwdArgs b;
    double* partial;                 // [gridDim.x][2][32]
    const float* mean; const float* invstd;      // [32] batch statistics (backward)
    const float* coef;               // [2][32]: mean(g), mean(g * xhat) (backward, pass B)
};

// ONE walk over the sweep's cell lists serves the four per-point reductions of the training step (KIND):
//   0  statistics of y (float64 sum, sum of squares)                         -> p.partial [block][2][32]
//   1  sums of g and g * xhat, g = d_image / cnt * [v > 0]                   -> p.partial
//   2  weight gradient with batch statistics: dy = scale * (g - mean(g) - xhat * mean(g xhat)) for EVERY in-range point (the two
//      mean terms reach the points the ReLU masked as well), dW[k][c] = sum f_k dy            -> a.partial [block][9][32]
//   3  weight gradient with frozen statistics: dW[k][c] = sum f_k * g * scale                 -> a.partial
// A block takes 256 consecutive cells at a time: their offsets are fetched by one load per thread and the non-empty ones compacted
// into a list (ascending) that the eight half-waves (32 lanes = 32 channels) work through -- most cells are empty, and walking them
// one dependent load chain at a time made each of these kernels 60-75 us per sweep.  A cell's points are taken in ascending point
// order (order2, written by the forward pass), 32 per gather, and broadcast by shuffle: the per-cell sums have the forward's order;
// no float atomics anywhere.
struct PfnBnBatch { PfnBnArgs s[kMaxSweeps]; };     // blockIdx.y = sweep: the walks of a sample's sweeps share a launch (latency chains on
                                                    // small grids, like the forward stage's kernels)
template <int KIND>
__global__ __launch_bounds__(256) void pfn_walk_kernel(PfnBnBatch m) {
    const PfnBnArgs& p = m.s[blockIdx.y];
    const PillarBwdArgs& a = p.b;
    constexpr int kChunk = 256;
    __shared__ int s_beg[kChunk + 1];
    __shared__ int s_list[kChunk];
    __shared__ int s_wn[4];
    constexpr bool kStats = KIND < 2;
    __shared__ double red_d[kStats ? 2 : 1][kStats ? kCellsPerBlock : 1][32];
    __shared__ float red_f[kStats ? 1 : kCellsPerBlock][kStats ? 1 : 9][32];
    const int sub = threadIdx.x >> 5, c = threadIdx.x & 31, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_cells = a.g.W * a.g.H;
    float w[9], dw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { w[k] = a.pfn_w[k * 32 + c]; dw[k] = 0.f; }
    const float scale = KIND ? a.pfn_scale[c] : 0.f, shift = KIND ? a.pfn_shift[c] : 0.f;
    const float mean = (KIND == 1 || KIND == 2) ? p.mean[c] : 0.f, invstd = (KIND == 1 || KIND == 2) ? p.invstd[c] : 0.f;
    const float k2 = KIND == 2 ? p.coef[c] : 0.f, k3 = KIND == 2 ? p.coef[32 + c] : 0.f;
    double s0 = 0.0, s1 = 0.0;
    for (int cell0 = blockIdx.x * kChunk; cell0 < n_cells; cell0 += gridDim.x * kChunk) {
        s_beg[threadIdx.x] = cell_offset_b(a, min(cell0 + (int)threadIdx.x, n_cells));
        if (threadIdx.x == 0) s_beg[kChunk] = cell_offset_b(a, min(cell0 + kChunk, n_cells));
        __syncthreads();
        const int my_cnt = cell0 + (int)threadIdx.x < n_cells ? s_beg[threadIdx.x + 1] - s_beg[threadIdx.x] : 0;
        const unsigned long long occ = __ballot(my_cnt > 0);
        if (lane == 0) s_wn[wv] = __popcll(occ);
        __syncthreads();
        int base = 0, nlist = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { if (u < wv) base += s_wn[u]; nlist += s_wn[u]; }
        if (my_cnt > 0) s_list[base + __popcll(occ & ((1ull << lane) - 1ull))] = threadIdx.x;
        __syncthreads();
        for (int k = sub; k < nlist; k += kCellsPerBlock) {
            const int lc = s_list[k];
            const int cell = cell0 + lc;
            const int beg = s_beg[lc], cnt = s_beg[lc + 1] - beg;
            const float fc = (float)cnt;
            const float gi = KIND ? a.d_image[(int64_t)cell * a.image_pitch + c] / fc : 0.f;
            const int iy = cell / a.g.W, ix = cell - iy * a.g.W;
            const float ccx = (float)ix * a.g.vx + a.g.cx0, ccy = (float)iy * a.g.vy + a.g.cy0, ccz = 0.f * a.g.vz + a.g.cz0;
            auto point = [&](float x, float y, float z, float mx, float my, float mz) {
                const float f[9] = {x, y, z, x - mx, y - my, z - mz, x - ccx, y - ccy, z - ccz};
                float v = f[0] * w[0];
#pragma unroll
                for (int q = 1; q < 9; ++q) v = fmaf(f[q], w[q], v);
                if (KIND == 0) {
                    s0 += v; s1 += (double)v * v;
                } else if (KIND == 1) {
                    const float g = (v * scale + shift) > 0.f ? gi : 0.f;
                    s0 += g; s1 += (double)g * ((v - mean) * invstd);
                } else if (KIND == 2) {
                    const float g = (v * scale + shift) > 0.f ? gi : 0.f;
                    const float dy = scale * ((g - k2) - ((v - mean) * invstd) * k3);
#pragma unroll
                    for (int q = 0; q < 9; ++q) dw[q] += f[q] * dy;
                } else {
                    if ((v * scale + shift) > 0.f) {
                        const float g = gi * scale;
#pragma unroll
                        for (int q = 0; q < 9; ++q) dw[q] += f[q] * g;
                    }
                }
            };
            if (cnt == 1) {                      // the mean IS the point (x / 1.0f == x)
                const float4 rc = a.cell_rec[beg];
                point(rc.x, rc.y, rc.z, rc.x, rc.y, rc.z);
                continue;
            }
            float sx = 0.f, sy = 0.f, sz = 0.f;
            float ux = 0.f, uy = 0.f, uz = 0.f;
            for (int j0 = 0; j0 < cnt; j0 += 32) {
                const bool have = j0 + c < cnt;
                const float* q = a.xyz_t + (int64_t)(have ? a.order2[beg + j0 + c] : 0) * 3;
                ux = have ? q[0] : 0.f; uy = have ? q[1] : 0.f; uz = have ? q[2] : 0.f;
                const int lim = min(32, cnt - j0);
                for (int j = 0; j < lim; ++j) { sx += shfl32(ux, j); sy += shfl32(uy, j); sz += shfl32(uz, j); }
            }
            const float mx = sx / fc, my = sy / fc, mz = sz / fc;
            for (int j0 = 0; j0 < cnt; j0 += 32) {
                if (cnt > 32) {                  // (a cell of <= 32 points still holds its only chunk)
                    const bool have = j0 + c < cnt;
                    const float* q = a.xyz_t + (int64_t)(have ? a.order2[beg + j0 + c] : 0) * 3;
                    ux = have ? q[0] : 0.f; uy = have ? q[1] : 0.f; uz = have ? q[2] : 0.f;
                }
                const int lim = min(32, cnt - j0);
                for (int j = 0; j < lim; ++j) point(shfl32(ux, j), shfl32(uy, j), shfl32(uz, j), mx, my, mz);
            }
        }
        __syncthreads();                         // s_beg / s_list are rewritten by the next chunk
    }
    if (kStats) {
        red_d[0][sub][c] = s0; red_d[1][sub][c] = s1;
        __syncthreads();
        if (threadIdx.x < 64) {
            const int which = threadIdx.x >> 5;
            double t = red_d[which][0][c];
#pragma unroll
            for (int q = 1; q < kCellsPerBlock; ++q) t += red_d[which][q][c];
            p.partial[((int64_t)blockIdx.x * 2 + which) * 32 + c] = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) red_f[sub][k][c] = dw[k];
        __syncthreads();
        for (int e = threadIdx.x; e < 9 * 32; e += 256) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < kCellsPerBlock; ++q) t += red_f[q][e / 32][e % 32];
            a.partial[(int64_t)blockIdx.x * 288 + e] = t;
        }
    }
}

// one block of 256 threads: 4 groups x 2 sums x 32 channels over the block partials in fixed order, then the constants.
// MODE 0 (forward): batch mean / invstd, the feature kernel's scale / shift, running-statistics update (unbiased variance).
// MODE 1 (backward): dgamma / dbeta (accumulated when asked) and the two means the weight-gradient pass needs.
// the sweeps whose points share one set of statistics (a per-process batch: the b-th sample's sweep of one frame slot, b = 0 .. n - 1)
constexpr int kMaxGroup = 16;
struct PfnGroup { const double* partial[kMaxGroup]; const int* block_sum[kMaxGroup]; int n; };

template <int MODE>
__global__ __launch_bounds__(1024) void pfn_bn_finalize_kernel(PfnGroup grp_src, int n_blocks, int n_cells, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float momentum,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2,
                                                              float* __restrict__ out3, int accumulate) {
    // 1024 threads: 16 groups x 2 sums x 32 channels, four independent chains per thread (a serial chain of 256 dependent loads per
    // thread made this kernel 64 us), fixed-order combine
    __shared__ double sh[2][16][32];
    const int c = threadIdx.x & 31, which = (threadIdx.x >> 5) & 1, grp = threadIdx.x >> 6;
    {
        double all = 0.0;                                    // the group's members in order; one member: 0.0 + x = x, the single-sweep bits
        for (int m = 0; m < grp_src.n; ++m) {
            const double* __restrict__ partial = grp_src.partial[m];
            double t[4] = {0, 0, 0, 0};
            int b = grp;
            for (; b + 48 < n_blocks; b += 64) {
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] += partial[((int64_t)(b + 16 * k) * 2 + which) * 32 + c];
            }
            for (; b < n_blocks; b += 16) t[0] += partial[((int64_t)b * 2 + which) * 32 + c];
            all += (t[0] + t[1]) + (t[2] + t[3]);
        }
        sh[which][grp][c] = all;
    }
    __syncthreads();
    if (threadIdx.x >= 32) return;
    double count = 0.0;                                      // in-range points of the group's sweeps
    for (int m = 0; m < grp_src.n; ++m) count += (double)grp_src.block_sum[m][(n_cells + kScanBlock - 1) / kScanBlock];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { a0 += sh[0][g][c]; a1 += sh[1][g][c]; }
    if (MODE == 0) {
        if (count < 1.0) {                                   // an empty sweep: nothing to normalise, statistics untouched
            const float sc = gamma[c] / sqrtf((running_var ? running_var[c] : 1.f) + eps);
            out0[c] = sc; out1[c] = beta[c] - (running_mean ? running_mean[c] : 0.f) * sc; out2[c] = 0.f; out3[c] = 0.f;
            return;
        }
        const double mean = a0 / count;
        double var = a1 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        out0[c] = sc;                                        // scale
        out1[c] = beta[c] - (float)mean * sc;                // shift
        out2[c] = (float)mean;
        out3[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    } else {
        out0[c] = accumulate ? out0[c] + (float)a1 : (float)a1;      // dgamma = sum g xhat
        out1[c] = accumulate ? out1[c] + (float)a0 : (float)a0;      // dbeta  = sum g
        out2[c] = count > 0.0 ? (float)(a0 / count) : 0.f;           // coef[0]: mean g
        out3[c] = count > 0.0 ? (float)(a1 / count) : 0.f;           // coef[1]: mean g xhat
    }
}

// adjoint of head_gather_kernel: the per-point gradient rows are summed per cell (ascending point order) into the
// image gradients; the cells of groups that are not gathered from, and empty cells, are written as zeros
__global__ __launch_bounds__(256) void head_scatter_kernel(PillarBwdArgs a) {
    const int sub = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int cell = blockIdx.x * kCellsPerBlock + sub;
    if (cell >= a.g.W * a.g.H) return;
    const int beg = cell_offset_b(a, cell);
    const int cnt = cell_offset_b(a, cell + 1) - beg;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int j = 0; j < cnt; ++j) {
        const float* r = a.dhx + (int64_t)a.order2[beg + j] * a.dhx_pitch;
        s0 += r[c]; s1 += r[32 + c]; s2 += r[64 + c]; s3 += r[96 + c];
    }
    float* b0 = a.d_b0 + (int64_t)cell * a.b0_pitch;
    for (int g = 0; g < a.n_groups; ++g) b0[g * 32 + c] = g == a.group0 ? s0 : g == a.group1 ? s1 : 0.f;
    float* dec = a.d_dec + (int64_t)cell * a.dec_pitch;
    dec[c] = s2; dec[32 + c] = s3;
}

}  // namespace himo

using namespace himo;

static size_t ws_cells(int cells) { return round_up((size_t)cells * 4, 16); }
static size_t ws_blocks(int cells) { return round_up(((size_t)(cells + kScanBlock - 1) / kScanBlock + 1) * 4, 16); }
static size_t ws_points(int64_t n) { return round_up((size_t)(n > 0 ? n : 1) * 4, 16); }
static size_t pillar_ws(int64_t n, int cells) { return 2 * ws_cells(cells) + ws_blocks(cells) + 2 * ws_points(n) + 3 * ws_points(n); }
// occupancy bitmap of the incremental images: one bit per cell, kept in the LAST bytes of the caller's per-sweep workspace
// (a position that does not depend on the sweep's point count, unlike the lists in front of it)
static size_t ws_occ(int cells) { return round_up(((size_t)cells + 63) / 64 * 8, 16); }

extern "C" size_t himo_pillar_workspace_bytes(int64_t max_points, int grid_w, int grid_h) {
    return pillar_ws(max_points, grid_w * grid_h) + 64 + ws_occ(grid_w * grid_h);
}

extern "C" int himo_pillar_occupancy_reset(void* d_workspace, size_t workspace_bytes, int grid_w, int grid_h, void* stream) {
    if (!d_workspace || grid_w < 1 || grid_h < 1 || workspace_bytes < ws_occ(grid_w * grid_h)) return HIMO_ERR_INVALID_ARGUMENT;
    const size_t nb = ws_occ(grid_w * grid_h);
    HIMO_HIP(hipMemsetAsync(reinterpret_cast<char*>(d_workspace) + (workspace_bytes - nb), 0xFF, nb, (hipStream_t)stream));
    return HIMO_OK;
}

// validate one sweep's arguments and fill its kernel argument block
static int pillar_args(PillarArgs& a, int64_t n, const float* d_pts, int pc_stride, const float* h_transform, const float* h_range,
                       const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                       const float* d_pfn_scale, const float* d_pfn_shift, float* d_xyz_t, int32_t* d_pid, float* d_offsets,
                       float* d_image, int image_pitch, void* d_workspace, size_t workspace_bytes) {
    if (n < 0 || pc_stride < 3 || grid_w < 1 || grid_h < 1 || !h_transform || !h_range || !h_voxel || !h_centre_offset)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (!d_pfn_weight || !d_pfn_scale || !d_pfn_shift || !d_image || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!d_pts || !d_xyz_t || !d_pid || !d_offsets)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n > 0x7fffffff || image_pitch < 32) return HIMO_ERR_UNSUPPORTED;
    if ((image_pitch & 3) || (reinterpret_cast<uintptr_t>(d_image) & 15)) return HIMO_ERR_UNSUPPORTED;     // 16-byte row stores
    const int cells = grid_w * grid_h;
    if (workspace_bytes < pillar_ws(n, cells) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    if ((cells + kScanBlock - 1) / kScanBlock > 1024) return HIMO_ERR_UNSUPPORTED;   // grids beyond 1M cells need a third scan level
    a = PillarArgs{};
    a.n = n; a.pts = d_pts; a.stride = pc_stride;
    for (int i = 0; i < 9; ++i) a.R[i] = h_transform[(i / 3) * 4 + (i % 3)];
    for (int i = 0; i < 3; ++i) a.t[i] = h_transform[i * 4 + 3];
    a.g.xmin = h_range[0]; a.g.ymin = h_range[1]; a.g.zmin = h_range[2];
    a.g.vx = h_voxel[0]; a.g.vy = h_voxel[1]; a.g.vz = h_voxel[2];
    a.g.cx0 = h_centre_offset[0]; a.g.cy0 = h_centre_offset[1]; a.g.cz0 = h_centre_offset[2];
    a.g.W = grid_w; a.g.H = grid_h;
    a.pfn_w = d_pfn_weight; a.pfn_scale = d_pfn_scale; a.pfn_shift = d_pfn_shift;
    a.xyz_t = d_xyz_t; a.pid = d_pid; a.offsets = d_offsets; a.image = d_image; a.image_pitch = image_pitch;
    char* ws = reinterpret_cast<char*>(d_workspace);
    a.cell_count = reinterpret_cast<int*>(ws);
    a.cell_cursor = reinterpret_cast<int*>(ws + ws_cells(cells));
    a.block_sum = reinterpret_cast<int*>(ws + 2 * ws_cells(cells));
    a.cell_rec = reinterpret_cast<float4*>(ws + 2 * ws_cells(cells) + ws_blocks(cells));
    a.order2 = reinterpret_cast<int*>(ws + 2 * ws_cells(cells) + ws_blocks(cells) + 4 * ws_points(n));
    return HIMO_OK;
}

// the stage's five launches over `count` sweeps (same grid size for all of them)
static int pillar_launch(const PillarBatch& m, int count, hipStream_t s) {
    const int cells = m.s[0].g.W * m.s[0].g.H;
    const int nblk = (cells + kScanBlock - 1) / kScanBlock;
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i)
        if (m.s[i].n > nmax) nmax = m.s[i].n;
    {
        const int n_vec4 = (int)(ws_cells(cells) / 16);            // the histogram only: the cursor array behind it is no longer used
        hipLaunchKernelGGL(pillar_clear_kernel, dim3(std::min((n_vec4 + 255) / 256, 512), count), dim3(256), 0, s, m, n_vec4);
    }
    HIMO_LAUNCH_CHECK("pillar_clear_kernel");
    const unsigned pblocks = (unsigned)((nmax + 255) / 256);
    if (nmax > 0) {
        ProfScope ps("pillar_assign_kernel", s);
        hipLaunchKernelGGL(pillar_assign_kernel, dim3(pblocks, count), dim3(256), 0, s, m);
    }
    HIMO_LAUNCH_CHECK("pillar_assign_kernel");
    {
        ProfScope ps("cell_scan_kernels", s);
        hipLaunchKernelGGL(cell_scan_local_kernel, dim3(nblk, count), dim3(256), 0, s, m);
        hipLaunchKernelGGL(cell_scan_top_kernel, dim3(1, count), dim3(1024), 0, s, m, nblk);
    }
    HIMO_LAUNCH_CHECK("cell_scan_kernels");
    if (nmax > 0) {
        ProfScope ps("pillar_fill_kernel", s);
        hipLaunchKernelGGL(pillar_fill_kernel, dim3(pblocks, count), dim3(256), 0, s, m);
    }
    HIMO_LAUNCH_CHECK("pillar_fill_kernel");
    {
        ProfScope ps("pillar_feature_kernel", s);
        hipLaunchKernelGGL(pillar_feature_kernel, dim3((cells + kFeatCells - 1) / kFeatCells, count), dim3(256), 0, s, m);
    }
    HIMO_LAUNCH_CHECK("pillar_feature_kernel");
    return HIMO_OK;
}

extern "C" int himo_pillarize(int64_t n, const float* d_pts, int pc_stride, const float* h_transform,
                              const float* h_range, const float* h_voxel, const float* h_centre_offset,
                              int grid_w, int grid_h,
                              const float* d_pfn_weight, const float* d_pfn_scale, const float* d_pfn_shift,
                              float* d_xyz_t, int32_t* d_pid, float* d_offsets, float* d_image, int image_pitch,
                              void* d_workspace, size_t workspace_bytes, void* stream) {
    PillarBatch m{};
    const int st = pillar_args(m.s[0], n, d_pts, pc_stride, h_transform, h_range, h_voxel, h_centre_offset, grid_w, grid_h, d_pfn_weight,
                               d_pfn_scale, d_pfn_shift, d_xyz_t, d_pid, d_offsets, d_image, image_pitch, d_workspace, workspace_bytes);
    if (st != HIMO_OK) return st;
    return pillar_launch(m, 1, (hipStream_t)stream);
}

// several sweeps of one sample (history, pc0, pc1) in the same five launches; every sweep has its own outputs, image
// channel group and workspace (workspace_bytes each)
extern "C" int himo_pillarize_multi(int n_sweeps, const himo_sweep* h_sweeps, const float* h_range, const float* h_voxel,
                                    const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                                    const float* d_pfn_scale, const float* d_pfn_shift, int image_pitch, size_t workspace_bytes,
                                    void* stream) {
    return himo_pillarize_multi_ex(n_sweeps, h_sweeps, h_range, h_voxel, h_centre_offset, grid_w, grid_h, d_pfn_weight, d_pfn_scale,
                                   d_pfn_shift, image_pitch, workspace_bytes, 0, stream);
}

extern "C" int himo_pillarize_multi_ex(int n_sweeps, const himo_sweep* h_sweeps, const float* h_range, const float* h_voxel,
                                       const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                                       const float* d_pfn_scale, const float* d_pfn_shift, int image_pitch, size_t workspace_bytes,
                                       int image_split, void* stream) {
    const bool incremental = (image_split & 2) != 0;                  // HIMO_IMAGE_INCREMENTAL
    image_split &= 1;
    if (image_split && (image_pitch & 15)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n_sweeps < 1 || n_sweeps > kMaxSweeps || !h_sweeps) return HIMO_ERR_INVALID_ARGUMENT;
    if (incremental && ((workspace_bytes & 15) || workspace_bytes < ws_occ(grid_w * grid_h))) return HIMO_ERR_WORKSPACE;
    PillarBatch m{};
    for (int i = 0; i < n_sweeps; ++i) {
        const himo_sweep& w = h_sweeps[i];
        const int st = pillar_args(m.s[i], w.n, w.d_pts, w.pc_stride, w.transform, h_range, h_voxel, h_centre_offset, grid_w, grid_h,
                                   d_pfn_weight, d_pfn_scale, d_pfn_shift, w.d_xyz_t, w.d_pid, w.d_offsets, w.d_image, image_pitch,
                                   w.d_workspace, workspace_bytes);
        if (st != HIMO_OK) return st;
        if (image_split && (reinterpret_cast<uintptr_t>(w.d_image) & 63)) return HIMO_ERR_INVALID_ARGUMENT;
        m.s[i].image_split = image_split ? 1 : 0;
        if (incremental) {
            const size_t nb = ws_occ(grid_w * grid_h);
            if (workspace_bytes < pillar_ws(w.n, grid_w * grid_h) + nb) return HIMO_ERR_WORKSPACE;
            m.s[i].occ = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w.d_workspace) + (workspace_bytes - nb));
        }
        for (int j = 0; j < i; ++j)
            if (h_sweeps[j].d_workspace == w.d_workspace) return HIMO_ERR_INVALID_ARGUMENT;     // one workspace per sweep
    }
    return pillar_launch(m, n_sweeps, (hipStream_t)stream);
}

static void carve_bwd(PillarBwdArgs& a, int64_t n, int cells, void* d_workspace) {
    char* ws = reinterpret_cast<char*>(d_workspace);
    a.cell_count = reinterpret_cast<const int*>(ws);
    a.block_sum = reinterpret_cast<const int*>(ws + 2 * ws_cells(cells));
    a.cell_rec = reinterpret_cast<const float4*>(ws + 2 * ws_cells(cells) + ws_blocks(cells));
    a.order2 = reinterpret_cast<const int*>(ws + 2 * ws_cells(cells) + ws_blocks(cells) + 4 * ws_points(n));
}

constexpr int kPfnBwdBlocks = 1024;

extern "C" size_t himo_pfn_backward_workspace_bytes(void) { return (size_t)kPfnBwdBlocks * 288 * 4 + 64; }

// d_pillar_workspace: the workspace himo_pillarize(n, ...) of the SAME sweep left behind (cell lists), untouched since
extern "C" int himo_pfn_backward(int64_t n, const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                                 const float* d_pfn_weight, const float* d_pfn_scale, const float* d_pfn_shift,
                                 const float* d_xyz_t, const void* d_pillar_workspace, const float* d_dimage, int image_pitch,
                                 float* d_dweight, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || !h_voxel || !h_centre_offset || grid_w < 1 || grid_h < 1 || !d_pfn_weight || !d_pfn_scale || !d_pfn_shift ||
        !d_pillar_workspace || !d_dimage || !d_dweight || !d_workspace || image_pitch < 32)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (n > 0 && !d_xyz_t) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_pfn_backward_workspace_bytes()) return HIMO_ERR_WORKSPACE;
    PillarBwdArgs a{};
    a.g.vx = h_voxel[0]; a.g.vy = h_voxel[1]; a.g.vz = h_voxel[2];
    a.g.cx0 = h_centre_offset[0]; a.g.cy0 = h_centre_offset[1]; a.g.cz0 = h_centre_offset[2];
    a.g.W = grid_w; a.g.H = grid_h;
    carve_bwd(a, n, grid_w * grid_h, const_cast<void*>(d_pillar_workspace));
    a.xyz_t = d_xyz_t; a.pfn_w = d_pfn_weight; a.pfn_scale = d_pfn_scale; a.pfn_shift = d_pfn_shift;
    a.d_image = d_dimage; a.image_pitch = image_pitch;
    a.partial = reinterpret_cast<float*>(d_workspace);
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope ps("pfn_backward_kernel", s);
        PfnBnBatch m{};
        m.s[0].b = a;
        hipLaunchKernelGGL(pfn_walk_kernel<3>, dim3(kPfnBwdBlocks), dim3(256), 0, s, m);
    }
    hipLaunchKernelGGL(pfn_backward_reduce_kernel, dim3(9), dim3(1024), 0, s, a.partial, kPfnBwdBlocks, d_dweight, (flags & 1u) ? 1 : 0);
    HIMO_LAUNCH_CHECK("pfn_backward kernels");
    return HIMO_OK;
}

// ---- training-mode BatchNorm of the pillar feature net: host side ---------------------------------------------------------
// workspace: the 1024 x 288 floats of himo_pfn_backward + block partials of the two reductions + 64 floats of coefficients
extern "C" size_t himo_pfn_bn_workspace_bytes(void) {
    return round_up((size_t)kPfnBwdBlocks * 288 * 4, 64) + (size_t)kPfnBwdBlocks * 2 * 32 * sizeof(double) + 64 * sizeof(float) + 64;
}

static int pfn_bn_args(PfnBnArgs& p, int64_t n, const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                       const float* d_pfn_weight, const float* d_xyz_t, const void* d_pillar_workspace, void* d_workspace,
                       size_t workspace_bytes) {
    if (n < 0 || !h_voxel || !h_centre_offset || grid_w < 1 || grid_h < 1 || !d_pfn_weight || !d_pillar_workspace || !d_workspace)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (n > 0 && !d_xyz_t) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_pfn_bn_workspace_bytes() || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    p = PfnBnArgs{};
    PillarBwdArgs& a = p.b;
    a.g.vx = h_voxel[0]; a.g.vy = h_voxel[1]; a.g.vz = h_voxel[2];
    a.g.cx0 = h_centre_offset[0]; a.g.cy0 = h_centre_offset[1]; a.g.cz0 = h_centre_offset[2];
    a.g.W = grid_w; a.g.H = grid_h;
    carve_bwd(a, n, grid_w * grid_h, const_cast<void*>(d_pillar_workspace));
    a.xyz_t = d_xyz_t; a.pfn_w = d_pfn_weight;
    a.partial = reinterpret_cast<float*>(d_workspace);
    p.partial = reinterpret_cast<double*>(reinterpret_cast<char*>(d_workspace) + round_up((size_t)kPfnBwdBlocks * 288 * 4, 64));
    return HIMO_OK;
}

// Batch statistics of y = feats W over the in-range points of each GROUP of sweeps (sweep i belongs to group i % n_groups: with the sweeps
// of a per-process batch in the order sample-major, frame-minor, group f = frame slot f of every sample -- what ONE call of the pillar
// net on a batch of sweeps normalises over) whose cell lists himo_pillarize* left in their d_pillar_workspace -> d_scale / d_shift (what
// the feature kernel and the backward pass consume: gamma invstd, beta - mean gamma invstd), d_mean / d_invstd (saved for the backward
// pass), rows of [n_groups][32] arrays; running statistics updated in place group after group (momentum, unbiased variance; NULL: not
// tracked).  Walk launches of up to 12 sweeps each; the per-group finalize kernels follow in order.  Workspace: n_sweeps x
// himo_pfn_bn_workspace_bytes().
extern "C" int himo_pfn_bn_stats_groups(int n_sweeps, int n_groups, const int64_t* h_n, const float* const* h_xyz_t,
                                        const void* const* h_pillar_workspace, const float* h_voxel, const float* h_centre_offset, int grid_w,
                                        int grid_h, const float* d_pfn_weight, const float* d_gamma, const float* d_beta, float eps,
                                        float momentum, float* d_running_mean, float* d_running_var, float* d_scale, float* d_shift,
                                        float* d_mean, float* d_invstd, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n_sweeps < 1 || n_groups < 1 || n_sweeps % n_groups || n_sweeps / n_groups > kMaxGroup || !h_n || !h_xyz_t || !h_pillar_workspace)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (!d_gamma || !d_beta || !d_scale || !d_shift || !d_mean || !d_invstd || (d_running_mean == nullptr) != (d_running_var == nullptr))
        return HIMO_ERR_INVALID_ARGUMENT;
    const size_t one = himo_pfn_bn_workspace_bytes();
    if (workspace_bytes < one * n_sweeps) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("pfn_bn_stats_kernel", s);
    PfnGroup groups[kMaxSweeps * kMaxGroup];
    if (n_groups > kMaxSweeps * kMaxGroup) return HIMO_ERR_UNSUPPORTED;
    for (int g = 0; g < n_groups; ++g) groups[g].n = 0;
    for (int lo = 0; lo < n_sweeps; lo += kMaxSweeps) {
        const int cnt = n_sweeps - lo < kMaxSweeps ? n_sweeps - lo : kMaxSweeps;
        PfnBnBatch m{};
        for (int j = 0; j < cnt; ++j) {
            const int i = lo + j;
            const int st = pfn_bn_args(m.s[j], h_n[i], h_voxel, h_centre_offset, grid_w, grid_h, d_pfn_weight, h_xyz_t[i], h_pillar_workspace[i],
                                       reinterpret_cast<char*>(d_workspace) + one * i, one);
            if (st != HIMO_OK) return st;
            PfnGroup& G = groups[i % n_groups];
            G.partial[G.n] = m.s[j].partial; G.block_sum[G.n] = m.s[j].b.block_sum; ++G.n;
        }
        hipLaunchKernelGGL(pfn_walk_kernel<0>, dim3(kPfnBwdBlocks, cnt), dim3(256), 0, s, m);
    }
    for (int g = 0; g < n_groups; ++g)
        hipLaunchKernelGGL(pfn_bn_finalize_kernel<0>, dim3(1), dim3(1024), 0, s, groups[g], kPfnBwdBlocks, grid_w * grid_h, d_gamma, d_beta, eps,
                           momentum, d_running_mean, d_running_var, d_scale + 32 * g, d_shift + 32 * g, d_mean + 32 * g, d_invstd + 32 * g, 0);
    HIMO_LAUNCH_CHECK("pfn_bn_stats kernels");
    return HIMO_OK;
}

// every sweep its own group: statistics per sweep (one call of the pillar net per sweep)
extern "C" int himo_pfn_bn_stats_multi(int n_sweeps, const int64_t* h_n, const float* const* h_xyz_t, const void* const* h_pillar_workspace,
                                       const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                                       const float* d_pfn_weight, const float* d_gamma, const float* d_beta, float eps, float momentum,
                                       float* d_running_mean, float* d_running_var, float* d_scale, float* d_shift, float* d_mean,
                                       float* d_invstd, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n_sweeps > kMaxSweeps) return HIMO_ERR_INVALID_ARGUMENT;
    return himo_pfn_bn_stats_groups(n_sweeps, n_sweeps, h_n, h_xyz_t, h_pillar_workspace, h_voxel, h_centre_offset, grid_w, grid_h, d_pfn_weight,
                                    d_gamma, d_beta, eps, momentum, d_running_mean, d_running_var, d_scale, d_shift, d_mean, d_invstd,
                                    d_workspace, workspace_bytes, stream);
}

extern "C" int himo_pfn_bn_stats(int64_t n, const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                                 const float* d_pfn_weight, const float* d_xyz_t, const void* d_pillar_workspace, const float* d_gamma,
                                 const float* d_beta, float eps, float momentum, float* d_running_mean, float* d_running_var,
                                 float* d_scale, float* d_shift, float* d_mean, float* d_invstd, void* d_workspace, size_t workspace_bytes,
                                 void* stream) {
    return himo_pfn_bn_stats_multi(1, &n, &d_xyz_t, &d_pillar_workspace, h_voxel, h_centre_offset, grid_w, grid_h, d_pfn_weight, d_gamma,
                                   d_beta, eps, momentum, d_running_mean, d_running_var, d_scale, d_shift, d_mean, d_invstd, d_workspace,
                                   workspace_bytes, stream);
}

// The feature kernel alone, with PER-SWEEP BatchNorm constants (d_scale / d_shift: [n_sweeps][32]): the second half of a
// training-mode pillar stage -- himo_pillarize_multi_ex built the cell lists (its images, written with stale constants, are
// overwritten here), himo_pfn_bn_stats turned them into each sweep's constants.
extern "C" int himo_pillar_features_multi(int n_sweeps, const himo_sweep* h_sweeps, const float* h_range, const float* h_voxel,
                                          const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                                          const float* d_scale, const float* d_shift, int image_pitch, size_t workspace_bytes,
                                          int image_split, void* stream) {
    const bool incremental = (image_split & 2) != 0;
    image_split &= 1;
    if (image_split && (image_pitch & 15)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n_sweeps < 1 || n_sweeps > kMaxSweeps || !h_sweeps || !d_scale || !d_shift) return HIMO_ERR_INVALID_ARGUMENT;
    if (incremental && ((workspace_bytes & 15) || workspace_bytes < ws_occ(grid_w * grid_h))) return HIMO_ERR_WORKSPACE;
    PillarBatch m{};
    for (int i = 0; i < n_sweeps; ++i) {
        const himo_sweep& w = h_sweeps[i];
        const int st = pillar_args(m.s[i], w.n, w.d_pts, w.pc_stride, w.transform, h_range, h_voxel, h_centre_offset, grid_w, grid_h,
                                   d_pfn_weight, d_scale + 32 * i, d_shift + 32 * i, w.d_xyz_t, w.d_pid, w.d_offsets, w.d_image, image_pitch,
                                   w.d_workspace, workspace_bytes);
        if (st != HIMO_OK) return st;
        if (image_split && (reinterpret_cast<uintptr_t>(w.d_image) & 63)) return HIMO_ERR_INVALID_ARGUMENT;
        m.s[i].image_split = image_split ? 1 : 0;
        if (incremental) {
            const size_t nb = ws_occ(grid_w * grid_h);
            if (workspace_bytes < pillar_ws(w.n, grid_w * grid_h) + nb) return HIMO_ERR_WORKSPACE;
            m.s[i].occ = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w.d_workspace) + (workspace_bytes - nb));
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const int cells = grid_w * grid_h;
    {
        ProfScope ps("pillar_feature_kernel", s);
        hipLaunchKernelGGL(pillar_feature_kernel, dim3((cells + kFeatCells - 1) / kFeatCells, n_sweeps), dim3(256), 0, s, m);
    }
    HIMO_LAUNCH_CHECK("pillar_feature_kernel");
    return HIMO_OK;
}

// himo_pfn_backward with batch statistics, for all sweeps at once: d loss / d pfn.weight, d gamma, d beta [32] SUMMED over the sweeps in
// order (flags bit 0: added to what the three hold).  Sweep i belongs to group i % n_groups (himo_pfn_bn_stats_groups); d_scale / d_shift
// / d_mean / d_invstd: rows of [n_groups][32] arrays as it left them.  The two means of the BatchNorm backward (mean g, mean g xhat) are
// taken over a group's points.  Walk launches of up to 12 sweeps; the per-group finalize and per-sweep reduce kernels follow in order,
// so with every sweep its own group the result has the bits of n_sweeps single calls.
extern "C" int himo_pfn_backward_bn_groups(int n_sweeps, int n_groups, const int64_t* h_n, const float* const* h_xyz_t,
                                           const void* const* h_pillar_workspace, const float* const* h_dimage, int image_pitch,
                                           const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h, const float* d_pfn_weight,
                                           const float* d_scale, const float* d_shift, const float* d_mean, const float* d_invstd,
                                           float* d_dweight, float* d_dgamma, float* d_dbeta, unsigned flags, void* d_workspace,
                                           size_t workspace_bytes, void* stream) {
    if (n_sweeps < 1 || n_groups < 1 || n_sweeps % n_groups || n_sweeps / n_groups > kMaxGroup || n_groups > kMaxSweeps * kMaxGroup ||
        n_sweeps > kMaxSweeps * kMaxGroup || !h_n || !h_xyz_t || !h_pillar_workspace || !h_dimage)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (!d_scale || !d_shift || !d_mean || !d_invstd || !d_dweight || !d_dgamma || !d_dbeta || image_pitch < 32) return HIMO_ERR_INVALID_ARGUMENT;
    const size_t one = himo_pfn_bn_workspace_bytes();
    if (workspace_bytes < one * n_sweeps) return HIMO_ERR_WORKSPACE;
    static thread_local PfnBnArgs args[kMaxSweeps * kMaxGroup];
    PfnGroup groups[kMaxSweeps * kMaxGroup];
    for (int g = 0; g < n_groups; ++g) groups[g].n = 0;
    for (int i = 0; i < n_sweeps; ++i) {
        if (!h_dimage[i]) return HIMO_ERR_INVALID_ARGUMENT;
        PfnBnArgs& p = args[i];
        const int st = pfn_bn_args(p, h_n[i], h_voxel, h_centre_offset, grid_w, grid_h, d_pfn_weight, h_xyz_t[i], h_pillar_workspace[i],
                                   reinterpret_cast<char*>(d_workspace) + one * i, one);
        if (st != HIMO_OK) return st;
        const int g = i % n_groups;
        p.b.pfn_scale = d_scale + 32 * g; p.b.pfn_shift = d_shift + 32 * g; p.b.d_image = h_dimage[i]; p.b.image_pitch = image_pitch;
        p.mean = d_mean + 32 * g; p.invstd = d_invstd + 32 * g;
        // the group's coefficients live behind the partials of its FIRST sweep's workspace block
        p.coef = i < n_groups ? reinterpret_cast<float*>(p.partial + (size_t)kPfnBwdBlocks * 2 * 32) : args[g].coef;
        PfnGroup& G = groups[g];
        G.partial[G.n] = p.partial; G.block_sum[G.n] = p.b.block_sum; ++G.n;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("pfn_backward_kernel", s);
    auto walk = [&](auto kernel) {
        for (int lo = 0; lo < n_sweeps; lo += kMaxSweeps) {
            const int cnt = n_sweeps - lo < kMaxSweeps ? n_sweeps - lo : kMaxSweeps;
            PfnBnBatch m{};
            for (int j = 0; j < cnt; ++j) m.s[j] = args[lo + j];
            hipLaunchKernelGGL(kernel, dim3(kPfnBwdBlocks, cnt), dim3(256), 0, s, m);
        }
    };
    walk(pfn_walk_kernel<1>);
    for (int g = 0; g < n_groups; ++g) {
        float* coef = const_cast<float*>(args[g].coef);
        hipLaunchKernelGGL(pfn_bn_finalize_kernel<1>, dim3(1), dim3(1024), 0, s, groups[g], kPfnBwdBlocks, grid_w * grid_h, (const float*)nullptr,
                           (const float*)nullptr, 0.f, 0.f, (float*)nullptr, (float*)nullptr, d_dgamma, d_dbeta, coef, coef + 32,
                           (g > 0 || (flags & 1u)) ? 1 : 0);
    }
    walk(pfn_walk_kernel<2>);
    for (int i = 0; i < n_sweeps; ++i)
        hipLaunchKernelGGL(pfn_backward_reduce_kernel, dim3(9), dim3(1024), 0, s, args[i].b.partial, kPfnBwdBlocks, d_dweight,
                           (i > 0 || (flags & 1u)) ? 1 : 0);
    HIMO_LAUNCH_CHECK("pfn_backward_bn kernels");
    return HIMO_OK;
}

extern "C" int himo_pfn_backward_bn_multi(int n_sweeps, const int64_t* h_n, const float* const* h_xyz_t, const void* const* h_pillar_workspace,
                                          const float* const* h_dimage, int image_pitch, const float* h_voxel, const float* h_centre_offset,
                                          int grid_w, int grid_h, const float* d_pfn_weight, const float* d_scale, const float* d_shift,
                                          const float* d_mean, const float* d_invstd, float* d_dweight, float* d_dgamma, float* d_dbeta,
                                          unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n_sweeps > kMaxSweeps) return HIMO_ERR_INVALID_ARGUMENT;
    return himo_pfn_backward_bn_groups(n_sweeps, n_sweeps, h_n, h_xyz_t, h_pillar_workspace, h_dimage, image_pitch, h_voxel, h_centre_offset,
                                       grid_w, grid_h, d_pfn_weight, d_scale, d_shift, d_mean, d_invstd, d_dweight, d_dgamma, d_dbeta, flags,
                                       d_workspace, workspace_bytes, stream);
}

extern "C" int himo_pfn_backward_bn(int64_t n, const float* h_voxel, const float* h_centre_offset, int grid_w, int grid_h,
                                    const float* d_pfn_weight, const float* d_scale, const float* d_shift, const float* d_mean,
                                    const float* d_invstd, const float* d_xyz_t, const void* d_pillar_workspace, const float* d_dimage,
                                    int image_pitch, float* d_dweight, float* d_dgamma, float* d_dbeta, unsigned flags, void* d_workspace,
                                    size_t workspace_bytes, void* stream) {
    return himo_pfn_backward_bn_multi(1, &n, &d_xyz_t, &d_pillar_workspace, &d_dimage, image_pitch, h_voxel, h_centre_offset, grid_w, grid_h,
                                      d_pfn_weight, d_scale, d_shift, d_mean, d_invstd, d_dweight, d_dgamma, d_dbeta, flags, d_workspace,
                                      workspace_bytes, stream);
}

extern "C" int himo_head_scatter(int64_t n, int grid_w, int grid_h, const void* d_pillar_workspace, const float* d_dhx, int dhx_pitch,
                                 float* d_db0, int b0_pitch, int group0, int group1, int n_groups, float* d_ddec, int dec_pitch,
                                 void* stream) {
    if (n < 0 || grid_w < 1 || grid_h < 1 || !d_pillar_workspace || !d_db0 || !d_ddec || dhx_pitch < 128 || n_groups < 1 ||
        b0_pitch < 32 * n_groups || dec_pitch < 64 || (n > 0 && !d_dhx))
        return HIMO_ERR_INVALID_ARGUMENT;
    PillarBwdArgs a{};
    a.g.W = grid_w; a.g.H = grid_h;
    carve_bwd(a, n, grid_w * grid_h, const_cast<void*>(d_pillar_workspace));
    a.dhx = d_dhx; a.dhx_pitch = dhx_pitch; a.d_b0 = d_db0; a.b0_pitch = b0_pitch; a.group0 = group0; a.group1 = group1;
    a.n_groups = n_groups; a.d_dec = d_ddec; a.dec_pitch = dec_pitch;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("head_scatter_kernel", s);
    hipLaunchKernelGGL(head_scatter_kernel, dim3((grid_w * grid_h + kCellsPerBlock - 1) / kCellsPerBlock), dim3(256), 0, s, a);
    HIMO_LAUNCH_CHECK("head_scatter_kernel");
    return HIMO_OK;
}
