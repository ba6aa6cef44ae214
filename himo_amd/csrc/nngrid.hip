// nngrid.hip -- exact k=1 nearest neighbour between two full LiDAR sweeps (~120k x 120k points) through
// a uniform BEV cell grid: the correspondence search of the self-supervised Chamfer losses (stage a11).
//
// The reference's loss lives in the absent OpenSceneFlow submodule (in-tree fact:
// `loss_fn=seflowppLoss` with `chamfer_dis`, `dynamic_chamfer_dis`, `cluster_based_pc0pc1` terms,
// assets/slurm/ssl-train-av2.sh:33).  This is this build's own design; the oracle is scipy's cKDTree.
//
// Why a grid here and brute force in nn.hip: per-instance sets (10..5000 points) are best swept
// exhaustively, but sweep-to-sweep search is 1.4e10 pairs (~2 ms) five times per training sample.
//
//   build:  BOTH point sets of a search are binned: cell histogram (integer atomics) -> exclusive scan -> counting-sort
//           scatter of the rows (xyz + original row packed as float4: a candidate is ONE 16-byte load).  The searched set is
//           kept twice, in row-major and in column-major cell order, so that ANY axis-aligned line of cells -- a row OR a
//           column of the grid -- is one contiguous run of candidates.
//   query:  WAVE-COOPERATIVE (round 5).  A 256-thread block owns 64 consecutive queries of the cell-sorted query set: one
//           query per lane, the same 64 in each of its four waves.  Neighbouring sorted queries share cells, so the block walks
//           Chebyshev rings around the RECTANGLE of cells its active queries occupy (a stretch of one grid row): ring 0 is
//           one run, every later ring is exactly four runs (top row, bottom row, left column, right column).  Run bounds and
//           candidates are wave-uniform: they arrive through the scalar data cache into SGPRs (no per-lane address, no
//           divergence), every lane tests every candidate of its wave against its own query, and the four waves take
//           alternate 8-candidate chunks of the ring.  After a ring the four waves merge (distance, row) through LDS; a lane
//           retires when its best distance is within the ring's guaranteed reach; the block leaves the segment when all its
//           lanes have.  Queries of the block that sit in another row / beyond 16 cells form the next segment.
//
// Why not one lane per query walking its own rings (rounds 1-4): a lane's walk is a chain of dependent loads whose length is
// its candidate count, and a wave waits for its longest lane.  On the uniform bench cloud every lane meets ~100 candidates;
// on a LiDAR-shaped sweep (himo_amd.synthetic.lidar_rings) cells near the sensor hold 600 points and a quarter of the queries
// (surfaces the other sweep does not see) must walk 3-19 m: 500 candidates on average, 12 000 for the worst lane, and the
// per-lane kernel went from 110 us to 1620 us per search (profiles/r05_train_rings_rocprofv3_kernel_stats_before.csv).
// Here the longest chain is a quarter of a ring's candidates at ~40 cycles each, shared by 64 queries.
//
// Ties keep the lowest reference row (the rule of nn.hip), independent of the order the candidates arrive in: the result
// does not depend on the atomics' order inside the counting sort.
#include "nngrid.h"
#include <math.h>

namespace himo {

constexpr int kNngWaves = 8;          // waves of a query block (a power of two): they share a ring's candidates
constexpr int kNngSegCells = 16;      // widest stretch of a grid row one segment covers
constexpr int kNngScanChunk = 4096;   // cells per iteration of the one-block scan

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ inline void cell_of(const NnGrid& g, float x, float y, int& cx, int& cy) {
    // points outside the grid are binned into the border cells (the ring bound below stays valid
    // because it is computed from the clamped cell geometry and the true query position)
    cx = clampi((int)floorf((x - g.x0) * g.inv_cell), 0, g.gw - 1);
    cy = clampi((int)floorf((y - g.y0) * g.inv_cell), 0, g.gh - 1);
}

struct NngArgs {
    NngSet set[kNngMaxSets];
    NngJob job[kNngMaxJobs];
    NnGrid g;
};

__global__ __launch_bounds__(256) void nng_count_kernel(NngArgs a) {
    const NngSet& S = a.set[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S.n) return;
    int cx, cy;
    cell_of(a.g, S.pts[(size_t)i * 3], S.pts[(size_t)i * 3 + 1], cx, cy);
    const int c = cy * a.g.gw + cx;
    S.cell_id[i] = c;
    atomicAdd(&S.offset[c], 1);
    if (S.searched) atomicAdd(&S.offset_t[cx * a.g.gh + cy], 1);
}

// in-place exclusive scan of one histogram per block (block 2k: set k row-major, 2k + 1: set k column-major); v[cells] = total
__global__ __launch_bounds__(1024) void nng_scan_kernel(NngArgs a, int cells) {
    const NngSet& S = a.set[blockIdx.x >> 1];
    if ((blockIdx.x & 1) && !S.searched) return;
    int* v = (blockIdx.x & 1) ? S.offset_t : S.offset;
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < cells; base += kNngScanChunk) {
        const int at = base + threadIdx.x * 4;
        int x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = at + k < cells ? v[at + k] : 0;
        const int mine = x[0] + x[1] + x[2] + x[3];
        int incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off, 64);
            if (lane >= off) incl += y;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int run = carry + incl - mine, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int t = wsum[k]; total += t; if (k < w) run += t; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (at + k < cells) v[at + k] = run; run += x[k]; }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) v[cells] = carry;
}

__global__ __launch_bounds__(256) void nng_fill_kernel(NngArgs a) {
    const NngSet& S = a.set[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S.n) return;
    const int c = S.cell_id[i];
    const float4 p = make_float4(S.pts[(size_t)i * 3], S.pts[(size_t)i * 3 + 1], S.pts[(size_t)i * 3 + 2], __int_as_float(i));
    S.sorted[S.offset[c] + atomicAdd(&S.cursor[c], 1)] = p;
    if (S.searched) {
        const int cy = c / a.g.gw, cx = c - cy * a.g.gw, ct = cx * a.g.gh + cy;
        S.sorted_t[S.offset_t[ct] + atomicAdd(&S.cursor_t[ct], 1)] = p;
    }
}

// (squared distance, row) as ONE unsigned 64-bit key: distances are >= +0, so their float bits order like the values, and the
// row in the low word breaks ties towards the lowest row; "no neighbour yet" = (+inf, 0xffffffff).  One compare + two selects
// per candidate instead of three compares.
__device__ inline unsigned long long nng_key(float d, int row) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)row;
}
constexpr unsigned long long kNngNone = 0x7f800000ffffffffull;
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64 * kNngWaves) void nng_query_kernel(NngArgs a) {
    __shared__ unsigned long long s_best[2][kNngWaves][64];
    __shared__ __attribute__((aligned(16))) float s_cand[kNngWaves][64 * 4];    // per wave: 32 candidate PAIRS [x0 x1 y0 y1 z0 z1 row0 row1]
    const NngJob& job = a.job[blockIdx.y];
    const NngSet& Q = a.set[job.q];
    const NngSet& R = a.set[job.r];
    const NnGrid& g = a.g;
    const int base = blockIdx.x * 64;
    if (base >= Q.n) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool valid = base + lane < Q.n;
    const float4 me = Q.sorted[min(base + lane, Q.n - 1)];
    const float qx = me.x, qy = me.y, qz = me.z;
    int cx, cy;
    cell_of(g, qx, qy, cx, cy);
    unsigned long long best = kNngNone;
    float best_d = INFINITY;                 // the distance half of `best`
    const float2v q2x{qx, qx}, q2y{qy, qy}, q2z{qz, qz};
    if (R.n > 0) {
        const int* __restrict__ off = R.offset;
        const int* __restrict__ off_t = R.offset_t;
        const float4* __restrict__ rows = R.sorted;
        const float4* __restrict__ cols = R.sorted_t;
        float* const mine = s_cand[wave];
        float* const my_slot = mine + (lane >> 1) * 8 + (lane & 1);
        // a bound computed in float32 from cell indices may exceed the true clearance by a rounding error of the coordinates
        const float slack = 1e-6f * (fabsf(qx) + fabsf(qy) + fabsf(g.x0) + fabsf(g.y0) + g.cell);
        int par = 0, turn = 0;          // turn: pieces handed out so far -- piece t goes to wave t % kNngWaves, across rings
        unsigned long long pending = __ballot(valid);
        while (pending) {
            // the next segment: the first unserved query's grid row, from its cell to at most kNngSegCells further (the sorted
            // order makes these lanes a contiguous stretch with non-decreasing cx)
            const int lead = __ffsll((long long)pending) - 1;
            const int y0 = __builtin_amdgcn_readlane(cy, lead), xa = __builtin_amdgcn_readlane(cx, lead);
            const bool act = ((pending >> lane) & 1ull) && cy == y0 && cx < xa + kNngSegCells;
            const unsigned long long actm = __ballot(act);
            const int xb = __builtin_amdgcn_readlane(cx, 63 - __clzll((long long)actm));
            // BEV clearance of the query inside the rectangle of cells [xa, xb] x {y0}: everything outside the rectangle grown
            // by `ring` cells is at least ring * cell + m away (m = 0 for queries outside the grid, binned into border cells)
            const float lxa = g.x0 + (float)xa * g.cell, lxb = g.x0 + (float)(xb + 1) * g.cell, ly = g.y0 + (float)y0 * g.cell;
            const float m = fmaxf(0.f, fminf(fminf(qx - lxa, lxb - qx), fminf(qy - ly, ly + g.cell - qy)) - slack);
            const int rmax = max(max(xa, g.gw - 1 - xb), max(y0, g.gh - 1 - y0));
            bool alive = act;
            bool fresh = false;         // candidates seen since the last merge
            int check = 0;              // the next ring after which the waves merge and the lanes test their bound
            int rbase = -8, ofs = 0;    // lane l of `ofs`: run bound (l & 7) of ring rbase + (l >> 3)
            for (int ring = 0; ring <= rmax; ++ring) {
                if (ring >= rbase + 8) {
                    // the run bounds of the next eight rings in ONE vector load.  A ring's runs: [0] top row, [1] bottom row
                    // (cells xl..xr of the row-major copy), [2] left column, [3] right column (rows yl+1..yh-1 of the column-major copy)
                    rbase = ring;
                    const int r = rbase + (lane >> 3), k = lane & 7;
                    const int cxl = clampi(xa - r, 0, g.gw - 1), cxr = clampi(xb + r, 0, g.gw - 1);
                    const int ry = clampi((k & 2) ? y0 + r : y0 - r, 0, g.gh - 1);
                    const int cyl = clampi(y0 - r + 1, 0, g.gh - 1), cyh = clampi(y0 + r - 1, 0, g.gh - 1);
                    const int row_major = ry * g.gw + ((k & 1) ? cxr + 1 : cxl);
                    const int col_major = ((k & 2) ? cxr : cxl) * g.gh + ((k & 1) ? cyh + 1 : cyl);
                    ofs = (k & 4) ? off_t[col_major] : off[row_major];
                }
                const int sl = (ring - rbase) * 8;
                const int b0 = __builtin_amdgcn_readlane(ofs, sl), b1 = __builtin_amdgcn_readlane(ofs, sl + 2);
                const int b2 = __builtin_amdgcn_readlane(ofs, sl + 4), b3 = __builtin_amdgcn_readlane(ofs, sl + 6);
                int n0 = __builtin_amdgcn_readlane(ofs, sl + 1) - b0, n1 = __builtin_amdgcn_readlane(ofs, sl + 3) - b1;
                int n2 = __builtin_amdgcn_readlane(ofs, sl + 5) - b2, n3 = __builtin_amdgcn_readlane(ofs, sl + 7) - b3;
                if (ring == 0) { n1 = 0; n2 = 0; n3 = 0; }         // ring 0 is the rectangle itself: one run
                else {
                    if (y0 - ring < 0) n0 = 0;                      // a side of the ring outside the grid
                    if (y0 + ring >= g.gh) n1 = 0;
                    if (xa - ring < 0) n2 = 0;
                    if (xb + ring >= g.gw) n3 = 0;
                }
                // the ring's candidates as ONE list (the four runs end to end), cut into pieces of 16 / 32 / 64; a piece is one
                // coalesced vector load of its wave (a lane per candidate), parked in the wave's LDS slice and read back as
                // broadcasts: every lane tests every candidate of the piece against its own query
                const int p1 = n0, p2 = p1 + n1, p3 = p2 + n2, total = p3 + n3;
                if (total > 0) {
                    fresh = true;
                    // (short walks split a ring finely so that all waves share it; from ring 4 on the merges are rings apart and
                    // whole 64-candidate pieces -- often a whole ring -- go to the waves in turn: a wave then pays the memory latency of
                    // every eighth ring instead of every ring's)
                    const int lg = (ring >= 4 || total >= 32 * kNngWaves) ? 6 : (total >= 16 * kNngWaves ? 5 : 4);
                    const int pieces = (total + (1 << lg) - 1) >> lg;
                    for (int t = (wave - turn) & (kNngWaves - 1); t < pieces; t += kNngWaves) {
                        const int x = min((t << lg) + lane, total - 1);                 // past the end: the last candidate again
                        const int in_run = x >= p3 ? b3 + (x - p3) : (x >= p2 ? b2 + (x - p2) : (x >= p1 ? b1 + (x - p1) : b0 + x));
                        if (lane < (1 << lg)) {
                            const float4 c = (x >= p2 ? cols : rows)[in_run];
                            my_slot[0] = c.x; my_slot[2] = c.y; my_slot[4] = c.z; my_slot[6] = c.w;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        // eight candidates (four pairs) per step, the LDS reads in flight before the first use; the slots past the
                        // piece's last candidate hold that candidate again (pieces are multiples of 16 slots).  FAST REJECT: after the
                        // first ring or two a lane's best rarely improves, so a step computes its eight distances on packed float32
                        // pairs, reduces them to their minimum, and only when that reaches SOME lane's best (ties included) does the
                        // wave take the exact (distance, row) update of the step -- 4.5 vector instructions per candidate instead of 9
                        const int steps = (min(1 << lg, total - (t << lg)) + 7) >> 3;
                        for (int j = 0; j < steps; ++j) {
                            float2v d[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float4 u = *reinterpret_cast<const float4*>(mine + (j * 4 + i) * 8);          // x0 x1 y0 y1
                                const float2v z = *reinterpret_cast<const float2v*>(mine + (j * 4 + i) * 8 + 4);    // z0 z1
                                const float2v dx = q2x - float2v{u.x, u.y}, dy = q2y - float2v{u.z, u.w}, dz = q2z - z;
                                d[i] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                            }
                            const float lo = fminf(fminf(fminf(d[0].x, d[0].y), fminf(d[1].x, d[1].y)), fminf(fminf(d[2].x, d[2].y), fminf(d[3].x, d[3].y)));
                            if (__ballot(lo <= best_d) != 0ull) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float2v row = *reinterpret_cast<const float2v*>(mine + (j * 4 + i) * 8 + 6);  // row0 row1
                                    const unsigned long long k0 = nng_key(d[i].x, __float_as_int(row.x)), k1 = nng_key(d[i].y, __float_as_int(row.y));
                                    best = k0 < best ? k0 : best;
                                    best = k1 < best ? k1 : best;
                                }
                                best_d = __uint_as_float((unsigned)(best >> 32));
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    turn += pieces;
                }
                if (ring < check && ring < rmax) continue;         // rings 0 1 2 3 5 8 13 20 31 ... : far walkers merge less often
                check = ring + 1 + (ring >= 3 ? ring / 2 : 0);
                if (fresh) {            // merge the waves' (distance, row)
                    s_best[par][wave][lane] = best;
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < kNngWaves; ++w) { const unsigned long long o = s_best[par][w][lane]; best = o < best ? o : best; }
                    best_d = __uint_as_float((unsigned)(best >> 32));
                    par ^= 1;           // the next merge writes the other buffer: one barrier per merge
                    fresh = false;
                }
                const float reach = (float)ring * g.cell + m;
                if (best != kNngNone && best_d <= reach * reach) alive = false;
                if (__ballot(alive) == 0ull) break;
            }
            pending &= ~actm;
        }
    }
    if (wave == 0 && valid) {
        const int o = __float_as_int(me.w);
        job.dist2[o] = __uint_as_float((unsigned)(best >> 32));
        if (job.idx) job.idx[o] = (int)(unsigned)best;
    }
}

static size_t nng_ints_bytes(int cells) { return round_up(((size_t)cells + 1) * 4, 16); }

size_t nng_workspace_bytes(int n_sets, int64_t n_max, int cells) {
    const size_t n = (size_t)(n_max > 0 ? n_max : 1);
    return (size_t)n_sets * (4 * nng_ints_bytes(cells) + round_up(n * 4, 16) + 2 * round_up(n * 16, 16)) + 64;
}

void nng_carve(void* workspace, NngSet* sets, int n_sets, const float* const* pts, const int* n, const int* searched, int cells) {
    char* w = reinterpret_cast<char*>(workspace);
    const size_t ib = nng_ints_bytes(cells);
    for (int k = 0; k < n_sets; ++k) {
        sets[k].pts = pts[k]; sets[k].n = n[k]; sets[k].searched = searched[k];
        sets[k].offset = reinterpret_cast<int*>(w); sets[k].offset_t = reinterpret_cast<int*>(w + ib);
        sets[k].cursor = reinterpret_cast<int*>(w + 2 * ib); sets[k].cursor_t = reinterpret_cast<int*>(w + 3 * ib);
        w += 4 * ib;
    }
    for (int k = 0; k < n_sets; ++k) {
        const size_t nn = (size_t)(n[k] > 0 ? n[k] : 1);
        sets[k].cell_id = reinterpret_cast<int*>(w); w += round_up(nn * 4, 16);
        sets[k].sorted = reinterpret_cast<float4*>(w); w += round_up(nn * 16, 16);
        sets[k].sorted_t = reinterpret_cast<float4*>(w); w += round_up(nn * 16, 16);
    }
}

static void nng_pack(NngArgs& a, const NngSet* sets, int n_sets, const NngJob* jobs, int n_jobs, const NnGrid& g) {
    a = NngArgs{};
    for (int k = 0; k < n_sets; ++k) a.set[k] = sets[k];
    for (int k = 0; k < n_jobs; ++k) a.job[k] = jobs[k];
    a.g = g;
}

int nng_build(const NngSet* sets, int n_sets, const NnGrid& g, hipStream_t s) {
    if (n_sets < 1 || n_sets > kNngMaxSets) return HIMO_ERR_INVALID_ARGUMENT;
    const int cells = g.gw * g.gh;
    NngArgs a;
    nng_pack(a, sets, n_sets, nullptr, 0, g);
    int n_max = 0;
    for (int k = 0; k < n_sets; ++k) n_max = sets[k].n > n_max ? sets[k].n : n_max;
    ProfScope ps("nn_grid_build", s);
    // nng_carve keeps the sets' integer arrays contiguous from set 0's offsets
    HIMO_HIP(hipMemsetAsync(sets[0].offset, 0, (size_t)n_sets * 4 * nng_ints_bytes(cells), s));
    if (n_max > 0) hipLaunchKernelGGL(nng_count_kernel, dim3((unsigned)((n_max + 255) / 256), n_sets), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nng_scan_kernel, dim3(2 * n_sets), dim3(1024), 0, s, a, cells);
    if (n_max > 0) hipLaunchKernelGGL(nng_fill_kernel, dim3((unsigned)((n_max + 255) / 256), n_sets), dim3(256), 0, s, a);
    HIMO_LAUNCH_CHECK("nn_grid_build");
    return HIMO_OK;
}

int nng_query(const NngSet* sets, int n_sets, const NngJob* jobs, int n_jobs, const NnGrid& g, hipStream_t s) {
    if (n_sets < 1 || n_sets > kNngMaxSets || n_jobs < 1 || n_jobs > kNngMaxJobs) return HIMO_ERR_INVALID_ARGUMENT;
    NngArgs a;
    nng_pack(a, sets, n_sets, jobs, n_jobs, g);
    int nq_max = 0;
    for (int k = 0; k < n_jobs; ++k) {
        if (jobs[k].q < 0 || jobs[k].q >= n_sets || jobs[k].r < 0 || jobs[k].r >= n_sets || !sets[jobs[k].r].searched)
            return HIMO_ERR_INVALID_ARGUMENT;
        nq_max = sets[jobs[k].q].n > nq_max ? sets[jobs[k].q].n : nq_max;
    }
    if (nq_max == 0) return HIMO_OK;
    {
        ProfScope ps("nn_grid_query_kernel", s);
        hipLaunchKernelGGL(nng_query_kernel, dim3((unsigned)((nq_max + 63) / 64), n_jobs), dim3(64 * kNngWaves), 0, s, a);
    }
    HIMO_LAUNCH_CHECK("nn_grid_query_kernel");
    return HIMO_OK;
}

}  // namespace himo

using namespace himo;

extern "C" size_t himo_nn_grid_workspace_bytes(int64_t n_max, int grid_w, int grid_h) {
    return nng_workspace_bytes(2, n_max, grid_w * grid_h);
}

extern "C" int himo_nn_grid(int64_t nq, const float* d_q, int64_t nr, const float* d_r, float x0, float y0, float cell,
                            int grid_w, int grid_h, float* d_dist2, int32_t* d_idx, void* d_workspace,
                            size_t workspace_bytes, void* stream) {
    if (nq < 0 || nr < 0 || grid_w < 1 || grid_h < 1 || !(cell > 0.f)) return HIMO_ERR_INVALID_ARGUMENT;
    if (nr > 0x7fffffff || nq > 0x7fffffff || (int64_t)grid_w * grid_h > (1 << 20)) return HIMO_ERR_UNSUPPORTED;
    if (nq == 0) return HIMO_OK;
    if (!d_q || !d_dist2 || !d_workspace || (nr > 0 && !d_r)) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_nn_grid_workspace_bytes(nq > nr ? nq : nr, grid_w, grid_h) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    NnGrid g{x0, y0, 1.0f / cell, cell, grid_w, grid_h};
    NngSet sets[2];
    const float* pts[2] = {d_q, d_r};
    const int n[2] = {(int)nq, (int)nr}, searched[2] = {0, 1};
    nng_carve(d_workspace, sets, 2, pts, n, searched, grid_w * grid_h);
    int st = nng_build(sets, 2, g, s);
    if (st != HIMO_OK) return st;
    NngJob job{0, 1, d_dist2, d_idx};
    return nng_query(sets, 2, &job, 1, g, s);
}
