// nn.hip -- exact k=1 nearest neighbour (the two halves of the Chamfer distance) for gfx950.
//
// Reference being replaced: the cKDTree build + query pairs of eval.py:50-62 and
// tools/test/score.py:180-192 (per-instance sets of 10..5000 points), and the correspondence search
// of the self-supervised Chamfer loss (two ~120k-point sweeps).
//
// Design: brute force, tiled through LDS.  A kd-tree is pointer-chasing with divergent control flow
// -- exactly what a 64-wide wavefront is bad at -- while the exhaustive search is a dense,
// perfectly regular N x M sweep: at 120k x 120k it is 1.4e10 distance evaluations, ~2 ms of VALU
// time on 256 CUs, and at the evaluator's per-instance sizes it is microseconds.  Each block owns
// QPT*256 queries (QPT registers-resident queries per lane) and streams the reference range of its
// queries through LDS in tiles; every lane reads the SAME LDS address at a time (a broadcast, which
// is conflict-free), so one ds_read_b128 feeds 64*QPT distance evaluations.  When the query count is
// too small to fill the chip the reference range is additionally split over SPLIT lanes of the
// wavefront and the partial minima are merged with wavefront shuffles.
//
// Numerics: float64 mode evaluates (dx*dx + dy*dy) + dz*dz with separately rounded products -- the
// order of scipy's cKDTree -- so sqrt(min) is bit-comparable with the reference; float32 mode uses
// fused multiply-adds.  Ties keep the lowest reference index.  Built with -ffp-contract=off.
#include "himo_common.h"
#include <math.h>

namespace himo {

constexpr int kNnThreads = 256;
constexpr int kNnTile = 1024;    // reference points per LDS tile

struct NnArgs {
    int64_t nq, nr;
    const void* q;           // [nq][3]
    const void* r;           // [nr][3]
    // range of references each query may match, one of:
    int n_segments;          //   (a) segment tables: query in [q_off[s], q_off[s+1]) searches [r_off[s], r_off[s+1])
    const int64_t* q_off;
    const int64_t* r_off;
    const int32_t* rbeg;     //   (b) explicit per-query [rbeg[i], rbeg[i] + rlen[i])
    const int32_t* rlen;
    void* dist2;             // [nq] squared distance (+inf when the range is empty)
    int32_t* idx;            // [nq] reference row or -1; may be nullptr
};

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

template <typename T>
__device__ inline T dist2(T qx, T qy, T qz, T rx, T ry, T rz);
template <>
__device__ inline float dist2<float>(float qx, float qy, float qz, float rx, float ry, float rz) {
    const float dx = qx - rx, dy = qy - ry, dz = qz - rz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}
template <>
__device__ inline double dist2<double>(double qx, double qy, double qz, double rx, double ry, double rz) {
    const double dx = qx - rx, dy = qy - ry, dz = qz - rz;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ inline int find_segment(const int64_t* __restrict__ off, int n, int64_t i) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// SPLIT lanes share one query; lane s of the group scans tile entries s, s+SPLIT, ...
template <typename T, int QPT, int SPLIT>
__global__ __launch_bounds__(kNnThreads) void nn_kernel(NnArgs a) {
    using V4 = typename Vec4<T>::type;
    __shared__ V4 tile[kNnTile];
    __shared__ int s_lo, s_hi;

    constexpr int kQueriesPerBlock = kNnThreads / SPLIT * QPT;
    const int lane_in_group = threadIdx.x % SPLIT;
    const int group = threadIdx.x / SPLIT;
    const int64_t qbase = (int64_t)blockIdx.x * kQueriesPerBlock;

    const T* __restrict__ Q = reinterpret_cast<const T*>(a.q);
    const T* __restrict__ R = reinterpret_cast<const T*>(a.r);

    T qx[QPT], qy[QPT], qz[QPT], best[QPT];
    int bi[QPT], rs[QPT], re[QPT];
    int lo = 0x7fffffff, hi = 0;
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int64_t i = qbase + (int64_t)k * (kNnThreads / SPLIT) + group;
        best[k] = (T)INFINITY; bi[k] = -1; rs[k] = 0; re[k] = 0;
        qx[k] = qy[k] = qz[k] = (T)0;
        if (i < a.nq) {
            qx[k] = Q[i * 3]; qy[k] = Q[i * 3 + 1]; qz[k] = Q[i * 3 + 2];
            if (a.rbeg) { rs[k] = a.rbeg[i]; re[k] = rs[k] + a.rlen[i]; }
            else {
                const int s = find_segment(a.q_off, a.n_segments, i);
                rs[k] = (int)a.r_off[s]; re[k] = (int)a.r_off[s + 1];
            }
            if (re[k] > rs[k]) { lo = min(lo, rs[k]); hi = max(hi, re[k]); }
        }
    }
    if (threadIdx.x == 0) { s_lo = 0x7fffffff; s_hi = 0; }
    __syncthreads();
    if (hi > lo) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    const int blo = s_lo, bhi = s_hi;

    for (int t = blo; t < bhi; t += kNnTile) {
        __syncthreads();
        for (int j = threadIdx.x; j < kNnTile; j += kNnThreads) {
            const int64_t g = (int64_t)t + j;
            V4 v;
            if (g < bhi) { v.x = R[g * 3]; v.y = R[g * 3 + 1]; v.z = R[g * 3 + 2]; v.w = (T)0; }
            else { v.x = v.y = v.z = (T)INFINITY; v.w = (T)0; }
            tile[j] = v;
        }
        __syncthreads();
        const int tcount = min(kNnTile, bhi - t);
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const int jlo = max(rs[k] - t, 0), jhi = min(re[k] - t, tcount);
            // first entry >= jlo that belongs to this lane of the group
            int j = jlo + ((lane_in_group - jlo % SPLIT) + SPLIT) % SPLIT;
#pragma unroll 4
            for (; j < jhi; j += SPLIT) {
                const V4 v = tile[j];
                const T d = dist2<T>(qx[k], qy[k], qz[k], v.x, v.y, v.z);
                if (d < best[k]) { best[k] = d; bi[k] = t + j; }
            }
        }
    }

    // merge the SPLIT partial minima of each query with wavefront shuffles (lowest index wins ties)
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
#pragma unroll
        for (int off = SPLIT >> 1; off > 0; off >>= 1) {
            const T od = __shfl_xor(best[k], off, 64);
            const int oi = __shfl_xor(bi[k], off, 64);
            const bool take = (od < best[k]) || (od == best[k] && oi >= 0 && (bi[k] < 0 || oi < bi[k]));
            if (take) { best[k] = od; bi[k] = oi; }
        }
        const int64_t i = qbase + (int64_t)k * (kNnThreads / SPLIT) + group;
        if (lane_in_group == 0 && i < a.nq) {
            reinterpret_cast<T*>(a.dist2)[i] = best[k];
            if (a.idx) a.idx[i] = bi[k];
        }
    }
}

template <typename T>
static int launch_nn(const NnArgs& a, hipStream_t s) {
    if (a.nq == 0) return HIMO_OK;
    // few queries: spend lanes on splitting the reference range instead (keeps >= ~1 block per CU)
    const int64_t blocks_q1 = (a.nq + kNnThreads - 1) / kNnThreads;
    ProfScope ps(sizeof(T) == 8 ? "nn_kernel_f64" : "nn_kernel_f32", s);
    if (blocks_q1 >= 2048) {
        constexpr int QPT = 2;
        const dim3 grid((unsigned)((a.nq + kNnThreads * QPT - 1) / (kNnThreads * QPT)));
        hipLaunchKernelGGL((nn_kernel<T, QPT, 1>), grid, dim3(kNnThreads), 0, s, a);
    } else if (blocks_q1 >= 256) {
        hipLaunchKernelGGL((nn_kernel<T, 1, 1>), dim3((unsigned)blocks_q1), dim3(kNnThreads), 0, s, a);
    } else if (blocks_q1 >= 64) {
        constexpr int SPLIT = 4;
        const int qpb = kNnThreads / SPLIT;
        hipLaunchKernelGGL((nn_kernel<T, 1, SPLIT>), dim3((unsigned)((a.nq + qpb - 1) / qpb)), dim3(kNnThreads), 0, s, a);
    } else {
        constexpr int SPLIT = 16;
        const int qpb = kNnThreads / SPLIT;
        hipLaunchKernelGGL((nn_kernel<T, 1, SPLIT>), dim3((unsigned)((a.nq + qpb - 1) / qpb)), dim3(kNnThreads), 0, s, a);
    }
    HIMO_LAUNCH_CHECK("nn_kernel");
    return HIMO_OK;
}

int nn_search_ranges(int64_t nq, int64_t nr, const void* q, const void* r, const int32_t* rbeg, const int32_t* rlen,
                     bool f64, void* dist2, int32_t* idx, hipStream_t s) {
    NnArgs a{};
    a.nq = nq; a.nr = nr; a.q = q; a.r = r; a.rbeg = rbeg; a.rlen = rlen; a.dist2 = dist2; a.idx = idx;
    return f64 ? launch_nn<double>(a, s) : launch_nn<float>(a, s);
}

template <typename T>
__global__ __launch_bounds__(256) void sqrt_kernel(int64_t n, T* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = sqrt(v[i]);          // correctly rounded in both precisions (cKDTree returns sqrt of the minimum)
}

}  // namespace himo

using namespace himo;

extern "C" int himo_sqrt_inplace(int64_t n, void* d_values, int dtype_is_f64, void* stream) {
    if (n < 0) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_values) return HIMO_ERR_INVALID_ARGUMENT;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype_is_f64) hipLaunchKernelGGL(sqrt_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, n, (double*)d_values);
    else hipLaunchKernelGGL(sqrt_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, n, (float*)d_values);
    HIMO_LAUNCH_CHECK("sqrt_kernel");
    return HIMO_OK;
}

extern "C" int himo_nn_search(int n_segments, const int64_t* d_q_offsets, const int64_t* d_r_offsets, int64_t nq,
                              int64_t nr, const void* d_q, const void* d_r, int dtype_is_f64, void* d_dist2,
                              int32_t* d_idx, void* stream) {
    if (n_segments < 1 || nq < 0 || nr < 0 || !d_q_offsets || !d_r_offsets) return HIMO_ERR_INVALID_ARGUMENT;
    if (nr > 0x7fffffff) return HIMO_ERR_UNSUPPORTED;
    if (nq == 0) return HIMO_OK;
    if (!d_q || !d_dist2 || (nr > 0 && !d_r)) return HIMO_ERR_INVALID_ARGUMENT;
    NnArgs a{};
    a.nq = nq; a.nr = nr; a.q = d_q; a.r = d_r; a.n_segments = n_segments; a.q_off = d_q_offsets; a.r_off = d_r_offsets;
    a.dist2 = d_dist2; a.idx = d_idx;
    return dtype_is_f64 ? launch_nn<double>(a, (hipStream_t)stream) : launch_nn<float>(a, (hipStream_t)stream);
}
