// compdis.hip -- stages a1-a6 of the HiMo hot path for gfx950 (MI355X).
//
//   flow (N,3) + pc0 (N,>=3) + lidar_dt (N,) + pose0/pose1  ->  comp_dis (N,3) [+ refined, eval mask]
//
// Reference arithmetic being replaced (paths under /root/reference):
//   save_zip.py:114-121, eval.py:283-299, utils/__init__.py:26-47.
//
// Shape of the problem: ~44-70 bytes and ~40 flops per point -- purely HBM-bound, and a single
// 120k-point sweep is only ~5 MB, i.e. <1 us of HBM time on this chip.  So the unit of work is a
// ragged BATCH of sweeps laid end to end in HBM, processed by two launches:
//
//   1. frame_prep_kernel   per-frame max(lidar_dt) (block reduction + one integer atomicMax per
//                          4096-point chunk) and, folded into the same launch, the per-frame
//                          ego transform inv(pose1) @ pose0 in float64 (one thread per frame).
//   2. compdis_kernel      one pass over the points: 4 consecutive points per lane so that every
//                          stream (xyzi rows, flow rows, dt, outputs) moves as 16-byte accesses.
//                          The frame of a 1024-point block is found once per block with a
//                          wave-uniform binary search; blocks that straddle a frame boundary
//                          take a per-point slow path.
//
// Numerics: by default the chain runs in float64 in numpy's operation order and is rounded to
// float32 where save_zip.py:70-72 casts, which is what makes the result bit-comparable with the
// reference for float64 poses.  HIMO_FLAG_F32_CHAIN runs the float32 chain numpy runs for float32
// poses.  This file is compiled with -ffp-contract=off; every fused multiply-add is explicit.
#include "compdis_math.h"

namespace himo {

// ego = inv(pose1) @ pose0 in float64 (save_zip.py:115): LU with partial pivoting (first maximal
// pivot, as LAPACK's idamax), explicit inverse by forward/back substitution on the identity, then
// a k-ordered 4x4 product -- the sequence numpy's inv + matmul perform.  Everything is unrolled with
// compile-time indices (row swaps are predicated selects) so it lives in registers.
__device__ inline void swap_if(bool c, double& x, double& y) { const double tx = c ? y : x, ty = c ? x : y; x = tx; y = ty; }

__device__ inline void compute_ego(const double* __restrict__ p0, const double* __restrict__ p1, FrameXf* out) {
    double a[4][4], b[4][4];   // b starts as the identity and ends as inv(pose1)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[r][c] = p1[r * 4 + c]; b[r][c] = r == c ? 1.0 : 0.0; }
    bool singular = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = k;
        double best = fabs(a[k][k]);
#pragma unroll
        for (int r = k + 1; r < 4; ++r)
            if (fabs(a[r][k]) > best) { best = fabs(a[r][k]); p = r; }
        singular = singular || !(best > 0.0);
#pragma unroll
        for (int r = k + 1; r < 4; ++r) {
            const bool sw = p == r;
#pragma unroll
            for (int c = 0; c < 4; ++c) { swap_if(sw, a[k][c], a[r][c]); swap_if(sw, b[k][c], b[r][c]); }
        }
        const double rp = 1.0 / a[k][k];
#pragma unroll
        for (int r = k + 1; r < 4; ++r) {
            a[r][k] *= rp;
#pragma unroll
            for (int c = k + 1; c < 4; ++c) a[r][c] -= a[r][k] * a[k][c];
        }
    }
    // L y = P I (unit lower), then U x = y, column by column
#pragma unroll
    for (int col = 0; col < 4; ++col) {
#pragma unroll
        for (int r = 1; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < r; ++c) b[r][col] -= a[r][c] * b[c][col];
#pragma unroll
        for (int r = 3; r >= 0; --r) {
#pragma unroll
            for (int c = r + 1; c < 4; ++c) b[r][col] -= a[r][c] * b[c][col];
            b[r][col] /= a[r][r];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double s = b[r][0] * p0[c];
#pragma unroll
            for (int k = 1; k < 4; ++k) s = fma(b[r][k], p0[k * 4 + c], s);
            if (singular) s = NAN;
            if (c < 3) out->R[r * 3 + c] = s; else out->t[r] = s;
        }
}

// the caller already holds ego = inv(pose1) @ pose0 (HIMO_FLAG_POSE_IS_EGO): copy it through
__device__ inline void copy_ego(const double* __restrict__ e, FrameXf* out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out->R[r * 3 + c] = e[r * 4 + c];
        out->t[r] = e[r * 4 + 3];
    }
}

// ------------------------------------------------------------------------------------------
// launch 1: per-frame max(lidar_dt) + ego transforms
// ------------------------------------------------------------------------------------------
__device__ inline float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ inline void block_max_to_key(float m, unsigned* key) {
    __shared__ float wmax[kPrepThreads / 64];
    m = wave_max(m);
    __syncthreads();   // wmax may still be read by the previous round
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = wmax[0];
#pragma unroll
        for (int w = 1; w < kPrepThreads / 64; ++w) r = fmaxf(r, wmax[w]);
        if (r > -INFINITY) atomicMax(key, float_to_key(r));
    }
}

// ego_mode: 0 = no poses (himo_dt0), 1 = pose0/pose1 given, 2 = pose0 holds the ego transform
__global__ __launch_bounds__(kPrepThreads) void frame_prep_kernel(
    int n_frames, int64_t total, int n_chunks, int ego_mode, const int64_t* __restrict__ offsets,
    const double* __restrict__ pose0, const double* __restrict__ pose1,
    const float* __restrict__ lidar_dt, unsigned* __restrict__ keys, FrameXf* __restrict__ xf) {
    const int b = blockIdx.x;
    if (b >= n_chunks) {   // surplus blocks: one thread per frame does the 4x4 work
        const int f = (b - n_chunks) * kPrepThreads + threadIdx.x;
        if (f < n_frames) {
            if (ego_mode == 1) compute_ego(pose0 + 16 * (size_t)f, pose1 + 16 * (size_t)f, &xf[f]);
            else if (ego_mode == 2) copy_ego(pose0 + 16 * (size_t)f, &xf[f]);
        }
        return;
    }
    const int64_t start = (int64_t)b * kPrepChunk;
    const int64_t end = start + kPrepChunk < total ? start + kPrepChunk : total;
    const int f0 = __builtin_amdgcn_readfirstlane(find_frame(offsets, n_frames, start));
    const bool vec = (reinterpret_cast<uintptr_t>(lidar_dt) & 15u) == 0;

    float v[kPrepChunk / kPrepThreads];   // this thread's 16 elements; -inf where out of range
#pragma unroll
    for (int j = 0; j < kPrepChunk / (kPrepThreads * 4); ++j) {
        const int64_t i = start + ((int64_t)j * kPrepThreads + threadIdx.x) * 4;
        if (vec && i + 3 < end) {
            const float4 q = *reinterpret_cast<const float4*>(lidar_dt + i);
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[4 * j + k] = i + k < end ? lidar_dt[i + k] : -INFINITY;
        }
    }
    // one round per frame present in the chunk (one round unless a frame boundary falls inside it)
    int f = f0;
    int64_t fend = offsets[f + 1];
    int64_t fbeg = start;
    while (true) {
        const int64_t hi = fend < end ? fend : end;
        float m = -INFINITY;   // fmaxf drops NaN operands
#pragma unroll
        for (int j = 0; j < kPrepChunk / (kPrepThreads * 4); ++j) {
            const int64_t i = start + ((int64_t)j * kPrepThreads + threadIdx.x) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k >= fbeg && i + k < hi) m = fmaxf(m, v[4 * j + k]);
        }
        block_max_to_key(m, &keys[f]);
        if (hi >= end) break;
        fbeg = hi;
        do { ++f; fend = offsets[f + 1]; } while (fend <= fbeg);   // skip empty frames
    }
}

// ------------------------------------------------------------------------------------------
// launch 2: the fused element-wise pass
// ------------------------------------------------------------------------------------------
struct CompdisArgs {
    int n_frames;
    int64_t total;
    const int64_t* offsets;
    const unsigned* keys;
    const FrameXf* xf;
    const float* pc0;
    int pc_stride;
    const float* flow;       // nullptr => RAW
    const float* lidar_dt;
    double sensor_dt;
    float* comp_dis;
    float* refined;          // may be nullptr
    uint8_t* eval_mask;      // may be nullptr
    const uint8_t* gm0;
    const uint8_t* valid;    // may be nullptr
    float bmin[3], bmax[3];
    float close_distance;
};

__device__ inline uint8_t mask_math(const CompdisArgs& a, float px, float py, float pz, uint8_t gm, uint8_t valid) {
    const float d = sqrtf(px * px + py * py);                        // eval.py:288 (float32 norm)
    const bool inside = (px > a.bmin[0]) & (px < a.bmax[0]) & (py > a.bmin[1]) & (py < a.bmax[1]) &
                        (pz > a.bmin[2]) & (pz < a.bmax[2]);         // utils/__init__.py:31-33
    return (uint8_t)((d <= a.close_distance) & (gm == 0) & (!inside) & (valid != 0));   // eval.py:289-296
}

template <bool F32>
__device__ inline void scalar_point(const CompdisArgs& a, const XfRegs& x, int64_t i) {
    const float* p = a.pc0 + i * (int64_t)a.pc_stride;
    const bool raw = a.flow == nullptr;
    float cd[3], rf[3];
    point_math<F32>(x, p[0], p[1], p[2], raw ? 0.f : a.flow[i * 3], raw ? 0.f : a.flow[i * 3 + 1],
                    raw ? 0.f : a.flow[i * 3 + 2], a.lidar_dt[i], a.sensor_dt, raw, cd, rf);
    a.comp_dis[i * 3] = cd[0]; a.comp_dis[i * 3 + 1] = cd[1]; a.comp_dis[i * 3 + 2] = cd[2];
    if (a.refined) { a.refined[i * 3] = rf[0]; a.refined[i * 3 + 1] = rf[1]; a.refined[i * 3 + 2] = rf[2]; }
    if (a.eval_mask) a.eval_mask[i] = mask_math(a, p[0], p[1], p[2], a.gm0[i], a.valid ? a.valid[i] : (uint8_t)1);
}

// STRIDE = 4 (xyzi rows) or 3 (xyz rows): 16-byte vector path; STRIDE = 0: any stride / alignment.
template <int STRIDE, bool F32>
__global__ __launch_bounds__(kThreads) void compdis_kernel(CompdisArgs a) {
    const int64_t bstart = (int64_t)blockIdx.x * kBlockPts;
    const int64_t bend = bstart + kBlockPts < a.total ? bstart + kBlockPts : a.total;
    const int f0 = __builtin_amdgcn_readfirstlane(find_frame(a.offsets, a.n_frames, bstart));
    const bool uniform = a.offsets[f0 + 1] >= bend;
    const int64_t g = bstart + (int64_t)threadIdx.x * kPtsPerThread;
    if (g >= bend) return;

    if (!uniform || STRIDE == 0 || g + kPtsPerThread > bend) {
        // frame boundary inside the block, generic layout, or the ragged tail: one point at a time
        int f = f0;
        for (int64_t i = g; i < g + kPtsPerThread && i < bend; ++i) {
            while (i >= a.offsets[f + 1]) ++f;
            const XfRegs x = load_xf(a.xf, a.keys, f);
            scalar_point<F32>(a, x, i);
        }
        return;
    }

    const XfRegs x = load_xf(a.xf, a.keys, f0);
    const bool raw = a.flow == nullptr;

    float px[4], py[4], pz[4];
    if (STRIDE == 4) {
        const float4* src = reinterpret_cast<const float4*>(a.pc0) + g;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float4 v = src[k]; px[k] = v.x; py[k] = v.y; pz[k] = v.z; }
    } else {
        const float4* src = reinterpret_cast<const float4*>(a.pc0 + g * 3);
        const float4 v0 = src[0], v1 = src[1], v2 = src[2];
        px[0] = v0.x; py[0] = v0.y; pz[0] = v0.z; px[1] = v0.w; py[1] = v1.x; pz[1] = v1.y;
        px[2] = v1.z; py[2] = v1.w; pz[2] = v2.x; px[3] = v2.y; py[3] = v2.z; pz[3] = v2.w;
    }
    float fl[12];
    if (!raw) {
        const float4* src = reinterpret_cast<const float4*>(a.flow + g * 3);
        const float4 v0 = src[0], v1 = src[1], v2 = src[2];
        fl[0] = v0.x; fl[1] = v0.y; fl[2] = v0.z; fl[3] = v0.w; fl[4] = v1.x; fl[5] = v1.y;
        fl[6] = v1.z; fl[7] = v1.w; fl[8] = v2.x; fl[9] = v2.y; fl[10] = v2.z; fl[11] = v2.w;
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) fl[k] = 0.f;
    }
    const float4 dtv = *reinterpret_cast<const float4*>(a.lidar_dt + g);
    const float dt[4] = {dtv.x, dtv.y, dtv.z, dtv.w};

    float cd[12], rf[12];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        point_math<F32>(x, px[k], py[k], pz[k], fl[3 * k], fl[3 * k + 1], fl[3 * k + 2], dt[k], a.sensor_dt, raw,
                        cd + 3 * k, rf + 3 * k);

    float4* dst = reinterpret_cast<float4*>(a.comp_dis + g * 3);
    dst[0] = make_float4(cd[0], cd[1], cd[2], cd[3]);
    dst[1] = make_float4(cd[4], cd[5], cd[6], cd[7]);
    dst[2] = make_float4(cd[8], cd[9], cd[10], cd[11]);
    if (a.refined) {
        float4* r = reinterpret_cast<float4*>(a.refined + g * 3);
        r[0] = make_float4(rf[0], rf[1], rf[2], rf[3]);
        r[1] = make_float4(rf[4], rf[5], rf[6], rf[7]);
        r[2] = make_float4(rf[8], rf[9], rf[10], rf[11]);
    }
    if (a.eval_mask) {
        const uchar4 gm = *reinterpret_cast<const uchar4*>(a.gm0 + g);
        uchar4 vl = make_uchar4(1, 1, 1, 1);
        if (a.valid) vl = *reinterpret_cast<const uchar4*>(a.valid + g);
        uchar4 m;
        m.x = mask_math(a, px[0], py[0], pz[0], gm.x, vl.x);
        m.y = mask_math(a, px[1], py[1], pz[1], gm.y, vl.y);
        m.z = mask_math(a, px[2], py[2], pz[2], gm.z, vl.z);
        m.w = mask_math(a, px[3], py[3], pz[3], gm.w, vl.w);
        *reinterpret_cast<uchar4*>(a.eval_mask + g) = m;
    }
}

// ------------------------------------------------------------------------------------------
// stand-alone element-wise operators (utils/__init__.py)
// ------------------------------------------------------------------------------------------
template <typename T, typename TD>
__global__ __launch_bounds__(256) void flow2compdis_kernel(int64_t n, const T* __restrict__ flow,
                                                           const TD* __restrict__ dt0, T sensor_dt, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one scalar element of the (n,3) array
    if (i >= n * 3) return;
    out[i] = flow[i] / sensor_dt * (T)dt0[i / 3];
}

template <typename T>
__global__ __launch_bounds__(256) void refine_pts_kernel(int64_t n, const float* __restrict__ pc, int stride,
                                                         const T* __restrict__ ds, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 3) return;
    const int64_t r = i / 3;
    out[i] = (T)pc[r * stride + (i - r * 3)] + ds[i];
}

__global__ __launch_bounds__(256) void ego_mask_kernel(int64_t n, const float* __restrict__ pts, int stride, float x0,
                                                       float y0, float z0, float x1, float y1, float z1,
                                                       uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = pts + i * stride;
    const bool inside = (p[0] > x0) & (p[0] < x1) & (p[1] > y0) & (p[1] < y1) & (p[2] > z0) & (p[2] < z1);
    out[i] = inside ? 0 : 1;
}

__global__ __launch_bounds__(256) void dt0_kernel(int64_t n, const float* __restrict__ dt, const unsigned* __restrict__ key,
                                                  float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = key_to_float(*key) - dt[i];
}

}  // namespace himo

using namespace himo;

extern "C" size_t himo_compdis_workspace_bytes(int n_frames) {
    if (n_frames < 1) n_frames = 1;
    // keys + transforms, plus room for the single-frame wrapper's offsets[2] and two poses
    return keys_bytes(n_frames) + (size_t)n_frames * sizeof(FrameXf) + 2 * sizeof(int64_t) + 32 * sizeof(double);
}

int himo::launch_frame_prep(int n_frames, int64_t total, const int64_t* d_offsets, const double* d_pose0,
                            const double* d_pose1, unsigned flags, const float* d_lidar_dt, void* d_workspace, hipStream_t s) {
    WorkspaceLayout w = carve(d_workspace, n_frames);
    HIMO_HIP(hipMemsetAsync(w.keys, 0, keys_bytes(n_frames), s));

    const int n_chunks = (int)((total + kPrepChunk - 1) / kPrepChunk);
    const int grid1 = n_chunks + (n_frames + kPrepThreads - 1) / kPrepThreads;
    {
        ProfScope ps("frame_prep_kernel", s);
        hipLaunchKernelGGL(frame_prep_kernel, dim3(grid1), dim3(kPrepThreads), 0, s, n_frames, total, n_chunks,
                           (flags & HIMO_FLAG_POSE_IS_EGO) ? 2 : 1, d_offsets, d_pose0, d_pose1, d_lidar_dt, w.keys, w.xf);
    }
    HIMO_LAUNCH_CHECK("frame_prep_kernel");
    return HIMO_OK;
}

static int launch_compdis(int n_frames, int64_t total, const int64_t* d_offsets, const double* d_pose0,
                          const double* d_pose1, const float* d_pc0, int pc_stride, const float* d_flow,
                          const float* d_lidar_dt, double sensor_dt, unsigned flags, float* d_comp_dis, float* d_refined,
                          uint8_t* d_eval_mask, const uint8_t* d_gm0, const uint8_t* d_valid, const float* h_bounds,
                          float close_distance, void* d_workspace, hipStream_t s) {
    WorkspaceLayout w = carve(d_workspace, n_frames);
    {
        int st = launch_frame_prep(n_frames, total, d_offsets, d_pose0, d_pose1, flags, d_lidar_dt, d_workspace, s);
        if (st != HIMO_OK) return st;
    }
    if (total == 0) return HIMO_OK;

    CompdisArgs a;
    a.n_frames = n_frames; a.total = total; a.offsets = d_offsets; a.keys = w.keys; a.xf = w.xf;
    a.pc0 = d_pc0; a.pc_stride = pc_stride; a.flow = (flags & HIMO_FLAG_RAW) ? nullptr : d_flow;
    a.lidar_dt = d_lidar_dt; a.sensor_dt = sensor_dt; a.comp_dis = d_comp_dis; a.refined = d_refined;
    a.eval_mask = d_eval_mask; a.gm0 = d_gm0; a.valid = d_valid;
    for (int i = 0; i < 3; ++i) { a.bmin[i] = h_bounds ? h_bounds[i] : 0.f; a.bmax[i] = h_bounds ? h_bounds[3 + i] : 0.f; }
    a.close_distance = close_distance;

    bool vec = (pc_stride == 3 || pc_stride == 4) && aligned16(d_pc0) && aligned16(d_lidar_dt) && aligned16(d_comp_dis) &&
               (a.flow == nullptr || aligned16(a.flow)) && (d_refined == nullptr || aligned16(d_refined));
    if (d_eval_mask) {
        auto al4 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; };
        vec = vec && al4(d_eval_mask) && al4(d_gm0) && (d_valid == nullptr || al4(d_valid));
    }
    const bool f32 = (flags & HIMO_FLAG_F32_CHAIN) != 0;
    const dim3 grid((unsigned)((total + kBlockPts - 1) / kBlockPts)), block(kThreads);
    const int stride_sel = vec ? pc_stride : 0;
#define HIMO_GO(S, F) hipLaunchKernelGGL((compdis_kernel<S, F>), grid, block, 0, s, a)
    {
        ProfScope ps("compdis_kernel", s);
        if (stride_sel == 4) { if (f32) HIMO_GO(4, true); else HIMO_GO(4, false); }
        else if (stride_sel == 3) { if (f32) HIMO_GO(3, true); else HIMO_GO(3, false); }
        else { if (f32) HIMO_GO(0, true); else HIMO_GO(0, false); }
    }
#undef HIMO_GO
    HIMO_LAUNCH_CHECK("compdis_kernel");
    return HIMO_OK;
}

extern "C" int himo_compdis_batch(int n_frames, int64_t total_points, const int64_t* d_offsets, const double* d_pose0,
                                  const double* d_pose1, const float* d_pc0, int pc_stride, const float* d_flow,
                                  const float* d_lidar_dt, double sensor_dt, unsigned flags, float* d_comp_dis,
                                  float* d_refined, uint8_t* d_eval_mask, const uint8_t* d_gm0,
                                  const uint8_t* d_flow_is_valid, const float* h_mask_bounds, float close_distance,
                                  void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n_frames < 1 || total_points < 0 || pc_stride < 3) return HIMO_ERR_INVALID_ARGUMENT;
    if (!d_offsets || !d_pose0 || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    if (!d_pose1 && !(flags & HIMO_FLAG_POSE_IS_EGO)) return HIMO_ERR_INVALID_ARGUMENT;
    if (total_points > 0 && (!d_pc0 || !d_lidar_dt || !d_comp_dis)) return HIMO_ERR_INVALID_ARGUMENT;
    if (total_points > 0 && !(flags & HIMO_FLAG_RAW) && !d_flow) return HIMO_ERR_INVALID_ARGUMENT;
    if (d_eval_mask && (!d_gm0 || !h_mask_bounds)) return HIMO_ERR_INVALID_ARGUMENT;
    if (d_eval_mask && (flags & HIMO_FLAG_SCANIA) && !d_flow_is_valid) return HIMO_ERR_INVALID_ARGUMENT;
    if (!(sensor_dt != 0.0)) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_compdis_workspace_bytes(n_frames) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    const uint8_t* valid = (flags & HIMO_FLAG_SCANIA) ? d_flow_is_valid : nullptr;   // eval.py:293-296
    return launch_compdis(n_frames, total_points, d_offsets, d_pose0, d_pose1, d_pc0, pc_stride, d_flow, d_lidar_dt,
                          sensor_dt, flags, d_comp_dis, d_refined, d_eval_mask, d_gm0, valid, h_mask_bounds,
                          close_distance, d_workspace, (hipStream_t)stream);
}

extern "C" int himo_compdis_frame(int64_t n_points, const double* h_pose0, const double* h_pose1, const float* d_pc0,
                                  int pc_stride, const float* d_flow, const float* d_lidar_dt, double sensor_dt,
                                  unsigned flags, float* d_comp_dis, float* d_refined, void* d_workspace,
                                  size_t workspace_bytes, void* stream) {
    if (n_points < 0 || pc_stride < 3 || !h_pose0 || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    if (!h_pose1 && !(flags & HIMO_FLAG_POSE_IS_EGO)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n_points == 0) return HIMO_ERR_EMPTY_FRAME;                       // max() of an empty sequence
    if (!d_pc0 || !d_lidar_dt || !d_comp_dis) return HIMO_ERR_INVALID_ARGUMENT;
    if (!(flags & HIMO_FLAG_RAW) && !d_flow) return HIMO_ERR_INVALID_ARGUMENT;
    if (!(sensor_dt != 0.0)) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_compdis_workspace_bytes(1) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    // singular pose1 -> the error numpy raises at save_zip.py:115; 4x4 determinant by cofactors
    if (!(flags & HIMO_FLAG_POSE_IS_EGO)) {
        const double* m = h_pose1;
        auto det3 = [](double a, double b, double c, double d, double e, double f, double g, double h, double i) {
            return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        };
        const double det = m[0] * det3(m[5], m[6], m[7], m[9], m[10], m[11], m[13], m[14], m[15]) -
                           m[1] * det3(m[4], m[6], m[7], m[8], m[10], m[11], m[12], m[14], m[15]) +
                           m[2] * det3(m[4], m[5], m[7], m[8], m[9], m[11], m[12], m[13], m[15]) -
                           m[3] * det3(m[4], m[5], m[6], m[8], m[9], m[10], m[12], m[13], m[14]);
        if (!(det != 0.0)) return HIMO_ERR_SINGULAR_POSE;
    }
    hipStream_t s = (hipStream_t)stream;
    char* tail = reinterpret_cast<char*>(d_workspace) + keys_bytes(1) + sizeof(FrameXf);
    int64_t h_off[2] = {0, n_points};
    double h_poses[32];
    for (int i = 0; i < 16; ++i) { h_poses[i] = h_pose0[i]; h_poses[16 + i] = h_pose1 ? h_pose1[i] : 0.0; }
    int64_t* d_off = reinterpret_cast<int64_t*>(tail);
    double* d_poses = reinterpret_cast<double*>(tail + 2 * sizeof(int64_t));
    HIMO_HIP(hipMemcpyAsync(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice, s));
    HIMO_HIP(hipMemcpyAsync(d_poses, h_poses, sizeof(h_poses), hipMemcpyHostToDevice, s));
    return launch_compdis(1, n_points, d_off, d_poses, d_poses + 16, d_pc0, pc_stride, d_flow, d_lidar_dt, sensor_dt, flags,
                          d_comp_dis, d_refined, nullptr, nullptr, nullptr, nullptr, 0.f, d_workspace, s);
}

extern "C" int himo_flow2compdis(int64_t n, const void* d_flow, const void* d_dt0, double sensor_dt, int dtype_flags,
                                 void* d_out, void* stream) {
    if (n < 0) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_flow || !d_dt0 || !d_out) return HIMO_ERR_INVALID_ARGUMENT;
    if ((dtype_flags & 2) && !(dtype_flags & 1)) return HIMO_ERR_INVALID_ARGUMENT;   // f64 dt0 promotes the result
    const dim3 grid((unsigned)((n * 3 + 255) / 256)), block(256);
    if (dtype_flags == 3)
        hipLaunchKernelGGL((flow2compdis_kernel<double, double>), grid, block, 0, (hipStream_t)stream, n,
                           (const double*)d_flow, (const double*)d_dt0, sensor_dt, (double*)d_out);
    else if (dtype_flags == 1)
        hipLaunchKernelGGL((flow2compdis_kernel<double, float>), grid, block, 0, (hipStream_t)stream, n,
                           (const double*)d_flow, (const float*)d_dt0, sensor_dt, (double*)d_out);
    else
        hipLaunchKernelGGL((flow2compdis_kernel<float, float>), grid, block, 0, (hipStream_t)stream, n,
                           (const float*)d_flow, (const float*)d_dt0, (float)sensor_dt, (float*)d_out);
    HIMO_LAUNCH_CHECK("flow2compdis_kernel");
    return HIMO_OK;
}

extern "C" int himo_refine_pts(int64_t n, const float* d_pc, int pc_stride, const void* d_ds, int dtype_is_f64,
                               void* d_out, void* stream) {
    if (n < 0 || pc_stride < 3) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_pc || !d_ds || !d_out) return HIMO_ERR_INVALID_ARGUMENT;
    const dim3 grid((unsigned)((n * 3 + 255) / 256)), block(256);
    if (dtype_is_f64)
        hipLaunchKernelGGL(refine_pts_kernel<double>, grid, block, 0, (hipStream_t)stream, n, d_pc, pc_stride,
                           (const double*)d_ds, (double*)d_out);
    else
        hipLaunchKernelGGL(refine_pts_kernel<float>, grid, block, 0, (hipStream_t)stream, n, d_pc, pc_stride,
                           (const float*)d_ds, (float*)d_out);
    HIMO_LAUNCH_CHECK("refine_pts_kernel");
    return HIMO_OK;
}

extern "C" int himo_ego_pts_mask(int64_t n, const float* d_pts, int pc_stride, const float* h_bounds, uint8_t* d_out,
                                 void* stream) {
    if (n < 0 || pc_stride < 3 || !h_bounds) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_pts || !d_out) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(ego_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d_pts,
                       pc_stride, h_bounds[0], h_bounds[1], h_bounds[2], h_bounds[3], h_bounds[4], h_bounds[5], d_out);
    HIMO_LAUNCH_CHECK("ego_mask_kernel");
    return HIMO_OK;
}

extern "C" int himo_dt0(int64_t n, const float* d_lidar_dt, float* d_dt0, void* d_workspace, void* stream) {
    if (n < 0 || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_ERR_EMPTY_FRAME;
    if (!d_lidar_dt || !d_dt0) return HIMO_ERR_INVALID_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    // workspace: [key (4 B) | pad | offsets[2] at +16]; poses are not needed (pose0 == nullptr)
    unsigned* key = reinterpret_cast<unsigned*>(d_workspace);
    HIMO_HIP(hipMemsetAsync(key, 0, 16, s));
    int64_t h_off[2] = {0, n};
    int64_t* d_off = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(d_workspace) + 16);
    HIMO_HIP(hipMemcpyAsync(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice, s));
    const int n_chunks = (int)((n + kPrepChunk - 1) / kPrepChunk);
    {
        ProfScope ps("frame_prep_kernel_dt0", s);
        hipLaunchKernelGGL(frame_prep_kernel, dim3(n_chunks), dim3(kPrepThreads), 0, s, 1, n, n_chunks, 0, d_off,
                           (const double*)nullptr, (const double*)nullptr, d_lidar_dt, key, (FrameXf*)nullptr);
    }
    HIMO_LAUNCH_CHECK("frame_prep_kernel");
    ProfScope ps("dt0_kernel", s);
    hipLaunchKernelGGL(dt0_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_lidar_dt, key, d_dt0);
    HIMO_LAUNCH_CHECK("dt0_kernel");
    return HIMO_OK;
}
