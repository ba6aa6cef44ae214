// compdis_math.h -- device math shared by the comp_dis kernels and the evaluator kernels.
// Reference arithmetic: save_zip.py:114-121, utils/__init__.py:36-47 (see compdis.hip for the notes
// on operation order).  Translation units including this must be built with -ffp-contract=off.
#pragma once
#include "himo_common.h"
#include <math.h>

namespace himo {

struct FrameXf {   // ego transform of one frame: p' = R p + t
    double R[9];
    double t[3];
};

constexpr int kPrepThreads = 256;
constexpr int kPrepChunk = 4096;      // points per block in the max pre-pass
constexpr int kThreads = 256;
constexpr int kPtsPerThread = 4;
constexpr int kBlockPts = kThreads * kPtsPerThread;

struct WorkspaceLayout {
    unsigned* keys;   // [n_frames] order-preserving keys of max(lidar_dt)
    FrameXf* xf;      // [n_frames]
};

__host__ __device__ inline size_t keys_bytes(int n_frames) {
    return ((size_t)n_frames * sizeof(unsigned) + 15) / 16 * 16;
}

// ------------------------------------------------------------------------------------------
// per-frame helpers
// ------------------------------------------------------------------------------------------
// largest f in [0, n_frames) with offsets[f] <= i   (i < offsets[n_frames])
__device__ inline int find_frame(const int64_t* __restrict__ offsets, int n_frames, int64_t i) {
    int lo = 0, hi = n_frames;   // invariant: offsets[lo] <= i < offsets[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

struct XfRegs {   // one frame's transform + max, held in registers (SGPRs on the uniform path)
    double R[9];
    double t[3];
    float fmax;
};

__device__ inline XfRegs load_xf(const FrameXf* __restrict__ xf, const unsigned* __restrict__ keys, int f) {
    XfRegs x;
#pragma unroll
    for (int i = 0; i < 9; ++i) x.R[i] = xf[f].R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) x.t[i] = xf[f].t[i];
    x.fmax = key_to_float(keys[f]);
    return x;
}

template <bool F32>
__device__ inline void point_math(const XfRegs& x, float px, float py, float pz, float fx, float fy, float fz,
                                  float dt, double sensor_dt, bool raw, float* cd, float* rf) {
    const float dt0 = x.fmax - dt;                                   // save_zip.py:120 (float32)
    if (F32) {
        const float r[9] = {(float)x.R[0], (float)x.R[1], (float)x.R[2], (float)x.R[3], (float)x.R[4],
                            (float)x.R[5], (float)x.R[6], (float)x.R[7], (float)x.R[8]};
        const float sdt = (float)sensor_dt;
        const float p[3] = {px, py, pz}, fl[3] = {fx, fy, fz};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pf = (fmaf(pz, r[c * 3 + 2], fmaf(py, r[c * 3 + 1], px * r[c * 3])) + (float)x.t[c]) - p[c];
            const float est = raw ? 0.0f : fl[c] - pf;
            const float v = est / sdt * dt0;                         // utils/__init__.py:43
            cd[c] = v;
            rf[c] = p[c] + v;                                        // utils/__init__.py:46
        }
    } else {
        const double p[3] = {(double)px, (double)py, (double)pz}, fl[3] = {(double)fx, (double)fy, (double)fz};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // save_zip.py:116 -- dgemm accumulates k-ordered fused multiply-adds
            const double pf = (fma(p[2], x.R[c * 3 + 2], fma(p[1], x.R[c * 3 + 1], p[0] * x.R[c * 3])) + x.t[c]) - p[c];
            const double est = raw ? 0.0 : fl[c] - pf;               // save_zip.py:117
            const double v = est / sensor_dt * (double)dt0;          // utils/__init__.py:43
            cd[c] = (float)v;                                        // save_zip.py:70-72
            rf[c] = (float)(p[c] + v);                               // utils/__init__.py:46
        }
    }
}


// host: zero the per-frame max keys, then launch frame_prep_kernel (max(lidar_dt) + ego transforms)
int launch_frame_prep(int n_frames, int64_t total, const int64_t* d_offsets, const double* d_pose0,
                      const double* d_pose1, unsigned flags, const float* d_lidar_dt, void* d_workspace, hipStream_t s);

inline WorkspaceLayout carve(void* ws, int n_frames) {
    WorkspaceLayout w;
    w.keys = reinterpret_cast<unsigned*>(ws);
    w.xf = reinterpret_cast<FrameXf*>(reinterpret_cast<char*>(ws) + keys_bytes(n_frames));
    return w;
}

}  // namespace himo
