// head.hip -- stage a10, per-point head glue: gather pillar features back to points (the "scatter/
// gather" pair of the pillar front end) and the final flow composition.  The GRU / MLP matrix work
// in between runs on the matrix cores through conv.hip's 1x1 row-GEMM with fused gate epilogues.
//
// Specification: himo_amd/seflow/spec.py steps 5-6; oracle: oracle/seflow_oracle.py::head / forward.
// Output contract (the only in-tree fact): (N,3) float32 flow INCLUDING ego motion, row-aligned
// with pc0 (save_zip.py:117, tools/test/score.py:583).
#include "himo_common.h"
#include <math.h>

namespace himo {

struct GatherArgs {
    int64_t n;
    const int* pid; const float* offsets;
    const float* img0; const float* img1; int img_pitch;      // pc0 / pc1 pillar images (32 ch at [cell*pitch])
    const float* dec; int dec_pitch;                          // decoder map (64 ch)
    const float* w_off; const float* b_off;                   // [3][64], [64]
    float* hx; float* rhx; int pitch;                         // [n][192]
};

// one lane per float4 of the 192-float row: 8 + 8 + 16 + 16 = 48 lanes per point
__global__ __launch_bounds__(256) void head_gather_kernel(GatherArgs a) {
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = item / 48;
    const int q = (int)(item % 48);
    if (i >= a.n) return;
    const int cell = a.pid[i];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < 32) {
        if (cell >= 0) {
            const float* src = q < 8 ? a.img0 + (int64_t)cell * a.img_pitch + q * 4
                             : q < 16 ? a.img1 + (int64_t)cell * a.img_pitch + (q - 8) * 4
                                      : a.dec + (int64_t)cell * a.dec_pitch + (q - 16) * 4;
            v = *reinterpret_cast<const float4*>(src);
        }
        *reinterpret_cast<float4*>(a.hx + i * a.pitch + q * 4) = v;
    } else {
        const int c = (q - 32) * 4;
        const float o0 = a.offsets[i * 3], o1 = a.offsets[i * 3 + 1], o2 = a.offsets[i * 3 + 2];
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            r[k] = fmaf(o2, a.w_off[128 + c + k], fmaf(o1, a.w_off[64 + c + k], o0 * a.w_off[c + k])) + a.b_off[c + k];
        v = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(a.hx + i * a.pitch + 128 + c) = v;
        *reinterpret_cast<float4*>(a.rhx + i * a.pitch + 128 + c) = v;
    }
}

struct FinalArgs {
    int64_t n;
    const float* y1; int y1_pitch;            // [n][32] after GELU
    const float* w2; const float* b2;         // [32][3], [3]
    const int* pid;
    const float* xyz_t;                       // [n][3] pc0 in pc1's frame
    const float* pts; int stride;             // raw pc0 rows
    float* flow;                              // [n][3]
};

__global__ __launch_bounds__(256) void head_final_kernel(FinalArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float res[3] = {0.f, 0.f, 0.f};
    if (a.pid[i] >= 0) {
        const float* y = a.y1 + i * a.y1_pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = y[0] * a.w2[c];
#pragma unroll
            for (int k = 1; k < 32; ++k) s = fmaf(y[k], a.w2[k * 3 + c], s);
            res[c] = s + a.b2[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float pose_flow = a.xyz_t[i * 3 + c] - a.pts[i * a.stride + c];
        a.flow[i * 3 + c] = a.pid[i] >= 0 ? pose_flow + res[c] : pose_flow;
    }
}

}  // namespace himo

using namespace himo;

extern "C" int himo_head_gather(int64_t n, const int32_t* d_pid, const float* d_offsets, const float* d_img0,
                                const float* d_img1, int img_pitch, const float* d_dec, int dec_pitch,
                                const float* d_w_off, const float* d_b_off, float* d_hx, float* d_rhx, int pitch,
                                void* stream) {
    if (n < 0 || pitch < 192 || (pitch & 3) || (img_pitch & 3) || (dec_pitch & 3)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_pid || !d_offsets || !d_img0 || !d_img1 || !d_dec || !d_w_off || !d_b_off || !d_hx || !d_rhx) return HIMO_ERR_INVALID_ARGUMENT;
    GatherArgs a{n, d_pid, d_offsets, d_img0, d_img1, img_pitch, d_dec, dec_pitch, d_w_off, d_b_off, d_hx, d_rhx, pitch};
    ProfScope ps("head_gather_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(head_gather_kernel, dim3((unsigned)((n * 48 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    HIMO_LAUNCH_CHECK("head_gather_kernel");
    return HIMO_OK;
}

extern "C" int himo_head_final(int64_t n, const float* d_y1, int y1_pitch, const float* d_w2, const float* d_b2,
                               const int32_t* d_pid, const float* d_xyz_t, const float* d_pts, int pc_stride,
                               float* d_flow, void* stream) {
    if (n < 0 || pc_stride < 3 || y1_pitch < 32) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_y1 || !d_w2 || !d_b2 || !d_pid || !d_xyz_t || !d_pts || !d_flow) return HIMO_ERR_INVALID_ARGUMENT;
    FinalArgs a{n, d_y1, y1_pitch, d_w2, d_b2, d_pid, d_xyz_t, d_pts, pc_stride, d_flow};
    ProfScope ps("head_final_kernel", (hipStream_t)stream);
    hipLaunchKernelGGL(head_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    HIMO_LAUNCH_CHECK("head_final_kernel");
    return HIMO_OK;
}
