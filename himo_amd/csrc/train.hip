// train.hip -- stage a11, training side: the element-wise / gather-scatter kernels of the network's backward
// pass (the matrix products run on conv.hip / convbf.hip row GEMMs and fastnsf.hip's split-K weight gradients).
//
// PARITY UNPINNED: the reference trains through `OpenSceneFlow/train.py` (assets/slurm/ssl-train-av2.sh:31), which is
// absent; this is the backward pass of this build's own network (himo_amd/seflow/spec.py).  Oracle: PyTorch CPU
// autograd through oracle/seflow_oracle.py.
//
// Training-mode conventions (himo_amd/seflow/train.py): BatchNorm statistics are frozen (scale / shift constants);
// GRU gates use exact expf / tanhf; every reduction is a fixed-order tree or an ordered per-cell sum -- no float
// atomics anywhere in the backward pass.
#include "himo_common.h"
#include <math.h>

namespace himo {

__device__ inline float sigmoid_exact(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ inline float gelu_f(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ inline float gelu_grad(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}

// ---- GRU cell, training forward: gates from the two pre-activation GEMMs -------------------------------------
// z = sigmoid(azr[:, :128]); r = sigmoid(azr[:, 128:]); rhx = [r * h, x]
__global__ __launch_bounds__(256) void gru_gate1_kernel(int64_t n, const float* __restrict__ azr, const float* __restrict__ hx,
                                                        float* __restrict__ z, float* __restrict__ r, float* __restrict__ rhx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 192) return;
    const int64_t i = e / 192;
    const int c = (int)(e % 192);
    if (c < 128) {
        const float zz = sigmoid_exact(azr[i * 256 + c]), rr = sigmoid_exact(azr[i * 256 + 128 + c]);
        z[i * 128 + c] = zz; r[i * 128 + c] = rr;
        rhx[i * 192 + c] = rr * hx[i * 192 + c];
    } else {
        rhx[i * 192 + c] = hx[i * 192 + c];
    }
}

// q = tanh(aq); hx_next = [(1 - z) * h + z * q, x]
__global__ __launch_bounds__(256) void gru_gate2_kernel(int64_t n, const float* __restrict__ aq, const float* __restrict__ z,
                                                        const float* __restrict__ hx, float* __restrict__ q, float* __restrict__ hx_next) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 192) return;
    const int64_t i = e / 192;
    const int c = (int)(e % 192);
    if (c < 128) {
        const float qq = tanhf(aq[i * 128 + c]), zz = z[i * 128 + c];
        q[i * 128 + c] = qq;
        hx_next[i * 192 + c] = (1.0f - zz) * hx[i * 192 + c] + zz * qq;
    } else {
        hx_next[i * 192 + c] = hx[i * 192 + c];
    }
}

// ---- GRU cell, backward ------------------------------------------------------------------------------------------
// dz = dh' * (q - h); daq = dh' * z * (1 - q^2); dhp = dh' * (1 - z)
__global__ __launch_bounds__(256) void gru_bwd1_kernel(int64_t n, const float* __restrict__ dhn, const float* __restrict__ z,
                                                       const float* __restrict__ q, const float* __restrict__ hx,
                                                       float* __restrict__ daq, float* __restrict__ dz, float* __restrict__ dhp) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 128) return;
    const int64_t i = e / 128;
    const int c = (int)(e % 128);
    const float g = dhn[e], zz = z[e], qq = q[e], h = hx[i * 192 + c];
    dz[e] = g * (qq - h);
    daq[e] = g * zz * (1.0f - qq * qq);
    dhp[e] = g * (1.0f - zz);
}

// d_rhx = daq Wq^T.  dr = d_rhx[:, :128] * h; dhp += d_rhx[:, :128] * r; dazr = [dz z (1-z), dr r (1-r)]; dx += d_rhx[:, 128:]
__global__ __launch_bounds__(256) void gru_bwd2_kernel(int64_t n, const float* __restrict__ d_rhx, const float* __restrict__ hx,
                                                       const float* __restrict__ z, const float* __restrict__ r,
                                                       const float* __restrict__ dz, float* __restrict__ dhp,
                                                       float* __restrict__ dazr, float* __restrict__ dx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 192) return;
    const int64_t i = e / 192;
    const int c = (int)(e % 192);
    if (c < 128) {
        const float drh = d_rhx[i * 192 + c], h = hx[i * 192 + c], rr = r[i * 128 + c], zz = z[i * 128 + c];
        dhp[i * 128 + c] += drh * rr;
        dazr[i * 256 + c] = dz[i * 128 + c] * zz * (1.0f - zz);
        dazr[i * 256 + 128 + c] = (drh * h) * rr * (1.0f - rr);
    } else {
        dx[i * 64 + (c - 128)] += d_rhx[i * 192 + c];
    }
}

// d_hx = dazr Wzr^T.  dh = dhp + d_hx[:, :128]; dx += d_hx[:, 128:]
__global__ __launch_bounds__(256) void gru_bwd3_kernel(int64_t n, const float* __restrict__ d_hx, const float* __restrict__ dhp,
                                                       float* __restrict__ dh, float* __restrict__ dx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 192) return;
    const int64_t i = e / 192;
    const int c = (int)(e % 192);
    if (c < 128) dh[i * 128 + c] = dhp[i * 128 + c] + d_hx[i * 192 + c];
    else dx[i * 64 + (c - 128)] += d_hx[i * 192 + c];
}

// ---- activations -------------------------------------------------------------------------------------------------------
// y = gelu(x * scale[c] + shift[c]) and the affine pre-activation is kept for the backward pass (scale may be null: 1, 0)
__global__ __launch_bounds__(256) void affine_gelu_fwd_kernel(int64_t rows, int ch, const float* __restrict__ x, int x_pitch,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ pre, int pre_pitch, float* __restrict__ y, int y_pitch) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * ch) return;
    const int64_t i = e / ch;
    const int c = (int)(e % ch);
    float v = x[i * x_pitch + c];
    if (scale) v = v * scale[c] + shift[c];
    pre[i * pre_pitch + c] = v;
    y[i * y_pitch + c] = gelu_f(v);
}

// dx = dy * gelu'(pre) * scale[c]   (in place on dy allowed)
__global__ __launch_bounds__(256) void affine_gelu_bwd_kernel(int64_t rows, int ch, const float* __restrict__ dy, int dy_pitch,
                                                              const float* __restrict__ pre, int pre_pitch, const float* __restrict__ scale,
                                                              float* __restrict__ dx, int dx_pitch) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * ch) return;
    const int64_t i = e / ch;
    const int c = (int)(e % ch);
    const float g = dy[i * dy_pitch + c] * gelu_grad(pre[i * pre_pitch + c]);
    dx[i * dx_pitch + c] = scale ? g * scale[c] : g;
}

// rows of `v` whose cell id is negative are zeroed (points the pillar grid dropped take no part in training)
__global__ __launch_bounds__(256) void mask_rows_kernel(int64_t n, int cols, const int* __restrict__ pid, float* __restrict__ v, int pitch) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * cols) return;
    const int64_t i = e / cols;
    if (pid[i] < 0) v[i * pitch + (int)(e % cols)] = 0.f;
}

}  // namespace himo

using namespace himo;

#define HIMO_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256), 0, (hipStream_t)stream

extern "C" int himo_gru_gates_fwd(int64_t n, int which, const float* d_pre, const float* d_z_in, const float* d_hx,
                                  float* d_z, float* d_r, float* d_q, float* d_out, void* stream) {
    if (n < 0 || !(which == 1 || which == 2)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_pre || !d_hx || !d_out) return HIMO_ERR_INVALID_ARGUMENT;
    if (which == 1) {
        if (!d_z || !d_r) return HIMO_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(gru_gate1_kernel, HIMO_GRID(n * 192), n, d_pre, d_hx, d_z, d_r, d_out);
    } else {
        if (!d_z_in || !d_q) return HIMO_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(gru_gate2_kernel, HIMO_GRID(n * 192), n, d_pre, d_z_in, d_hx, d_q, d_out);
    }
    HIMO_LAUNCH_CHECK("gru_gate_kernel");
    return HIMO_OK;
}

extern "C" int himo_gru_bwd1(int64_t n, const float* d_dh_next, const float* d_z, const float* d_q, const float* d_hx,
                             float* d_daq, float* d_dz, float* d_dhp, void* stream) {
    if (n < 0) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_dh_next || !d_z || !d_q || !d_hx || !d_daq || !d_dz || !d_dhp) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gru_bwd1_kernel, HIMO_GRID(n * 128), n, d_dh_next, d_z, d_q, d_hx, d_daq, d_dz, d_dhp);
    HIMO_LAUNCH_CHECK("gru_bwd1_kernel");
    return HIMO_OK;
}

extern "C" int himo_gru_bwd2(int64_t n, const float* d_d_rhx, const float* d_hx, const float* d_z, const float* d_r,
                             const float* d_dz, float* d_dhp, float* d_dazr, float* d_dx, void* stream) {
    if (n < 0) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_d_rhx || !d_hx || !d_z || !d_r || !d_dz || !d_dhp || !d_dazr || !d_dx) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gru_bwd2_kernel, HIMO_GRID(n * 192), n, d_d_rhx, d_hx, d_z, d_r, d_dz, d_dhp, d_dazr, d_dx);
    HIMO_LAUNCH_CHECK("gru_bwd2_kernel");
    return HIMO_OK;
}

extern "C" int himo_gru_bwd3(int64_t n, const float* d_d_hx, const float* d_dhp, float* d_dh, float* d_dx, void* stream) {
    if (n < 0) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_d_hx || !d_dhp || !d_dh || !d_dx) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gru_bwd3_kernel, HIMO_GRID(n * 192), n, d_d_hx, d_dhp, d_dh, d_dx);
    HIMO_LAUNCH_CHECK("gru_bwd3_kernel");
    return HIMO_OK;
}

extern "C" int himo_affine_gelu_fwd(int64_t rows, int ch, const float* d_x, int x_pitch, const float* d_scale, const float* d_shift,
                                    float* d_pre, int pre_pitch, float* d_y, int y_pitch, void* stream) {
    if (rows < 0 || ch < 1) return HIMO_ERR_INVALID_ARGUMENT;
    if (rows == 0) return HIMO_OK;
    if (!d_x || !d_pre || !d_y || ((d_scale == nullptr) != (d_shift == nullptr))) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(affine_gelu_fwd_kernel, HIMO_GRID(rows * ch), rows, ch, d_x, x_pitch, d_scale, d_shift, d_pre, pre_pitch, d_y, y_pitch);
    HIMO_LAUNCH_CHECK("affine_gelu_fwd_kernel");
    return HIMO_OK;
}

extern "C" int himo_affine_gelu_bwd(int64_t rows, int ch, const float* d_dy, int dy_pitch, const float* d_pre, int pre_pitch,
                                    const float* d_scale, float* d_dx, int dx_pitch, void* stream) {
    if (rows < 0 || ch < 1) return HIMO_ERR_INVALID_ARGUMENT;
    if (rows == 0) return HIMO_OK;
    if (!d_dy || !d_pre || !d_dx) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(affine_gelu_bwd_kernel, HIMO_GRID(rows * ch), rows, ch, d_dy, dy_pitch, d_pre, pre_pitch, d_scale, d_dx, dx_pitch);
    HIMO_LAUNCH_CHECK("affine_gelu_bwd_kernel");
    return HIMO_OK;
}

extern "C" int himo_mask_rows(int64_t n, int cols, const int32_t* d_pid, float* d_v, int pitch, void* stream) {
    if (n < 0 || cols < 1 || pitch < cols) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_pid || !d_v) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(mask_rows_kernel, HIMO_GRID(n * cols), n, cols, d_pid, d_v, pitch);
    HIMO_LAUNCH_CHECK("mask_rows_kernel");
    return HIMO_OK;
}
