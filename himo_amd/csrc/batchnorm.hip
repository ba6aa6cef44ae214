// batchnorm.hip -- stage a11, BatchNorm in TRAINING mode for the encoder's conv -> BN -> GELU layers (BASELINE config 5).
//
// Why it exists: the reference's job trains from scratch (assets/slurm/ssl-train-av2.sh:31-34 passes no checkpoint=,
// 12 epochs, batch_size=8), so its BatchNorm layers normalise with BATCH statistics, learn gamma / beta and update their
// running statistics.  PARITY UNPINNED: the network source (OpenSceneFlow) is absent; the semantics below are
// torch.nn.BatchNorm2d's (biased variance for the normalisation, unbiased for the running estimate, momentum 0.1) applied
// to this build's own specification (himo_amd/seflow/spec.py: the statistics of a layer are taken over the images of ONE
// forward call -- the F frames of a sample, frames as the batch); oracle: oracle/seflow_oracle.py forward_train(training=True).
//
// All three stages are HBM-bound streams over an [n_img][rows][ch] NHWC map with a per-channel reduction:
//   forward   stats (read x once, float64 sums)  ->  finalize  ->  normalise + GELU (read x, write xhat and y)
//   backward  float64 sums of g = dy * gelu'(gamma xhat + beta) and of g xhat (read dy, xhat)  ->  finalize  ->
//             dx = gamma invstd (g - mean(g) - xhat mean(g xhat)), g evaluated again    (read dy, xhat; write dx)
// Every reduction is a fixed two-level tree (block partials -> one block per 128 channels): bit-deterministic, no atomics.
// Algorithmic bytes per element: forward 4 + 4 + 8 = 16 B, backward 8 + 8 + 4 = 20 B.
#include "conv_common.h"
#include <math.h>

namespace himo {

struct BnMap {                 // [n_img][rows][ch] view: element (i, r, c) at p + i * img_stride + r * pitch + c
    const float* p; int64_t img_stride; int pitch;
};
struct BnMapW { float* p; int64_t img_stride; int pitch; };

// GELU and its derivative as the convolution epilogues evaluate them (conv_common.h: erf to 1.5e-7 absolute, hardware exp2 / rcp): the
// library's erff / expf made both element-wise passes instruction-bound (~80 instructions per activation against 12-24 B of traffic)
__device__ inline float bn_gelu(float v) { return gelu_exact(v); }
__device__ inline float bn_gelu_grad(float v) { return gelu_grad_exact(v); }

// block = 8 row groups x 32 float4 columns (a 128-channel tile, blockIdx.y) over `rows_pb` consecutive global rows
__device__ inline int64_t bn_addr(int64_t r, int64_t rows, int64_t img_stride, int pitch) {
    const int64_t img = r / rows;
    return img * img_stride + (r - img * rows) * pitch;
}
// The row walks below split their FIRST global row into (image, row in image) with 32-bit divisions (the entry points admit only
// total rows x quads < 2^31) and then step: a 64-bit division per row and quad -- ~150 instructions, emulated -- used to be most of
// these kernels' instruction count.
struct BnRow {
    int img, rr;
    __device__ inline BnRow(unsigned r, unsigned rows) : img((int)(r / rows)), rr((int)(r - (r / rows) * rows)) {}
    __device__ inline void step(int by, int rows) { rr += by; while (rr >= rows) { rr -= rows; ++img; } }
    __device__ inline int64_t at(int64_t img_stride, int pitch) const { return (int64_t)img * img_stride + (int64_t)rr * pitch; }
};

// narrow layers keep every lane busy: a block covers 2^qs float4 columns (32 for > 64 channels, 16 for > 32, else 8) x 256 >> qs row groups
__device__ __host__ inline int bn_quad_shift(int ch) { return ch > 64 ? 5 : (ch > 32 ? 4 : 3); }

// the block's two per-channel partial sums -> partial[tile][block][2][128], groups combined in fixed order
__device__ inline void bn_block_partials(const double (&s0)[4], const double (&s1)[4], int qs, int q, int grp, int G, double* __restrict__ partial) {
    __shared__ double sh[2][1024];                       // [which][grp][4 << qs]
    const int w = 4 << qs;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[0][grp * w + q * 4 + k] = s0[k]; sh[1][grp * w + q * 4 + k] = s1[k]; }
    __syncthreads();
    const int which = threadIdx.x >> 7, lc = threadIdx.x & 127;
    double t = 0.0;
    if (lc < w) {
        t = sh[which][lc];
        for (int g = 1; g < G; ++g) t += sh[which][g * w + lc];
    }
    partial[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + which) * 128 + lc] = t;
}

// ---- forward: per-channel sum and sum of squares (float64) -------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(int64_t total, int64_t rows, int ch, BnMap x, double* __restrict__ partial,
                                                               int rows_pb) {
    const int qs = bn_quad_shift(ch), q = threadIdx.x & ((1 << qs) - 1), grp = threadIdx.x >> qs, G = 256 >> qs;
    const int col = (int)blockIdx.y * 128 + q * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_pb;
    const int64_t r1 = r0 + rows_pb < total ? r0 + rows_pb : total;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (col < ch) {
        BnRow row((unsigned)(r0 + grp), (unsigned)rows);
#pragma unroll 4
        for (int64_t r = r0 + grp; r < r1; r += G, row.step(G, (int)rows)) {
            const float4 v = *reinterpret_cast<const float4*>(x.p + row.at(x.img_stride, x.pitch) + col);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
        }
    }
    bn_block_partials(s, ss, qs, q, grp, G, partial);
}

// the two per-channel sums over the block partials.  One 1024-thread block per 16 channels (blockIdx.x = tile * 8 + chunk): 64 groups x
// 16 channels, each thread four independent chains over every 64th partial, fixed-order combine -- a single serial chain of dependent
// float64 loads per thread made this 40 us.  Result valid in threads 0..15; returns the thread's channel.
__device__ inline int bn_sum_partials(const double* __restrict__ partial, int n_blocks, int ch, double& s0, double& s1) {
    __shared__ double sh[2][64][16];
    const int lc = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int tile = (int)blockIdx.x >> 3, chunk = (int)blockIdx.x & 7;
    const int col = tile * 128 + chunk * 16 + lc;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (col < ch) {
        const double* p = partial + (int64_t)tile * n_blocks * 256 + chunk * 16 + lc;
        int i = grp;
        for (; i + 192 < n_blocks; i += 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] += p[(int64_t)(i + 64 * k) * 256]; b[k] += p[(int64_t)(i + 64 * k) * 256 + 128]; }
        }
        for (; i < n_blocks; i += 64) { a[0] += p[(int64_t)i * 256]; b[0] += p[(int64_t)i * 256 + 128]; }
    }
    sh[0][grp][lc] = (a[0] + a[1]) + (a[2] + a[3]);
    sh[1][grp][lc] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (threadIdx.x < 16) {
        for (int g = 0; g < 64; ++g) { s0 += sh[0][g][lc]; s1 += sh[1][g][lc]; }
    }
    return col;
}

// fixed-order sum of the block partials, then the layer's constants.
// consts [4][ch]: mean, invstd (forward) -- the backward reuses the buffer for its own three coefficient rows.
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const double* __restrict__ partial, int n_blocks, int ch, double count,
                                                                float eps, float momentum, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, float* __restrict__ mean_out,
                                                                float* __restrict__ invstd_out) {
    double s, ss;
    const int col = bn_sum_partials(partial, n_blocks, ch, s, ss);
    if (threadIdx.x < 16 && col < ch) {
        const double mean = s / count;
        double var = ss / count - mean * mean;            // biased: what the normalisation uses
        if (var < 0.0) var = 0.0;
        mean_out[col] = (float)mean;
        invstd_out[col] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {                               // torch: running estimate uses the UNBIASED variance
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[col] = (float)((1.0 - (double)momentum) * (double)running_mean[col] + (double)momentum * mean);
            running_var[col] = (float)((1.0 - (double)momentum) * (double)running_var[col] + (double)momentum * unbiased);
        }
    }
}

// xhat = (x - mean) * invstd; y = gelu(gamma * xhat + beta)
__global__ __launch_bounds__(256) void bn_normalize_gelu_kernel(int64_t total, int64_t rows, int ch, BnMap x, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, BnMapW xhat, BnMapW y) {
    const unsigned c4 = (unsigned)ch >> 2;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (unsigned)total * c4) return;
    const unsigned r = e / c4;
    const int col = (int)(e - r * c4) * 4;
    const BnRow row(r, (unsigned)rows);
    const int64_t img = row.img, rr = row.rr;
    const float4 v = *reinterpret_cast<const float4*>(x.p + img * x.img_stride + rr * x.pitch + col);
    const float4 m = *reinterpret_cast<const float4*>(mean + col), is = *reinterpret_cast<const float4*>(invstd + col);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + col), be = *reinterpret_cast<const float4*>(beta + col);
    float4 h, o;
    h.x = (v.x - m.x) * is.x; h.y = (v.y - m.y) * is.y; h.z = (v.z - m.z) * is.z; h.w = (v.w - m.w) * is.w;
    o.x = bn_gelu(ga.x * h.x + be.x); o.y = bn_gelu(ga.y * h.y + be.y); o.z = bn_gelu(ga.z * h.z + be.z); o.w = bn_gelu(ga.w * h.w + be.w);
    if (xhat.p) *reinterpret_cast<float4*>(xhat.p + img * xhat.img_stride + rr * xhat.pitch + col) = h;      // (optional: see himo_bn_train_bwd_x)
    *reinterpret_cast<float4*>(y.p + img * y.img_stride + rr * y.pitch + col) = o;
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
// g = dy * gelu'(gamma * xhat + beta) -> dx; partial float64 sums of g and g * xhat per channel
// FROM_X: `xhat` holds the layer's INPUT x and xhat = (x - mean) * invstd is formed here, as the forward pass formed it (the forward
// pass then need not write xhat at all: 4 of its 16 B per activation)
template <bool FROM_X>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(int64_t total, int64_t rows, int ch, BnMap dy, BnMap xhat,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             double* __restrict__ partial, int rows_pb) {
    const int qs = bn_quad_shift(ch), q = threadIdx.x & ((1 << qs) - 1), grp = threadIdx.x >> qs, G = 256 >> qs;
    const int col = (int)blockIdx.y * 128 + q * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_pb;
    const int64_t r1 = r0 + rows_pb < total ? r0 + rows_pb : total;
    // a thread's own rows (a few dozen at most: rows_pb / G) are summed in float32, everything across threads and blocks in float64
    float fs[4] = {0, 0, 0, 0}, fx[4] = {0, 0, 0, 0};
    if (col < ch) {
        const float4 ga = *reinterpret_cast<const float4*>(gamma + col), be = *reinterpret_cast<const float4*>(beta + col);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), is = make_float4(1.f, 1.f, 1.f, 1.f);
        if (FROM_X) { m = *reinterpret_cast<const float4*>(mean + col); is = *reinterpret_cast<const float4*>(invstd + col); }
        BnRow row((unsigned)(r0 + grp), (unsigned)rows);
#pragma unroll 4
        for (int64_t r = r0 + grp; r < r1; r += G, row.step(G, (int)rows)) {
            const int64_t img = row.img, rr = row.rr;
            const float4 d = *reinterpret_cast<const float4*>(dy.p + img * dy.img_stride + rr * dy.pitch + col);
            float4 h = *reinterpret_cast<const float4*>(xhat.p + img * xhat.img_stride + rr * xhat.pitch + col);
            if (FROM_X) { h.x = (h.x - m.x) * is.x; h.y = (h.y - m.y) * is.y; h.z = (h.z - m.z) * is.z; h.w = (h.w - m.w) * is.w; }
            float4 g;
            g.x = d.x * bn_gelu_grad(ga.x * h.x + be.x); g.y = d.y * bn_gelu_grad(ga.y * h.y + be.y);
            g.z = d.z * bn_gelu_grad(ga.z * h.z + be.z); g.w = d.w * bn_gelu_grad(ga.w * h.w + be.w);
            fs[0] += g.x; fs[1] += g.y; fs[2] += g.z; fs[3] += g.w;
            fx[0] = fmaf(g.x, h.x, fx[0]); fx[1] = fmaf(g.y, h.y, fx[1]); fx[2] = fmaf(g.z, h.z, fx[2]); fx[3] = fmaf(g.w, h.w, fx[3]);
        }
    }
    const double s[4] = {fs[0], fs[1], fs[2], fs[3]}, sx[4] = {fx[0], fx[1], fx[2], fx[3]};
    bn_block_partials(s, sx, qs, q, grp, G, partial);
}

// dbeta = sum g, dgamma = sum g xhat; coefficient rows for the element-wise pass: k1 = gamma invstd, k2 = dbeta / N, k3 = dgamma / N
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int n_blocks, int ch, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ coef) {
    double sg, sgx;
    const int col = bn_sum_partials(partial, n_blocks, ch, sg, sgx);
    if (threadIdx.x < 16 && col < ch) {
        dbeta[col] = accumulate ? dbeta[col] + (float)sg : (float)sg;
        dgamma[col] = accumulate ? dgamma[col] + (float)sgx : (float)sgx;
        coef[col] = gamma[col] * invstd[col];
        coef[ch + col] = (float)(sg / count);
        coef[2 * ch + col] = (float)(sgx / count);
    }
}

// dx = k1 * (g - k2 - xhat * k3) with g = dy * gelu'(gamma xhat + beta) evaluated AGAIN (the same expression on the same operands
// as in the sums: the same bits) -- the partial pass used to park g in dx (4 B per activation written, 4 B read back here; GELU' is
// ~20 instructions since round 5, cheaper than the round trip).  dx may alias dy (each element is read before it is written).
template <bool FROM_X>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(int64_t total, int64_t rows, int ch, BnMap dy, BnMap xhat,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, BnMapW dx) {
    const unsigned c4 = (unsigned)ch >> 2;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (unsigned)total * c4) return;
    const unsigned r = e / c4;
    const int col = (int)(e - r * c4) * 4;
    const BnRow row(r, (unsigned)rows);
    const int64_t img = row.img, rr = row.rr;
    float* p = dx.p + img * dx.img_stride + rr * dx.pitch + col;
    const float4 d = *reinterpret_cast<const float4*>(dy.p + img * dy.img_stride + rr * dy.pitch + col);
    float4 h = *reinterpret_cast<const float4*>(xhat.p + img * xhat.img_stride + rr * xhat.pitch + col);
    if (FROM_X) {
        const float4 m = *reinterpret_cast<const float4*>(mean + col), is = *reinterpret_cast<const float4*>(invstd + col);
        h.x = (h.x - m.x) * is.x; h.y = (h.y - m.y) * is.y; h.z = (h.z - m.z) * is.z; h.w = (h.w - m.w) * is.w;
    }
    const float4 ga = *reinterpret_cast<const float4*>(gamma + col), be = *reinterpret_cast<const float4*>(beta + col);
    float4 g;
    g.x = d.x * bn_gelu_grad(ga.x * h.x + be.x); g.y = d.y * bn_gelu_grad(ga.y * h.y + be.y);
    g.z = d.z * bn_gelu_grad(ga.z * h.z + be.z); g.w = d.w * bn_gelu_grad(ga.w * h.w + be.w);
    const float4 k1 = *reinterpret_cast<const float4*>(coef + col), k2 = *reinterpret_cast<const float4*>(coef + ch + col),
                 k3 = *reinterpret_cast<const float4*>(coef + 2 * ch + col);
    float4 o;
    o.x = k1.x * ((g.x - k2.x) - h.x * k3.x); o.y = k1.y * ((g.y - k2.y) - h.y * k3.y);
    o.z = k1.z * ((g.z - k2.z) - h.z * k3.z); o.w = k1.w * ((g.w - k2.w) - h.w * k3.w);
    *reinterpret_cast<float4*>(p) = o;
}

// eval-mode constants from the current running statistics: scale = gamma / sqrt(var + eps), shift = beta - mean * scale
__global__ __launch_bounds__(256) void bn_fold_kernel(int ch, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ch) return;
    const float s = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
}

static int bn_rows_per_block(int64_t total) {
    int64_t r = (total + 1023) / 1024;                    // at most 1024 blocks of partials (4 per CU)
    r = (r + 31) / 32 * 32;
    return (int)(r < 64 ? 64 : r);
}
static size_t bn_ws(int64_t total, int ch) {
    const size_t nb = (size_t)((total + 63) / 64) < 1025 ? (size_t)((total + 63) / 64) + 1 : 1025;
    const size_t tiles = (size_t)(ch + 127) / 128;
    return tiles * nb * 2 * 128 * sizeof(double) + (size_t)4 * ch * sizeof(float) + 64;
}
static bool bn_map_ok(const void* p, int64_t img_stride, int pitch, int ch) {
    return p && !(reinterpret_cast<uintptr_t>(p) & 15) && !(img_stride & 3) && !(pitch & 3) && pitch >= ch;
}

}  // namespace himo

using namespace himo;

extern "C" size_t himo_bn_workspace_bytes(int64_t total_rows, int ch) { return total_rows > 0 && ch > 0 ? bn_ws(total_rows, ch) : 0; }

extern "C" int himo_bn_train_fwd(int n_img, int64_t rows, int ch, const float* d_x, int64_t x_img_stride, int x_pitch,
                                 const float* d_gamma, const float* d_beta, float eps, float momentum, float* d_running_mean,
                                 float* d_running_var, float* d_mean, float* d_invstd, float* d_xhat, int64_t xhat_img_stride, int xhat_pitch,
                                 float* d_y, int64_t y_img_stride, int y_pitch, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n_img < 1 || rows < 1 || ch < 4 || (ch & 3) || !d_gamma || !d_beta || !d_mean || !d_invstd || !d_workspace)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (!bn_map_ok(d_x, x_img_stride, x_pitch, ch) || (d_xhat && !bn_map_ok(d_xhat, xhat_img_stride, xhat_pitch, ch)) ||
        !bn_map_ok(d_y, y_img_stride, y_pitch, ch))
        return HIMO_ERR_UNSUPPORTED;                        // 16-byte accesses: aligned bases, strides multiples of 4 floats
    if ((d_running_mean == nullptr) != (d_running_var == nullptr)) return HIMO_ERR_INVALID_ARGUMENT;
    const int64_t total = (int64_t)n_img * rows;
    if (total * (ch >> 2) >= ((int64_t)1 << 31)) return HIMO_ERR_UNSUPPORTED;      // 32-bit element indices in the kernels
    if (workspace_bytes < bn_ws(total, ch) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int rows_pb = bn_rows_per_block(total);
    const int nb = (int)((total + rows_pb - 1) / rows_pb), tiles = (ch + 127) / 128;
    double* partial = reinterpret_cast<double*>(d_workspace);
    const BnMap x{d_x, x_img_stride, x_pitch};
    {
        ProfScope ps("bn_stats_kernel", s);
        hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb, tiles), dim3(256), 0, s, total, rows, ch, x, partial, rows_pb);
        hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(tiles * 8), dim3(1024), 0, s, partial, nb, ch, (double)total, eps, momentum, d_running_mean,
                           d_running_var, d_mean, d_invstd);
    }
    HIMO_LAUNCH_CHECK("bn_stats kernels");
    {
        ProfScope ps("bn_normalize_gelu_kernel", s);
        const int64_t n4 = total * (ch >> 2);
        hipLaunchKernelGGL(bn_normalize_gelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, total, rows, ch, x, d_mean, d_invstd,
                           d_gamma, d_beta, BnMapW{d_xhat, xhat_img_stride, xhat_pitch}, BnMapW{d_y, y_img_stride, y_pitch});
    }
    HIMO_LAUNCH_CHECK("bn_normalize_gelu_kernel");
    return HIMO_OK;
}

static int bn_train_bwd(int n_img, int64_t rows, int ch, const float* d_dy, int64_t dy_img_stride, int dy_pitch,
                        const float* d_xhat, int64_t xhat_img_stride, int xhat_pitch, const float* d_gamma, const float* d_beta,
                        const float* d_mean, const float* d_invstd, float* d_dx, int64_t dx_img_stride, int dx_pitch, float* d_dgamma, float* d_dbeta,
                        unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);

extern "C" int himo_bn_train_bwd(int n_img, int64_t rows, int ch, const float* d_dy, int64_t dy_img_stride, int dy_pitch,
                                 const float* d_xhat, int64_t xhat_img_stride, int xhat_pitch, const float* d_gamma, const float* d_beta,
                                 const float* d_invstd, float* d_dx, int64_t dx_img_stride, int dx_pitch, float* d_dgamma, float* d_dbeta,
                                 unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    return bn_train_bwd(n_img, rows, ch, d_dy, dy_img_stride, dy_pitch, d_xhat, xhat_img_stride, xhat_pitch, d_gamma, d_beta, nullptr, d_invstd,
                        d_dx, dx_img_stride, dx_pitch, d_dgamma, d_dbeta, flags, d_workspace, workspace_bytes, stream);
}

// ... from the layer's INPUT x instead of xhat (himo_bn_train_fwd called with d_xhat = NULL): xhat = (x - mean) * invstd is formed in the
// kernels exactly as the forward pass formed it -- same results bit for bit, 4 B per activation less written by the forward pass
extern "C" int himo_bn_train_bwd_x(int n_img, int64_t rows, int ch, const float* d_dy, int64_t dy_img_stride, int dy_pitch,
                                   const float* d_x, int64_t x_img_stride, int x_pitch, const float* d_gamma, const float* d_beta,
                                   const float* d_mean, const float* d_invstd, float* d_dx, int64_t dx_img_stride, int dx_pitch,
                                   float* d_dgamma, float* d_dbeta, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!d_mean) return HIMO_ERR_INVALID_ARGUMENT;
    return bn_train_bwd(n_img, rows, ch, d_dy, dy_img_stride, dy_pitch, d_x, x_img_stride, x_pitch, d_gamma, d_beta, d_mean, d_invstd,
                        d_dx, dx_img_stride, dx_pitch, d_dgamma, d_dbeta, flags, d_workspace, workspace_bytes, stream);
}

static int bn_train_bwd(int n_img, int64_t rows, int ch, const float* d_dy, int64_t dy_img_stride, int dy_pitch,
                        const float* d_xhat, int64_t xhat_img_stride, int xhat_pitch, const float* d_gamma, const float* d_beta,
                        const float* d_mean, const float* d_invstd, float* d_dx, int64_t dx_img_stride, int dx_pitch, float* d_dgamma, float* d_dbeta,
                        unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n_img < 1 || rows < 1 || ch < 4 || (ch & 3) || !d_gamma || !d_beta || !d_invstd || !d_dgamma || !d_dbeta || !d_workspace)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (!bn_map_ok(d_dy, dy_img_stride, dy_pitch, ch) || !bn_map_ok(d_xhat, xhat_img_stride, xhat_pitch, ch) || !bn_map_ok(d_dx, dx_img_stride, dx_pitch, ch))
        return HIMO_ERR_UNSUPPORTED;
    const int64_t total = (int64_t)n_img * rows;
    if (total * (ch >> 2) >= ((int64_t)1 << 31)) return HIMO_ERR_UNSUPPORTED;      // 32-bit element indices in the kernels
    if (workspace_bytes < bn_ws(total, ch) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int rows_pb = bn_rows_per_block(total);
    const int nb = (int)((total + rows_pb - 1) / rows_pb), tiles = (ch + 127) / 128;
    double* partial = reinterpret_cast<double*>(d_workspace);
    float* coef = reinterpret_cast<float*>(partial + (size_t)tiles * nb * 2 * 128);
    const BnMap xh{d_xhat, xhat_img_stride, xhat_pitch};
    const BnMapW dx{d_dx, dx_img_stride, dx_pitch};
    {
        ProfScope ps("bn_bwd_kernel", s);
        const BnMap dym{d_dy, dy_img_stride, dy_pitch};
        if (d_mean) hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(nb, tiles), dim3(256), 0, s, total, rows, ch, dym, xh, d_gamma, d_beta, d_mean, d_invstd, partial, rows_pb);
        else hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(nb, tiles), dim3(256), 0, s, total, rows, ch, dym, xh, d_gamma, d_beta, d_mean, d_invstd, partial, rows_pb);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tiles * 8), dim3(1024), 0, s, partial, nb, ch, (double)total, d_gamma, d_invstd, d_dgamma,
                           d_dbeta, (flags & 1u) ? 1 : 0, coef);
        const int64_t n4 = total * (ch >> 2);
        if (d_mean) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, total, rows, ch, dym, xh, d_gamma, d_beta, d_mean, d_invstd, coef, dx);
        else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, total, rows, ch, dym, xh, d_gamma, d_beta, d_mean, d_invstd, coef, dx);
    }
    HIMO_LAUNCH_CHECK("bn_bwd kernels");
    return HIMO_OK;
}

extern "C" int himo_bn_fold(int ch, const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_var, float eps,
                            float* d_scale, float* d_shift, void* stream) {
    if (ch < 1 || !d_gamma || !d_beta || !d_mean || !d_var || !d_scale || !d_shift) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((ch + 255) / 256), dim3(256), 0, (hipStream_t)stream, ch, d_gamma, d_beta, d_mean, d_var, eps,
                       d_scale, d_shift);
    HIMO_LAUNCH_CHECK("bn_fold_kernel");
    return HIMO_OK;
}
