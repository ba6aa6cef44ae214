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
#include "bf16x3.h"

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

// y[i][c] += b[i][c] (pitched 2-D views; gradient fan-in of a tensor with two consumers)
__global__ __launch_bounds__(256) void add2d_kernel(int64_t rows, int cols, const float* __restrict__ b, int b_pitch,
                                                    float* __restrict__ y, int y_pitch) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * cols) return;
    const int64_t i = e / cols;
    const int c = (int)(e % cols);
    y[i * y_pitch + c] += b[i * b_pitch + c];
}

// ---- convolution backward helpers ---------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));

// w [k][k][cin][cout] -> wf [k][k][cout][cin] with the taps mirrored: the data-gradient of a stride-1 "same" convolution is
// the same convolution of dY with these weights
__global__ __launch_bounds__(256) void weight_flip_kernel(const float* __restrict__ w, int ks, int cin, int cout, float* __restrict__ wf) {
    const int64_t total = (int64_t)ks * ks * cin * cout;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int co = (int)(e % cout);
    const int ci = (int)((e / cout) % cin);
    const int tap = (int)(e / ((int64_t)cout * cin));
    const int ftap = ks * ks - 1 - tap;
    wf[((int64_t)ftap * cout + co) * cin + ci] = w[e];
}

// z [2H][2W][C] with z[2y][2x] = dy[y][x], zeros elsewhere (H, W = dy size; out size 2H x 2W):
// a stride-2 convolution's data gradient is the stride-1 data gradient of the zero-stuffed dY
__global__ __launch_bounds__(256) void zero_stuff_kernel(int n_img, int H, int W, int C, const float* __restrict__ dy, int64_t dy_bs,
                                                         int dy_pitch, float* __restrict__ z, int64_t z_bs, int z_pitch) {
    const int c4 = C / 4;
    const int64_t total = (int64_t)n_img * (2 * H) * (2 * W) * c4;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int q = (int)(e % c4);
    int64_t pix = e / c4;
    const int ox = (int)(pix % (2 * W)); pix /= 2 * W;
    const int oy = (int)(pix % (2 * H));
    const int img = (int)(pix / (2 * H));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(oy & 1) && !(ox & 1))
        v = *reinterpret_cast<const float4*>(dy + img * dy_bs + ((int64_t)(oy / 2) * W + ox / 2) * dy_pitch + q * 4);
    *reinterpret_cast<float4*>(z + img * z_bs + ((int64_t)oy * (2 * W) + ox) * z_pitch + q * 4) = v;
}

// adjoint of upsample2x_kernel (bilinear x2, align_corners): dx[y][x] = sum over the output pixels whose footprint
// contains (y, x) of weight * dy -- a gather over at most 4 x 4 candidates, so no atomics
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(int H, int W, int C, const float* __restrict__ dy, int dy_pitch,
                                                             float* __restrict__ dx, int dx_pitch, float ry, float rx) {
    const int c4 = C / 4;
    const int64_t total = (int64_t)H * W * c4;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int q = (int)(e % c4);
    const int64_t pix = e / c4;
    const int x = (int)(pix % W), y = (int)(pix / W);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // output rows whose source coordinate sy = ry * oy lies in (y - 1, y + 1)
    // (a single-row / single-column image has ratio 0: every output row reads source row 0)
    const int oy_lo = ry > 0.f ? max(0, (int)floorf((float)(y - 1) / ry)) : 0;
    const int oy_hi = ry > 0.f ? min(2 * H - 1, (int)ceilf((float)(y + 1) / ry)) : 2 * H - 1;
    const int ox_lo = rx > 0.f ? max(0, (int)floorf((float)(x - 1) / rx)) : 0;
    const int ox_hi = rx > 0.f ? min(2 * W - 1, (int)ceilf((float)(x + 1) / rx)) : 2 * W - 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float sy = ry * (float)oy;
        const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        float wy = 0.f;
        if (y0 == y) wy += ly0;
        if (y1 == y) wy += ly1;
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const float sx = rx * (float)ox;
            const int x0 = (int)sx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            float wx = 0.f;
            if (x0 == x) wx += lx0;
            if (x1 == x) wx += lx1;
            if (wx == 0.f) continue;
            const float4 g = *reinterpret_cast<const float4*>(dy + ((int64_t)oy * (2 * W) + ox) * dy_pitch + q * 4);
            const float w = wy * wx;
            acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
        }
    }
    *reinterpret_cast<float4*>(dx + pix * dx_pitch + q * 4) = acc;
}

// weight gradient of a 3x3 (pad 1, stride S) convolution: dW[tap][ci][co] = sum_pixels X[pix*S + tap - 1][ci] dY[pix][co]
// as a split-K matrix product per (tap, 128x128 tile): fragments straight from global memory (see fastnsf.hip)
struct ConvWgradArgs {
    const float* x; int x_pitch; int H, W;          // input image (one image per launch)
    const float* dy; int dy_pitch; int Ho, Wo;      // output-gradient image
    int cin, cout, stride, chunk;                   // chunk = output pixels per block (even)
    float* partial;                                 // [tap][tile][chunks][128][128]
};

__global__ __launch_bounds__(256) void conv_wgrad_partial_kernel(ConvWgradArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int co_tiles = (a.cout + 127) / 128, ci_tiles = (a.cin + 127) / 128;
    const int tiles = ci_tiles * co_tiles;
    const int tap = (int)blockIdx.y / tiles, tile = (int)blockIdx.y % tiles;
    const int ci0 = (tile / co_tiles) * 128, co0 = (tile % co_tiles) * 128;
    const int ky = tap / 3 - 1, kx = tap % 3 - 1;
    const int64_t P = (int64_t)a.Ho * a.Wo;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk;
    const int64_t p1 = p0 + a.chunk < P ? p0 + a.chunk : P;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int ci[2], co[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { ci[t] = ci0 + wm * 64 + t * 32 + li; co[t] = co0 + wn * 64 + t * 32 + li; }
    if (ci0 + wm * 64 < a.cin && co0 + wn * 64 < a.cout) {
#pragma unroll 2
        for (int64_t p = p0; p < p1; p += 2) {
            const int64_t pp = p + lh;
            const int oy = (int)(pp / a.Wo), ox = (int)(pp % a.Wo);
            const int iy = oy * a.stride + ky, ix = ox * a.stride + kx;
            const bool okp = pp < p1;
            const bool okx = okp && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const int64_t xin = ((int64_t)iy * a.W + ix) * a.x_pitch;
            float af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = (okx && ci[t] < a.cin) ? a.x[xin + ci[t]] : 0.f;
                bf[t] = (okp && co[t] < a.cout) ? a.dy[pp * a.dy_pitch + co[t]] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = a.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 128 * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + wn * 64 + j * 32 + li] = acc[i][j][r];
}

// ---- weight gradient, LDS-tiled ------------------------------------------------------------------------------------
// Block tile: 64 input channels x 64 output channels x all 9 taps, accumulated over a run of pixel tiles (TR output rows
// x 32 output columns each).  Per pixel tile the block stages dY [TR*32 px][64 co] and the X halo [HR rows x HC cols][64 ci]
// in LDS ONCE and all nine taps read it (shifted), so a k-step is 9 MFMAs on 10 LDS reads -- matrix-core bound, where
// the fragment-from-global kernel above is load bound.  Wave (wm, wn) owns the 32x32 sub-tile for all taps: 144
// accumulators.  Columns are XOR-swizzled so that the two half-waves of a fragment (two neighbouring output pixels)
// land in different banks.  Stride 1: TR = 2, halo 4 x 34; stride 2: TR = 1, halo 3 x 65.
struct ConvWgradTiledArgs {
    const float* x; int64_t x_bs; int x_pitch;
    const float* dy; int64_t dy_bs; int dy_pitch;
    int n_img, H, W, Ho, Wo, cin, cout;   // H, W: input image; Ho, Wo: output-gradient image
    int tiles_per_chunk;           // pixel tiles per block
    float* partial;                // [ci_tile][co_tile][chunk][9][64][64]
    float* db_partial;             // or NULL: [co_tile][chunk][64] column sums of dY (the bias gradient), written by the ci_tile 0 blocks
};

template <int S> struct WgTile {
    static constexpr int TR = S == 1 ? 2 : 1;
    static constexpr int HR = (TR - 1) * S + 3, HC = 31 * S + 3;
    static constexpr int HaloPx = HR * HC, TilePx = TR * 32;
};

// halo pixels of the two half-waves differ by S in index: swizzle on the bit that differs
template <int S> __device__ inline int swz(int px, int col) { return px * 64 + (col ^ (((px >> (S - 1)) & 1) << 5)); }

template <int S>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tiled_kernel(ConvWgradTiledArgs a) {
    using T = WgTile<S>;
    __shared__ float Xs[T::HaloPx * 64];
    __shared__ float Ys[T::TilePx * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int co_tiles = a.cout / 64;
    const int ci0 = ((int)blockIdx.y / co_tiles) * 64, co0 = ((int)blockIdx.y % co_tiles) * 64;
    const int ci_quads = min(16, (a.cin - ci0) / 4);            // a 32-channel input (enc1.0) fills half the tile
    const int col_blocks = a.Wo / 32, row_groups = a.Ho / T::TR;
    const int tiles_per_img = col_blocks * row_groups;
    const int n_tiles = a.n_img * tiles_per_img;
    const int t0 = (int)blockIdx.x * a.tiles_per_chunk;
    const int t1 = min(t0 + a.tiles_per_chunk, n_tiles);

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int t = t0; t < t1; ++t) {
        const int img = t / tiles_per_img;
        const int rem = t - img * tiles_per_img;
        const int y0 = (rem / col_blocks) * T::TR, x0 = (rem % col_blocks) * 32;       // output coordinates
        const float* xi = a.x + img * a.x_bs + ci0;
        const float* di = a.dy + img * a.dy_bs + co0;
        __syncthreads();
        for (int e = threadIdx.x; e < T::HaloPx * 16; e += 256) {
            const int px = e >> 4, q = e & 15;
            const int hy = px / T::HC, hx = px - hy * T::HC;
            const int iy = y0 * S - 1 + hy, ix = x0 * S - 1 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < ci_quads && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v = *reinterpret_cast<const float4*>(xi + ((int64_t)iy * a.W + ix) * a.x_pitch + q * 4);
            *reinterpret_cast<float4*>(&Xs[swz<S>(px, q * 4)]) = v;
        }
        for (int e = threadIdx.x; e < T::TilePx * 16; e += 256) {
            const int px = e >> 4, q = e & 15;
            const int ry = px >> 5, rx = px & 31;
            const float4 v = *reinterpret_cast<const float4*>(di + ((int64_t)(y0 + ry) * a.Wo + x0 + rx) * a.dy_pitch + q * 4);
            *reinterpret_cast<float4*>(&Ys[swz<1>(px, q * 4)]) = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < T::TilePx / 2; ++kk) {
            const int r = kk >> 4, c = ((kk & 15) << 1) + lh;
            const float b = Ys[swz<1>(r * 32 + c, wn * 32 + li)];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float av = Xs[swz<S>((r * S + ky) * T::HC + c * S + kx, wm * 32 + li)];
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[ky * 3 + kx], 0, 0, 0);
                }
        }
    }
    float* out = a.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 * 64 * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[t * 4096 + (wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + wn * 32 + li] = acc[t][r];
}

// ---- the same weight gradient with SPLIT-bf16 operands on the 16-bit matrix instructions (himo_conv3x3_wgrad_batch flag 2) ----
// x = h + m (two bf16, 16 significant bits), products h*h + h*m + m*h into one float32 accumulator: THREE
// v_mfma_f32_32x32x16_bf16 per 16 pixels of a tap instead of EIGHT v_mfma_f32_32x32x2_f32 -- 5.3x less matrix time.  The sum
// runs over hundreds of thousands of pixels with unbiased operand roundings, and the training step's gradients are compared
// at 2e-3 of the tensor's max-abs (tests/test_train_gpu.py); the float32 kernel above stays the default of the ABI.
// The reduction dimension is the PIXEL index, so the operands are staged TRANSPOSED: Xt[plane][halo row][ci][px],
// Yt[plane][row][co][px], 8 consecutive pixels = one 16-byte fragment read.  A tap's kx shifts the pixel window by 0, 1, 2
// elements: every (row, ky) reads ONE aligned fragment + the following 32-bit word and forms the three windows in registers
// (kx = 1: four v_alignbit; kx = 2: a register rename).  Rows are padded to 40 pixels = 80 bytes: consecutive channels sit 20
// banks apart, so the 16 lanes of a ds_read_b128 phase cover the 64 banks exactly once, and so do the transposed stores.
// This kernel: stride 1 (stride 2: conv_wgrad_split2_kernel below); same tile, partial layout and reduce kernel as above.
constexpr int kWsPxp = 40;

__device__ inline void split2_bf16_pair(float a, float b, unsigned& hw, unsigned& mw) {
    typedef float wg_f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 wg_b2 __attribute__((ext_vector_type(2)));
    wg_f2 v; v[0] = a; v[1] = b;
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg_b2));          // v_cvt_pk_bf16_f32 (round to nearest even)
    wg_f2 r;
    r[0] = a - __builtin_bit_cast(float, hw << 16);
    r[1] = b - __builtin_bit_cast(float, hw & 0xffff0000u);
    mw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wg_b2));
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(ConvWgradTiledArgs a) {
    constexpr int TR = 2, HR = 4, HC = 34;
    __shared__ __attribute__((aligned(16))) unsigned short Xt[2][HR][64][kWsPxp];
    __shared__ __attribute__((aligned(16))) unsigned short Yt[2][TR][64][kWsPxp];
    // the 32-bit word that FOLLOWS an 8-pixel fragment (pixels 8 c + 8, + 9 of a channel: the first word of chunk c + 1), a second time
    // as [plane][row][c][channel]: read from Xt it is a ds_read_b32 at a 20-dword channel pitch -- 32 banks, lanes 8 apart on the same
    // one, 4-way conflicts that made these six small reads of a step cost more LDS cycles than its eight 16-byte reads (PMC: 62 % of
    // the kernel's LDS cycles were bank conflicts); here consecutive lanes read consecutive words
    __shared__ unsigned Xn[2][HR][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int co_tiles = a.cout / 64;
    const int ci0 = ((int)blockIdx.y / co_tiles) * 64, co0 = ((int)blockIdx.y % co_tiles) * 64;
    const int col_blocks = a.Wo / 32, row_groups = a.Ho / TR;
    const int tiles_per_img = col_blocks * row_groups;
    const int n_tiles = a.n_img * tiles_per_img;
    const int t0 = (int)blockIdx.x * a.tiles_per_chunk;
    const int t1 = min(t0 + a.tiles_per_chunk, n_tiles);
    // staging: this thread's channel; its wave stages halo row sg of X (five 8-pixel chunks) and chunk sg of both dY rows.  The wave
    // index goes through readfirstlane so that every coordinate, bound test and row address below is SCALAR: a load is one
    // global_load_dword (scalar row base + the lane's channel offset) and a tile's bound tests are a handful of scalar branches
    // (per-lane 64-bit address arithmetic and an exec-mask branch around each of the 56 loads cost more than the tile's matrix loop)
    const int sc = threadIdx.x & 63;
    const int sg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // the bias gradient rides along (dY passes through this block's registers anyway): this thread's share of the column sum of
    // channel co0 + sc lives in LDS between tiles -- as a register it cost 29 spills in a kernel that sits at 256
    __shared__ float bacc[256];
    const bool want_db = a.db_partial != nullptr && ci0 == 0;
    bacc[threadIdx.x] = 0.f;

    // a tile's operands travel global -> registers one tile AHEAD (issued before the previous tile's matrix loop, which hides
    // their latency), then registers -> split -> transposed LDS between two barriers
    constexpr int kXItems = 5, kYItems = TR;
    float vx[kXItems][8], vy[kYItems][8];
    auto fetch = [&](int t) {
        const int img = t / tiles_per_img;
        const int rem = t - img * tiles_per_img;
        const int y0 = (rem / col_blocks) * TR, x0 = (rem % col_blocks) * 32;
        const int iy = y0 - 1 + sg;
        const bool row_ok = iy >= 0 && iy < a.H, left_ok = x0 > 0, right_ok = x0 + 32 < a.W;
        // halo column hx of this row sits at xrow[hx * pitch]; 64 lanes = 64 consecutive channels of one pixel
        const float* xrow = a.x + img * a.x_bs + ci0 + ((int64_t)iy * a.W + (x0 - 1)) * a.x_pitch;
#pragma unroll
        for (int ch = 0; ch < kXItems; ++ch)
#pragma unroll
            for (int j = 0; j < 8; ++j) vx[ch][j] = 0.f;
        if (row_ok) {
#pragma unroll
            for (int hx = 1; hx < HC - 1; ++hx) vx[hx >> 3][hx & 7] = xrow[(int64_t)hx * a.x_pitch + sc];
            if (left_ok) vx[0][0] = xrow[sc];
            if (right_ok) vx[(HC - 1) >> 3][(HC - 1) & 7] = xrow[(int64_t)(HC - 1) * a.x_pitch + sc];
        }
#pragma unroll
        for (int r = 0; r < kYItems; ++r) {
            const float* drow = a.dy + img * a.dy_bs + co0 + ((int64_t)(y0 + r) * a.Wo + x0 + sg * 8) * a.dy_pitch;
#pragma unroll
            for (int j = 0; j < 8; ++j) vy[r][j] = drow[(int64_t)j * a.dy_pitch + sc];
        }
    };
    if (t0 < t1) fetch(t0);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();                                   // the previous tile's fragment reads are done
        if (want_db) {
            float bs = 0.f;
#pragma unroll
            for (int it = 0; it < kYItems; ++it)
                bs += ((vy[it][0] + vy[it][1]) + (vy[it][2] + vy[it][3])) + ((vy[it][4] + vy[it][5]) + (vy[it][6] + vy[it][7]));
            bacc[threadIdx.x] += bs;
        }
#pragma unroll
        for (int ch = 0; ch < kXItems; ++ch) {
            uint4 h, m;
            split2_bf16_pair(vx[ch][0], vx[ch][1], h.x, m.x); split2_bf16_pair(vx[ch][2], vx[ch][3], h.y, m.y);
            split2_bf16_pair(vx[ch][4], vx[ch][5], h.z, m.z); split2_bf16_pair(vx[ch][6], vx[ch][7], h.w, m.w);
            *reinterpret_cast<uint4*>(&Xt[0][sg][sc][ch * 8]) = h;
            *reinterpret_cast<uint4*>(&Xt[1][sg][sc][ch * 8]) = m;
            if (ch > 0) { Xn[0][sg][ch - 1][sc] = h.x; Xn[1][sg][ch - 1][sc] = m.x; }
        }
#pragma unroll
        for (int r = 0; r < kYItems; ++r) {
            uint4 h, m;
            split2_bf16_pair(vy[r][0], vy[r][1], h.x, m.x); split2_bf16_pair(vy[r][2], vy[r][3], h.y, m.y);
            split2_bf16_pair(vy[r][4], vy[r][5], h.z, m.z); split2_bf16_pair(vy[r][6], vy[r][7], h.w, m.w);
            *reinterpret_cast<uint4*>(&Yt[0][r][sc][sg * 8]) = h;
            *reinterpret_cast<uint4*>(&Yt[1][r][sc][sg * 8]) = m;
        }
        if (t + 1 < t1) fetch(t + 1);
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < TR; ++r)
#pragma unroll 1
            for (int s = 0; s < 2; ++s) {
                const int px = 16 * s + 8 * lh;
                bf16x8 bfr[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) bfr[p] = *reinterpret_cast<const bf16x8*>(&Yt[p][r][wn * 32 + li][px]);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    uint4 q[2]; unsigned q4[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        q[p] = *reinterpret_cast<const uint4*>(&Xt[p][r + ky][wm * 32 + li][px]);
                        q4[p] = Xn[p][r + ky][2 * s + lh][wm * 32 + li];
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        bf16x8 af[2];
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            uint4 w = q[p];
                            if (kx == 1) w = make_uint4(__builtin_amdgcn_alignbit(q[p].y, q[p].x, 16), __builtin_amdgcn_alignbit(q[p].z, q[p].y, 16),
                                                        __builtin_amdgcn_alignbit(q[p].w, q[p].z, 16), __builtin_amdgcn_alignbit(q4[p], q[p].w, 16));
                            if (kx == 2) w = make_uint4(q[p].y, q[p].z, q[p].w, q4[p]);
                            af[p] = __builtin_bit_cast(bf16x8, w);
                        }
                        floatx16& c = acc[ky * 3 + kx];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], c, 0, 0, 0);
                    }
                }
            }
    }
    if (want_db) {                                         // thread = sg * 64 + sc: the four groups' shares, fixed order
        __syncthreads();
        if (threadIdx.x < 64)
            a.db_partial[((int64_t)(co0 / 64) * gridDim.x + blockIdx.x) * 64 + sc] = (bacc[sc] + bacc[64 + sc]) + (bacc[128 + sc] + bacc[192 + sc]);
    }
    float* out = a.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 * 64 * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[t * 4096 + (wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + wn * 32 + li] = acc[t][r];
}

// STRIDE 2 with split-bf16 operands (himo_conv3x3_wgrad_batch flag 2).  The tap windows of 8 consecutive OUTPUT pixels are every other
// input pixel, so the halo (3 rows x 65 columns) is staged DE-INTERLEAVED: Xo[..][j] = x[2 (x0 + j) - 1] (j = 0 .. 32) and
// Xe[..][j] = x[2 (x0 + j)] (j = 0 .. 31); kx = 0 reads Xo[px ..], kx = 1 Xe[px ..], kx = 2 Xo[px + 1 ..] (one v_alignbit per word).
// One output row x 32 output columns per tile; same block tile, partial layout and reduce kernel as the stride-1 kernel.
__global__ __launch_bounds__(256, 2) void conv_wgrad_split2_kernel(ConvWgradTiledArgs a) {
    constexpr int HR = 3;
    __shared__ __attribute__((aligned(16))) unsigned short Xo[2][HR][64][kWsPxp];
    __shared__ __attribute__((aligned(16))) unsigned short Xe[2][HR][64][kWsPxp];
    __shared__ __attribute__((aligned(16))) unsigned short Yt[2][64][kWsPxp];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int co_tiles = a.cout / 64;
    const int ci0 = ((int)blockIdx.y / co_tiles) * 64, co0 = ((int)blockIdx.y % co_tiles) * 64;
    const int col_blocks = a.Wo / 32;
    const int tiles_per_img = col_blocks * a.Ho;
    const int n_tiles = a.n_img * tiles_per_img;
    const int t0 = (int)blockIdx.x * a.tiles_per_chunk;
    const int t1 = min(t0 + a.tiles_per_chunk, n_tiles);
    const int sc = threadIdx.x & 63;                             // staging: this thread's channel; its wave's share of the items
    const int sg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar, as in the stride-1 kernel
    const bool ci_ok = ci0 + sc < a.cin;                         // a 32-channel input (enc1.0) fills half the tile

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // item = (halo row, 16 input pixels) of this thread's channel: wave sg stages pixels 16 sg .. 16 sg + 15 of the three halo rows;
    // the 65th column (odd plane, index 32) of halo row sg is one more load for waves 0 .. 2.  Lanes beyond a 32-channel input load
    // channel 0 and are zeroed when the values are split.
    float vx[3][16], vlast, vy[8];
    auto fetch = [&](int t) {
        const int img = t / tiles_per_img;
        const int rem = t - img * tiles_per_img;
        const int y0 = rem / col_blocks, x0 = (rem % col_blocks) * 32;
        const int lane_off = ci_ok ? sc : 0;
        const float* xi = a.x + img * a.x_bs + ci0;
#pragma unroll
        for (int hr = 0; hr < 3; ++hr) {
            const int iy = 2 * y0 - 1 + hr;
            const int ix0 = 2 * x0 - 1 + 16 * sg;
            const float* xrow = xi + ((int64_t)iy * a.W + ix0) * a.x_pitch;
#pragma unroll
            for (int j = 0; j < 16; ++j) vx[hr][j] = 0.f;
            if (iy >= 0) {
#pragma unroll
                for (int j = 1; j < 16; ++j) vx[hr][j] = xrow[(int64_t)j * a.x_pitch + lane_off];
                if (ix0 >= 0) vx[hr][0] = xrow[lane_off];
            }
        }
        {
            const int iy = 2 * y0 - 1 + sg;
            vlast = 0.f;
            if (sg < 3 && iy >= 0) vlast = xi[((int64_t)iy * a.W + 2 * x0 + 63) * a.x_pitch + lane_off];
        }
        const float* drow = a.dy + img * a.dy_bs + co0 + ((int64_t)y0 * a.Wo + x0 + sg * 8) * a.dy_pitch;
#pragma unroll
        for (int j = 0; j < 8; ++j) vy[j] = drow[(int64_t)j * a.dy_pitch + sc];
    };
    if (t0 < t1) fetch(t0);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();                                   // the previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int hr = it, ch = sg;
            if (!ci_ok)
#pragma unroll
                for (int j = 0; j < 16; ++j) vx[it][j] = 0.f;
            uint4 h, m;                                    // j even -> odd input columns, j odd -> even input columns
            split2_bf16_pair(vx[it][0], vx[it][2], h.x, m.x); split2_bf16_pair(vx[it][4], vx[it][6], h.y, m.y);
            split2_bf16_pair(vx[it][8], vx[it][10], h.z, m.z); split2_bf16_pair(vx[it][12], vx[it][14], h.w, m.w);
            *reinterpret_cast<uint4*>(&Xo[0][hr][sc][ch * 8]) = h;
            *reinterpret_cast<uint4*>(&Xo[1][hr][sc][ch * 8]) = m;
            split2_bf16_pair(vx[it][1], vx[it][3], h.x, m.x); split2_bf16_pair(vx[it][5], vx[it][7], h.y, m.y);
            split2_bf16_pair(vx[it][9], vx[it][11], h.z, m.z); split2_bf16_pair(vx[it][13], vx[it][15], h.w, m.w);
            *reinterpret_cast<uint4*>(&Xe[0][hr][sc][ch * 8]) = h;
            *reinterpret_cast<uint4*>(&Xe[1][hr][sc][ch * 8]) = m;
        }
        if (sg < 3) {
            unsigned hw, mw;
            split2_bf16_pair(ci_ok ? vlast : 0.f, 0.f, hw, mw);
            Xo[0][sg][sc][32] = (unsigned short)hw; Xo[1][sg][sc][32] = (unsigned short)mw;
        }
        {
            uint4 h, m;
            split2_bf16_pair(vy[0], vy[1], h.x, m.x); split2_bf16_pair(vy[2], vy[3], h.y, m.y);
            split2_bf16_pair(vy[4], vy[5], h.z, m.z); split2_bf16_pair(vy[6], vy[7], h.w, m.w);
            *reinterpret_cast<uint4*>(&Yt[0][sc][sg * 8]) = h;
            *reinterpret_cast<uint4*>(&Yt[1][sc][sg * 8]) = m;
        }
        if (t + 1 < t1) fetch(t + 1);
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            const int px = 16 * s + 8 * lh;
            bf16x8 bfr[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) bfr[p] = *reinterpret_cast<const bf16x8*>(&Yt[p][wn * 32 + li][px]);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                uint4 qo[2], qe[2]; unsigned q4[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    qo[p] = *reinterpret_cast<const uint4*>(&Xo[p][ky][wm * 32 + li][px]);
                    q4[p] = *reinterpret_cast<const unsigned*>(&Xo[p][ky][wm * 32 + li][px + 8]);
                    qe[p] = *reinterpret_cast<const uint4*>(&Xe[p][ky][wm * 32 + li][px]);
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    bf16x8 af[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        uint4 w = kx == 1 ? qe[p] : qo[p];
                        if (kx == 2) w = make_uint4(__builtin_amdgcn_alignbit(qo[p].y, qo[p].x, 16), __builtin_amdgcn_alignbit(qo[p].z, qo[p].y, 16),
                                                    __builtin_amdgcn_alignbit(qo[p].w, qo[p].z, 16), __builtin_amdgcn_alignbit(q4[p], qo[p].w, 16));
                        af[p] = __builtin_bit_cast(bf16x8, w);
                    }
                    floatx16& c = acc[ky * 3 + kx];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], c, 0, 0, 0);
                }
            }
        }
    }
    float* out = a.partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 * 64 * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[t * 4096 + (wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + wn * 32 + li] = acc[t][r];
}

// the bias gradient's chunk partials -> db [cout]: block = 64 channels, four thread groups over every fourth chunk, fixed-order combine
__global__ __launch_bounds__(256) void conv_wgrad_db_reduce_kernel(const float* __restrict__ db_partial, int chunks, float* __restrict__ db,
                                                                   int accumulate) {
    __shared__ float sh[4][64];
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const float* p = db_partial + (int64_t)blockIdx.x * chunks * 64 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = grp;
    for (; b + 12 < chunks; b += 16) { s0 += p[(int64_t)b * 64]; s1 += p[(int64_t)(b + 4) * 64]; s2 += p[(int64_t)(b + 8) * 64]; s3 += p[(int64_t)(b + 12) * 64]; }
    for (; b < chunks; b += 4) s0 += p[(int64_t)b * 64];
    sh[grp][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0) {
        const float t = (sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o]);
        const int c = blockIdx.x * 64 + o;
        db[c] = accumulate ? db[c] + t : t;
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_tiled_reduce_kernel(const float* __restrict__ partial, int chunks, int cin, int cout,
                                                                      float* __restrict__ dW, int accumulate) {
    // a block owns 64 consecutive elements; its four thread groups take every fourth chunk, four independent chains each
    // (16 loads in flight per element: a sum of hundreds of partials as one chain costs a memory latency per term), combined in a
    // fixed order -- deterministic
    __shared__ float sh[4][64];
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + o;
    const bool ok = e < 9ll * cin * cout;
    float s = 0.f;
    if (ok) {
        const int co = (int)(e % cout);
        const int ci = (int)((e / cout) % cin);
        const int tap = (int)(e / ((int64_t)cout * cin));
        const int tile = (ci / 64) * (cout / 64) + co / 64;
        const float* p = partial + ((int64_t)tile * chunks) * 9 * 4096 + tap * 4096 + (ci % 64) * 64 + (co % 64);
        constexpr int64_t kStride = 9 * 4096;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = grp;
        for (; b + 12 < chunks; b += 16) {
            s0 += p[(int64_t)b * kStride]; s1 += p[(int64_t)(b + 4) * kStride];
            s2 += p[(int64_t)(b + 8) * kStride]; s3 += p[(int64_t)(b + 12) * kStride];
        }
        for (; b < chunks; b += 4) s0 += p[(int64_t)b * kStride];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[grp][o] = s;
    __syncthreads();
    if (grp == 0 && ok) {
        const float t = (sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o]);
        dW[e] = accumulate ? dW[e] + t : t;
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int chunks, int cin, int cout,
                                                                float* __restrict__ dW, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = 9ll * cin * cout;
    if (e >= total) return;
    const int co = (int)(e % cout);
    const int ci = (int)((e / cout) % cin);
    const int tap = (int)(e / ((int64_t)cout * cin));
    const int co_tiles = (cout + 127) / 128, tiles = ((cin + 127) / 128) * co_tiles;
    const int tile = (ci / 128) * co_tiles + co / 128;
    const float* p = partial + ((int64_t)(tap * tiles + tile) * chunks) * 128 * 128 + (ci % 128) * 128 + (co % 128);
    float s = 0.f;
    for (int b = 0; b < chunks; ++b) s += p[(int64_t)b * 128 * 128];
    dW[e] = accumulate ? dW[e] + s : s;
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

extern "C" int himo_weight_flip(const float* d_w, int ksize, int cin, int cout, float* d_wf, void* stream) {
    if (!d_w || !d_wf || ksize < 1 || cin < 1 || cout < 1) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(weight_flip_kernel, HIMO_GRID((int64_t)ksize * ksize * cin * cout), d_w, ksize, cin, cout, d_wf);
    HIMO_LAUNCH_CHECK("weight_flip_kernel");
    return HIMO_OK;
}

extern "C" int himo_zero_stuff2x(int n_img, int h, int w, int c, const float* d_dy, int64_t dy_batch_stride, int dy_pitch,
                                 float* d_z, int64_t z_batch_stride, int z_pitch, void* stream) {
    if (n_img < 1 || h < 1 || w < 1 || c < 4 || (c & 3) || (dy_pitch & 3) || (z_pitch & 3) || !d_dy || !d_z) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(zero_stuff_kernel, HIMO_GRID((int64_t)n_img * 4 * h * w * (c / 4)), n_img, h, w, c, d_dy, dy_batch_stride,
                       dy_pitch, d_z, z_batch_stride, z_pitch);
    HIMO_LAUNCH_CHECK("zero_stuff_kernel");
    return HIMO_OK;
}

extern "C" int himo_upsample2x_bwd(const float* d_dy, int dy_pitch, int h, int w, int c, float* d_dx, int dx_pitch, void* stream) {
    if (!d_dy || !d_dx || h < 1 || w < 1 || c < 4 || (c & 3) || (dy_pitch & 3) || (dx_pitch & 3)) return HIMO_ERR_INVALID_ARGUMENT;
    const float ry = h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f, rx = w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, HIMO_GRID((int64_t)h * w * (c / 4)), h, w, c, d_dy, dy_pitch, d_dx, dx_pitch, ry, rx);
    HIMO_LAUNCH_CHECK("upsample2x_bwd_kernel");
    return HIMO_OK;
}

static int conv_wgrad_chunk(int64_t pixels) {
    int64_t c = (pixels + 47) / 48;                 // aim for <= 48 chunks
    if (c < 256) c = 256;
    return (int)((c + 1) / 2 * 2);
}

extern "C" size_t himo_conv_wgrad_workspace_bytes(int ho, int wo, int cin, int cout) {
    const int64_t P = (int64_t)ho * wo;
    const int chunk = conv_wgrad_chunk(P);
    const size_t chunks = (size_t)((P + chunk - 1) / chunk);
    const size_t tiles = (size_t)((cin + 127) / 128) * ((cout + 127) / 128);
    return 9 * tiles * chunks * 128 * 128 * 4 + 64;
}

// one image: X [h][w] pixels (pitch x_pitch), dY [ho][wo]; 3x3 pad 1, stride 1|2; flags bit 0 = accumulate into d_dw
extern "C" int himo_conv3x3_wgrad(const float* d_x, int x_pitch, int h, int w, int cin, const float* d_dy, int dy_pitch, int cout,
                                  int stride, float* d_dw, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!d_x || !d_dy || !d_dw || !d_workspace || h < 1 || w < 1 || cin < 1 || cout < 1 || !(stride == 1 || stride == 2))
        return HIMO_ERR_INVALID_ARGUMENT;
    const int ho = stride == 2 ? (h + 1) / 2 : h, wo = stride == 2 ? (w + 1) / 2 : w;
    if (workspace_bytes < himo_conv_wgrad_workspace_bytes(ho, wo, cin, cout) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    ConvWgradArgs a{d_x, x_pitch, h, w, d_dy, dy_pitch, ho, wo, cin, cout, stride, conv_wgrad_chunk((int64_t)ho * wo),
                    reinterpret_cast<float*>(d_workspace)};
    const int64_t P = (int64_t)ho * wo;
    const int chunks = (int)((P + a.chunk - 1) / a.chunk);
    const int tiles = ((cin + 127) / 128) * ((cout + 127) / 128);
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope ps("conv_wgrad_partial_kernel", s);
        hipLaunchKernelGGL(conv_wgrad_partial_kernel, dim3(chunks, 9 * tiles), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((9ll * cin * cout + 255) / 256)), dim3(256), 0, s, a.partial, chunks, cin,
                       cout, d_dw, (flags & 1u) ? 1 : 0);
    HIMO_LAUNCH_CHECK("conv_wgrad kernels");
    return HIMO_OK;
}

extern "C" int himo_add2d(int64_t rows, int cols, const float* d_b, int b_pitch, float* d_y, int y_pitch, void* stream) {
    if (rows < 0 || cols < 1 || b_pitch < cols || y_pitch < cols) return HIMO_ERR_INVALID_ARGUMENT;
    if (rows == 0) return HIMO_OK;
    if (!d_b || !d_y) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(add2d_kernel, HIMO_GRID(rows * cols), rows, cols, d_b, b_pitch, d_y, y_pitch);
    HIMO_LAUNCH_CHECK("add2d_kernel");
    return HIMO_OK;
}

static int wgrad_tiled_chunks(int n_tiles, int tiles_xy, bool beside = false) {
    // ~2 blocks per CU over the whole launch, at least 4 pixel tiles per block (each block writes 144 KB of partials).
    // `beside` (flags bit 2): the launch shares the device with another stream's kernels (the training step's data-gradient chain) --
    // ONE block per CU: two of these blocks hold 496 of a SIMD lane's 512 registers and 141 of 160 KB of LDS, nothing else could be
    // resident beside them; with one, a block of the other stream's convolutions fits on every CU (same-box: +3.1 % step)
    int chunks = ((beside ? 1 : 2) * 256 + tiles_xy - 1) / tiles_xy;     // (256 / 384 / 768 blocks: within 1-2 % either way, scripts/ab_train.sh)
    if (chunks > n_tiles / 4) chunks = n_tiles / 4;
    if (chunks < 1) chunks = 1;
    return chunks;
}

static bool wgrad_tiled_ok(int h, int w, int cin, int cout, int stride) {
    if (stride == 1) return !(h & 1) && !(w % 32) && !(cin % 64) && !(cout % 64);
    return stride == 2 && !(h & 1) && !(w % 64) && (cin == 32 || !(cin % 64)) && !(cout % 64);
}

extern "C" size_t himo_conv_wgrad_batch_workspace_bytes(int n_img, int h, int w, int cin, int cout, int stride) {
    if (!wgrad_tiled_ok(h, w, cin, cout, stride)) return 0;
    const int ho = h / stride, wo = w / stride;
    const int n_tiles = n_img * (ho / (stride == 1 ? 2 : 1)) * (wo / 32), tiles_xy = ((cin + 63) / 64) * (cout / 64);
    const size_t chunks = wgrad_tiled_chunks(n_tiles, tiles_xy);
    return (size_t)tiles_xy * chunks * 9 * 4096 * 4 + (size_t)(cout / 64) * chunks * 64 * 4 + 64;      // weight partials + bias partials
}

// 3x3 (pad 1) weight gradient over a batch of images (frames), LDS-tiled.  stride 1: h even, w % 32 == 0, cin % 64 == 0;
// stride 2: h even, w % 64 == 0, cin == 32 or cin % 64 == 0; cout % 64 == 0, pitches % 4 == 0
// (HIMO_ERR_UNSUPPORTED otherwise: use himo_conv3x3_wgrad)
static int conv3x3_wgrad_batch(int n_img, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int cin,
                               const float* d_dy, int64_t dy_batch_stride, int dy_pitch, int cout, int stride, float* d_dw, float* d_db,
                               unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream);

extern "C" int himo_conv3x3_wgrad_batch(int n_img, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int cin,
                                        const float* d_dy, int64_t dy_batch_stride, int dy_pitch, int cout, int stride, float* d_dw,
                                        unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    return conv3x3_wgrad_batch(n_img, d_x, x_batch_stride, x_pitch, h, w, cin, d_dy, dy_batch_stride, dy_pitch, cout, stride, d_dw, nullptr,
                               flags, d_workspace, workspace_bytes, stream);
}

// ... and the bias gradient d_db [cout] = column sums of dY from the same pass over dY (flags bit 0 accumulates it too).  Stride 1 with
// split-bf16 operands (flags bit 1) only: HIMO_ERR_UNSUPPORTED otherwise (himo_colsum is the stand-alone form).
extern "C" int himo_conv3x3_wgrad_batch_bias(int n_img, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int cin,
                                             const float* d_dy, int64_t dy_batch_stride, int dy_pitch, int cout, int stride, float* d_dw,
                                             float* d_db, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!d_db) return HIMO_ERR_INVALID_ARGUMENT;
    if (stride != 1 || !(flags & 2u)) return HIMO_ERR_UNSUPPORTED;
    return conv3x3_wgrad_batch(n_img, d_x, x_batch_stride, x_pitch, h, w, cin, d_dy, dy_batch_stride, dy_pitch, cout, stride, d_dw, d_db,
                               flags, d_workspace, workspace_bytes, stream);
}

static int conv3x3_wgrad_batch(int n_img, const float* d_x, int64_t x_batch_stride, int x_pitch, int h, int w, int cin,
                               const float* d_dy, int64_t dy_batch_stride, int dy_pitch, int cout, int stride, float* d_dw, float* d_db,
                               unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!d_x || !d_dy || !d_dw || !d_workspace || n_img < 1 || h < 1 || w < 1 || cin < 1 || cout < 1) return HIMO_ERR_INVALID_ARGUMENT;
    if (!wgrad_tiled_ok(h, w, cin, cout, stride) || (x_pitch & 3) || (dy_pitch & 3) || (x_batch_stride & 3) || (dy_batch_stride & 3))
        return HIMO_ERR_UNSUPPORTED;
    if (workspace_bytes < himo_conv_wgrad_batch_workspace_bytes(n_img, h, w, cin, cout, stride) || !aligned16(d_workspace))
        return HIMO_ERR_WORKSPACE;
    const int ho = h / stride, wo = w / stride;
    const int n_tiles = n_img * (ho / (stride == 1 ? 2 : 1)) * (wo / 32), tiles_xy = ((cin + 63) / 64) * (cout / 64);
    const int chunks = wgrad_tiled_chunks(n_tiles, tiles_xy, (flags & 4u) != 0);      // (the workspace is sized for the larger count)
    ConvWgradTiledArgs a{d_x, x_batch_stride, x_pitch, d_dy, dy_batch_stride, dy_pitch, n_img, h, w, ho, wo, cin, cout,
                         (n_tiles + chunks - 1) / chunks, reinterpret_cast<float*>(d_workspace), nullptr};
    const int grid_x = (n_tiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk;
    if (d_db) a.db_partial = a.partial + (size_t)tiles_xy * chunks * 9 * 4096;        // behind the weight partials (grid_x <= chunks)
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope ps("conv_wgrad_tiled_kernel", s);
        if (stride == 1 && (flags & 2u)) hipLaunchKernelGGL(conv_wgrad_split_kernel, dim3(grid_x, tiles_xy), dim3(256), 0, s, a);
        else if (flags & 2u) hipLaunchKernelGGL(conv_wgrad_split2_kernel, dim3(grid_x, tiles_xy), dim3(256), 0, s, a);
        else if (stride == 1) hipLaunchKernelGGL(conv_wgrad_tiled_kernel<1>, dim3(grid_x, tiles_xy), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_wgrad_tiled_kernel<2>, dim3(grid_x, tiles_xy), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(conv_wgrad_tiled_reduce_kernel, dim3((unsigned)((9ll * cin * cout + 63) / 64)), dim3(256), 0, s, a.partial, grid_x,
                       cin, cout, d_dw, (flags & 1u) ? 1 : 0);
    if (d_db) hipLaunchKernelGGL(conv_wgrad_db_reduce_kernel, dim3(cout / 64), dim3(256), 0, s, a.db_partial, grid_x, d_db, (flags & 1u) ? 1 : 0);
    HIMO_LAUNCH_CHECK("conv_wgrad_tiled kernels");
    return HIMO_OK;
}
