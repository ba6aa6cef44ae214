// mlpfused.hip -- stage a12 (FastNSF, BASELINE config 4): the coordinate MLP 3 -> 128 (x 8, ReLU) -> 3 over all points of a sweep as
// TWO kernels per optimiser iteration -- the whole forward pass, and the whole chain of input gradients -- instead of one row GEMM
// per layer and direction (17 launches whose activations all went through HBM twice).
//
// PARITY UNPINNED (himo_amd/fastnsf.py is this build's own specification; oracle: oracle/fastnsf_oracle.py, PyTorch autograd).
//
// Structure (the fused head's, csrc/gruhead.hip): a block owns 64 points for ALL layers.  The layer's input [64 x 128] lives in LDS
// as the matrix instruction's A operand, already split into two 16-bit planes (forward: fp16 h + l, 22-bit products; backward:
// bf16 h + m, float32's range -- gradients sit far below fp16's subnormal floor); wave w owns output columns [32 w, 32 w + 32) of all
// 64 rows (two 32 x 32 accumulator tiles), weight fragments stream from L2 (packed by himo_mlp_repack: [slab][plane][cout][16]) with
// a one-slab register prefetch, three matrix instructions per product block (h*h, h*l, l*h), float32 accumulation.  What leaves the
// chip is only what the weight gradients need: the post-ReLU activations H_k (forward) and the masked gradients dZ_k (backward),
// 32 KB per block and layer each; nothing is read back between layers.
// Per point: 7 x 2 x 128 x 128 = 229 kFLOP per direction (x3 issued); HBM: 8 x 512 B written (forward), 8 x 512 B read + written
// (backward).  The first (3 -> 128) and last (128 -> 3) layers are vector arithmetic.
#include "himo_common.h"
#include "bf16x3.h"
#include <math.h>

namespace himo {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kMlpRows = 64, kMlpHidden = 128, kMlpSlabs = kMlpHidden / 16, kMlpMaxHidden = 12;
constexpr int kMlpPlane = kMlpSlabs * kMlpRows * 32;             // bytes per 16-bit plane of the A operand

struct MlpBiasOut { float* db[16]; };                           // per hidden layer: [128]

struct MlpFusedArgs {
    int64_t n;
    int n_hidden;                                               // hidden layers (128 wide): H_0 .. H_{n_hidden-1}
    const float* x0;                                            // [n][4] input rows (forward)
    const float* w_first; const float* b_first;                 // [4][128] (rows beyond cin zero), [128]
    const unsigned short* w_hidden[kMlpMaxHidden];              // forward: packed W_k (fp16 split); backward: packed W_k^T (bf16 x2); index k = 1 .. n_hidden-1
    const float* b_hidden[kMlpMaxHidden];
    const float* w_last; const float* b_last;                   // [128][4], [4]
    float* H[kMlpMaxHidden];                                    // [n][128] post-ReLU activations (written forward, read backward)
    float* dZ[kMlpMaxHidden];                                   // [n][128] masked gradients at the hidden layers' outputs (backward)
    float* out;                                                 // forward: [n][4] network output
    const float* dout;                                          // backward: [n][4] gradient of the output
    float* db_partial;                                          // backward, or NULL: [n_hidden][blocks][128] column sums of dZ_k over a block's rows
};

// A operand: [plane][slab][row][16 x 16 bit], the two 16-byte halves of a row swapped for rows 16..31 of each 32-row tile (every
// ds_read_b128 lane group covers the 256-byte bank row once; csrc/gruhead.hip)
__device__ inline int mlp_slot(int s, int slab, int row, int half) {
    return ((s * kMlpSlabs + slab) * kMlpRows + row) * 32 + ((half ^ ((row >> 4) & 1)) << 4);
}
template <bool BF16>
__device__ inline void mlp_a_store(unsigned char* A, int row, int k, float v) {
    unsigned h, l;
    if (BF16) { h = bf16_rne_bits(v); l = bf16_rne_bits(v - bf16_bits_to_float(h)); }
    else split2(v, h, l);
    const int off = mlp_slot(0, k >> 4, row, (k & 15) >> 3) + (k & 7) * 2;
    *reinterpret_cast<unsigned short*>(A + off) = (unsigned short)h;
    *reinterpret_cast<unsigned short*>(A + off + kMlpPlane) = (unsigned short)l;
}

// acc[rt] += A[rows of tile rt][0..128) * W[:, col0 + li]; three products per block (low x high, high x low, high x high)
template <bool BF16>
__device__ inline void mlp_gemm(const unsigned char* A, const unsigned short* __restrict__ wpk, int col0, floatx16 (&acc)[2], int li, int lh) {
    uint4 bcur[2], bnxt[2];
    auto load_b = [&](int slab, uint4 (&b)[2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            b[s] = *reinterpret_cast<const uint4*>(wpk + (((int64_t)slab * 2 + s) * kMlpHidden + col0 + li) * 16 + lh * 8);
    };
    load_b(0, bcur);
#pragma unroll 2
    for (int slab = 0; slab < kMlpSlabs; ++slab) {
        if (slab + 1 < kMlpSlabs) load_b(slab + 1, bnxt);
        bf16x8 af[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[rt][s] = *reinterpret_cast<const bf16x8*>(A + mlp_slot(s, slab, rt * 32 + li, lh));
#define HIMO_MLP_TERM(SA, SB)                                                                                                        \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                                                                \
        if (BF16) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][SA], __builtin_bit_cast(bf16x8, bcur[SB]), acc[rt], 0, 0, 0); \
        else acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[rt][SA]), __builtin_bit_cast(f16x8, bcur[SB]), acc[rt], 0, 0, 0); \
    }
        HIMO_MLP_TERM(1, 0) HIMO_MLP_TERM(0, 1) HIMO_MLP_TERM(0, 0)
#undef HIMO_MLP_TERM
        bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
    }
}

__device__ inline int mlp_row(int rt, int r, int lh) { return rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh; }

// ---- forward: x0 -> H_0 .. H_{L-1} -> out -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void mlp_forward_kernel(MlpFusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char A[2 * kMlpPlane + 64 * 4];      // (+ padding so that Y [64][129] fits)
    __shared__ float s_x[kMlpRows][4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kMlpRows;
    const int col = wave * 32 + li;
    // every [n][.] buffer of the interface holds ceil(n / 64) * 64 rows: the last block's padding rows are computed and stored like
    // the others (no per-element bounds checks, and the 32 row addresses of a lane are ONE 32-bit lane offset + constants)
    if (threadIdx.x < kMlpRows) {
        const float4 v = *reinterpret_cast<const float4*>(a.x0 + (r0 + threadIdx.x) * 4);
        s_x[threadIdx.x][0] = v.x; s_x[threadIdx.x][1] = v.y; s_x[threadIdx.x][2] = v.z; s_x[threadIdx.x][3] = v.w;
    }
    __syncthreads();
    const unsigned lane_off = (unsigned)(4 * lh) * kMlpHidden + (unsigned)col;
    float h[2][16];
    {   // first layer: K = 4, vector arithmetic in accumulator layout
        const float w0 = a.w_first[col], w1 = a.w_first[kMlpHidden + col], w2 = a.w_first[2 * kMlpHidden + col], w3 = a.w_first[3 * kMlpHidden + col];
        const float b = a.b_first[col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mlp_row(rt, r, lh);
                const float v = fmaf(s_x[row][3], w3, fmaf(s_x[row][2], w2, fmaf(s_x[row][1], w1, s_x[row][0] * w0))) + b;
                h[rt][r] = fmaxf(v, 0.f);
            }
    }
    const int L = a.n_hidden;
#pragma unroll 1
    for (int k = 0; k < L; ++k) {
        // h = H_k in registers: to HBM (the weight gradients and the backward mask read it) and, split, into the A operand
        float* __restrict__ Hb = a.H[k] + r0 * kMlpHidden;      // uniform base of this block's rows
        const bool last = k + 1 == L;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Hb[lane_off + (unsigned)((rt * 32 + (r & 3) + 8 * (r >> 2)) * kMlpHidden)] = h[rt][r];
                if (!last) mlp_a_store<false>(A, mlp_row(rt, r, lh), col, h[rt][r]);
            }
        if (last) break;
        __syncthreads();                                        // A = H_k
        floatx16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        mlp_gemm<false>(A, a.w_hidden[k + 1], wave * 32, acc, li, lh);
        const float b = a.b_hidden[k + 1][col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[rt][r] = fmaxf(fmaf(acc[rt][r], kF16AccScale, b), 0.f);
        __syncthreads();                                        // every wave has read A
    }
    // last layer (128 -> up to 4 outputs): H_{L-1} as float32 rows in LDS (pitch 129: conflict-free), one thread per (row, output)
    __syncthreads();
    float* Y = reinterpret_cast<float*>(A);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[mlp_row(rt, r, lh) * 129 + col] = h[rt][r];
    __syncthreads();
    {
        const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
        const float* y = Y + row * 129;
        float s = a.b_last[c];
#pragma unroll 8
        for (int k = 0; k < kMlpHidden; ++k) s = fmaf(y[k], a.w_last[k * 4 + c], s);
        a.out[(r0 + row) * 4 + c] = s;
    }
}

// ---- backward: dout -> dZ_{L-1} .. dZ_0 (dZ_k = gradient at H_k, masked by H_k > 0) ---------------------------------------------
__global__ __launch_bounds__(256, 3) void mlp_backward_kernel(MlpFusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char A[2 * kMlpPlane];
    __shared__ float s_d[kMlpRows][4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kMlpRows;
    const int col = wave * 32 + li;
    if (threadIdx.x < kMlpRows) {                               // (padded buffers: see the forward kernel)
        const float4 v = *reinterpret_cast<const float4*>(a.dout + (r0 + threadIdx.x) * 4);
        s_d[threadIdx.x][0] = v.x; s_d[threadIdx.x][1] = v.y; s_d[threadIdx.x][2] = v.z; s_d[threadIdx.x][3] = v.w;
    }
    __syncthreads();
    const unsigned lane_off = (unsigned)(4 * lh) * kMlpHidden + (unsigned)col;
    const int L = a.n_hidden;
    floatx16 g[2];                                              // the running gradient, in accumulator layout (it IS the accumulator)
    {   // gradient at H_{L-1} through the last layer (K = 4): sum_c dout[row][c] * W_last[col][c]
        const float4 w = *reinterpret_cast<const float4*>(a.w_last + col * 4);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mlp_row(rt, r, lh);
                g[rt][r] = fmaf(s_d[row][3], w.w, fmaf(s_d[row][2], w.z, fmaf(s_d[row][1], w.y, s_d[row][0] * w.x)));
            }
    }
#pragma unroll 1
    for (int k = L - 1; k >= 0; --k) {
        // mask with H_k > 0 (a row tile's 16 loads in flight together: no branches around them), write dZ_k, and -- unless this is
        // the first layer -- hand it to the next product as the split-bf16 A operand
        const float* __restrict__ Hb = a.H[k] + r0 * kMlpHidden;
        float* __restrict__ Zb = a.dZ[k] + r0 * kMlpHidden;
        float csum = 0.f;                                       // this lane's 32 rows of column `col`: the layer's bias gradient rides along
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = Hb[lane_off + (unsigned)((rt * 32 + (r & 3) + 8 * (r >> 2)) * kMlpHidden)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = hv[r] > 0.f ? g[rt][r] : 0.f;
                Zb[lane_off + (unsigned)((rt * 32 + (r & 3) + 8 * (r >> 2)) * kMlpHidden)] = v;
                if (k > 0) mlp_a_store<true>(A, mlp_row(rt, r, lh), col, v);
                csum += v;
            }
        }
        if (a.db_partial) {                                     // (padding rows carry zeros: d_dout's are zero and the mask keeps them so)
            csum += __shfl_xor(csum, 32, 64);
            if (lh == 0) a.db_partial[((int64_t)k * gridDim.x + blockIdx.x) * kMlpHidden + col] = csum;
        }
        if (k == 0) break;
        __syncthreads();                                        // A = dZ_k
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[rt][r] = 0.f;
        mlp_gemm<true>(A, a.w_hidden[k], wave * 32, g, li, lh);         // packed W_k^T: rows = W_k's outputs, columns = its inputs
        __syncthreads();                                        // every wave has read A
    }
}

// the bias gradients of all hidden layers from the backward kernel's block partials: block = layer, 8 groups x 128 columns, four
// independent chains per thread, fixed-order combine
__global__ __launch_bounds__(1024) void mlp_bias_reduce_kernel(const float* __restrict__ partial, int n_blocks, MlpBiasOut out) {
    __shared__ float sh[8][kMlpHidden];
    const int c = threadIdx.x & 127, grp = threadIdx.x >> 7, k = blockIdx.x;
    const float* p = partial + (int64_t)k * n_blocks * kMlpHidden + c;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    int b = grp;
    for (; b + 24 < n_blocks; b += 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] += p[(int64_t)(b + 8 * q) * kMlpHidden];
    }
    for (; b < n_blocks; b += 8) t[0] += p[(int64_t)b * kMlpHidden];
    sh[grp][c] = (t[0] + t[1]) + (t[2] + t[3]);
    __syncthreads();
    if (grp == 0) {
        float r = sh[0][c];
#pragma unroll
        for (int g = 1; g < 8; ++g) r += sh[g][c];
        out.db[k][c] = r;
    }
}

}  // namespace himo

using namespace himo;

static int mlp_fused_args(MlpFusedArgs& a, int64_t n, int n_hidden, const float* d_w_first, const float* d_b_first,
                          const void* const* h_w_hidden_packed, const float* const* h_b_hidden, const float* d_w_last,
                          const float* d_b_last, float* const* h_H) {
    if (n < 0 || n_hidden < 1 || n_hidden > kMlpMaxHidden || !d_w_first || !d_w_last || !h_w_hidden_packed || !h_H) return HIMO_ERR_INVALID_ARGUMENT;
    a = MlpFusedArgs{};
    a.n = n; a.n_hidden = n_hidden; a.w_first = d_w_first; a.b_first = d_b_first; a.w_last = d_w_last; a.b_last = d_b_last;
    for (int k = 0; k < n_hidden; ++k) {
        if (!h_H[k]) return HIMO_ERR_INVALID_ARGUMENT;
        a.H[k] = h_H[k];
        if (k > 0) {
            if (!h_w_hidden_packed[k] || (reinterpret_cast<uintptr_t>(h_w_hidden_packed[k]) & 15)) return HIMO_ERR_INVALID_ARGUMENT;
            a.w_hidden[k] = (const unsigned short*)h_w_hidden_packed[k];
            a.b_hidden[k] = h_b_hidden ? h_b_hidden[k] : nullptr;
        }
    }
    return HIMO_OK;
}

// EVERY [n][.] buffer (d_x0, h_H[k], d_out; d_dout, h_dZ[k]) must hold ceil(n / 64) * 64 rows: the kernels process whole 64-row
// blocks without bounds checks (the padding rows of d_x0 / d_dout should be finite, e.g. zero; what lands in the others' is unused).
// d_x0 [n][4]; first layer W [4][128] float32 (rows beyond the real input width zero) + bias; hidden layer k = 1 .. n_hidden - 1:
// himo_mlp_repack's forward copy of W_k [128][128] + bias (index 0 of the arrays is ignored); last layer W [128][4] + bias [4];
// h_H: n_hidden device pointers [n][128] (written); d_out [n][4].
extern "C" int himo_mlp_forward_fused(int64_t n, const float* d_x0, int n_hidden, const float* d_w_first, const float* d_b_first,
                                      const void* const* h_w_hidden_packed, const float* const* h_b_hidden, const float* d_w_last,
                                      const float* d_b_last, float* const* h_H, float* d_out, void* stream) {
    MlpFusedArgs a;
    const int st = mlp_fused_args(a, n, n_hidden, d_w_first, d_b_first, h_w_hidden_packed, h_b_hidden, d_w_last, d_b_last, h_H);
    if (st != HIMO_OK) return st;
    if (!d_b_first || !d_b_last || !h_b_hidden || !d_out || (n > 0 && !d_x0) || !aligned16(d_x0)) return HIMO_ERR_INVALID_ARGUMENT;
    for (int k = 1; k < n_hidden; ++k)
        if (!h_b_hidden[k]) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    a.x0 = d_x0; a.out = d_out;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("mlp_forward_kernel", s);
    hipLaunchKernelGGL(mlp_forward_kernel, dim3((unsigned)((n + kMlpRows - 1) / kMlpRows)), dim3(256), 0, s, a);
    HIMO_LAUNCH_CHECK("mlp_forward_kernel");
    return HIMO_OK;
}

// d_dout [n][4] = gradient of the output; hidden layer k = 1 .. n_hidden - 1: himo_mlp_repack's BACKWARD copy (W_k^T, two-term bf16);
// h_H as written by the forward pass; h_dZ: n_hidden device pointers [n][128] (written): dZ_k = gradient at H_k, masked -- the
// operand of layer k's weight gradient (X = H_{k-1}) and, through W_k^T, of dZ_{k-1}.
extern "C" size_t himo_mlp_bias_workspace_bytes(int64_t n, int n_hidden) {
    return (size_t)(n_hidden > 0 ? n_hidden : 1) * (size_t)((n + kMlpRows - 1) / kMlpRows) * kMlpHidden * 4 + 64;
}

static int mlp_backward_fused(int64_t n, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed, const float* d_w_last,
                              float* const* h_H, float* const* h_dZ, float* const* h_db, void* d_workspace, size_t workspace_bytes, void* stream);

extern "C" int himo_mlp_backward_fused(int64_t n, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed,
                                       const float* d_w_last, float* const* h_H, float* const* h_dZ, void* stream) {
    return mlp_backward_fused(n, d_dout, n_hidden, h_wT_hidden_packed, d_w_last, h_H, h_dZ, nullptr, nullptr, 0, stream);
}

// ... and the hidden layers' bias gradients h_db[k] [128] = column sums of dZ_k, collected while dZ_k passes through the kernel
// (workspace: himo_mlp_bias_workspace_bytes(n, n_hidden)); d_dout's padding rows must be ZERO.
extern "C" int himo_mlp_backward_fused_bias(int64_t n, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed,
                                            const float* d_w_last, float* const* h_H, float* const* h_dZ, float* const* h_db,
                                            void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!h_db || !d_workspace || n_hidden > 16 || workspace_bytes < himo_mlp_bias_workspace_bytes(n, n_hidden) || !aligned16(d_workspace))
        return HIMO_ERR_INVALID_ARGUMENT;
    return mlp_backward_fused(n, d_dout, n_hidden, h_wT_hidden_packed, d_w_last, h_H, h_dZ, h_db, d_workspace, workspace_bytes, stream);
}

static int mlp_backward_fused(int64_t n, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed, const float* d_w_last,
                              float* const* h_H, float* const* h_dZ, float* const* h_db, void* d_workspace, size_t workspace_bytes, void* stream) {
    MlpFusedArgs a;
    if (!h_dZ) return HIMO_ERR_INVALID_ARGUMENT;
    const int st = mlp_fused_args(a, n, n_hidden, d_w_last /* unused slot */, nullptr, h_wT_hidden_packed, nullptr, d_w_last, nullptr, h_H);
    if (st != HIMO_OK) return st;
    if ((n > 0 && !d_dout) || !aligned16(d_dout) || !aligned16(d_w_last)) return HIMO_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < n_hidden; ++k) {
        if (!h_dZ[k]) return HIMO_ERR_INVALID_ARGUMENT;
        a.dZ[k] = h_dZ[k];
    }
    if (n == 0) return HIMO_OK;
    a.dout = d_dout;
    MlpBiasOut bo{};
    if (h_db) {
        for (int k = 0; k < n_hidden; ++k) {
            if (!h_db[k]) return HIMO_ERR_INVALID_ARGUMENT;
            bo.db[k] = h_db[k];
        }
        a.db_partial = reinterpret_cast<float*>(d_workspace);
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("mlp_backward_kernel", s);
    const unsigned blocks = (unsigned)((n + kMlpRows - 1) / kMlpRows);
    hipLaunchKernelGGL(mlp_backward_kernel, dim3(blocks), dim3(256), 0, s, a);
    if (h_db) hipLaunchKernelGGL(mlp_bias_reduce_kernel, dim3(n_hidden), dim3(1024), 0, s, a.db_partial, (int)blocks, bo);
    HIMO_LAUNCH_CHECK("mlp_backward_kernel");
    return HIMO_OK;
}
