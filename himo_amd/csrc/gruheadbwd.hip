// gruheadbwd.hip -- backpropagation through the GRU iterations of the per-point head in ONE kernel (training, BASELINE config 5).
//
// PARITY UNPINNED (reference network source absent): the forward it differentiates is himo_amd/seflow/spec.py steps 5-6 as run by
// csrc/gruhead.hip (himo_gru_head_train), the unfused statement of the same arithmetic is himo_amd/seflow/train.py
// HeadTrainer.backward (three element-wise kernels around two transposed row products per iteration) and the oracle is torch
// autograd (tests/test_train_gpu.py).  Replaces, per iteration, gru_bwd1 / gru_bwd2 / gru_bwd3 (csrc/train.hip) and the two
// himo_conv2d data-gradient products: ~1.7 GB of state traffic per iteration at 120k points becomes the 0.43 GB that must move
// (z, r, q, h read once; the two gate gradients written once for the weight-gradient products).
//
// Design.  A block owns 32 points for the whole backward sweep.  The running gradient dh [32][128] stays in REGISTERS in matrix
// accumulator layout -- wave w owns hidden columns [32w, 32w + 32) -- and so do dhp and the gate gradients while they are formed;
// only the two products' A operands (d aq: K = 128; d azr: K = 256) pass through LDS, split into bf16 planes (x = h + m: three
// 16-bit products per block, HIMO_PACK_BF16X2; or three planes / six products, HIMO_PACK_BF16X3).  Both products have 192 output
// columns: the 128 hidden ones map onto the waves' own columns; the 64 x columns (d x, summed over the iterations) are two more
// column tiles that the four waves share as (tile, K half), so every wave issues the same 1.5 tiles of matrix work and the halves
// meet once at the end.  Weight fragments come straight from L2 (himo_conv_pack_weights_ex layout of W^T, cout = 192).
#include "conv_common.h"
#include "bf16x3.h"

namespace himo {

constexpr int kHbRows = 32;

struct GruHeadBwdArgs {
    int64_t n, rows;                                     // points; padded row count = iteration stride of the stacked tensors
    int iters;
    const float* dhx_last;                               // [rows][192]  d loss / d [h_T | x] (from the decoder)
    const float* hx; const float* z; const float* r; const float* q;      // himo_head_saved stacks
    const unsigned short* wq_t;                          // pack(Wq^T):  K = 128, cout = 192
    const unsigned short* wzr_t;                         // pack(Wzr^T): K = 256, cout = 192
    float* daq;                                          // [iters][rows][128]  d loss / d (q pre-activation)
    float* dazr;                                         // [iters][rows][256]  d loss / d (z | r pre-activations)
    float* dhx0;                                         // [rows][192]  d loss / d [h_0 | x]
};

template <int SLABS>
__device__ inline int hb_slot(int s, int slab, int row, int half) {
    return ((s * SLABS + slab) * kHbRows + row) * 32 + ((half ^ ((row >> 4) & 1)) << 4);
}

// one value of an A operand, column k of `row`, into the NP bf16 planes
template <int NP, int SLABS>
__device__ inline void hb_store(unsigned char* A, int row, int k, float v) {
    constexpr int kPlane = SLABS * kHbRows * 32;
    unsigned h, m, l = 0;
    if (NP == 3) split3(v, h, m, l);
    else { h = bf16_rne_bits(v); m = bf16_rne_bits(v - bf16_bits_to_float(h)); }
    const int off = hb_slot<SLABS>(0, k >> 4, row, (k & 15) >> 3) + (k & 7) * 2;
    *reinterpret_cast<unsigned short*>(A + off) = (unsigned short)h;
    *reinterpret_cast<unsigned short*>(A + off + kPlane) = (unsigned short)m;
    if (NP == 3) *reinterpret_cast<unsigned short*>(A + off + 2 * kPlane) = (unsigned short)l;
}

// acc += A[32 rows][16 slabs s0 .. s1) x W[:, col + li]   (W packed [slab][plane][192][16])
template <int NP, int SLABS>
__device__ inline void hb_gemm(const unsigned char* A, const unsigned short* __restrict__ wpk, int col, int s0, int s1, floatx16& acc,
                               int li, int lh) {
    uint4 bcur[NP], bnxt[NP];
    auto load_b = [&](int slab, uint4 (&b)[NP]) {
#pragma unroll
        for (int s = 0; s < NP; ++s)
            b[s] = *reinterpret_cast<const uint4*>(wpk + (((int64_t)slab * NP + s) * 192 + col + li) * 16 + lh * 8);
    };
    load_b(s0, bcur);
#pragma unroll 2
    for (int slab = s0; slab < s1; ++slab) {
        if (slab + 1 < s1) load_b(slab + 1, bnxt);
        bf16x8 af[NP];
#pragma unroll
        for (int s = 0; s < NP; ++s) af[s] = *reinterpret_cast<const bf16x8*>(A + hb_slot<SLABS>(s, slab, li, lh));
#define HIMO_TERM(SA, SB) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SA], __builtin_bit_cast(bf16x8, bcur[SB]), acc, 0, 0, 0);
        if constexpr (NP == 3) { HIMO_TERM(2, 0) HIMO_TERM(0, 2) HIMO_TERM(1, 1) HIMO_TERM(1, 0) HIMO_TERM(0, 1) HIMO_TERM(0, 0) }
        else { HIMO_TERM(1, 0) HIMO_TERM(0, 1) HIMO_TERM(0, 0) }
#undef HIMO_TERM
#pragma unroll
        for (int s = 0; s < NP; ++s) bcur[s] = bnxt[s];
    }
}

__device__ inline float hb_load(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ inline void hb_save(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

template <int NP>
__global__ __launch_bounds__(256, 2) void gru_head_bwd_kernel(GruHeadBwdArgs a) {
    constexpr int kA1 = NP * 8 * kHbRows * 32, kA2 = NP * 16 * kHbRows * 32;
    __shared__ __attribute__((aligned(16))) unsigned char A1[kA1];        // d aq   [32][128] in planes
    __shared__ __attribute__((aligned(16))) unsigned char A2[kA2];        // d azr  [32][256]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int cw = wave * 32, ctx = wave & 1, kh = wave >> 1;          // own hidden columns; x-column tile and K half
    const unsigned row0 = (unsigned)blockIdx.x * kHbRows + 4 * lh;     // accumulator element r sits (r & 3) + 8 (r >> 2) rows further
    // byte offsets of (row0, own column) in the [rows][128] / [192] / [256] tensors and of the x tile's column in a [192] one
    unsigned o128 = (row0 * 128u + cw + li) * 4u, o192 = (row0 * 192u + cw + li) * 4u, o256 = (row0 * 256u + cw + li) * 4u;
    unsigned ox = (row0 * 192u + 128u + 32u * ctx + li) * 4u;
    int sli = li, slh = lh;

    float dh[16], dxa[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned dr = (r & 3) + 8 * (r >> 2);
        dh[r] = hb_load(a.dhx_last, o192 + dr * 768u);
        dxa[r] = hb_load(a.dhx_last, ox + dr * 768u);
    }
    if (kh) {                                              // the second K half starts from zero: the halves are summed at the end
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[r] = 0.f;
    }
    const float* hx_x = a.hx;

#pragma unroll 1
    for (int t = a.iters - 1; t >= 0; --t) {
        asm volatile("" : "+v"(o128), "+v"(o192), "+v"(o256), "+v"(sli), "+v"(slh));      // keep per-element addresses out of the loop-invariant set
        const float* zt = a.z + t * a.rows * 128; const float* rt_ = a.r + t * a.rows * 128; const float* qt = a.q + t * a.rows * 128;
        const float* ht = hx_x + t * a.rows * 192;
        float* daq_t = a.daq + t * a.rows * 128; float* dazr_t = a.dazr + t * a.rows * 256;
        float dhp[16];
        {
            float zz[16], qq[16], hh[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned dr = (r & 3) + 8 * (r >> 2);
                zz[r] = hb_load(zt, o128 + dr * 512u); qq[r] = hb_load(qt, o128 + dr * 512u); hh[r] = hb_load(ht, o192 + dr * 768u);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned dr = (r & 3) + 8 * (r >> 2);
                const int row = (int)dr + 4 * slh;
                const bool valid = (int64_t)blockIdx.x * kHbRows + row < a.n;      // padding rows carry zeros into the weight gradients
                const float g = dh[r];
                const float dz = g * (qq[r] - hh[r]);
                const float daq = valid ? g * zz[r] * (1.0f - qq[r] * qq[r]) : 0.f;
                const float dazz = valid ? dz * zz[r] * (1.0f - zz[r]) : 0.f;
                dhp[r] = g * (1.0f - zz[r]);
                hb_store<NP, 8>(A1, row, cw + sli, daq);
                hb_store<NP, 16>(A2, row, cw + sli, dazz);
                hb_save(daq_t, o128 + dr * 512u, daq);
                hb_save(dazr_t, o256 + dr * 1024u, dazz);
            }
        }
        __syncthreads();                                        // A1 = d aq
        floatx16 accH, accX;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accH[r] = 0.f; accX[r] = 0.f; }
        hb_gemm<NP, 8>(A1, a.wq_t, cw, 0, 8, accH, li, lh);                          // d (r h) for the own columns
        hb_gemm<NP, 8>(A1, a.wq_t, 128 + 32 * ctx, 4 * kh, 4 * kh + 4, accX, li, lh);     // d x, this wave's K half
        asm volatile("" : "+v"(o128), "+v"(o192), "+v"(o256), "+v"(sli), "+v"(slh));
        {
            float rr[16], hh[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned dr = (r & 3) + 8 * (r >> 2);
                rr[r] = hb_load(rt_, o128 + dr * 512u); hh[r] = hb_load(ht, o192 + dr * 768u);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned dr = (r & 3) + 8 * (r >> 2);
                const int row = (int)dr + 4 * slh;
                const bool valid = (int64_t)blockIdx.x * kHbRows + row < a.n;
                const float drh = accH[r];
                dhp[r] += drh * rr[r];
                const float dazr_r = valid ? (drh * hh[r]) * rr[r] * (1.0f - rr[r]) : 0.f;
                hb_store<NP, 16>(A2, row, 128 + cw + sli, dazr_r);
                hb_save(dazr_t, o256 + 512u + dr * 1024u, dazr_r);
                dxa[r] += accX[r];
            }
        }
        __syncthreads();                                        // A2 = d azr
#pragma unroll
        for (int r = 0; r < 16; ++r) { accH[r] = 0.f; accX[r] = 0.f; }
        hb_gemm<NP, 16>(A2, a.wzr_t, cw, 0, 16, accH, li, lh);
        hb_gemm<NP, 16>(A2, a.wzr_t, 128 + 32 * ctx, 8 * kh, 8 * kh + 8, accX, li, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) { dh[r] = dhp[r] + accH[r]; dxa[r] += accX[r]; }
        __syncthreads();                                        // every wave has read A2 (and A1) before the next iteration rewrites them
    }

    // d [h_0 | x]: the hidden columns from the registers, the x columns as the sum of the two K halves (through LDS)
    float* X = reinterpret_cast<float*>(A2);                    // [2 tiles][32 rows][32 columns]
    if (kh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(ctx * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = dxa[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned dr = (r & 3) + 8 * (r >> 2);
        hb_save(a.dhx0, o192 + dr * 768u, dh[r]);
        if (!kh) hb_save(a.dhx0, ox + dr * 768u, dxa[r] + X[(ctx * 32 + dr + 4 * lh) * 32 + li]);
    }
}

}  // namespace himo

using namespace himo;

extern "C" int himo_gru_head_backward(int64_t n, int iters, const float* d_dhx_last, const himo_head_saved* h_saved,
                                      const void* d_wq_t_packed, const void* d_wzr_t_packed, int packed_format, float* d_daq,
                                      float* d_dazr, float* d_dhx0, void* stream) {
    if (n < 0 || iters < 1 || iters > 4 || !h_saved || !(packed_format == 0 || packed_format == 2)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    const int64_t rows = h_saved->rows;                          // the iteration stride: >= ceil(n / 64) * 64
    if (!d_dhx_last || !d_wq_t_packed || !d_wzr_t_packed || !d_daq || !d_dazr || !d_dhx0 || rows < (n + 63) / 64 * 64 || (rows & 63) || !h_saved->d_hx ||
        !h_saved->d_z || !h_saved->d_r || !h_saved->d_q)
        return HIMO_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(d_wq_t_packed) | reinterpret_cast<uintptr_t>(d_wzr_t_packed)) & 15) return HIMO_ERR_INVALID_ARGUMENT;
    if (rows * 1024 >= ((int64_t)1 << 32)) return HIMO_ERR_UNSUPPORTED;            // 32-bit byte offsets inside one iteration's tensors
    GruHeadBwdArgs a{};
    a.n = n; a.rows = rows; a.iters = iters; a.dhx_last = d_dhx_last;
    a.hx = h_saved->d_hx; a.z = h_saved->d_z; a.r = h_saved->d_r; a.q = h_saved->d_q;
    a.wq_t = (const unsigned short*)d_wq_t_packed; a.wzr_t = (const unsigned short*)d_wzr_t_packed;
    a.daq = d_daq; a.dazr = d_dazr; a.dhx0 = d_dhx0;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("gru_head_bwd_kernel", s);
    const dim3 grid((unsigned)((n + 63) / 64 * 64 / kHbRows));       // whole 64-row blocks of the n points (rows beyond: the caller's business)
    if (packed_format == 2) hipLaunchKernelGGL(gru_head_bwd_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gru_head_bwd_kernel<3>, grid, dim3(256), 0, s, a);
    HIMO_LAUNCH_CHECK("gru_head_bwd_kernel");
    return HIMO_OK;
}
