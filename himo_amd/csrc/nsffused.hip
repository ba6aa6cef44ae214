// nsffused.hip -- stage a12 (FastNSF, BASELINE config 4): one optimiser iteration of the coordinate MLP 3 -> 128 (x 8, ReLU) -> 3 over
// all points of a sweep as THREE launches -- forward + objective, backward + ALL weight gradients, reduce + Adam + re-pack -- instead
// of the 29 of round 3 (fused forward, lookup, fused backward, then eight split-K weight-gradient products that re-read every
// activation and every masked gradient from HBM: 49 % of the fit).
//
// PARITY UNPINNED (himo_amd/fastnsf.py is this build's own specification; oracle: oracle/fastnsf_oracle.py, PyTorch autograd).
//
//   nsf_forward_kernel   csrc/mlpfused.hip's forward (a block = 64 points through all layers, activations in LDS as the split-fp16
//                        A operand) with two changes: the activations H_k leave the chip ALREADY in the form their only consumer
//                        wants -- two-term bf16, in matrix-FRAGMENT order (below), plus one ReLU-mask bit per value -- and the block
//                        finishes its points: moved = x + f(x), distance-transform lookup (csrc/dtlookup.h), d loss / d out
//                        (unnormalised), the tile's loss / count sums AND the last layer's gradients (H_7 is still in LDS).
//   nsf_backward_kernel  a block = 4 tiles of 64 points, ONE wave per SIMD (4 waves = 4 column tiles, 512 registers each), layers OUTER /
//                        tiles inner, so that
//                        a layer's weight gradient accumulates over 256 points in registers and leaves the chip once per block and
//                        layer.  Per (layer k, tile): mask the running gradient with H_k > 0 -> dZ_k; its weight gradient
//                        H_{k-1}^T dZ_k needs NO transpose and NO LDS: in the accumulator layout of v_mfma_f32_32x32x16 a lane
//                        holds one COLUMN and 16 rows, i.e. registers 8j .. 8j+7 of a lane ARE an A / B fragment whose K index is
//                        the point row -- the same row permutation for H (spilled that way by the forward) and for dZ (just
//                        computed), and a contraction does not care about the order of its index.  Only the input gradient
//                        dZ_k W_k^T needs dZ_k as a row-major A operand: through LDS (double-buffered per tile: one barrier per
//                        tile), against W_k^T staged in LDS once per layer and block.
//   nsf_update_kernel    fixed-order sum of the blocks' partials x 1 / (points in the volume), Adam, and the two packed copies of
//                        every hidden W the next iteration's kernels read (csrc/convbf.hip mlp_repack_kernel's layouts).
// Algorithmic HBM bytes per point and iteration: H spill 7 x 512 B written, 7 x 512 B read; 8 x 16 B of mask bits written + read;
// 16 + 16 + 16 B of x / out / dout; per BLOCK of 256 points 459 KB of partial gradients written + read: ~11.6 kB / point against
// round 3's 28.3 kB.  Matrix work: 22 products of 128 x 128 per point, 3 instructions per float32 product block.
#include "himo_common.h"
#include "bf16x3.h"
#include "dtlookup.h"
#include "lanetranspose.h"
#include <math.h>

namespace himo {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned nsf_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kNsfRows = 64, kNsfHidden = 128, kNsfSlabs = kNsfHidden / 16, kNsfMaxHidden = 12;
constexpr int kNsfPlane = kNsfSlabs * kNsfRows * 32;            // bytes per 16-bit plane of a tile's row-major A operand
constexpr int kNsfTileBytes = kNsfRows * kNsfHidden * 4;        // one tile of one H_k in the spill: two bf16 planes
constexpr int kNsfT = 4;                                        // tiles per backward block
constexpr int kNsfWBytes = 2 * kNsfHidden * kNsfHidden * 2;     // one packed W_k^T (two planes)

// ---- the spill: H_k of tile t at spill + ((k * tiles + t) * 32 KiB), as [column tile c 4][row half h 2][k-step j 2][plane p 2]
// [lane 64][8 x bf16]: lane (li, lh) of the wave that owns columns [32 c, 32 c + 32) holds, for rows 32 h + ..., the eight values of
// accumulator registers 8 j .. 8 j + 7 (rows (r & 3) + 8 (r >> 2) + 4 lh), high (p = 0) or middle (p = 1) bf16 term.
__device__ inline int nsf_frag(int c, int h, int j, int p) { return ((((c * 2 + h) * 2 + j) * 2 + p) * 64) * 16; }
__device__ inline int nsf_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }          // within a 32-row half

// eight values -> their two-term bf16 split, packed as the matrix instruction wants them (element i in bits 16 (i & 1) of dword i / 2):
// v_cvt_pk_bf16_f32 rounds two values to nearest-even in one instruction (gfx950), the residual x - hi is exact in float32
typedef __bf16 nsf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float nsf_f32x2 __attribute__((ext_vector_type(2)));
__device__ inline void nsf_split_pack(const float (&v)[8], uint4& hi, uint4& mid) {
    unsigned h[4], m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const nsf_f32x2 p = {v[2 * q], v[2 * q + 1]};
        const nsf_bf16x2 hh = __builtin_convertvector(p, nsf_bf16x2);
        const nsf_f32x2 r = p - __builtin_convertvector(hh, nsf_f32x2);
        const nsf_bf16x2 mm = __builtin_convertvector(r, nsf_bf16x2);
        h[q] = __builtin_bit_cast(unsigned, hh); m[q] = __builtin_bit_cast(unsigned, mm);
    }
    hi = uint4{h[0], h[1], h[2], h[3]};
    mid = uint4{m[0], m[1], m[2], m[3]};
}
__device__ inline unsigned nsf_u16_at(const uint4& v, int i) {
    const unsigned w = i < 2 ? v.x : i < 4 ? v.y : i < 6 ? v.z : v.w;
    return (i & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ inline float nsf_bf16_at(const uint4& v, int i) {
    const unsigned w = i < 2 ? v.x : i < 4 ? v.y : i < 6 ? v.z : v.w;
    return bf16_bits_to_float((i & 1) ? (w >> 16) : (w & 0xffffu));
}
__device__ inline bool nsf_nonzero_at(const uint4& v, int i) {
    const unsigned w = i < 2 ? v.x : i < 4 ? v.y : i < 6 ? v.z : v.w;
    return ((i & 1) ? (w >> 16) : (w & 0xffffu)) != 0u;
}

// row-major A operand (csrc/mlpfused.hip): [plane][slab][row][16 x 16 bit], 16-byte halves swapped for rows 16..31 of a 32-row tile
__device__ inline int nsf_slot(int s, int slab, int row, int half) {
    return ((s * kNsfSlabs + slab) * kNsfRows + row) * 32 + ((half ^ ((row >> 4) & 1)) << 4);
}
template <bool BF16>
__device__ inline void nsf_a_store(unsigned char* A, int row, int k, float v) {
    unsigned h, l;
    if (BF16) { h = bf16_rne_bits(v); l = bf16_rne_bits(v - bf16_bits_to_float(h)); }
    else split2(v, h, l);
    const int off = nsf_slot(0, k >> 4, row, (k & 15) >> 3) + (k & 7) * 2;
    *reinterpret_cast<unsigned short*>(A + off) = (unsigned short)h;
    *reinterpret_cast<unsigned short*>(A + off + kNsfPlane) = (unsigned short)l;
}

__device__ inline void nsf_a_store_bits(unsigned char* A, int row, int k, unsigned h, unsigned l) {
    const int off = nsf_slot(0, k >> 4, row, (k & 15) >> 3) + (k & 7) * 2;
    *reinterpret_cast<unsigned short*>(A + off) = (unsigned short)h;
    *reinterpret_cast<unsigned short*>(A + off + kNsfPlane) = (unsigned short)l;
}

// ---- the row-major A operand from accumulator-layout registers WITHOUT 2-byte stores.  A lane holds one COLUMN; the 16 bytes of
// a row's eight consecutive columns belong to eight consecutive lanes.  64 ds_write_b16 per lane and step were the largest single
// cost of the backward kernel's vector phase (measured: 100 of 358 us; an LDS store costs its instruction, not its bytes).  Instead
// the 8 lanes x 8 rows block of 16-bit values (a uint4 per lane: element i = row i of the block) is TRANSPOSED across the lanes
// -- xor-1 at half-word granularity (DPP quad_perm + v_perm), xor-2 and xor-4 at dword granularity (DPP / ds_swizzle + selects) --
// so that lane q of the group holds row q's eight columns: one ds_write_b128.
// block (rt, j) of a tile (rows rt * 32 + nsf_row(8 j + i, lh), i = 0 .. 7), plane p, this lane's column col: vec = the lane's eight values
__device__ __forceinline__ void nsf_a_store_block(unsigned char* A, int p, const uint4& vec, int rt, int j, int lane, int col) {
    const uint4 t = lane_transpose8(vec, lane);
    const int q = lane & 7, lh = lane >> 5, c0 = col & ~7;
    const int row = rt * 32 + nsf_row(8 * j + q, lh);
    *reinterpret_cast<uint4*>(A + nsf_slot(p, c0 >> 4, row, (c0 & 15) >> 3)) = t;
}

struct NsfFwdArgs {
    int64_t n;
    int n_hidden, tiles;
    const float* x0;                                            // [n_pad][4]
    const float* w_first; const float* b_first;                 // [4][128], [128]
    const unsigned short* w_hidden[kNsfMaxHidden];              // packed W_k (fp16 split x 2^6), k = 1 .. n_hidden - 1
    const float* b_hidden[kNsfMaxHidden];
    const float* w_last; const float* b_last;                   // [128][4], [4]
    unsigned char* spill;
    unsigned* maskbits;                                         // [n_hidden][tiles][256]: bit 16 rt + r of thread (wave, lane) = H_k value > 0
    float* out;                                                 // [n_pad][4]
    float* dout;                                                // [n_pad][4] or NULL (no objective: inference of the fitted field)
    DtGrid grid; const unsigned short* vol; float trunc;
    double* loss_partial; int* count_partial;                   // [tiles]
    float* last_partial;                                        // [tiles][kNsfLastStride]: dW_last [128][4] then db_last [4] of the tile's points
};
constexpr int kNsfLastStride = 4 * kNsfHidden + 8;

// acc[rt] += A[rows of half rt][0..128) * W[:, col0 + li] (weights streamed from L2, one-slab register prefetch)
__device__ inline void nsf_fwd_gemm(const unsigned char* A, const unsigned short* __restrict__ wpk, int col0, floatx16 (&acc)[2], int li, int lh) {
    uint4 bcur[2], bnxt[2];
    auto load_b = [&](int slab, uint4 (&b)[2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            b[s] = *reinterpret_cast<const uint4*>(wpk + (((int64_t)slab * 2 + s) * kNsfHidden + col0 + li) * 16 + lh * 8);
    };
    load_b(0, bcur);
#pragma unroll 2
    for (int slab = 0; slab < kNsfSlabs; ++slab) {
        if (slab + 1 < kNsfSlabs) load_b(slab + 1, bnxt);
        f16x8 af[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[rt][s] = *reinterpret_cast<const f16x8*>(A + nsf_slot(s, slab, rt * 32 + li, lh));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rt][1], __builtin_bit_cast(f16x8, bcur[0]), acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rt][0], __builtin_bit_cast(f16x8, bcur[1]), acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[rt][0], __builtin_bit_cast(f16x8, bcur[0]), acc[rt], 0, 0, 0);
        }
        bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
    }
}

__global__ __launch_bounds__(256, 3) void nsf_forward_kernel(NsfFwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char A[2 * kNsfPlane + 64 * 4];      // (+ padding so that Y [64][129] fits)
    __shared__ float s_x[kNsfRows][4];
    __shared__ float s_o[kNsfRows][4];
    __shared__ float s_w[kNsfHidden][4];
    __shared__ double s_l[kNsfRows];
    __shared__ int s_c[kNsfRows];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kNsfRows;
    const int col = wave * 32 + li;
    if (threadIdx.x < kNsfRows) {                               // every [n][.] buffer holds whole 64-row tiles
        const float4 v = *reinterpret_cast<const float4*>(a.x0 + (r0 + threadIdx.x) * 4);
        s_x[threadIdx.x][0] = v.x; s_x[threadIdx.x][1] = v.y; s_x[threadIdx.x][2] = v.z; s_x[threadIdx.x][3] = v.w;
    }
    __syncthreads();
    float h[2][16];
    {   // first layer: K = 4, vector arithmetic in accumulator layout
        const float w0 = a.w_first[col], w1 = a.w_first[kNsfHidden + col], w2 = a.w_first[2 * kNsfHidden + col], w3 = a.w_first[3 * kNsfHidden + col];
        const float b = a.b_first[col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + nsf_row(r, lh);
                const float v = fmaf(s_x[row][3], w3, fmaf(s_x[row][2], w2, fmaf(s_x[row][1], w1, s_x[row][0] * w0))) + b;
                h[rt][r] = fmaxf(v, 0.f);
            }
    }
    const int L = a.n_hidden;
#pragma unroll 1
    for (int k = 0; k < L; ++k) {
        // h = H_k in registers: to the spill as bf16 fragments, and (split fp16) into the next product's A operand
        unsigned char* __restrict__ sp = a.spill + ((int64_t)k * a.tiles + blockIdx.x) * kNsfTileBytes + lane * 16;
        const bool last = k + 1 == L;
        unsigned bits = 0;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) bits |= h[rt][r] > 0.f ? 1u << (16 * rt + r) : 0u;
        a.maskbits[((int64_t)k * a.tiles + blockIdx.x) * 256 + threadIdx.x] = bits;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (!last) {    // (H_{L-1} is not spilled: its only consumers are the mask bits above and the last layer's gradients below)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = h[rt][8 * j + i];
                    uint4 hi, mid;
                    nsf_split_pack(v, hi, mid);
                    *reinterpret_cast<uint4*>(sp + nsf_frag(wave, rt, j, 0)) = hi;
                    *reinterpret_cast<uint4*>(sp + nsf_frag(wave, rt, j, 1)) = mid;
                }
            }
            if (!last) {    // (2-byte stores here: with three blocks per CU they hide under the other blocks; the lane transpose: 164 -> 176 us)
#pragma unroll
                for (int r = 0; r < 16; ++r) nsf_a_store<false>(A, rt * 32 + nsf_row(r, lh), col, h[rt][r]);
            }
        }
        if (last) break;
        __syncthreads();                                        // A = H_k
        floatx16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        nsf_fwd_gemm(A, a.w_hidden[k + 1], wave * 32, acc, li, lh);
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
        for (int r = 0; r < 16; ++r) Y[(rt * 32 + nsf_row(r, lh)) * 129 + col] = h[rt][r];
    __syncthreads();
    {
        const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
        const float* y = Y + row * 129;
        float s = a.b_last[c];
#pragma unroll 8
        for (int k = 0; k < kNsfHidden; ++k) s = fmaf(y[k], a.w_last[k * 4 + c], s);
        a.out[(r0 + row) * 4 + c] = s;
        s_o[row][c] = s;
    }
    if (!a.dout) return;
    __syncthreads();
    // the objective for this block's points: moved = x + f(x) -> distance-transform lookup -> d loss / d out (to be scaled by
    // 1 / points-in-volume, which only the update kernel knows), loss and count sums in a fixed order
    if (threadIdx.x < kNsfRows) {
        const int row = threadIdx.x;
        float gr[3] = {0.f, 0.f, 0.f}, D = 0.f;
        bool inside = false;
        if (r0 + row < a.n) {
            const float p[3] = {s_x[row][0] + s_o[row][0], s_x[row][1] + s_o[row][1], s_x[row][2] + s_o[row][2]};
            inside = dt_lookup(p, a.grid, a.vol, a.trunc, D, gr);
        }
        *reinterpret_cast<float4*>(a.dout + (r0 + row) * 4) = float4{gr[0], gr[1], gr[2], 0.f};
        s_o[row][0] = gr[0]; s_o[row][1] = gr[1]; s_o[row][2] = gr[2]; s_o[row][3] = 0.f;       // (out is on its way; s_o now holds dout)
        s_l[row] = (inside && D <= a.trunc) ? (double)D : 0.0;
        s_c[row] = inside ? 1 : 0;
    }
    __syncthreads();
    // the last layer's gradients for this tile: dW_last[col][c] = sum_rows H_{L-1}[row][col] dout[row][c] (H_{L-1} is Y), db_last = sum dout
    {
        const int cc = threadIdx.x & 127, half = threadIdx.x >> 7;
        float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int row = half * 32; row < half * 32 + 32; ++row) {
            const float hv = Y[row * 129 + cc];
#pragma unroll
            for (int c = 0; c < 4; ++c) t4[c] = fmaf(hv, s_o[row][c], t4[c]);
        }
        if (half == 1) { s_w[cc][0] = t4[0]; s_w[cc][1] = t4[1]; s_w[cc][2] = t4[2]; s_w[cc][3] = t4[3]; }
        __syncthreads();
        float* __restrict__ lp = a.last_partial + (int64_t)blockIdx.x * kNsfLastStride;
        if (half == 0)
            *reinterpret_cast<float4*>(lp + cc * 4) = float4{t4[0] + s_w[cc][0], t4[1] + s_w[cc][1], t4[2] + s_w[cc][2], t4[3] + s_w[cc][3]};
        if (threadIdx.x >= 252) {
            const int c = threadIdx.x - 252;
            float sgm = 0.f;
            for (int row = 0; row < kNsfRows; ++row) sgm += s_o[row][c];
            lp[4 * kNsfHidden + c] = sgm;
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s_l[threadIdx.x] += s_l[threadIdx.x + s]; s_c[threadIdx.x] += s_c[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.loss_partial[blockIdx.x] = s_l[0]; a.count_partial[blockIdx.x] = s_c[0]; }
}

// ---- backward + every weight gradient ----------------------------------------------------------------------------------------
struct NsfBwdArgs {
    int n_hidden, tiles;
    const float* x0; const float* dout;                         // [n_pad][4]
    const unsigned short* wT_hidden[kNsfMaxHidden];             // packed W_k^T (two-term bf16), k = 1 .. n_hidden - 1
    const float* w_last;                                        // [128][4]
    const unsigned char* spill;
    const unsigned* maskbits;
    const float* last_partial;                                  // [tiles][kNsfLastStride] (written by the forward kernel)
    int off_w[kNsfMaxHidden + 2], off_b[kNsfMaxHidden + 2];     // float offsets of layer i's W / b in the flat parameter vector, i = 0 .. n_hidden
    int64_t partial_stride;                                     // floats per block (>= total parameters)
    float* partial;                                             // [blocks][partial_stride]: UNSCALED gradient sums of this block's points
};

// One wave per SIMD (256 threads, the whole 512-entry register file per lane): wave w owns column tile w of all 64 rows of a tile.
// Every block has kNsfT whole tiles (the interface pads the point buffers to that; a padding tile's dout is zero).
// With a single wave per SIMD nothing hides a memory round trip, and ANY scratch access would queue behind the prefetches in
// flight (one in-order counter): the kernel must not spill.  What keeps it from spilling: one mask dword per step instead of
// four fragment registers, wave-uniform (scalar) base addresses with immediate offsets for every fragment load, and no
// special-cased layer inside the tile loop (the last layer's own gradients are the forward kernel's job).
struct NsfBwdCtx {
    unsigned char* Wl; unsigned char* Abuf;                     // LDS
    int w, lane, li, lh, col;
    int64_t tile0;
    float* P;
};

// The spill reads run HALF A STEP ahead of the matrix instructions in two register sets (an HBM round trip under this kernel's load
// was measured at ~4000 cycles, a (layer, tile) step at ~8000): a step is  mask + split (vector work)  |  weight gradient, column
// tiles 0, 1  |  barrier  |  input gradient  |  weight gradient, column tiles 2, 3  -- the sets are re-requested for tiles 2, 3 right
// after tiles 0, 1 are consumed, and for the NEXT step's tiles 0, 1 right after 2, 3; the mask dword of the next step is requested
// as soon as this step's is applied.  "Next" is (k, t + 1), or (k - 1, 0) after a layer's last tile (the fragments are H_{k-1}'s).
// For loads to stay in flight ACROSS a barrier the
// barrier must not be __syncthreads(): its release fence is s_waitcnt vmcnt(0) on gfx9 (loads and stores share the counter) --
// every barrier would drain the prefetch.  The steps only exchange LDS data, so they wait for their own LDS operations and then
// s_barrier.
__device__ __forceinline__ void nsf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// W_k^T (64 KiB, the packed global layout as it is) global -> LDS by LDS-DMA: no staging registers, and nothing waits here -- the
// consumer (the layer's first input-gradient product, most of a step later) waits with a COUNTED s_waitcnt: at least 9 younger
// vector-memory loads (a mask dword and two fragment sets) are issued between this and that wait, and loads return in order.
__device__ __forceinline__ void nsf_stage_w(const unsigned short* __restrict__ src, const NsfBwdCtx& x) {
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(x.Wl) + (unsigned)x.w * 16384u;
    const unsigned char* g = reinterpret_cast<const unsigned char*>(src) + x.w * 16384 + x.lane * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g + i * 1024), "s"(lds_base + (unsigned)i * 1024u) : "memory");
    }
}

// fragments (rt, j, p) of column tile c: `tile` is a wave-uniform pointer to the tile's 32 KiB, voff = 16 * lane
__device__ __forceinline__ void nsf_load_hf(const unsigned char* __restrict__ tile, unsigned voff, int c, uint4 (&f)[2][2][2]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const unsigned char* __restrict__ base = tile + nsf_frag(c, rt, 0, 0);             // scalar; (j, p) ride in the instruction's offset
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) f[rt][j][p] = *reinterpret_cast<const uint4*>(base + (voff + (unsigned)((j * 2 + p) * 1024)));
    }
}

// one hidden layer k >= 1 of the backward pass for the block's tiles: dZ_k = mask(g), bias / weight gradients of layer k, g <- dZ_k W_k^T
__device__ __forceinline__ void nsf_bwd_layer(const NsfBwdArgs& a, const NsfBwdCtx& x, int k, floatx16 (&g)[kNsfT][2], uint4 (&hf)[2][2][2][2],
                                              unsigned& mbits) {
    const int li = x.li, lh = x.lh, col = x.col;
    const unsigned voff = (unsigned)x.lane * 16u;
    floatx16 accW[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[c][r] = 0.f;
    float csum = 0.f;                                           // bias gradient of layer k: this lane's rows of column col
    // wave-uniform pointers: H_{k-1} / mask bits of the block's first tile
    const unsigned char* __restrict__ hprev = a.spill + ((int64_t)(k - 1) * a.tiles + x.tile0) * kNsfTileBytes;
    const unsigned* __restrict__ mrow = a.maskbits + ((int64_t)k * a.tiles + x.tile0) * 256;
    const int64_t layer_bytes = (int64_t)a.tiles * kNsfTileBytes;
#pragma unroll
    for (int t = 0; t < kNsfT; ++t) {
        // where the NEXT step's data lives: tile t + 1 of this layer, or tile 0 of layer k - 1 (its H is H_{k-2})
        const unsigned char* __restrict__ hnext = t + 1 < kNsfT ? hprev + (t + 1) * kNsfTileBytes : hprev - layer_bytes;
        const unsigned* __restrict__ mnext = t + 1 < kNsfT ? mrow + (t + 1) * 256 : mrow - (int64_t)a.tiles * 256;
        const bool next_has_hf = t + 1 < kNsfT || k >= 2;
        // dZ_k: as B fragments (K = point rows, in accumulator order) and into the row-major A operand
        uint4 bz[2][2][2];
        unsigned char* A = x.Abuf + (t & 1) * (2 * kNsfPlane);
        const unsigned bits = mbits;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float dz[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dz[r] = (bits >> (16 * rt + r)) & 1u ? g[t][rt][r] : 0.f;
                csum += dz[r];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = dz[8 * j + i];
                nsf_split_pack(v, bz[rt][j][0], bz[rt][j][1]);
                nsf_a_store_block(A, 0, bz[rt][j][0], rt, j, x.lane, col);
                nsf_a_store_block(A, 1, bz[rt][j][1], rt, j, x.lane, col);
            }
        }
        mbits = mnext[threadIdx.x];                             // (k >= 1: layer k - 1 exists)
        // weight gradient of layer k: accW[c] += H_{k-1}[64 rows, column tile c]^T dZ_k[64 rows, column tile w] -- column tiles 0, 1
        // now (their fragments were requested half a step ago), 2, 3 after the input gradient (requested here)
        auto wgrad = [&](int c) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accW[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hf[c & 1][rt][j][1]), __builtin_bit_cast(bf16x8, bz[rt][j][0]), accW[c], 0, 0, 0);
                    accW[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hf[c & 1][rt][j][0]), __builtin_bit_cast(bf16x8, bz[rt][j][1]), accW[c], 0, 0, 0);
                    accW[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hf[c & 1][rt][j][0]), __builtin_bit_cast(bf16x8, bz[rt][j][0]), accW[c], 0, 0, 0);
                }
        };
        wgrad(0);
        nsf_load_hf(hprev + t * kNsfTileBytes, voff, 2, hf[0]);
        wgrad(1);
        nsf_load_hf(hprev + t * kNsfTileBytes, voff, 3, hf[1]);
        if (t == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // this layer's W^T has landed in LDS (nsf_stage_w: >= 9 younger loads)
        nsf_lds_barrier();                                      // A = dZ_k(t): every column tile's wave has written its part
        // input gradient: g(t) = dZ_k(t) W_k^T[:, column tile w]; a weight fragment serves both row halves
        floatx16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
#pragma unroll
        for (int slab = 0; slab < kNsfSlabs; ++slab) {
            bf16x8 af[2][2], bf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf[s] = *reinterpret_cast<const bf16x8*>(x.Wl + ((((slab * 2 + s) * kNsfHidden) + col) * 16 + lh * 8) * 2);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) af[rt][s] = *reinterpret_cast<const bf16x8*>(A + nsf_slot(s, slab, rt * 32 + li, lh));
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][1], bf[0], acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[1], acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][0], bf[0], acc[rt], 0, 0, 0);
            }
        }
        g[t][0] = acc[0]; g[t][1] = acc[1];
        wgrad(2);
        if (next_has_hf) nsf_load_hf(hnext, voff, 0, hf[0]);
        wgrad(3);
        if (next_has_hf) nsf_load_hf(hnext, voff, 1, hf[1]);
    }
    // ---- end of layer k: the block's partial gradients leave; the next layer's W^T is requested (LDS-DMA, below)
    csum += __shfl_xor(csum, 32, 64);
    {   // (wave-uniform base per group of 8 rows + this lane's offset + an immediate: no per-row address registers)
        float* __restrict__ dst = x.P + a.off_w[k];
        const unsigned vo = (unsigned)(4 * lh * kNsfHidden + col);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* __restrict__ base = dst + (c * 32 + 8 * (r >> 2)) * kNsfHidden;
                base[vo + (unsigned)((r & 3) * kNsfHidden)] = accW[c][r];
            }
    }
    if (lh == 0) x.P[a.off_b[k] + col] = csum;
    nsf_lds_barrier();                                          // every wave is done with this layer's W^T
    if (k >= 2) nsf_stage_w(a.wT_hidden[k - 1], x);
}

__global__ __launch_bounds__(256, 1) void nsf_backward_kernel(NsfBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char Wl[kNsfWBytes];                 // W_k^T of the current layer (64 KiB)
    __shared__ __attribute__((aligned(16))) unsigned char Abuf[2][2 * kNsfPlane];         // dZ_k of tile t as the A operand (2 x 32 KiB)
    __shared__ float s_d[kNsfT][kNsfRows][4];
    __shared__ float s_x[kNsfT][kNsfRows][4];
    NsfBwdCtx x;
    x.Wl = Wl; x.Abuf = &Abuf[0][0];
    x.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); x.lane = threadIdx.x & 63;
    x.li = x.lane & 31; x.lh = x.lane >> 5; x.col = x.w * 32 + x.li;
    x.tile0 = (int64_t)blockIdx.x * kNsfT;
    x.P = a.partial + (int64_t)blockIdx.x * a.partial_stride;
    const int lh = x.lh, col = x.col;
    const int L = a.n_hidden;

    // prologue: the tiles' dout and x rows; W_{L-1}^T
    for (int i = threadIdx.x; i < kNsfT * kNsfRows * 2; i += 256) {
        const int which = i / (kNsfT * kNsfRows), rr = i % (kNsfT * kNsfRows);
        const float4 v = *reinterpret_cast<const float4*>((which ? a.x0 : a.dout) + (x.tile0 * kNsfRows + rr) * 4);
        float* dst = which ? &s_x[0][0][0] + rr * 4 : &s_d[0][0][0] + rr * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    nsf_stage_w(a.wT_hidden[L - 1], x);                         // (drained by the __syncthreads below)
    {   // the last layer's gradients of this block's four tiles (the forward kernel left them per tile)
        const float* __restrict__ lp = a.last_partial + x.tile0 * kNsfLastStride;
        for (int i = threadIdx.x; i < 4 * kNsfHidden + 4; i += 256) {
            float sacc = lp[i];
#pragma unroll
            for (int t = 1; t < kNsfT; ++t) sacc += lp[t * kNsfLastStride + i];
            x.P[(i < 4 * kNsfHidden ? a.off_w[L] + i : a.off_b[L] + i - 4 * kNsfHidden)] = sacc;
        }
    }
    __syncthreads();

    floatx16 g[kNsfT][2];                                       // running gradient at H_k of (tile, row half, column col)
    {   // through the last layer (K = 4): sum_c dout[row][c] * W_last[col][c]
        const float4 wl = *reinterpret_cast<const float4*>(a.w_last + col * 4);
#pragma unroll
        for (int t = 0; t < kNsfT; ++t)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* d = s_d[t][rt * 32 + nsf_row(r, lh)];
                    g[t][rt][r] = fmaf(d[3], wl.w, fmaf(d[2], wl.z, fmaf(d[1], wl.y, d[0] * wl.x)));
                }
    }

    uint4 hf[2][2][2][2];                                       // the running prefetch (above): first step = (layer L - 1, tile 0)
    unsigned mbits = a.maskbits[((int64_t)(L - 1) * a.tiles + x.tile0) * 256 + threadIdx.x];
#pragma unroll
    for (int c = 0; c < 2; ++c) nsf_load_hf(a.spill + ((int64_t)(L - 2) * a.tiles + x.tile0) * kNsfTileBytes, (unsigned)x.lane * 16u, c, hf[c]);
#pragma unroll 1
    for (int k = L - 1; k >= 1; --k) nsf_bwd_layer(a, x, k, g, hf, mbits);

    {   // layer 0: dZ_0 = mask(g); db_0 = its column sums; dW_first[c][col] = sum_rows x[row][c] dZ_0[row][col]
        float csum = 0.f, thin[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned* __restrict__ mrow = a.maskbits + x.tile0 * 256;
#pragma unroll
        for (int t = 0; t < kNsfT; ++t) {
            const unsigned bits = mbits;                        // (tile 0's arrived with layer 1's last step)
            if (t + 1 < kNsfT) mbits = mrow[(t + 1) * 256 + threadIdx.x];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dz = (bits >> (16 * rt + r)) & 1u ? g[t][rt][r] : 0.f;
                    csum += dz;
                    const float* xr = s_x[t][rt * 32 + nsf_row(r, lh)];
#pragma unroll
                    for (int c = 0; c < 4; ++c) thin[c] = fmaf(xr[c], dz, thin[c]);
                }
        }
        csum += __shfl_xor(csum, 32, 64);
#pragma unroll
        for (int c = 0; c < 4; ++c) thin[c] += __shfl_xor(thin[c], 32, 64);
        if (lh == 0) {
            x.P[a.off_b[0] + col] = csum;
#pragma unroll
            for (int c = 0; c < 4; ++c) x.P[a.off_w[0] + c * kNsfHidden + col] = thin[c];
        }
    }
}

// ---- reduce + Adam + re-pack ---------------------------------------------------------------------------------------------------
struct NsfUpdArgs {
    int total, n_partials, n_fwd_blocks, n_layers;              // n_layers = n_hidden + 1 (first, hidden ..., last)
    int64_t partial_stride;
    const float* partial; const double* loss_partial; const int* count_partial;
    float* p; float* g; float* m; float* v;
    float lr, b1, b2, eps, bc1, bc2_sqrt;
    int off_w[kNsfMaxHidden + 2];
    unsigned short* fwd[kNsfMaxHidden + 2]; unsigned short* bwd[kNsfMaxHidden + 2];      // packed copies of hidden layer i (NULL: none)
    double* loss;                                               // mean distance of the points in the volume
    int* count;                                                 // points in the volume
};

constexpr int kNsfUpdSplit = 4;                                 // threads per parameter: each sums a quarter of the partial list

// One block = 64 parameters x kNsfUpdSplit runs of the partial list.  The sum of ~470 partials is a chain of HBM latencies, not a
// byte rate: four threads per parameter with eight independent chains each keep 32 loads per parameter in flight (one thread with
// eight chains: 56 us per launch at 3.4 TB/s).  Fixed order: run q covers partials [q * per, (q + 1) * per), chain c of a run
// its every eighth, chains and runs are combined pairwise in index order -- the same bits on every launch.
__global__ __launch_bounds__(256) void nsf_update_kernel(NsfUpdArgs a) {
    __shared__ int s_c[256];
    __shared__ double s_l[256];
    __shared__ float s_t[kNsfUpdSplit][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    // the count partials first: their loads are in flight under the gradient partials'
    int c = 0;
    for (int base = 0; base < a.n_fwd_blocks; base += 2048) {
        int cv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = base + k * 256 + (int)threadIdx.x;
            cv[k] = b < a.n_fwd_blocks ? a.count_partial[b] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) c += cv[k];
    }
    float run = 0.f;
    if (i < a.total) {
        const int per = (a.n_partials + kNsfUpdSplit - 1) / kNsfUpdSplit;
        int b = q * per;
        const int b1 = b + per < a.n_partials ? b + per : a.n_partials;
        const float* __restrict__ src = a.partial + i;
        float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (; b + 7 < b1; b += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) t8[k] += src[(int64_t)(b + k) * a.partial_stride];
        }
        for (; b < b1; ++b) t8[0] += src[(int64_t)b * a.partial_stride];
        const float t4[4] = {t8[0] + t8[4], t8[1] + t8[5], t8[2] + t8[6], t8[3] + t8[7]};
        run = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
    s_t[q][lane] = run;
    s_c[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_c[threadIdx.x] += s_c[threadIdx.x + s];
        __syncthreads();
    }
    const int m_in = s_c[0];
    const float inv = m_in > 0 ? 1.0f / (float)m_in : 0.f;
    if (blockIdx.x == 0) {
        double t = 0.0;
        for (int b = threadIdx.x; b < a.n_fwd_blocks; b += 256) t += a.loss_partial[b];
        s_l[threadIdx.x] = t;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) s_l[threadIdx.x] += s_l[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) { *a.loss = m_in > 0 ? s_l[0] / (double)m_in : 0.0; *a.count = m_in; }
    }
    if (q != 0 || i >= a.total) return;
    const float gi = ((s_t[0][lane] + s_t[1][lane]) + (s_t[2][lane] + s_t[3][lane])) * inv;
    a.g[i] = gi;
    const float mi = a.b1 * a.m[i] + (1.f - a.b1) * gi;          // Adam, torch.optim.Adam's update order (csrc/fastnsf.hip adam_kernel)
    const float vi = a.b2 * a.v[i] + (1.f - a.b2) * gi * gi;
    a.m[i] = mi; a.v[i] = vi;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    const float pn = a.p[i] - (a.lr / a.bc1) * (mi / denom);
    a.p[i] = pn;
    // the packed copies of a hidden W (csrc/convbf.hip mlp_repack_kernel's layouts)
    for (int layer = 1; layer < a.n_layers - 1; ++layer) {
        const int e = i - a.off_w[layer];
        if (e < 0 || e >= kNsfHidden * kNsfHidden || !a.fwd[layer]) continue;
        const int ci = e / kNsfHidden, co = e % kNsfHidden;
        unsigned hh, ll;
        split2(pn * kF16WeightScale, hh, ll);
        const int fb = ((ci >> 4) * 2) * kNsfHidden * 16 + co * 16 + (ci & 15);
        a.fwd[layer][fb] = (unsigned short)hh; a.fwd[layer][fb + kNsfHidden * 16] = (unsigned short)ll;
        const unsigned bh = bf16_rne_bits(pn), bl = bf16_rne_bits(pn - bf16_bits_to_float(bh));
        const int bb = ((co >> 4) * 2) * kNsfHidden * 16 + ci * 16 + (co & 15);
        a.bwd[layer][bb] = (unsigned short)bh; a.bwd[layer][bb + kNsfHidden * 16] = (unsigned short)bl;
        break;
    }
}

}  // namespace himo

using namespace himo;

// tiles of 64 points, rounded up to whole backward blocks (kNsfT tiles): EVERY [n][.] buffer of this interface holds
// himo_nsf_padded_rows(n) rows; the kernels work on whole tiles / blocks without bounds checks
static inline int64_t nsf_tiles(int64_t n) { return ((n + kNsfRows - 1) / kNsfRows + kNsfT - 1) / kNsfT * kNsfT; }

extern "C" int64_t himo_nsf_padded_rows(int64_t n) { return n <= 0 ? 0 : nsf_tiles(n) * kNsfRows; }

// the workspace the three kernels share: [H_k fragments: n_hidden x tiles x 32 KiB][mask bits: n_hidden x tiles x 1 KiB]
// [the forward's last-layer gradient partials: tiles x kNsfLastStride floats]
static inline size_t nsf_mask_offset(int64_t tiles, int n_hidden) { return (size_t)n_hidden * (size_t)tiles * kNsfTileBytes; }
static inline size_t nsf_last_offset(int64_t tiles, int n_hidden) { return nsf_mask_offset(tiles, n_hidden) + (size_t)n_hidden * (size_t)tiles * 1024; }
extern "C" size_t himo_nsf_spill_bytes(int64_t n, int n_hidden) {
    if (n < 0 || n_hidden < 1) return 0;
    return nsf_last_offset(nsf_tiles(n), n_hidden) + (size_t)nsf_tiles(n) * kNsfLastStride * 4 + 256;
}

extern "C" int himo_nsf_backward_blocks(int64_t n) { return n <= 0 ? 0 : (int)(nsf_tiles(n) / kNsfT); }

static bool nsf_dt_grid(const float* h_origin, float cell, const int* h_dims, int window, DtGrid& g) {
    if (!h_origin || !h_dims || !(cell > 0.f) || window < 1) return false;
    if (h_dims[0] < 2 || h_dims[1] < 2 || h_dims[2] < 2) return false;
    g = DtGrid{h_origin[0], h_origin[1], h_origin[2], cell, h_dims[0], h_dims[1], h_dims[2], window};
    return true;
}

// Forward pass of the MLP over n points + (when d_dout is given) the distance-transform objective of the moved points.
// EVERY [n][.] buffer holds himo_nsf_padded_rows(n) rows; d_x0's padding rows must be finite (zero).  d_spill: himo_nsf_spill_bytes.
// d_dout [n][4]: d loss / d out WITHOUT the 1 / (points in the volume) factor (himo_nsf_update applies it to the summed gradients;
// rows >= n are written as zeros); d_loss_partial / d_count_partial: himo_nsf_padded_rows(n) / 64 entries each.
extern "C" int himo_nsf_forward(int64_t n, const float* d_x0, int n_hidden, const float* d_w_first, const float* d_b_first,
                                const void* const* h_w_hidden_packed, const float* const* h_b_hidden, const float* d_w_last,
                                const float* d_b_last, void* d_spill, float* d_out, const float* h_origin, float cell, const int* h_dims,
                                int window, const void* d_volume, float trunc_dist, float* d_dout, double* d_loss_partial,
                                int* d_count_partial, void* stream) {
    if (n < 0 || n_hidden < 1 || n_hidden > kNsfMaxHidden || !d_w_first || !d_b_first || !d_w_last || !d_b_last || !h_w_hidden_packed ||
        !h_b_hidden || !d_spill || !d_out)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_x0 || !aligned16(d_x0) || !aligned16(d_spill) || !aligned16(d_out)) return HIMO_ERR_INVALID_ARGUMENT;
    NsfFwdArgs a{};
    a.n = n; a.n_hidden = n_hidden; a.tiles = (int)nsf_tiles(n); a.x0 = d_x0; a.w_first = d_w_first; a.b_first = d_b_first;
    a.w_last = d_w_last; a.b_last = d_b_last; a.spill = reinterpret_cast<unsigned char*>(d_spill); a.out = d_out;
    a.maskbits = reinterpret_cast<unsigned*>(a.spill + nsf_mask_offset(a.tiles, n_hidden));
    a.last_partial = reinterpret_cast<float*>(a.spill + nsf_last_offset(a.tiles, n_hidden));
    for (int k = 1; k < n_hidden; ++k) {
        if (!h_w_hidden_packed[k] || !h_b_hidden[k] || !aligned16(h_w_hidden_packed[k])) return HIMO_ERR_INVALID_ARGUMENT;
        a.w_hidden[k] = (const unsigned short*)h_w_hidden_packed[k]; a.b_hidden[k] = h_b_hidden[k];
    }
    if (d_dout) {
        if (!d_volume || !d_loss_partial || !d_count_partial || !aligned16(d_dout) || !nsf_dt_grid(h_origin, cell, h_dims, window, a.grid))
            return HIMO_ERR_INVALID_ARGUMENT;
        a.dout = d_dout; a.vol = reinterpret_cast<const unsigned short*>(d_volume); a.trunc = trunc_dist;
        a.loss_partial = d_loss_partial; a.count_partial = d_count_partial;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("nsf_forward_kernel", s);
    hipLaunchKernelGGL(nsf_forward_kernel, dim3((unsigned)a.tiles), dim3(256), 0, s, a);
    HIMO_LAUNCH_CHECK("nsf_forward_kernel");
    return HIMO_OK;
}

// Backward pass + the gradient of EVERY parameter, per block of 256 points: d_partial [himo_nsf_backward_blocks(n)][partial_stride]
// floats, each row laid out like the flat parameter vector (h_off_w[i] / h_off_b[i] = float offsets of layer i's W [cin][cout] / b,
// i = 0 first (W [4][128]), 1 .. n_hidden - 1 hidden, n_hidden last (W [128][4])).  d_dout's padding rows must be zero.
extern "C" int himo_nsf_backward(int64_t n, const float* d_x0, const float* d_dout, int n_hidden, const void* const* h_wT_hidden_packed,
                                 const float* d_w_last, const void* d_spill, const int* h_off_w, const int* h_off_b, int64_t partial_stride,
                                 float* d_partial, void* stream) {
    if (n < 0 || n_hidden < 2 || n_hidden > kNsfMaxHidden || !h_wT_hidden_packed || !d_w_last || !d_spill || !h_off_w || !h_off_b || !d_partial)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_x0 || !d_dout || !aligned16(d_x0) || !aligned16(d_dout) || !aligned16(d_w_last) || !aligned16(d_spill)) return HIMO_ERR_INVALID_ARGUMENT;
    NsfBwdArgs a{};
    a.n_hidden = n_hidden; a.tiles = (int)nsf_tiles(n); a.x0 = d_x0; a.dout = d_dout; a.w_last = d_w_last;
    a.spill = reinterpret_cast<const unsigned char*>(d_spill); a.partial_stride = partial_stride; a.partial = d_partial;
    a.maskbits = reinterpret_cast<const unsigned*>(a.spill + nsf_mask_offset(a.tiles, n_hidden));
    a.last_partial = reinterpret_cast<const float*>(a.spill + nsf_last_offset(a.tiles, n_hidden));
    for (int k = 1; k < n_hidden; ++k) {
        if (!h_wT_hidden_packed[k] || !aligned16(h_wT_hidden_packed[k])) return HIMO_ERR_INVALID_ARGUMENT;
        a.wT_hidden[k] = (const unsigned short*)h_wT_hidden_packed[k];
    }
    for (int i = 0; i <= n_hidden; ++i) {
        if (h_off_w[i] < 0 || h_off_b[i] < 0 || h_off_w[i] + (i == 0 || i == n_hidden ? 4 * kNsfHidden : kNsfHidden * kNsfHidden) > partial_stride)
            return HIMO_ERR_INVALID_ARGUMENT;
        a.off_w[i] = h_off_w[i]; a.off_b[i] = h_off_b[i];
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("nsf_backward_kernel", s);
    hipLaunchKernelGGL(nsf_backward_kernel, dim3((unsigned)himo_nsf_backward_blocks(n)), dim3(256), 0, s, a);
    HIMO_LAUNCH_CHECK("nsf_backward_kernel");
    return HIMO_OK;
}

// gradient = (fixed-order sum of the n_partials block rows) / (points in the volume; n_fwd_blocks = himo_nsf_padded_rows(n) / 64
// per-tile counts) -> d_grad; Adam step `step` on d_param / d_m /
// d_v (csrc/fastnsf.hip adam_kernel's arithmetic); the fp16-split copy of every hidden W and the two-term bf16 copy of its transpose
// (himo_mlp_repack's layouts; h_fwd_packed[i] / h_bwd_packed[i] for layer i = 1 .. n_hidden - 1, others ignored);
// d_loss = mean distance over the points in the volume, d_count = their number.
extern "C" int himo_nsf_update(int total, int n_partials, int64_t partial_stride, const float* d_partial, int n_fwd_blocks,
                               const double* d_loss_partial, const int* d_count_partial, float* d_param, float* d_grad, float* d_m, float* d_v,
                               float lr, float beta1, float beta2, float eps, int step, int n_hidden, const int* h_off_w,
                               void* const* h_fwd_packed, void* const* h_bwd_packed, double* d_loss, int* d_count, void* stream) {
    if (total < 1 || n_partials < 0 || n_fwd_blocks < 0 || step < 1 || n_hidden < 1 || n_hidden > kNsfMaxHidden || !d_partial || !d_loss_partial ||
        !d_count_partial || !d_param || !d_grad || !d_m || !d_v || !h_off_w || !h_fwd_packed || !h_bwd_packed || !d_loss || !d_count ||
        partial_stride < total)
        return HIMO_ERR_INVALID_ARGUMENT;
    NsfUpdArgs a{};
    a.total = total; a.n_partials = n_partials; a.n_fwd_blocks = n_fwd_blocks; a.n_layers = n_hidden + 1; a.partial_stride = partial_stride;
    a.partial = d_partial; a.loss_partial = d_loss_partial; a.count_partial = d_count_partial;
    a.p = d_param; a.g = d_grad; a.m = d_m; a.v = d_v;
    a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
    a.bc1 = 1.0f - powf(beta1, (float)step); a.bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    for (int i = 0; i <= n_hidden; ++i) a.off_w[i] = h_off_w[i];
    for (int i = 1; i < n_hidden; ++i) {
        if (!h_fwd_packed[i] || !h_bwd_packed[i]) return HIMO_ERR_INVALID_ARGUMENT;
        a.fwd[i] = (unsigned short*)h_fwd_packed[i]; a.bwd[i] = (unsigned short*)h_bwd_packed[i];
    }
    a.loss = d_loss; a.count = d_count;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("nsf_update_kernel", s);
    hipLaunchKernelGGL(nsf_update_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a);
    HIMO_LAUNCH_CHECK("nsf_update_kernel");
    return HIMO_OK;
}
