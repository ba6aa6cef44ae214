// gruhead.hip -- the whole per-point head of the scene-flow network in ONE kernel (inference, split-bf16):
// gather -> 4 GRU iterations (2 GEMMs + gates each) -> Linear(192,32) + GELU -> Linear(32,3) -> flow output.
//
// PARITY UNPINNED (reference network absent): specification himo_amd/seflow/spec.py step 5-6, oracle
// oracle/seflow_oracle.py::head.  Replaces 11 launches (head_gather, 8 GRU row-GEMMs with gate epilogues, dec1,
// head_final) whose state traffic -- ~5 KB per point per iteration through HBM/L2 -- was the binding resource.
//
// Design.  A block owns 64 points for the whole head.  The GEMM A operand [h | x] (192 columns) lives in LDS, already
// split into the three bf16 planes, for all four iterations; the hidden state itself stays in REGISTERS in MFMA
// accumulator layout: wave w owns hidden columns [32w, 32w+32) of all 64 rows, and in the z|r GEMM it owns exactly the
// z and r column tiles of those columns, in the q GEMM the q tile of those columns -- so z, r*h, q and the state update
// never leave the wave; only the new A operand (r*h, then h') is written back to LDS.  Weights are pre-split
// (himo_conv_pack_weights layout [slab][term][cout][16]); every wave needs different output columns, so B fragments
// are read straight from global memory/L2 (a fragment load is 32 columns x 32 bytes = 1 KB contiguous) with a
// one-slab register prefetch -- staging them in LDS would buy no reuse.
// LDS planes are [term][slab][row][16 bf16] with the two 16-byte halves of a row swapped for rows 16..31 of each
// 32-row tile: with ds_read_b128's lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) every group then touches
// all 16 slots of the 256-byte bank row exactly once.
#include "conv_common.h"
#include "bf16x3.h"

namespace himo {

struct GruHeadArgs {
    int64_t n;
    const int* pid; const float* offsets;
    const float* img0; const float* img1; int img_pitch;
    const float* dec; int dec_pitch;
    const float* w_off; const float* b_off;             // Linear(3,64)
    const unsigned short* wzr; const float* bzr;        // packed 192 -> 256 (z | r)
    const unsigned short* wq; const float* bq;          // packed 192 -> 128
    const unsigned short* w1; const float* b1;          // packed 192 -> 32
    const float* w2; const float* b2;                   // [32][3], [3]
    const float* xyz_t; const float* pts; int stride;
    float* flow;
    int iters;
    int img_split;                                      // img0 / img1 rows in the split activation format (convsg.hip)
    unsigned* nonfinite;                                // or NULL: set to 1 when any flow value written is NaN / inf (fp16-range guard)
    int w2_pitch;                                       // row pitch of w2: 3 (inference) or 4 (the trainer's zero-padded copy)
};

// TRAINING forward (himo_gru_head_train, SAVE = true): the same kernel also leaves every tensor the backward pass reads -- the GRU
// operands and gates of each iteration, the decoder's pre-activation -- and writes the network's residual flow [n][4] instead of the
// final flow.  All buffers have ceil(n / 64) * 64 rows, so no store needs a row guard.
constexpr int kGhMaxIters = 4;
struct GruHeadSave {                 // tensors of successive iterations are stacked: element (t, row, c) at base + (t * rows + row) * pitch + c
    float* hx;                       // [iters + 1][rows][192]  [h_t | x]: the z|r product's operand (hx[iters] feeds the decoder)
    float* rhx;                      // [iters][rows][192]      [r_t h_t | x]: the q product's operand
    float* z; float* r; float* q;    // [iters][rows][128]
    float* pre1; float* y1;          // [rows][32]  decoder pre-activation and its GELU
    float* res;                      // [rows][4]   Linear(32,3)(y1) for in-range points, zeros otherwise; column 3 = 0
    int64_t rows;                    // ceil(n / 64) * 64
};

// several samples in one launch (himo_gru_head_batch): a block finds its sample from the running block counts -- one
// sample's 120k points are only ~2.4 rounds of blocks on 256 CUs, so per-sample launches end in a half-empty round each
constexpr int kGhMaxSamples = 16;
struct GruHeadSample {
    int64_t n;
    const int* pid; const float* offsets; const float* img0; const float* img1; const float* dec;
    const float* xyz_t; const float* pts; float* flow; int stride;
};
struct GruHeadBatch {
    int n_samples;
    int block_start[kGhMaxSamples + 1];
    GruHeadSample s[kGhMaxSamples];
};

// SLABS = 12: the GEMMs run over all 192 columns of [h | x].  SLABS = 9 ("folded", himo_gru_head_batch_folded): x =
// Linear(3,64)(offset) is an AFFINE function of the point's 3-D offset o and never changes over the iterations, so its share
// of every gate pre-activation is x W_x = [o, 1] [W_off W_x ; b_off W_x] -- a K = 4 product.  The host folds the 64 x-rows of
// every weight matrix into those 4 rows (padded to one 16-row slab), the A operand's ninth slab holds (o0, o1, o2, 1, 0 ...)
// for the whole kernel, and every GEMM of the head runs 9 slabs instead of 12: a quarter less matrix work, fragment reads,
// weight stream and operand LDS, with no extra vector work.
constexpr int kGhRows = 64;
constexpr bool kGhLaunder = false;

template <int SLABS>
__device__ inline int a_slot(int s, int slab, int row, int half) {
    return ((s * SLABS + slab) * kGhRows + row) * 32 + ((half ^ ((row >> 4) & 1)) << 4);
}

// one value of the A operand, column k of `row`, into the FMT planes (FMT = 3: bf16 h, m, l; FMT = 2: fp16 h, l')
template <int FMT, int SLABS>
__device__ inline void a_store(unsigned char* A, int row, int k, float v) {
    constexpr int kGhPlane = SLABS * kGhRows * 32;       // bytes per 16-bit plane
    unsigned h, m = 0, l;
    if (FMT == 3) split3(v, h, m, l); else split2(v, h, l);
    const int slab = k >> 4, kk = k & 15;
    const int off = a_slot<SLABS>(0, slab, row, kk >> 3) + (kk & 7) * 2;
    *reinterpret_cast<unsigned short*>(A + off) = (unsigned short)h;
    if (FMT == 3) *reinterpret_cast<unsigned short*>(A + off + kGhPlane) = (unsigned short)m;
    *reinterpret_cast<unsigned short*>(A + off + (FMT - 1) * kGhPlane) = (unsigned short)l;
}

// acc[rt][t] += A[rows of tile rt][0..192) * W[:, col[t] + li] for the wave's NT column tiles; RT row tiles starting at rt0
// FMT = 2: one accumulator for the three terms of the fp16 split (bf16x3.h)
template <int RT, int NT, int FMT, int SLABS>
__device__ inline void gemm192(const unsigned char* A, const unsigned short* __restrict__ wpk, int cout, const int (&col)[NT],
                               floatx16 (&acc)[RT][NT], int rt0, int li, int lh) {
    uint4 bcur[NT][FMT], bnxt[NT][FMT];
    auto load_b = [&](int slab, uint4 (&b)[NT][FMT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < FMT; ++s)
                b[t][s] = *reinterpret_cast<const uint4*>(wpk + (((int64_t)slab * FMT + s) * cout + col[t] + li) * 16 + lh * 8);
    };
    load_b(0, bcur);
    constexpr int kUnroll = SLABS % 2 == 0 ? 2 : 3;      // a divisor of the trip count: no remainder loop, no full unroll
#pragma unroll kUnroll
    for (int slab = 0; slab < SLABS; ++slab) {
        if (slab + 1 < SLABS) load_b(slab + 1, bnxt);
        bf16x8 af[RT][FMT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int s = 0; s < FMT; ++s)
                af[rt][s] = *reinterpret_cast<const bf16x8*>(A + a_slot<SLABS>(s, slab, (rt0 + rt) * 32 + li, lh));
#define HIMO_TERM(SA, SB)                                                                                          \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) _Pragma("unroll") for (int t = 0; t < NT; ++t)                  \
        acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt][SA], __builtin_bit_cast(bf16x8, bcur[t][SB]), acc[rt][t], 0, 0, 0);
#define HIMO_TERM16(ACC, SA, SB)                                                                                   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) _Pragma("unroll") for (int t = 0; t < NT; ++t)                  \
        ACC[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[rt][SA]),                    \
                                                           __builtin_bit_cast(f16x8, bcur[t][SB]), ACC[rt][t], 0, 0, 0);
        if constexpr (FMT == 3) {
            HIMO_TERM(2, 0) HIMO_TERM(0, 2) HIMO_TERM(1, 1) HIMO_TERM(1, 0) HIMO_TERM(0, 1) HIMO_TERM(0, 0)
        } else {
            HIMO_TERM16(acc, 1, 0) HIMO_TERM16(acc, 0, 1) HIMO_TERM16(acc, 0, 0)
        }
#undef HIMO_TERM16
#undef HIMO_TERM
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < FMT; ++s) bcur[t][s] = bnxt[t][s];
    }
    if (FMT == 2) {                                     // fold the cross terms in and undo the weights' packing scale
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[rt][t][r] = acc[rt][t][r] * kF16AccScale;
    }
}

// (Round 3, measured and dropped -- profiles/r03_exp_head_gate_pipelining.txt: the z | r product as two products with the r
// gate issued slab by slab under the z product's matrix instructions and the z gate under the q product's.  The unrolled slab
// loop it needs costs 20 spilled registers at the 168 budget; 204-208 us per 120k points against 200.7.)
// a saved value: uniform tensor base (scalar registers) + a 32-bit per-lane byte offset -- written as 64-bit pointers per element the
// compiler hoists ~160 loop-invariant address registers out of the iteration loop and spills them
template <int GROUP>
__device__ inline void sv_store(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

template <int FMT, int SLABS, bool SAVE = false>
// (SAVE: 74 KB of LDS admit two blocks per CU anyway -- the three-block register cap only bought 13 spilled registers)
__global__ __launch_bounds__(256, (FMT == 2 && !SAVE) ? 3 : 2) void gru_head_kernel(GruHeadArgs a, GruHeadBatch batch, GruHeadSave sv) {
    constexpr bool FOLD = SLABS == 9;
    static_assert(!(SAVE && FOLD), "the training forward keeps x explicit");
    constexpr int kGhPlane = SLABS * kGhRows * 32;
    __shared__ __attribute__((aligned(16))) unsigned char A[FMT * kGhPlane];
    __shared__ int s_pid[kGhRows];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // wave id in an SGPR: uniform branches
    const int li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    {   // this block's sample (uniform: scalar loads from the argument block)
        int smp = 0;
        while (smp + 1 < batch.n_samples && bid >= batch.block_start[smp + 1]) ++smp;
        bid -= batch.block_start[smp];
        const GruHeadSample& g = batch.s[smp];
        a.n = g.n; a.pid = g.pid; a.offsets = g.offsets; a.img0 = g.img0; a.img1 = g.img1; a.dec = g.dec;
        a.xyz_t = g.xyz_t; a.pts = g.pts; a.flow = g.flow; a.stride = g.stride;
    }
    const int64_t r0 = (int64_t)bid * kGhRows;

    if (threadIdx.x < kGhRows) {
        const int64_t i = r0 + threadIdx.x;
        s_pid[threadIdx.x] = i < a.n ? a.pid[i] : -1;
    }
    if constexpr (FOLD) {
        // ninth slab of the A operand, columns 128 .. 143 of every row: (o0, o1, o2, 1, 0, ..., 0); thread -> (row, 4 columns)
        const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
        const int64_t i = r0 + row;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) {
            if (i < a.n) { v[0] = a.offsets[i * 3]; v[1] = a.offsets[i * 3 + 1]; v[2] = a.offsets[i * 3 + 2]; }
            v[3] = 1.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) a_store<FMT, SLABS>(A, row, 128 + q * 4 + k, v[k]);
    } else if constexpr (SAVE) {
        // x = Linear(3,64)(offset to the pillar centre), also copied into every saved operand: thread -> (column, every 4th row), so a
        // wave's stores cover one row's 64 columns = 256 contiguous bytes (in the 16-columns-per-thread arrangement below a store
        // instruction touched 64 separate 64-byte segments and the copies cost 670 us per 120k points)
        const int c = threadIdx.x & 63;
        const float w0 = a.w_off[c], w1 = a.w_off[64 + c], w2 = a.w_off[128 + c], bc = a.b_off[c];
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int row = wave + 4 * j;                   // wave-uniform: the offsets arrive by scalar loads
            const int64_t i = r0 + row;
            const int64_t ic = i < a.n ? i : a.n - 1;       // (padding rows repeat the last point: their values are never read)
            const float o0 = a.offsets[ic * 3], o1 = a.offsets[ic * 3 + 1], o2 = a.offsets[ic * 3 + 2];
            const float v = fmaf(o2, w2, fmaf(o1, w1, o0 * w0)) + bc;
            a_store<FMT, SLABS>(A, row, 128 + c, v);
            const unsigned at = ((unsigned)i * 192u + 128u + (unsigned)c) * 4u;
            for (int t = 0; t <= a.iters; ++t) sv_store<0>(sv.hx + t * sv.rows * 192, at, v);
            for (int t = 0; t < a.iters; ++t) sv_store<0>(sv.rhx + t * sv.rows * 192, at, v);
        }
    } else {
        // x = Linear(3,64)(offset to the pillar centre): thread -> (row, 16-column slab), columns 128 + 16 q ..
        const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
        const int64_t i = r0 + row;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if (i < a.n) { o0 = a.offsets[i * 3]; o1 = a.offsets[i * 3 + 1]; o2 = a.offsets[i * 3 + 2]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = q * 16 + k;
            const float v = fmaf(o2, a.w_off[128 + c], fmaf(o1, a.w_off[64 + c], o0 * a.w_off[c])) + a.b_off[c];
            a_store<FMT, SLABS>(A, row, 128 + c, v);
        }
    }
    __syncthreads();

    // h0: wave w gathers its 32 hidden columns (0: pc0 image, 1: pc1 image, 2,3: decoder map) in accumulator layout.
    // BRANCH-FREE: every lane issues all of its 32 row loads (dropped points read cell 0 and discard it) and waits ONCE.
    // Written as `if (cell >= 0) v = src[...]` the compiler branched around each load and waited vmcnt(0) per element:
    // 32 serialized memory round trips per block, ~26 us of the block's life (the 64 us "fixed cost" of the r02 breakdown).
    const float* src = wave == 0 ? a.img0 : wave == 1 ? a.img1 : a.dec + (wave - 2) * 32;
    const int src_pitch = wave < 2 ? a.img_pitch : a.dec_pitch;
    float h[2][16];
    {
        int cell[2][16];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) cell[rt][r] = s_pid[rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        if (FMT == 2 && wave < 2 && a.img_split) {
            // fp16 split image rows: the value IS its two-term pair (h at half-word li & 15, l 32 bytes further) of the 16-channel record
            unsigned short hh[2][16], ll[2][16];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned short* rec = reinterpret_cast<const unsigned short*>(src + (int64_t)max(cell[rt][r], 0) * src_pitch + (li & ~15));
                    hh[rt][r] = rec[li & 15]; ll[rt][r] = rec[16 + (li & 15)];
                }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    h[rt][r] = cell[rt][r] >= 0 ? (float)__builtin_bit_cast(_Float16, hh[rt][r]) + (float)__builtin_bit_cast(_Float16, ll[rt][r]) : 0.f;
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[rt][r] = src[(int64_t)max(cell[rt][r], 0) * src_pitch + li];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[rt][r] = cell[rt][r] >= 0 ? h[rt][r] : 0.f;
            if (FMT == 2 && wave < 2) {
                // float32 image rows in the fp16-split network: rounded to the two-term value here, so both layouts give the same bits
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        unsigned hh, ll;
                        split2(h[rt][r], hh, ll);
                        h[rt][r] = (float)__builtin_bit_cast(_Float16, (unsigned short)hh) + (float)__builtin_bit_cast(_Float16, (unsigned short)ll);
                    }
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                a_store<FMT, SLABS>(A, row, wave * 32 + li, h[rt][r]);
                if constexpr (SAVE) sv_store<1>(sv.hx, (((unsigned)r0 + row) * 192u + wave * 32 + li) * 4u, h[rt][r]);
            }
    }
    __syncthreads();
    // byte offsets of (row r0 + 4 lh, column 32 wave + li) in a [rows][128] / [rows][192] saved tensor; accumulator element r of row
    // tile rt sits kGhRowOf(rt, r) rows further
    unsigned o128 = (((unsigned)r0 + 4 * lh) * 128u + wave * 32 + li) * 4u, o192 = (((unsigned)r0 + 4 * lh) * 192u + wave * 32 + li) * 4u;

    const int col_zr[2] = {wave * 32, 128 + wave * 32};
    const int col_q[1] = {wave * 32};
    const float bz = a.bzr[wave * 32 + li], br = a.bzr[128 + wave * 32 + li], bq = a.bq[wave * 32 + li];

#pragma unroll 1
    for (int it = 0; it < a.iters; ++it) {
        int sli = li, slh = lh;               // lane coordinates of the gate stages' LDS / global stores
        if constexpr (SAVE || kGhLaunder) asm volatile("" : "+v"(o128), "+v"(o192), "+v"(sli), "+v"(slh));      // keep the ~100 per-element addresses out of the loop-invariant set
        float* const sz = SAVE ? sv.z + it * sv.rows * 128 : nullptr; float* const sr = SAVE ? sv.r + it * sv.rows * 128 : nullptr;
        float* const sq = SAVE ? sv.q + it * sv.rows * 128 : nullptr; float* const srhx = SAVE ? sv.rhx + it * sv.rows * 192 : nullptr;
        float* const shx = SAVE ? sv.hx + (it + 1) * sv.rows * 192 : nullptr;
        floatx16 acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;
        gemm192<2, 2, FMT, SLABS>(A, a.wzr, 256, col_zr, acc, 0, li, lh);
        __syncthreads();                                        // every wave has read [h | x]      the bound on what a second operand buffer could buy
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[rt][0][r] = sigmoid_f(acc[rt][0][r] + bz);                      // z replaces its accumulator element
                const float rr = sigmoid_f(acc[rt][1][r] + br);
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * slh;
                a_store<FMT, SLABS>(A, row, wave * 32 + sli, rr * h[rt][r]);
                if constexpr (SAVE) {
                    const unsigned dr = rt * 32 + (r & 3) + 8 * (r >> 2);
                    sv_store<2>(sz, o128 + dr * 512u, acc[rt][0][r]); sv_store<2>(sr, o128 + dr * 512u, rr);
                    sv_store<3>(srhx, o192 + dr * 768u, rr * h[rt][r]);
                }
            }
        __syncthreads();                                        // A = [r*h | x]
        // (again: the two gate stages address the same elements, and common addresses would stay live across the q product)
        if constexpr (SAVE || kGhLaunder) asm volatile("" : "+v"(o128), "+v"(o192), "+v"(sli), "+v"(slh));
        floatx16 acq[2][1];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acq[rt][0][r] = 0.f;
        gemm192<2, 1, FMT, SLABS>(A, a.wq, 128, col_q, acq, 0, li, lh);
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float q = tanh_f(acq[rt][0][r] + bq);
                const float zz = acc[rt][0][r];
                const float hn = (1.0f - zz) * h[rt][r] + zz * q;
                h[rt][r] = hn;
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * slh;
                a_store<FMT, SLABS>(A, row, wave * 32 + sli, hn);
                if constexpr (SAVE) {
                    const unsigned dr = rt * 32 + (r & 3) + 8 * (r >> 2);
                    sv_store<4>(sq, o128 + dr * 512u, q);
                    sv_store<5>(shx, o192 + dr * 768u, hn);
                }
            }
        __syncthreads();                                        // A = [h' | x]
    }

    // y1 = gelu([h | x] W1 + b1): waves 0 and 1 take one 32-row tile each
    floatx16 ac1[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) ac1[0][0][r] = 0.f;
    const int col_1[1] = {0};
    if (wave < 2) gemm192<1, 1, FMT, SLABS>(A, a.w1, 32, col_1, ac1, wave, li, lh);
    __syncthreads();                                            // A is dead from here: reuse it for y1 [64][32] float32
    float* Y = reinterpret_cast<float*>(A);
    if (wave < 2) {
        const float b1 = a.b1[li];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float pre = ac1[0][0][r] + b1, y = gelu_exact(pre);
            Y[row * 32 + li] = y;
            if constexpr (SAVE) {
                const unsigned at = (((unsigned)r0 + row) * 32u + li) * 4u;
                sv_store<6>(sv.pre1, at, pre); sv_store<6>(sv.y1, at, y);
            }
        }
    }
    __syncthreads();
    if constexpr (SAVE) {
        // residual flow rows [n][4]: Linear(32,3)(y1) for in-range points, zeros for the dropped ones (himo_mask_rows) and in column 3
        const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
        const int64_t i = r0 + row;
        float out = 0.f;
        if (c < 3 && s_pid[row] >= 0) {
            const float* y = Y + row * 32;
            float s = y[0] * a.w2[c];
#pragma unroll
            for (int k = 1; k < 32; ++k) s = fmaf(y[k], a.w2[k * a.w2_pitch + c], s);
            out = s + a.b2[c];
        }
        sv.res[i * 4 + c] = out;
        if (a.nonfinite && !(fabsf(out) <= 3.4028234664e38f)) atomicOr(a.nonfinite, 1u);
        return;
    }
    // flow = pose_flow (+ Linear(32,3)(y1) for in-range points): head_final_kernel's arithmetic
    if (threadIdx.x < 3 * kGhRows) {
        const int row = threadIdx.x / 3, c = threadIdx.x % 3;
        const int64_t i = r0 + row;
        if (i < a.n) {
            const float pose_flow = a.xyz_t[i * 3 + c] - a.pts[i * a.stride + c];
            float out = pose_flow;
            if (s_pid[row] >= 0) {
                const float* y = Y + row * 32;
                float s = y[0] * a.w2[c];
#pragma unroll
                for (int k = 1; k < 32; ++k) s = fmaf(y[k], a.w2[k * a.w2_pitch + c], s);
                out = pose_flow + (s + a.b2[c]);
            }
            a.flow[i * 3 + c] = out;
            // the finite-flow guard of the fp16-split arithmetic, at the only place a flow value is born (an overflowed activation
            // is inf -> NaN at the next split and reaches every output it feeds): taken only when something IS wrong
            if (a.nonfinite && !(fabsf(out) <= 3.4028234664e38f)) atomicOr(a.nonfinite, 1u);
        }
    }
}

}  // namespace himo

using namespace himo;

// hidden 128 (= 32 + 32 + 64 gathered channels), x 64, dec1 width 32: the head of himo_amd/seflow/spec.py.  Packed
// weights: himo_conv_pack_weights_ex(w, 1, 192, cout, packed_format) of zr [192][256], q [192][128], dec1 [192][32]
// -- or, folded: of [144][cout] matrices = the 128 hidden rows, then W_off W_x (3 rows), b_off W_x (1 row), 12 zero rows.
static int gru_head_launch(int n_samples, const himo_head_sample* h_samples, int img_pitch, int dec_pitch,
                           const float* d_w_off, const float* d_b_off, bool folded,
                           const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                           const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                           int iters, int packed_format, int img_split, unsigned* d_nonfinite, void* stream) {
    if (img_split && (packed_format != 1 || (img_pitch & 15))) return HIMO_ERR_INVALID_ARGUMENT;
    if (n_samples < 0 || n_samples > kGhMaxSamples || (n_samples && !h_samples)) return HIMO_ERR_INVALID_ARGUMENT;
    if (iters < 0 || !(packed_format == 0 || packed_format == 1) || img_pitch < 32 || dec_pitch < 64) return HIMO_ERR_INVALID_ARGUMENT;
    if ((!folded && (!d_w_off || !d_b_off)) || !d_wzr_packed || !d_bzr || !d_wq_packed || !d_bq || !d_w1_packed || !d_b1 || !d_w2 || !d_b2)
        return HIMO_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(d_wzr_packed) | reinterpret_cast<uintptr_t>(d_wq_packed) | reinterpret_cast<uintptr_t>(d_w1_packed)) & 15)
        return HIMO_ERR_INVALID_ARGUMENT;
    GruHeadBatch b{};
    int64_t blocks = 0;
    for (int i = 0; i < n_samples; ++i) {
        const himo_head_sample& h = h_samples[i];
        if (h.n < 0 || h.pc_stride < 3) return HIMO_ERR_INVALID_ARGUMENT;
        if (h.n == 0) continue;                         // empty samples take no blocks
        if (!h.d_pid || !h.d_offsets || !h.d_img0 || !h.d_img1 || !h.d_dec || !h.d_xyz_t || !h.d_pts || !h.d_flow) return HIMO_ERR_INVALID_ARGUMENT;
        GruHeadSample& g = b.s[b.n_samples];
        g.n = h.n; g.pid = h.d_pid; g.offsets = h.d_offsets; g.img0 = h.d_img0; g.img1 = h.d_img1; g.dec = h.d_dec;
        g.xyz_t = h.d_xyz_t; g.pts = h.d_pts; g.flow = h.d_flow; g.stride = h.pc_stride;
        b.block_start[b.n_samples] = (int)blocks;
        blocks += (h.n + kGhRows - 1) / kGhRows;
        if (blocks > 0x7fffffff) return HIMO_ERR_INVALID_ARGUMENT;
        b.block_start[++b.n_samples] = (int)blocks;
    }
    if (!blocks) return HIMO_OK;
    GruHeadArgs a{};
    a.img_pitch = img_pitch; a.dec_pitch = dec_pitch; a.w_off = d_w_off; a.b_off = d_b_off;
    a.wzr = (const unsigned short*)d_wzr_packed; a.bzr = d_bzr; a.wq = (const unsigned short*)d_wq_packed; a.bq = d_bq;
    a.w1 = (const unsigned short*)d_w1_packed; a.b1 = d_b1; a.w2 = d_w2; a.b2 = d_b2; a.iters = iters;
    a.img_split = img_split ? 1 : 0;
    a.nonfinite = d_nonfinite;
    a.w2_pitch = 3;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("gru_head_kernel", s);
    const dim3 grid((unsigned)blocks);
    const GruHeadSave none{};
    if (folded) {
        if (packed_format == 1) hipLaunchKernelGGL((gru_head_kernel<2, 9>), grid, dim3(256), 0, s, a, b, none);
        else hipLaunchKernelGGL((gru_head_kernel<3, 9>), grid, dim3(256), 0, s, a, b, none);
    } else {
        if (packed_format == 1) hipLaunchKernelGGL((gru_head_kernel<2, 12>), grid, dim3(256), 0, s, a, b, none);
        else hipLaunchKernelGGL((gru_head_kernel<3, 12>), grid, dim3(256), 0, s, a, b, none);
    }
    HIMO_LAUNCH_CHECK("gru_head_kernel");
    return HIMO_OK;
}

extern "C" int himo_gru_head_batch(int n_samples, const himo_head_sample* h_samples, int img_pitch, int dec_pitch,
                                   const float* d_w_off, const float* d_b_off,
                                   const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                                   const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                                   int iters, int packed_format, int img_split, void* stream) {
    return gru_head_launch(n_samples, h_samples, img_pitch, dec_pitch, d_w_off, d_b_off, false, d_wzr_packed, d_bzr, d_wq_packed, d_bq,
                           d_w1_packed, d_b1, d_w2, d_b2, iters, packed_format, img_split, nullptr, stream);
}

extern "C" int himo_gru_head_batch_folded(int n_samples, const himo_head_sample* h_samples, int img_pitch, int dec_pitch,
                                          const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                                          const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                                          int iters, int packed_format, int img_split, void* stream) {
    return gru_head_launch(n_samples, h_samples, img_pitch, dec_pitch, nullptr, nullptr, true, d_wzr_packed, d_bzr, d_wq_packed, d_bq,
                           d_w1_packed, d_b1, d_w2, d_b2, iters, packed_format, img_split, nullptr, stream);
}

// Both forms above plus the finite-flow guard: d_w_off / d_b_off NULL selects the folded weights ([144][cout] matrices);
// d_nonfinite (device uint32, or NULL) is OR-ed with 1 when any flow value this launch writes is NaN or infinite -- the caller
// clears it (himo_clear_u32) before the launches it wants to cover and reads it back when it needs the verdict.
extern "C" int himo_gru_head_batch_guarded(int n_samples, const himo_head_sample* h_samples, int img_pitch, int dec_pitch,
                                           const float* d_w_off, const float* d_b_off,
                                           const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                                           const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                                           int iters, int packed_format, int img_split, uint32_t* d_nonfinite, void* stream) {
    if ((d_w_off == nullptr) != (d_b_off == nullptr)) return HIMO_ERR_INVALID_ARGUMENT;
    return gru_head_launch(n_samples, h_samples, img_pitch, dec_pitch, d_w_off, d_b_off, d_w_off == nullptr, d_wzr_packed, d_bzr, d_wq_packed,
                           d_bq, d_w1_packed, d_b1, d_w2, d_b2, iters, packed_format, img_split, d_nonfinite, stream);
}

// The head's TRAINING forward: one launch = himo_head_gather + the GRU row products with their gate kernels + the decoder of the
// unfused trainer path (himo_amd/seflow/train.py HeadTrainer.forward), leaving every tensor its backward pass reads.
extern "C" int himo_gru_head_train(int64_t n, const int32_t* d_pid, const float* d_offsets, const float* d_img0, const float* d_img1,
                                   int img_pitch, const float* d_dec, int dec_pitch, const float* d_w_off, const float* d_b_off,
                                   const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                                   const void* d_w1_packed, const float* d_b1, const float* d_w2, int w2_pitch, const float* d_b2,
                                   int iters, int packed_format, const himo_head_saved* h_saved, uint32_t* d_nonfinite, void* stream) {
    if (n < 0 || iters < 1 || iters > kGhMaxIters || !(packed_format == 0 || packed_format == 1) || img_pitch < 32 || dec_pitch < 64 ||
        !(w2_pitch == 3 || w2_pitch == 4) || !h_saved)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (!d_w_off || !d_b_off || !d_wzr_packed || !d_bzr || !d_wq_packed || !d_bq || !d_w1_packed || !d_b1 || !d_w2 || !d_b2)
        return HIMO_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(d_wzr_packed) | reinterpret_cast<uintptr_t>(d_wq_packed) | reinterpret_cast<uintptr_t>(d_w1_packed)) & 15)
        return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_pid || !d_offsets || !d_img0 || !d_img1 || !d_dec) return HIMO_ERR_INVALID_ARGUMENT;
    if (h_saved && h_saved->rows * 768 >= ((int64_t)1 << 32)) return HIMO_ERR_UNSUPPORTED;      // 32-bit byte offsets inside a saved tensor
    GruHeadSave sv{};
    sv.hx = h_saved->d_hx; sv.rhx = h_saved->d_rhx; sv.z = h_saved->d_z; sv.r = h_saved->d_r; sv.q = h_saved->d_q;
    sv.pre1 = h_saved->d_pre1; sv.y1 = h_saved->d_y1; sv.res = h_saved->d_res;
    sv.rows = h_saved->rows;                                   // the iteration stride: >= ceil(n / 64) * 64 (a caller may keep capacity)
    if (!sv.hx || !sv.rhx || !sv.z || !sv.r || !sv.q || !sv.pre1 || !sv.y1 || !sv.res || sv.rows < (n + 63) / 64 * 64 || (sv.rows & 63))
        return HIMO_ERR_INVALID_ARGUMENT;
    GruHeadBatch b{};
    b.n_samples = 1;
    GruHeadSample& g = b.s[0];
    g.n = n; g.pid = d_pid; g.offsets = d_offsets; g.img0 = d_img0; g.img1 = d_img1; g.dec = d_dec;
    const int64_t blocks = (n + kGhRows - 1) / kGhRows;
    if (blocks > 0x7fffffff) return HIMO_ERR_INVALID_ARGUMENT;
    b.block_start[0] = 0; b.block_start[1] = (int)blocks;
    GruHeadArgs a{};
    a.img_pitch = img_pitch; a.dec_pitch = dec_pitch; a.w_off = d_w_off; a.b_off = d_b_off;
    a.wzr = (const unsigned short*)d_wzr_packed; a.bzr = d_bzr; a.wq = (const unsigned short*)d_wq_packed; a.bq = d_bq;
    a.w1 = (const unsigned short*)d_w1_packed; a.b1 = d_b1; a.w2 = d_w2; a.b2 = d_b2; a.iters = iters; a.w2_pitch = w2_pitch;
    a.nonfinite = d_nonfinite;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("gru_head_train_kernel", s);
    if (packed_format == 1) hipLaunchKernelGGL((gru_head_kernel<2, 12, true>), dim3((unsigned)blocks), dim3(256), 0, s, a, b, sv);
    else hipLaunchKernelGGL((gru_head_kernel<3, 12, true>), dim3((unsigned)blocks), dim3(256), 0, s, a, b, sv);
    HIMO_LAUNCH_CHECK("gru_head_kernel<train>");
    return HIMO_OK;
}

extern "C" int himo_clear_u32(uint32_t* d_words, int n, void* stream) {
    if (!d_words || n < 1) return HIMO_ERR_INVALID_ARGUMENT;
    HIMO_HIP(hipMemsetAsync(d_words, 0, (size_t)n * 4, (hipStream_t)stream));
    return HIMO_OK;
}

extern "C" int himo_gru_head(int64_t n, const int32_t* d_pid, const float* d_offsets, const float* d_img0, const float* d_img1,
                             int img_pitch, const float* d_dec, int dec_pitch, const float* d_w_off, const float* d_b_off,
                             const void* d_wzr_packed, const float* d_bzr, const void* d_wq_packed, const float* d_bq,
                             const void* d_w1_packed, const float* d_b1, const float* d_w2, const float* d_b2,
                             const float* d_xyz_t, const float* d_pts, int pc_stride, float* d_flow, int iters, int packed_format,
                             void* stream) {
    himo_head_sample h{n, d_pid, d_offsets, d_img0, d_img1, d_dec, d_xyz_t, d_pts, pc_stride, d_flow};
    return himo_gru_head_batch(1, &h, img_pitch, dec_pitch, d_w_off, d_b_off, d_wzr_packed, d_bzr, d_wq_packed, d_bq,
                               d_w1_packed, d_b1, d_w2, d_b2, iters, packed_format, 0, stream);
}
