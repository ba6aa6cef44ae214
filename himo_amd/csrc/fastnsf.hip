// fastnsf.hip -- stage a12: device pieces of the optimisation-based scene flow ("fastnsf": a per-scene
// coordinate MLP fitted by gradient descent on a truncated Chamfer objective).
//
// PARITY UNPINNED: the reference only names the method (`python save.py model=fastnsf`, README.md:53; result keys
// `fastnsf10` / `nsfp`, tools/view_instance.py:155); the implementation is in the absent OpenSceneFlow submodule.
// Specification: himo_amd/fastnsf.py (this build's own, after the published Neural Scene Flow Prior family);
// oracle: oracle/fastnsf_oracle.py (PyTorch CPU autograd + Adam).
//
// The MLP forward / input-gradient GEMMs run on conv.hip's row GEMM (epilogues BIAS_RELU / RELU_MASK); this file
// holds what is specific to fitting:
//   wgrad_partial_kernel   dW = X^T dZ as a split-K matrix product on the float32 matrix cores: every block owns
//                          256 point rows and accumulates a full Cin x Cout (<= 128 x 128) partial; MFMA A/B
//                          fragments are read STRAIGHT from global memory -- a fragment is 32 consecutive floats of
//                          one row, i.e. one coalesced 128-byte access -- so no LDS staging is needed;
//   wgrad_reduce_kernel    fixed-order sum of the partials (+ the bias gradient = column sums of dZ);
//   transpose_kernel       W -> W^T for the input-gradient GEMM;
//   adam_kernel            the optimiser step;
//   chamfer_trunc_*        the objective and its gradient with respect to the moved points.
#include "himo_common.h"
#include <math.h>

namespace himo {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kWgRows = 256;     // point rows per block

// partial[tile][b][128][128]: tile = (ci tile, co tile) = blockIdx.y, b = row chunk = blockIdx.x
__global__ __launch_bounds__(256) void wgrad_partial_kernel(int64_t n, const float* __restrict__ X, int x_pitch, int cin,
                                                           const float* __restrict__ dZ, int z_pitch, int cout,
                                                           float* __restrict__ partial, int rows_pb) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;            // wave tile: ci in [64 wm, +64), co in [64 wn, +64)
    const int li = lane & 31, lh = lane >> 5;
    const int co_tiles = (cout + 127) / 128;
    const int ci0 = ((int)blockIdx.y / co_tiles) * 128, co0 = ((int)blockIdx.y % co_tiles) * 128;
    X += ci0; dZ += co0; cin -= ci0; cout -= co0;       // this block's 128 x 128 window
    const int64_t r0 = (int64_t)blockIdx.x * rows_pb;
    const int64_t r1 = r0 + rows_pb < n ? r0 + rows_pb : n;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    int ci[2], co[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { ci[t] = wm * 64 + t * 32 + li; co[t] = wn * 64 + t * 32 + li; }
    const bool any_ci = wm * 64 < cin, any_co = wn * 64 < cout;
    if (any_ci && any_co) {
#pragma unroll 4
        for (int64_t r = r0; r < r1; r += 2) {
            const int64_t row = r + lh;                    // k index of this half-wave
            const bool ok = row < r1;
            float af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = (ok && ci[t] < cin) ? X[row * x_pitch + ci[t]] : 0.f;      // A[i = ci][k = row]
                bf[t] = (ok && co[t] < cout) ? dZ[row * z_pitch + co[t]] : 0.f;    // B[k = row][j = co]
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    float* out = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 128 * 128;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // ci
                const int col = wn * 64 + b * 32 + li;                                  // co
                out[row * 128 + col] = acc[a][b][r];
            }
}

// The same partial tiles with the operands staged through LDS: 32 rows x 128 columns of X and of dZ per stage, loaded
// with 16-byte accesses (the fragment-from-global kernel above issues one 4-byte load per lane per fragment and is bound
// by the number of load instructions, not by bytes), double-buffered; fragments are conflict-free ds_read_b32 (the two
// half-waves of a fragment read two different rows).  Needs 16-byte aligned rows; the caller falls back otherwise.
constexpr int kWtRows = 32;

__global__ __launch_bounds__(256, 2) void wgrad_partial_lds_kernel(int64_t n, const float* __restrict__ X, int x_pitch, int cin,
                                                                   const float* __restrict__ dZ, int z_pitch, int cout,
                                                                   float* __restrict__ partial, int rows_pb) {
    __shared__ __attribute__((aligned(16))) float Xs[2][kWtRows][128];
    __shared__ __attribute__((aligned(16))) float Zs[2][kWtRows][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int co_tiles = (cout + 127) / 128;
    const int ci0 = ((int)blockIdx.y / co_tiles) * 128, co0 = ((int)blockIdx.y % co_tiles) * 128;
    const int64_t r0 = (int64_t)blockIdx.x * rows_pb;
    const int64_t r1 = r0 + rows_pb < n ? r0 + rows_pb : n;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging: thread -> (row = it * 8 + t / 32, float4 column q = t % 32), 4 rows x 2 matrices per thread per stage
    const int q = threadIdx.x & 31, srow = threadIdx.x >> 5;
    float4 px[4], pz[4];
    auto load_stage = [&](int64_t base) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int64_t row = base + it * 8 + srow;
            const bool ok = row < r1;
            px[it] = (ok && ci0 + q * 4 < cin) ? *reinterpret_cast<const float4*>(X + row * x_pitch + ci0 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            pz[it] = (ok && co0 + q * 4 < cout) ? *reinterpret_cast<const float4*>(dZ + row * z_pitch + co0 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            *reinterpret_cast<float4*>(&Xs[buf][it * 8 + srow][q * 4]) = px[it];
            *reinterpret_cast<float4*>(&Zs[buf][it * 8 + srow][q * 4]) = pz[it];
        }
    };
    load_stage(r0);
    store_stage(0);
    __syncthreads();
    int buf = 0;
    for (int64_t base = r0; base < r1; base += kWtRows) {
        const bool more = base + kWtRows < r1;
        if (more) load_stage(base + kWtRows);
#pragma unroll 4
        for (int kk = 0; kk < kWtRows / 2; ++kk) {
            const int row = 2 * kk + lh;
            float af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = Xs[buf][row][wm * 64 + t * 32 + li];
                bf[t] = Zs[buf][row][wn * 64 + t * 32 + li];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) {
            store_stage(buf ^ 1);             // the other buffer was last read before the previous barrier
            __syncthreads();
            buf ^= 1;
        }
    }
    float* out = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 128 * 128;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + wn * 64 + b * 32 + li] = acc[a][b][r];
}

// The same partial tiles with SPLIT-bf16 operands (himo_linear_wgrad_ex flag 2; the mixed-precision training step): x = h + m
// in bf16 (16 significant bits, float32 range), products h*h + h*m + m*h on v_mfma_f32_32x32x16_bf16 -- 3 instructions per 16
// rows where the float32 kernel issues 8 -- which leaves the kernel bound by its operand stream.  The reduction index is the
// ROW, so both operands are staged transposed ([plane][column][row], 8 rows = one 16-byte fragment; rows padded to 40 = 80
// bytes so that consecutive columns sit 20 banks apart: conflict-free 16-byte stores and fragment reads), 32 rows per stage,
// the next stage's 32 coalesced 4-byte loads per lane in flight during the current stage's matrix instructions.  Needs
// cin, cout multiples of 4 and 16-byte aligned rows like the float32 LDS kernel; same partial layout and reduction.
constexpr int kWsRowsPad = 40;
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void wg_split_pair(float a, float b, unsigned& hw, unsigned& mw) {
    typedef float wg_f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 wg_b2 __attribute__((ext_vector_type(2)));
    wg_f2 v; v[0] = a; v[1] = b;
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg_b2));
    wg_f2 r;
    r[0] = a - __builtin_bit_cast(float, hw << 16);
    r[1] = b - __builtin_bit_cast(float, hw & 0xffff0000u);
    mw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wg_b2));
}

__global__ __launch_bounds__(256, 2) void wgrad_partial_split_kernel(int64_t n, const float* __restrict__ X, int x_pitch, int cin,
                                                                     const float* __restrict__ dZ, int z_pitch, int cout,
                                                                     float* __restrict__ partial, int rows_pb, float* __restrict__ colpart,
                                                                     int n_chunks) {
    __shared__ __attribute__((aligned(16))) unsigned short Xt[2][128][kWsRowsPad];
    __shared__ __attribute__((aligned(16))) unsigned short Zt[2][128][kWsRowsPad];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int co_tiles = (cout + 127) / 128, tiles = ((cin + 127) / 128) * co_tiles;
    // 1-D grid: the (input tile, output tile) blocks of ONE row chunk get the linear ids m * 8 + x for consecutive m -- the dispatcher
    // places block b on XCD b % 8, so they run on the same XCD at the same time and the X / dZ rows that two of them read (each operand
    // tile is read once per tile of the other operand: 1.7 GB for the head's 480k x 192 x 256 gate gradient, which ran at HBM speed)
    // meet in that XCD's L2 instead of coming from HBM once per reader
    const int lin = (int)blockIdx.x, xcd = lin & 7, m = lin >> 3;
    const int tile = m % tiles, chunk = (m / tiles) * 8 + xcd;
    if (chunk >= n_chunks) return;
    const int ci0 = (tile / co_tiles) * 128, co0 = (tile % co_tiles) * 128;
    const int64_t r0 = (int64_t)chunk * rows_pb;
    const int64_t r1 = r0 + rows_pb < n ? r0 + rows_pb : n;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging: this thread's column of the tile, its two 8-row chunks of the 32-row stage (chunk = sg + 2 it).  The chunk index is
    // wave-uniform and goes through readfirstlane: row numbers, the end-of-range test and the row base addresses are SCALAR, a load is
    // one global_load_dword (scalar row base + the lane's column); columns beyond the matrix load column 0 and are zeroed when staged
    // (per-lane 64-bit row arithmetic and an exec-mask branch around each of a stage's 32 loads outweighed its 24 matrix instructions)
    const int sc = threadIdx.x & 127;
    const int sg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
    const bool x_ok = ci0 + sc < cin, z_ok = co0 + sc < cout;
    const int x_col = x_ok ? sc : 0, z_col = z_ok ? sc : 0;
    float vx[2][8], vz[2][8];
    // the bias gradient (column sums of dZ) rides along in the blocks of input tile 0: dZ passes through these registers anyway, a
    // separate column-sum pass read it from HBM a second time (colpart [co tile][block][128], reduced by colsum_reduce_kernel)
    const bool want_db = colpart != nullptr && ci0 == 0;
    float bsum = 0.f;
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int64_t row0 = base + (sg + 2 * it) * 8;
            const float* xr = X + row0 * x_pitch + ci0;
            const float* zr = dZ + row0 * z_pitch + co0;
            if (row0 + 8 <= r1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { vx[it][j] = xr[(int64_t)j * x_pitch + x_col]; vz[it][j] = zr[(int64_t)j * z_pitch + z_col]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    vx[it][j] = 0.f; vz[it][j] = 0.f;
                    if (row0 + j < r1) { vx[it][j] = xr[(int64_t)j * x_pitch + x_col]; vz[it][j] = zr[(int64_t)j * z_pitch + z_col]; }
                }
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int ch = sg + 2 * it;
            if (!x_ok)
#pragma unroll
                for (int j = 0; j < 8; ++j) vx[it][j] = 0.f;
            if (!z_ok)
#pragma unroll
                for (int j = 0; j < 8; ++j) vz[it][j] = 0.f;
            if (want_db) bsum += ((vz[it][0] + vz[it][1]) + (vz[it][2] + vz[it][3])) + ((vz[it][4] + vz[it][5]) + (vz[it][6] + vz[it][7]));
            uint4 h, m;
            wg_split_pair(vx[it][0], vx[it][1], h.x, m.x); wg_split_pair(vx[it][2], vx[it][3], h.y, m.y);
            wg_split_pair(vx[it][4], vx[it][5], h.z, m.z); wg_split_pair(vx[it][6], vx[it][7], h.w, m.w);
            *reinterpret_cast<uint4*>(&Xt[0][sc][ch * 8]) = h;
            *reinterpret_cast<uint4*>(&Xt[1][sc][ch * 8]) = m;
            wg_split_pair(vz[it][0], vz[it][1], h.x, m.x); wg_split_pair(vz[it][2], vz[it][3], h.y, m.y);
            wg_split_pair(vz[it][4], vz[it][5], h.z, m.z); wg_split_pair(vz[it][6], vz[it][7], h.w, m.w);
            *reinterpret_cast<uint4*>(&Zt[0][sc][ch * 8]) = h;
            *reinterpret_cast<uint4*>(&Zt[1][sc][ch * 8]) = m;
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t base = r0; base < r1; base += 32) {
        __syncthreads();                                   // the previous stage's fragment reads are done
        stage();
        __syncthreads();
        if (base + 32 < r1) fetch(base + 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wg_bf16x8 af[2][2], bf[2][2];                  // [tile][plane]
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    af[t][pl] = *reinterpret_cast<const wg_bf16x8*>(&Xt[pl][wm * 64 + t * 32 + li][ks * 16 + lh * 8]);
                    bf[t][pl] = *reinterpret_cast<const wg_bf16x8*>(&Zt[pl][wn * 64 + t * 32 + li][ks * 16 + lh * 8]);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                }
        }
    }
    if (want_db) {                                         // the two chunk groups' shares of a column, fixed order
        __syncthreads();
        float* bsh = reinterpret_cast<float*>(&Xt[0][0][0]);
        bsh[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < 128) colpart[((int64_t)(co0 / 128) * n_chunks + chunk) * 128 + sc] = bsh[sc] + bsh[128 + sc];
    }
    float* out = partial + ((int64_t)tile * n_chunks + chunk) * 128 * 128;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + wn * 64 + b * 32 + li] = acc[a][b][r];
}

// ---- the split-bf16 weight gradient with EVERY operand row read ONCE (round 6; himo_linear_wgrad_ex flag 2 with 128 < cin <= 192 and
// cout = 128 | 256: the head's gate gradients over the four stacked GRU iterations, 480k rows x 192 -> 128 | 256).  The 128 x 128 block
// tiles above read each operand tile once per tile of the OTHER operand (X twice, dZ twice for 192 -> 256: 1.7 GB; co-scheduling the tiles
// of a row chunk on one XCD made the second read an L2 hit, and the kernel still ran at the rate of its operand stream: 347 us, 0.16 of
// the matrix peak).  Here ONE block of eight waves owns all cin x cout outputs for its run of rows -- wave (wm, wn): input-channel tiles
// [3 wm, 3 wm + 3) x output tiles [NB wn, NB wn + NB), 3 NB accumulators of 32 x 32 -- so X and dZ leave HBM once (0.86 GB), a stage
// of 32 rows is (192 + cout) columns of split operands in LDS (72 KB), and a wave issues 9 NB matrix instructions on 2 (3 + NB) fragment
// reads per 16 rows.  One block per CU (256 blocks: one round), the same partial layout (128 x 128 tiles) and reduce kernels as above;
// the per-element summation order over the rows of a block is the 128-tile kernel's.
template <int NB, int STAGE>                               // output tiles per wave: 1 (cout 128) or 2 (cout 256); rows per stage: 32 | 64
__global__ __launch_bounds__(512, 1) void wgrad_full_split_kernel(int64_t n, const float* __restrict__ X, int x_pitch, int cin,
                                                                   const float* __restrict__ dZ, int z_pitch, float* __restrict__ partial,
                                                                   int rows_pb, float* __restrict__ colpart, int n_chunks) {
    constexpr int CI = 192, CO = NB * 128, kChunks = STAGE / 8, kPad = STAGE + 8;
    constexpr int kXItems = (CI * kChunks + 511) / 512, kZItems = CO * kChunks / 512;      // staging items per thread: (column, 8-row chunk) pairs
    __shared__ __attribute__((aligned(16))) unsigned short Xt[2][CI][kPad];
    __shared__ __attribute__((aligned(16))) unsigned short Zt[2][CO][kPad];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int chunk = (int)blockIdx.x;
    const int64_t r0 = (int64_t)chunk * rows_pb;
    const int64_t r1 = r0 + rows_pb < n ? r0 + rows_pb : n;
    floatx16 acc[3][NB];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // staging item i = thread + 512 k: X items 0 .. 767 = (column i % 192, chunk i / 192), dZ items 0 .. 4 CO - 1 = (column i % CO, chunk
    // i / CO).  64 divides 192 and CO, so a wave's 64 items share their chunk: row numbers and row bases are scalar, a load is one
    // global_load_dword with the lane's column as its offset (columns >= cin load column 0 and are staged as zeros)
    int xcol[kXItems], xch[kXItems], zcol[kZItems], zch[kZItems];
    bool xlive[kXItems], xok[kXItems];
#pragma unroll
    for (int k = 0; k < kXItems; ++k) {
        const int i = (int)threadIdx.x + 512 * k;
        xlive[k] = __builtin_amdgcn_readfirstlane((int)(i < kChunks * CI));
        xch[k] = __builtin_amdgcn_readfirstlane(i / CI);
        const int c = i % CI;
        xok[k] = c < cin; xcol[k] = c;
    }
#pragma unroll
    for (int k = 0; k < kZItems; ++k) {
        const int i = (int)threadIdx.x + 512 * k;
        zch[k] = __builtin_amdgcn_readfirstlane(i / CO);
        zcol[k] = i % CO;
    }
    float vx[kXItems][8], vz[kZItems][8];
    const bool want_db = colpart != nullptr;
    float bsum = 0.f;
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int k = 0; k < kXItems; ++k) {
            if (!xlive[k]) continue;
            const int64_t row0 = base + xch[k] * 8;
            const float* xr = X + row0 * x_pitch;
            const int c = xok[k] ? xcol[k] : 0;
            if (row0 + 8 <= r1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) vx[k][j] = xr[(int64_t)j * x_pitch + c];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) vx[k][j] = row0 + j < r1 ? xr[(int64_t)j * x_pitch + c] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < kZItems; ++k) {
            const int64_t row0 = base + zch[k] * 8;
            const float* zr = dZ + row0 * z_pitch;
            if (row0 + 8 <= r1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) vz[k][j] = zr[(int64_t)j * z_pitch + zcol[k]];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) vz[k][j] = row0 + j < r1 ? zr[(int64_t)j * z_pitch + zcol[k]] : 0.f;
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < kXItems; ++k) {
            if (!xlive[k]) continue;
            if (!xok[k])
#pragma unroll
                for (int j = 0; j < 8; ++j) vx[k][j] = 0.f;
            uint4 h, m;
            wg_split_pair(vx[k][0], vx[k][1], h.x, m.x); wg_split_pair(vx[k][2], vx[k][3], h.y, m.y);
            wg_split_pair(vx[k][4], vx[k][5], h.z, m.z); wg_split_pair(vx[k][6], vx[k][7], h.w, m.w);
            *reinterpret_cast<uint4*>(&Xt[0][xcol[k]][xch[k] * 8]) = h;
            *reinterpret_cast<uint4*>(&Xt[1][xcol[k]][xch[k] * 8]) = m;
        }
#pragma unroll
        for (int k = 0; k < kZItems; ++k) {
            if (want_db) bsum += ((vz[k][0] + vz[k][1]) + (vz[k][2] + vz[k][3])) + ((vz[k][4] + vz[k][5]) + (vz[k][6] + vz[k][7]));
            uint4 h, m;
            wg_split_pair(vz[k][0], vz[k][1], h.x, m.x); wg_split_pair(vz[k][2], vz[k][3], h.y, m.y);
            wg_split_pair(vz[k][4], vz[k][5], h.z, m.z); wg_split_pair(vz[k][6], vz[k][7], h.w, m.w);
            *reinterpret_cast<uint4*>(&Zt[0][zcol[k]][zch[k] * 8]) = h;
            *reinterpret_cast<uint4*>(&Zt[1][zcol[k]][zch[k] * 8]) = m;
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t base = r0; base < r1; base += STAGE) {
        __syncthreads();                                   // the previous stage's fragment reads are done
        stage();
        __syncthreads();
        if (base + STAGE < r1) fetch(base + STAGE);
#pragma unroll
        for (int ks = 0; ks < STAGE / 16; ++ks) {
            wg_bf16x8 af[3][2], bf[NB][2];                 // [tile][plane]
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int t = 0; t < 3; ++t) af[t][pl] = *reinterpret_cast<const wg_bf16x8*>(&Xt[pl][(wm * 3 + t) * 32 + li][ks * 16 + lh * 8]);
#pragma unroll
                for (int u = 0; u < NB; ++u) bf[u][pl] = *reinterpret_cast<const wg_bf16x8*>(&Zt[pl][(wn * NB + u) * 32 + li][ks * 16 + lh * 8]);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                }
        }
    }
    if (want_db) {                                         // a column's shares (512 / CO threads x their items), fixed order
        __syncthreads();
        float* bsh = reinterpret_cast<float*>(&Xt[0][0][0]);
        bsh[threadIdx.x] = bsum;
        __syncthreads();
        if ((int)threadIdx.x < CO) {
            const int c = threadIdx.x;
            const float t = NB == 2 ? bsh[c] + bsh[256 + c] : (bsh[c] + bsh[128 + c]) + (bsh[256 + c] + bsh[384 + c]);      // (512 / CO threads per column)
            colpart[((int64_t)(c / 128) * n_chunks + chunk) * 128 + (c & 127)] = t;
        }
    }
    // partial tiles in the 128 x 128 layout of the kernels above: tile (ci / 128) * (CO / 128) + co / 128
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ci_t = (wm * 3 + a) * 32, co = (wn * NB + b) * 32 + li;
            float* out = partial + ((int64_t)((ci_t >> 7) * NB + (co >> 7)) * n_chunks + chunk) * 128 * 128;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci_t + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(ci & 127) * 128 + (co & 127)] = acc[a][b][r];
            }
        }
}

// dW[ci][co] (+)= sum_b partial[tile][b][ci % 128][co % 128], deterministic: a block owns 64 consecutive elements, its
// four thread groups sum every fourth chunk (four loads in flight per element instead of one serial chain), and the
// groups are combined in a fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int n_blocks, int cin, int cout,
                                                           float* __restrict__ dW, int accumulate) {
    __shared__ float sh[4][64];
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + o;
    float s = 0.f;
    if (e < cin * cout) {
        const int ci = e / cout, co = e % cout;
        const int tile = (ci / 128) * ((cout + 127) / 128) + co / 128;
        const float* p = partial + (int64_t)tile * n_blocks * 128 * 128 + (ci % 128) * 128 + (co % 128);
        // four independent chains per thread (16 loads in flight per element over the four groups): the sum of ~500 partials
        // as ONE chain costs a memory latency per term, not a byte rate
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = grp;
        for (; b + 12 < n_blocks; b += 16) {
            s0 += p[(int64_t)b * 128 * 128]; s1 += p[(int64_t)(b + 4) * 128 * 128];
            s2 += p[(int64_t)(b + 8) * 128 * 128]; s3 += p[(int64_t)(b + 12) * 128 * 128];
        }
        for (; b < n_blocks; b += 4) s0 += p[(int64_t)b * 128 * 128];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[grp][o] = s;
    __syncthreads();
    if (grp == 0 && e < cin * cout) {
        const float t = (sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o]);
        dW[e] = accumulate ? dW[e] + t : t;
    }
}

// bias gradient: column sums of dZ, two stages with fixed order
__global__ __launch_bounds__(256) void colsum_partial_kernel(int64_t n, const float* __restrict__ dZ, int z_pitch, int cout,
                                                            float* __restrict__ partial, int rows_pb) {
    // block = rows_pb rows x 128 columns (column tile blockIdx.y): thread t sums column (t % 128) over rows of parity (t / 128)
    const int col = (int)blockIdx.y * 128 + (threadIdx.x & 127), half = threadIdx.x >> 7;
    const int64_t r0 = (int64_t)blockIdx.x * rows_pb;
    const int64_t r1 = r0 + rows_pb < n ? r0 + rows_pb : n;
    float s = 0.f;
    if (col < cout)
        for (int64_t r = r0 + half; r < r1; r += 2) s += dZ[r * z_pitch + col];
    __shared__ float sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    const int lc = threadIdx.x & 127;
    if (half == 0) partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 128 + lc] = sh[lc] + sh[lc + 128];
}

// the same partial sums with 16-byte loads: 8 row groups x 32 float4 columns per block (needs 16-byte aligned rows)
__global__ __launch_bounds__(256) void colsum_partial_v4_kernel(int64_t n, const float* __restrict__ dZ, int z_pitch, int cout,
                                                               float* __restrict__ partial, int rows_pb) {
    const int q = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int col = (int)blockIdx.y * 128 + q * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_pb;
    const int64_t r1 = r0 + rows_pb < n ? r0 + rows_pb : n;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < cout) {
#pragma unroll 4
        for (int64_t r = r0 + grp; r < r1; r += 8) {
            const float4 v = *reinterpret_cast<const float4*>(dZ + r * z_pitch + col);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    __shared__ float4 sh[8][32];
    sh[grp][q] = s;
    __syncthreads();
    if (grp == 0) {
        float4 t = sh[0][q];
#pragma unroll
        for (int g = 1; g < 8; ++g) { t.x += sh[g][q].x; t.y += sh[g][q].y; t.z += sh[g][q].z; t.w += sh[g][q].w; }
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 128 + q * 4) = t;
    }
}

// 8 groups x 128 columns: group g sums partials g, g+8, ...; fixed-order combine
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ partial, int n_blocks, int cout,
                                                            float* __restrict__ db, int accumulate) {
    __shared__ float sh[8][128];
    const int lc = threadIdx.x & 127, grp = threadIdx.x >> 7;
    const int col = (int)blockIdx.x * 128 + lc;
    float s = 0.f;
    if (col < cout) {
        const float* p = partial + (int64_t)blockIdx.x * n_blocks * 128 + lc;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four chains per thread, fixed order (see wgrad_reduce_kernel)
        int b = grp;
        for (; b + 24 < n_blocks; b += 32) {
            s0 += p[(int64_t)b * 128]; s1 += p[(int64_t)(b + 8) * 128]; s2 += p[(int64_t)(b + 16) * 128]; s3 += p[(int64_t)(b + 24) * 128];
        }
        for (; b < n_blocks; b += 8) s0 += p[(int64_t)b * 128];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[grp][lc] = s;
    __syncthreads();
    if (grp == 0 && col < cout) {
        float t = sh[0][lc];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += sh[g][lc];
        db[col] = accumulate ? db[col] + t : t;
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * cols) return;
    const int r = e / cols, c = e % cols;
    wt[c * rows + r] = w[e];
}

// Adam (no weight decay, bias-corrected), torch.optim.Adam's update order
__global__ __launch_bounds__(256) void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}

// truncated Chamfer: terms with squared distance above trunc2 are dropped (contribute neither loss nor gradient)
struct ChamferArgs {
    int n0, n1;
    const float* moved; const float* pc1;
    const float* d_a; const int* i_a;      // moved -> pc1
    const float* d_b; const int* i_b;      // pc1 -> moved
    float trunc2;
    float* grad;                           // [n0][3] d loss / d moved
    double* partial;                       // [blocks0 + blocks1]
    unsigned long long* scat;              // [n0][3]: the pc1 -> moved half's scattered gradient, 2^-40 fixed point (order-independent sum)
};
constexpr double kChamferScatScale = 1099511627776.0;      // 2^40 (csrc/sslloss.hip uses the same scheme)

__device__ inline void block_sum1(double v, double* out) {
    __shared__ double red[256];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

__global__ __launch_bounds__(256) void chamfer_trunc_a_kernel(ChamferArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double t = 0.0;
    if (i < a.n0) {
        float g[3] = {0.f, 0.f, 0.f};
        if (a.n1 > 0 && a.d_a[i] <= a.trunc2) {
            t = (double)a.d_a[i] / (double)a.n0;
            const int j = a.i_a[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = 2.0f / (float)a.n0 * (a.moved[i * 3 + c] - a.pc1[j * 3 + c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)                        // (the scattering kernel ran first)
            a.grad[i * 3 + c] = g[c] + (float)((double)(long long)a.scat[i * 3 + c] * (1.0 / kChamferScatScale));
    }
    block_sum1(t, a.partial + blockIdx.x);
}

__global__ __launch_bounds__(256) void chamfer_trunc_b_kernel(ChamferArgs a, int blocks0) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    double t = 0.0;
    if (j < a.n1 && a.n0 > 0 && a.d_b[j] <= a.trunc2) {
        t = (double)a.d_b[j] / (double)a.n1;
        const int i = a.i_b[j];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            atomicAdd(a.scat + i * 3 + c, (unsigned long long)__double2ll_rn((double)(2.0f / (float)a.n1 * (a.moved[i * 3 + c] - a.pc1[j * 3 + c])) * kChamferScatScale));
    }
    block_sum1(t, a.partial + blocks0 + blockIdx.x);
}

// p' = R p + t with separately rounded products and sums (the rule of seflow/spec.py step 0); out pitch >= 3, tail zeroed
__global__ __launch_bounds__(256) void rigid_kernel(int64_t n, const float* __restrict__ p, int stride, const float* __restrict__ T,
                                                    float* __restrict__ y, int y_pitch) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = p[i * stride], yy = p[i * stride + 1], z = p[i * stride + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) y[i * y_pitch + r] = ((x * T[r * 4] + yy * T[r * 4 + 1]) + z * T[r * 4 + 2]) + T[r * 4 + 3];
    for (int c = 3; c < y_pitch; ++c) y[i * y_pitch + c] = 0.f;
}

// y[i][c] = a[i][c] + b_scale * b[i][c] for c < cols (b may be null); columns cols..y_pitch-1 are zeroed when zero_tail
__global__ __launch_bounds__(256) void rows_add_kernel(int64_t n, int cols, const float* __restrict__ a, int a_pitch,
                                                       const float* __restrict__ b, int b_pitch, float b_scale,
                                                       float* __restrict__ y, int y_pitch, int zero_tail) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < cols; ++c) y[i * y_pitch + c] = a[i * a_pitch + c] + (b ? b_scale * b[i * b_pitch + c] : 0.f);
    if (zero_tail)
        for (int c = cols; c < y_pitch; ++c) y[i * y_pitch + c] = 0.f;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
    double t = 0.0;
    for (int b = threadIdx.x; b < n; b += 256) t += partial[b];
    __shared__ double res;
    block_sum1(t, &res);
    if (threadIdx.x == 0) *out = res;
}

}  // namespace himo

using namespace himo;

static int wgrad_rows_per_block(int64_t n, int tiles) {
    const int64_t target_blocks = 512 / tiles > 0 ? 512 / tiles : 1;
    int64_t rows = (n + target_blocks - 1) / target_blocks;
    if (rows < kWgRows) rows = kWgRows;
    return (int)((rows + 31) / 32 * 32);
}

// column sums into d_out; colpart holds ceil(cout/128) * nb * 128 floats with nb <= n / 256 + 1
static int colsum_launch(int64_t n, const float* d_z, int z_pitch, int cout, float* d_out, int acc, float* colpart, hipStream_t s) {
    const int co_tiles = (cout + 127) / 128;
    const int rows_pb = wgrad_rows_per_block(n, 1);
    const int nb = (int)((n + rows_pb - 1) / rows_pb);
    if ((z_pitch & 3) == 0 && (cout & 3) == 0 && (reinterpret_cast<uintptr_t>(d_z) & 15) == 0)
        hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3(nb, co_tiles), dim3(256), 0, s, n, d_z, z_pitch, cout, colpart, rows_pb);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb, co_tiles), dim3(256), 0, s, n, d_z, z_pitch, cout, colpart, rows_pb);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(co_tiles), dim3(1024), 0, s, colpart, nb, cout, d_out, acc);
    HIMO_LAUNCH_CHECK("colsum kernels");
    return HIMO_OK;
}

static size_t wgrad_ws(int64_t n_rows, int cin, int cout) {
    const size_t nb = (size_t)((n_rows + kWgRows - 1) / kWgRows) + 1;
    const size_t tiles = (size_t)((cin + 127) / 128) * ((cout + 127) / 128), ctiles = (size_t)(cout + 127) / 128;
    return tiles * nb * 128 * 128 * 4 + ctiles * nb * 128 * 4 + 64;
}

// dW = X^T dZ and db = column sums of dZ for a THIN input (cin <= 4: the head's Linear(3, 64) on the point offsets).  As a matrix-core
// product the 3-row operand filled 3 of 128 tile rows (83 us per 120k points); here it is what it is, a column reduction of dZ with cin + 1
// weights per row: block = 4 row groups x 64 columns, partials [block][cin + 1][cout], fixed-order reductions.
namespace himo {
__global__ __launch_bounds__(256) void wgrad_thin_partial_kernel(int64_t n, const float* __restrict__ X, int x_pitch, int cin,
                                                                 const float* __restrict__ dZ, int z_pitch, int cout,
                                                                 float* __restrict__ partial, int rows_pb) {
    __shared__ float sh[4][5][64];
    const int c = (int)blockIdx.y * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_pb;
    const int64_t r1 = r0 + rows_pb < n ? r0 + rows_pb : n;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int kk[4];                                             // rows >= cin re-read row 0 (their sums are never written): no conditional loads,
#pragma unroll                                             // which the compiler would serialise one by one
    for (int k = 0; k < 4; ++k) kk[k] = k < cin ? k : 0;
    const int cc = c < cout ? c : cout - 1;
#pragma unroll 4
    for (int64_t r = r0 + grp; r < r1; r += 4) {
        const float g = dZ[r * z_pitch + cc];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += X[r * x_pitch + kk[k]] * g;
        a[4] += g;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) sh[grp][k][threadIdx.x & 63] = a[k];
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * 64; e += 256) {
        const int k = e >> 6, cc = e & 63;
        const int col = (int)blockIdx.y * 64 + cc;
        if (col < cout) partial[((int64_t)blockIdx.x * 5 + k) * cout + col] = (sh[0][k][cc] + sh[1][k][cc]) + (sh[2][k][cc] + sh[3][k][cc]);
    }
}
// element e of [5][cout] (rows 0 .. cin - 1: dW, row 4: db): four thread groups over every fourth block partial, four chains each
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float* __restrict__ partial, int nb, int cin, int cout,
                                                                float* __restrict__ dW, float* __restrict__ db, int accumulate) {
    __shared__ float sh[4][64];
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = (int)blockIdx.x * 64 + o;
    const bool ok = e < 5 * cout;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        const float* p = partial + e;
        const int64_t st = 5ll * cout;
        int b = grp;
        for (; b + 12 < nb; b += 16) { s0 += p[b * st]; s1 += p[(b + 4) * st]; s2 += p[(b + 8) * st]; s3 += p[(b + 12) * st]; }
        for (; b < nb; b += 4) s0 += p[b * st];
    }
    sh[grp][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && ok) {
        const float t = (sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o]);
        const int k = e / cout, col = e - k * cout;
        if (k < cin) dW[k * cout + col] = accumulate ? dW[k * cout + col] + t : t;
        else if (k == 4 && db) db[col] = accumulate ? db[col] + t : t;
    }
}
}  // namespace himo

extern "C" size_t himo_wgrad_workspace_bytes(int64_t n_rows) { return wgrad_ws(n_rows, 128, 128); }
extern "C" size_t himo_wgrad_workspace_bytes_ex(int64_t n_rows, int cin, int cout) { return wgrad_ws(n_rows, cin, cout); }

// flags bit 0: accumulate into dW / db instead of overwriting; bit 1 (2): split-bf16 operands (wgrad_partial_split_kernel)
extern "C" int himo_linear_wgrad_ex(int64_t n, const float* d_x, int x_pitch, int cin, const float* d_dz, int z_pitch, int cout,
                                    float* d_dw, float* d_db, unsigned flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n < 1 || cin < 1 || cout < 1 || !d_x || !d_dz || !d_dw || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < wgrad_ws(n, cin, cout) || !aligned16(d_workspace)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (cin <= 4) {                                        // thin input: a column reduction, not a matrix product (wgrad_thin_*)
        const int64_t want = (n + 63) / 64;
        const int nb = (int)(want < 512 ? want : 512);
        const int rows_pb = (int)((n + nb - 1) / nb);
        if ((size_t)nb * 5 * cout * 4 <= workspace_bytes) {
            float* partial = reinterpret_cast<float*>(d_workspace);
            ProfScope ps("wgrad_partial_kernel", s);
            hipLaunchKernelGGL(wgrad_thin_partial_kernel, dim3(nb, (cout + 63) / 64), dim3(256), 0, s, n, d_x, x_pitch, cin, d_dz, z_pitch, cout,
                               partial, rows_pb);
            hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((5 * cout + 63) / 64), dim3(256), 0, s, partial, nb, cin, cout, d_dw, d_db,
                               (flags & 1u) ? 1 : 0);
            HIMO_LAUNCH_CHECK("wgrad_thin kernels");
            return HIMO_OK;
        }
    }
    const int ci_tiles = (cin + 127) / 128, co_tiles = (cout + 127) / 128;
    // rows per block: 256 for short inputs, longer runs when that would mean > ~512 blocks (the partial tiles, 64 KB
    // each, are written and read back: fewer, longer blocks keep that traffic below the operand traffic)
    const int rows_pb = wgrad_rows_per_block(n, ci_tiles * co_tiles);
    const int nb = (int)((n + rows_pb - 1) / rows_pb);
    float* partial = reinterpret_cast<float*>(d_workspace);
    float* colpart = partial + (size_t)ci_tiles * co_tiles * (nb + 1) * 128 * 128;
    const int acc = (flags & 1u) ? 1 : 0;
    {
        ProfScope ps("wgrad_partial_kernel", s);
        const bool vec = !(x_pitch & 3) && !(z_pitch & 3) && !(cin & 3) && !(cout & 3) &&
                         !((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_dz)) & 15);
        if ((flags & 2u) && cin > 128 && cin <= 192 && (cout == 128 || cout == 256) && n >= 256 * 64) {
            // every operand row read once: one block of eight waves per run of rows, one block per CU (wgrad_full_split_kernel)
            int rows_full = (int)(((n + 255) / 256 + 31) / 32 * 32);
            if (rows_full < kWgRows) rows_full = kWgRows;      // (the workspace is sized for runs of at least kWgRows rows)
            const int nb_full = (int)((n + rows_full - 1) / rows_full);
            float* colpart_full = partial + (size_t)ci_tiles * co_tiles * (nb_full + 1) * 128 * 128;
#define HIMO_WG_FULL(NBV, ST) hipLaunchKernelGGL((wgrad_full_split_kernel<NBV, ST>), dim3(nb_full), dim3(512), 0, s, n, d_x, x_pitch, cin, d_dz, z_pitch, \
                                                 partial, rows_full, d_db ? colpart_full : nullptr, nb_full)
            if (cout == 256) HIMO_WG_FULL(2, 32); else HIMO_WG_FULL(1, 32);      // (64-row stages: the same time, 129 instead of 72 KB of LDS)
#undef HIMO_WG_FULL
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((cin * cout + 63) / 64), dim3(256), 0, s, partial, nb_full, cin, cout, d_dw, acc);
            if (d_db) hipLaunchKernelGGL(colsum_reduce_kernel, dim3(co_tiles), dim3(1024), 0, s, colpart_full, nb_full, cout, d_db, acc);
            HIMO_LAUNCH_CHECK("wgrad_full_split kernels");
            return HIMO_OK;
        }
        if (flags & 2u)
            hipLaunchKernelGGL(wgrad_partial_split_kernel, dim3((unsigned)((nb + 7) / 8 * 8 * ci_tiles * co_tiles)), dim3(256), 0, s, n, d_x, x_pitch,
                               cin, d_dz, z_pitch, cout, partial, rows_pb, d_db ? colpart : nullptr, nb);
        else if (vec)
            hipLaunchKernelGGL(wgrad_partial_lds_kernel, dim3(nb, ci_tiles * co_tiles), dim3(256), 0, s, n, d_x, x_pitch, cin, d_dz, z_pitch,
                               cout, partial, rows_pb);
        else
            hipLaunchKernelGGL(wgrad_partial_kernel, dim3(nb, ci_tiles * co_tiles), dim3(256), 0, s, n, d_x, x_pitch, cin, d_dz, z_pitch, cout,
                               partial, rows_pb);
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((cin * cout + 63) / 64), dim3(256), 0, s, partial, nb, cin, cout, d_dw, acc);
    HIMO_LAUNCH_CHECK("wgrad kernels");
    if (d_db && (flags & 2u)) {                            // the split kernel left the bias partials behind the weight partials
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3(co_tiles), dim3(1024), 0, s, colpart, nb, cout, d_db, acc);
        HIMO_LAUNCH_CHECK("colsum_reduce_kernel");
        return HIMO_OK;
    }
    if (d_db) return colsum_launch(n, d_dz, z_pitch, cout, d_db, acc, colpart, s);
    return HIMO_OK;
}

extern "C" int himo_linear_wgrad(int64_t n, const float* d_x, int x_pitch, int cin, const float* d_dz, int z_pitch, int cout,
                                 float* d_dw, float* d_db, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (cin > 128 || cout > 128) return HIMO_ERR_INVALID_ARGUMENT;
    return himo_linear_wgrad_ex(n, d_x, x_pitch, cin, d_dz, z_pitch, cout, d_dw, d_db, 0u, d_workspace, workspace_bytes, stream);
}

// column sums of a pitched [n][cout] matrix (bias gradients); workspace as for himo_linear_wgrad_ex(n, 1, cout)
extern "C" int himo_colsum(int64_t n, const float* d_z, int z_pitch, int cout, float* d_out, unsigned flags, void* d_workspace,
                           size_t workspace_bytes, void* stream) {
    if (n < 1 || cout < 1 || z_pitch < cout || !d_z || !d_out || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    const int nb = (int)((n + kWgRows - 1) / kWgRows), co_tiles = (cout + 127) / 128;
    if (workspace_bytes < (size_t)co_tiles * nb * 128 * 4) return HIMO_ERR_WORKSPACE;
    return colsum_launch(n, d_z, z_pitch, cout, d_out, (flags & 1u) ? 1 : 0, reinterpret_cast<float*>(d_workspace), (hipStream_t)stream);
}

extern "C" int himo_transpose(const float* d_w, int rows, int cols, float* d_wt, void* stream) {
    if (!d_w || !d_wt || rows < 1 || cols < 1) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(transpose_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_w, rows, cols, d_wt);
    HIMO_LAUNCH_CHECK("transpose_kernel");
    return HIMO_OK;
}

extern "C" int himo_adam_step(int64_t n, float* d_param, const float* d_grad, float* d_m, float* d_v, float lr, float beta1,
                              float beta2, float eps, int step, void* stream) {
    if (n < 0 || step < 1) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_param || !d_grad || !d_m || !d_v) return HIMO_ERR_INVALID_ARGUMENT;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d_param, d_grad, d_m,
                       d_v, lr, beta1, beta2, eps, bc1, sqrtf(bc2));
    HIMO_LAUNCH_CHECK("adam_kernel");
    return HIMO_OK;
}

extern "C" size_t himo_chamfer_trunc_workspace_bytes(int n0, int n1) {
    return round_up(((size_t)(n0 + 255) / 256 + (size_t)(n1 + 255) / 256 + 2) * 8, 16) + (size_t)(n0 > 0 ? n0 : 1) * 24 + 64;
}

extern "C" int himo_chamfer_trunc(int n0, int n1, const float* d_moved, const float* d_pc1, const float* d_dist_a,
                                  const int32_t* d_idx_a, const float* d_dist_b, const int32_t* d_idx_b, float trunc_dist,
                                  double* d_loss, float* d_grad_moved, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n0 < 0 || n1 < 0 || !d_loss || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    if (n0 > 0 && (!d_moved || !d_grad_moved)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n0 > 0 && n1 > 0 && (!d_pc1 || !d_dist_a || !d_idx_a || !d_dist_b || !d_idx_b)) return HIMO_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < himo_chamfer_trunc_workspace_bytes(n0, n1)) return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int b0 = (n0 + 255) / 256, b1 = (n1 + 255) / 256;
    ChamferArgs a{n0, n1, d_moved, d_pc1, d_dist_a, d_idx_a, d_dist_b, d_idx_b, trunc_dist * trunc_dist, d_grad_moved,
                  reinterpret_cast<double*>(d_workspace),
                  reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(d_workspace) + round_up(((size_t)b0 + b1 + 2) * 8, 16))};
    if (n0 > 0) HIMO_HIP(hipMemsetAsync(a.scat, 0, (size_t)n0 * 24, s));
    if (b1) hipLaunchKernelGGL(chamfer_trunc_b_kernel, dim3(b1), dim3(256), 0, s, a, b0);
    if (b0) hipLaunchKernelGGL(chamfer_trunc_a_kernel, dim3(b0), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, a.partial, b0 + b1, d_loss);
    HIMO_LAUNCH_CHECK("chamfer_trunc kernels");
    return HIMO_OK;
}

extern "C" int himo_rigid_transform(int64_t n, const float* d_pts, int pc_stride, const float* d_transform, float* d_out,
                                    int out_pitch, void* stream) {
    if (n < 0 || pc_stride < 3 || out_pitch < 3) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_pts || !d_transform || !d_out) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(rigid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d_pts, pc_stride,
                       d_transform, d_out, out_pitch);
    HIMO_LAUNCH_CHECK("rigid_kernel");
    return HIMO_OK;
}

extern "C" int himo_rows_add(int64_t n, int cols, const float* d_a, int a_pitch, const float* d_b, int b_pitch, float b_scale,
                             float* d_y, int y_pitch, int zero_tail, void* stream) {
    if (n < 0 || cols < 1 || a_pitch < cols || y_pitch < cols || (d_b && b_pitch < cols)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n == 0) return HIMO_OK;
    if (!d_a || !d_y) return HIMO_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(rows_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, cols, d_a, a_pitch,
                       d_b, b_pitch, b_scale, d_y, y_pitch, zero_tail);
    HIMO_LAUNCH_CHECK("rows_add_kernel");
    return HIMO_OK;
}
