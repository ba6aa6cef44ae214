// evalmetrics.hip -- stages a7/a8: per-instance refinement metrics (MPE + Chamfer) on gfx950.
//
// Reference being replaced: InstanceMetrics.step_eval, eval.py:64-114 (per category -> per instance:
// point count, mean speed, mean range, mean point error, Chamfer distance via two cKDTree
// build+query pairs) and ScoreMetrics.step, tools/test/score.py:223-321.  The bucket bookkeeping of
// eval.py:99-147 (a few dozen records per sweep) stays on the host, in himo_amd/eval.py.
//
// Pipeline for a ragged batch of sweeps (all on one stream):
//   1. select_count / scan / select_compact   ordered stream compaction of the points that are in
//                                             the evaluation mask AND belong to CAR / OTHER_VEHICLES
//   2. rank_frames_kernel  counting sort by (frame, class group, instance id), one block per frame:
//                        the frame's distinct labels in an LDS hash table, ranked among themselves,
//                        then the points in index order.  Instances become contiguous segments in np.unique
//                        order; each point learns its segment [start, start+len); segment heads
//                        emit one record.
//   3. payload_kernel    the float64 comp_dis chain for the GT flow and the estimate (the same
//                        arithmetic as compdis.hip) -> refined GT / estimated positions, per-point
//                        speed, range and point error, written in sorted order.
//   4. nn_kernel x2      exact 1-NN gt->est and est->gt inside each segment (nn.hip).
//   5. seg_reduce_kernel fixed-order tree sums per segment -> record means.
// Everything is deterministic: no floating-point atomics, fixed reduction trees.
#include "compdis_math.h"

namespace himo {

int nn_search_ranges(int64_t nq, int64_t nr, const void* q, const void* r, const int32_t* rbeg, const int32_t* rlen,
                     bool f64, void* dist2, int32_t* idx, hipStream_t s);

constexpr int kSelBlock = 1024;   // points per block in the selection passes (256 threads x 4)
constexpr int kSelThreads = 256;

enum EvalMode { kModeFlow = 0, kModeCompDis = 1, kModeRaw = 2, kModeScore = 3, kModeDirect = 4 };

struct EvalArgs {
    int n_frames;
    int64_t total;
    const int64_t* offsets;
    const unsigned* keys;
    const FrameXf* xf;
    const float* pc0;        // may be nullptr in score mode
    int pc_stride;
    const float* gt;         // gt flow (incl. ego motion) [T,3]; score mode: gt comp_dis
    const float* est;        // est flow / est comp_dis [T,3]; unused when raw
    const float* lidar_dt;   // [T]; score mode: gt_flow_norm (may be nullptr)
    const uint8_t* category;
    const int64_t* instance;
    const uint8_t* eval_mask;
    uint8_t lut[256];        // category -> class group (0 = not evaluated)
    double sensor_dt;
    int mode;
    bool direct_est_is_dis;  // kModeDirect: est is float32 comp_dis instead of float64 flow
    // workspace
    int* block_counts;       // [nblk + 1]  (exclusive scan in place)
    int* frame_counts;       // [n_frames + 1] -> compact offsets per frame after the scan
    unsigned long long* labels;   // [M] compact order
    int* orig;               // [M] original point row
    int* spos;               // [M] compact -> sorted position
    int* seg_start;          // [M] sorted order
    int* seg_len;            // [M]
    double* gt_ref;          // [M][3] sorted order
    double* est_ref;         // [M][3]
    double* vel;             // [M]
    double* dis;             // [M]
    double* err;             // [M]
    double* d12;             // [M] squared NN distance gt -> est
    double* d21;             // [M]
    int* rec_start;          // [max_records]
    himo_instance_record* records;
    int64_t max_records;
    int64_t* counts;         // [0] = M (selected points), [1] = number of records
};

__device__ inline bool selected(const EvalArgs& a, int64_t i) {
    return a.eval_mask[i] != 0 && a.lut[a.category[i]] != 0;
}

// ---- 1a. count selected points per 1024-point block and per frame ---------------------------------
__global__ __launch_bounds__(kSelThreads) void select_count_kernel(EvalArgs a) {
    const int64_t bstart = (int64_t)blockIdx.x * kSelBlock;
    const int64_t bend = bstart + kSelBlock < a.total ? bstart + kSelBlock : a.total;
    const int f0 = __builtin_amdgcn_readfirstlane(find_frame(a.offsets, a.n_frames, bstart));
    const bool uniform = a.offsets[f0 + 1] >= bend;
    int cnt = 0;
    int f = f0;
    for (int64_t i = bstart + threadIdx.x * 4; i < bstart + threadIdx.x * 4 + 4 && i < bend; ++i) {
        const bool sel = selected(a, i);
        cnt += sel;
        if (!uniform && sel) {
            while (i >= a.offsets[f + 1]) ++f;
            atomicAdd(&a.frame_counts[f], 1);          // integer atomics: deterministic totals
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    __shared__ int w[kSelThreads / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = w[0] + w[1] + w[2] + w[3];
        a.block_counts[blockIdx.x] = tot;
        if (uniform && tot) atomicAdd(&a.frame_counts[f0], tot);
    }
}

// ---- 1b. exclusive scans (one block) ---------------------------------------------------------------
__global__ __launch_bounds__(1024) void select_scan_kernel(int* block_counts, int nblk, int* frame_counts, int n_frames,
                                                           int64_t* counts) {
    __shared__ int part[1024];
    __shared__ int carry;
    for (int pass = 0; pass < 2; ++pass) {
        int* v = pass == 0 ? block_counts : frame_counts;
        const int n = pass == 0 ? nblk : n_frames;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int base = 0; base < n; base += 1024) {
            const int i = base + threadIdx.x;
            const int x = i < n ? v[i] : 0;
            part[threadIdx.x] = x;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {       // Hillis-Steele inclusive scan
                const int y = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
                __syncthreads();
                part[threadIdx.x] += y;
                __syncthreads();
            }
            const int incl = part[threadIdx.x], c = carry;
            if (i < n) v[i] = c + incl - x;
            __syncthreads();
            if (threadIdx.x == 1023) carry = c + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            v[n] = carry;
            if (pass == 0) counts[0] = carry;
        }
        __syncthreads();
    }
}

// ---- 1c. ordered compaction: label + original row ---------------------------------------------------
__global__ __launch_bounds__(kSelThreads) void select_compact_kernel(EvalArgs a) {
    const int64_t bstart = (int64_t)blockIdx.x * kSelBlock;
    const int64_t bend = bstart + kSelBlock < a.total ? bstart + kSelBlock : a.total;
    const int f0 = find_frame(a.offsets, a.n_frames, bstart);
    bool sel[4];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = bstart + threadIdx.x * 4 + k;
        sel[k] = i < bend && selected(a, i);
        cnt += sel[k];
    }
    // exclusive prefix of cnt over the block: wave scan + wave totals
    int incl = cnt;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    __shared__ int wtot[kSelThreads / 64];
    if (lane == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int before = a.block_counts[blockIdx.x] + incl - cnt;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wtot[w];
    int f = f0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!sel[k]) continue;
        const int64_t i = bstart + threadIdx.x * 4 + k;
        while (i >= a.offsets[f + 1]) ++f;
        const unsigned long long lab = ((unsigned long long)f << 34) | ((unsigned long long)a.lut[a.category[i]] << 32) |
                                       (unsigned long long)(uint32_t)a.instance[i];
        a.labels[before] = lab;
        a.orig[before] = (int)i;
        ++before;
    }
}

// ---- 2. counting sort position of every selected point inside its frame -----------------------------
// One 1024-thread block per frame.  A sweep holds a few dozen to a few hundred distinct (group, instance) labels among its 10^3..10^4
// selected points, so the block (a) enters the frame's labels into an LDS hash table (2048 slots, 64-bit compare-and-swap) and counts
// them, (b) ranks the K distinct labels among themselves -- start of a label's segment = frame start + the counts of the smaller
// labels, (c) walks the frame's points tile by tile in index order: position = segment start + the label's points in earlier tiles (a
// running count per slot) + the label's points earlier in this tile.  All counts are integers: positions, segment bounds and counts
// are those of the all-pairs form this replaces (every point compared its label with every label of its frame: O(M_f^2), 84 % of the
// evaluator's device time at 10^4 selected points per sweep) -- which stays as the path of a frame whose distinct labels do not fit
// the table.  Record slots are handed out by an integer atomic in arrival order; the host sorts the records.
constexpr int kRankThreads = 1024;
constexpr int kRankSlots = 2048;                  // distinct labels of one frame the table holds (at most kRankSlots * 3 / 4 are admitted)
constexpr unsigned long long kRankEmpty = ~0ull;  // (no label: a frame index has 29 bits)

__device__ inline void rank_emit(const EvalArgs& a, int c, unsigned long long lab, int start, int before, int eq_total) {
    const int pos = start + before;
    a.spos[c] = pos;
    a.seg_start[pos] = start;
    a.seg_len[pos] = eq_total;
    if (before == 0) {   // segment head: one record per (frame, group, instance)
        const unsigned long long slot = atomicAdd((unsigned long long*)&a.counts[1], 1ull);
        if ((int64_t)slot < a.max_records) {
            himo_instance_record r;
            r.frame = (int)(lab >> 34);
            r.group = (int)((lab >> 32) & 3);
            r.instance = (int64_t)(lab & 0xffffffffull);
            r.num_pts = eq_total;
            r.vel = r.dis = r.mpe = r.cham = 0.0;
            a.records[slot] = r;
            a.rec_start[slot] = start;
        }
    }
}

__global__ __launch_bounds__(kRankThreads) void rank_frames_kernel(EvalArgs a) {
    __shared__ unsigned long long keys[kRankSlots];
    __shared__ int counts[kRankSlots], start[kRankSlots], running[kRankSlots], occupied[kRankSlots];
    __shared__ int tile_slot[kRankThreads];
    __shared__ int s_k, s_overflow;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int fbeg = a.frame_counts[f], fend = a.frame_counts[f + 1];
    if (fend <= fbeg) return;
    for (int j = tid; j < kRankSlots; j += kRankThreads) { keys[j] = kRankEmpty; counts[j] = 0; running[j] = 0; }
    if (tid == 0) { s_k = 0; s_overflow = 0; }
    __syncthreads();
    // (a) the frame's labels into the table; a point remembers its slot in spos (overwritten with its position below)
    for (int c = fbeg + tid; c < fend; c += kRankThreads) {
        const unsigned long long lab = a.labels[c];
        unsigned h = (unsigned)((lab * 0x9E3779B97F4A7C15ull) >> 53);          // 11 bits
        int slot = -1;
        for (int probe = 0; probe < kRankSlots; ++probe, h = (h + 1) & (kRankSlots - 1)) {
            const unsigned long long seen = keys[h];
            if (seen == lab) { slot = (int)h; break; }
            if (seen == kRankEmpty) {
                if (s_k >= kRankSlots * 3 / 4) break;                           // full enough: long probe chains would follow
                const unsigned long long prev = atomicCAS(&keys[h], kRankEmpty, lab);
                if (prev == kRankEmpty) { atomicAdd(&s_k, 1); slot = (int)h; break; }
                if (prev == lab) { slot = (int)h; break; }
            }
        }
        if (slot < 0) s_overflow = 1;
        else { atomicAdd(&counts[slot], 1); a.spos[c] = slot; }
    }
    __syncthreads();
    if (s_overflow) {
        // more distinct labels than the table admits: the all-pairs form for this frame (labels from L2)
        for (int c = fbeg + tid; c < fend; c += kRankThreads) {
            const unsigned long long lab = a.labels[c];
            int less = 0, eq_before = 0, eq_total = 0;
            for (int j = fbeg; j < fend; ++j) {
                const unsigned long long o = a.labels[j];
                less += o < lab;
                const bool eq = o == lab;
                eq_total += eq;
                eq_before += eq & (j < c);
            }
            rank_emit(a, c, lab, fbeg + less, eq_before, eq_total);
        }
        return;
    }
    // (b) the distinct labels among themselves
    if (tid == 0) s_k = 0;
    __syncthreads();
    for (int j = tid; j < kRankSlots; j += kRankThreads)
        if (keys[j] != kRankEmpty) occupied[atomicAdd(&s_k, 1)] = j;
    __syncthreads();
    const int K = s_k;
    for (int j = tid; j < kRankSlots; j += kRankThreads) {
        const unsigned long long mine = keys[j];
        if (mine == kRankEmpty) continue;
        int less = 0;
        for (int k = 0; k < K; ++k) {
            const int o = occupied[k];
            less += keys[o] < mine ? counts[o] : 0;
        }
        start[j] = fbeg + less;
    }
    __syncthreads();
    // (c) the points in index order, a tile of kRankThreads at a time
    for (int base = fbeg; base < fend; base += kRankThreads) {
        const int c = base + tid;
        const bool live = c < fend;
        const int slot = live ? a.spos[c] : -1;
        tile_slot[tid] = slot;
        __syncthreads();
        if (live) {
            int before = running[slot];
            for (int j = 0; j < tid; ++j) before += tile_slot[j] == slot;
            rank_emit(a, c, keys[slot], start[slot], before, counts[slot]);
        }
        __syncthreads();
        if (live) atomicAdd(&running[slot], 1);
        __syncthreads();
    }
}

// ---- 3. per-point float64 chain, written in sorted order ---------------------------------------------
__device__ inline double norm3(double x, double y, double z) { return sqrt((x * x + y * y) + z * z); }

__global__ __launch_bounds__(256) void payload_kernel(EvalArgs a, int M) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= M) return;
    const int64_t i = a.orig[c];
    const int pos = a.spos[c];
    double g[3], e[3], vel, dis, err;
    if (a.mode == kModeScore) {
        // tools/test/score.py:299-306: float32 arithmetic on the zip payloads
        float p[3] = {0.f, 0.f, 0.f};
        if (a.pc0) { p[0] = a.pc0[i * a.pc_stride]; p[1] = a.pc0[i * a.pc_stride + 1]; p[2] = a.pc0[i * a.pc_stride + 2]; }
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gd = a.gt[i * 3 + k], ed = a.est[i * 3 + k];
            g[k] = (double)(p[k] + gd);
            e[k] = (double)(p[k] + ed);
            const float df = gd - ed;
            d2 = d2 + df * df;
        }
        err = (double)sqrtf(d2);
        vel = a.lidar_dt ? (double)a.lidar_dt[i] : 0.0;
        dis = 0.0;
    } else if (a.mode == kModeDirect) {
        // step_eval's own arguments (eval.py:64): ego-motion-free float64 flows, dt0 already formed
        const float* prow = a.pc0 + i * (int64_t)a.pc_stride;
        const double* gt64 = reinterpret_cast<const double*>(a.gt);
        const float dt0 = a.lidar_dt[i];
        double gtf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double p = (double)prow[k];
            gtf[k] = gt64[i * 3 + k];
            g[k] = p + gtf[k] / a.sensor_dt * (double)dt0;                         // eval.py:72
            if (a.direct_est_is_dis) e[k] = (double)(prow[k] + a.est[i * 3 + k]);  // eval.py:70
            else e[k] = p + reinterpret_cast<const double*>(a.est)[i * 3 + k] / a.sensor_dt * (double)dt0;   // eval.py:68
        }
        vel = norm3(gtf[0], gtf[1], gtf[2]);
        float s = 0.f;
        for (int k = 0; k < a.pc_stride; ++k) s = s + prow[k] * prow[k];
        dis = (double)sqrtf(s);
        err = norm3(g[0] - e[0], g[1] - e[1], g[2] - e[2]);
    } else {
        const int f = (int)(a.labels[c] >> 34);
        const XfRegs x = load_xf(a.xf, a.keys, f);
        const float* prow = a.pc0 + i * (int64_t)a.pc_stride;
        const float pf32[3] = {prow[0], prow[1], prow[2]};
        const float dt0 = x.fmax - a.lidar_dt[i];                                  // eval.py:299
        double gtf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double p = (double)pf32[k];
            const double pf = (fma((double)pf32[2], x.R[k * 3 + 2], fma((double)pf32[1], x.R[k * 3 + 1], (double)pf32[0] * x.R[k * 3])) + x.t[k]) - p;
            gtf[k] = (double)a.gt[i * 3 + k] - pf;                                 // eval.py:286
            g[k] = p + gtf[k] / a.sensor_dt * (double)dt0;                         // eval.py:72
            if (a.mode == kModeCompDis) {
                e[k] = (double)(pf32[k] + a.est[i * 3 + k]);                       // eval.py:70 (float32 + float32)
            } else {
                const double ef = a.mode == kModeRaw ? 0.0 : (double)a.est[i * 3 + k] - pf;   // eval.py:302
                e[k] = p + ef / a.sensor_dt * (double)dt0;                         // eval.py:68
            }
        }
        vel = norm3(gtf[0], gtf[1], gtf[2]);                                       // eval.py:91 (per point)
        float s = 0.f;                                                             // eval.py:94: norm over ALL columns (float32)
        for (int k = 0; k < a.pc_stride; ++k) s = s + prow[k] * prow[k];
        dis = (double)sqrtf(s);
        err = norm3(g[0] - e[0], g[1] - e[1], g[2] - e[2]);                        // eval.py:95 (per point)
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.gt_ref[(int64_t)pos * 3 + k] = g[k]; a.est_ref[(int64_t)pos * 3 + k] = e[k]; }
    a.vel[pos] = vel; a.dis[pos] = dis; a.err[pos] = err;
}

// ---- 5. fixed-order segment means ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_reduce_kernel(EvalArgs a) {
    __shared__ double red[5][256];
    const int64_t K = a.counts[1] < a.max_records ? a.counts[1] : a.max_records;
    for (int64_t rec = blockIdx.x; rec < K; rec += gridDim.x) {
        const int start = a.rec_start[rec];
        const int len = (int)a.records[rec].num_pts;
        double s[5] = {0, 0, 0, 0, 0};
        for (int j = threadIdx.x; j < len; j += 256) {
            const int p = start + j;
            s[0] += a.vel[p]; s[1] += a.dis[p]; s[2] += a.err[p]; s[3] += sqrt(a.d12[p]); s[4] += sqrt(a.d21[p]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = s[k];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off)
#pragma unroll
                for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            himo_instance_record r = a.records[rec];
            const double n = (double)len;
            r.vel = red[0][0] / n / a.sensor_dt;                                   // eval.py:91
            r.dis = red[1][0] / n;
            r.mpe = red[2][0] / n;
            r.cham = (red[3][0] / n + red[4][0] / n) / 2.0;                        // eval.py:61
            a.records[rec] = r;
        }
    }
}

static size_t a16(size_t x) { return round_up(x, 16); }

struct EvalLayout { size_t prep, bc, fc, labels, orig, spos, sstart, slen, gt, est, vel, dis, err, d12, d21, recs, end; };

static EvalLayout eval_layout(int n_frames, int64_t total, int64_t max_records) {
    EvalLayout L;
    const size_t T = (size_t)total, nblk = (T + kSelBlock - 1) / kSelBlock;
    size_t o = 0;
    L.prep = o; o += a16(himo_compdis_workspace_bytes(n_frames));
    L.bc = o; o += a16((nblk + 1) * 4);
    L.fc = o; o += a16(((size_t)n_frames + 1) * 4);
    L.labels = o; o += a16(T * 8);
    L.orig = o; o += a16(T * 4);
    L.spos = o; o += a16(T * 4);
    L.sstart = o; o += a16(T * 4);
    L.slen = o; o += a16(T * 4);
    L.gt = o; o += a16(T * 24);
    L.est = o; o += a16(T * 24);
    L.vel = o; o += a16(T * 8);
    L.dis = o; o += a16(T * 8);
    L.err = o; o += a16(T * 8);
    L.d12 = o; o += a16(T * 8);
    L.d21 = o; o += a16(T * 8);
    L.recs = o; o += a16((size_t)max_records * 4);
    L.end = o;
    return L;
}

}  // namespace himo

using namespace himo;

extern "C" size_t himo_eval_workspace_bytes(int n_frames, int64_t total_points, int64_t max_records) {
    if (n_frames < 1) n_frames = 1;
    if (total_points < 0) total_points = 0;
    if (max_records < 0) max_records = 0;
    return eval_layout(n_frames, total_points, max_records).end + 64;
}

extern "C" int himo_eval_instances(int n_frames, int64_t total_points, const int64_t* d_offsets, const double* d_pose0,
                                   const double* d_pose1, const float* d_pc0, int pc_stride, const void* d_gt,
                                   const void* d_est, const float* d_lidar_dt, const uint8_t* d_category,
                                   const int64_t* d_instance, const uint8_t* d_eval_mask, const uint8_t* h_class_lut,
                                   double sensor_dt, int mode, unsigned flags, himo_instance_record* d_records,
                                   int64_t max_records, int64_t* d_counts, void* d_workspace, size_t workspace_bytes,
                                   void* stream) {
    if (n_frames < 1 || total_points < 0 || mode < 0 || mode > 4 || max_records < 1) return HIMO_ERR_INVALID_ARGUMENT;
    if (total_points > 0x7fffffff || n_frames >= (1 << 29)) return HIMO_ERR_UNSUPPORTED;
    if (!d_offsets || !h_class_lut || !d_records || !d_counts || !d_workspace) return HIMO_ERR_INVALID_ARGUMENT;
    if (!(sensor_dt != 0.0)) return HIMO_ERR_INVALID_ARGUMENT;
    const bool score = mode == kModeScore, direct = mode == kModeDirect;
    if (total_points > 0) {
        if (!d_gt || !d_category || !d_instance || !d_eval_mask) return HIMO_ERR_INVALID_ARGUMENT;
        if (mode != kModeRaw && !d_est) return HIMO_ERR_INVALID_ARGUMENT;
        if (!score && (!d_pc0 || !d_lidar_dt || pc_stride < 3)) return HIMO_ERR_INVALID_ARGUMENT;
        if (!score && !direct && !d_pose0) return HIMO_ERR_INVALID_ARGUMENT;
        if (!score && !direct && !d_pose1 && !(flags & HIMO_FLAG_POSE_IS_EGO)) return HIMO_ERR_INVALID_ARGUMENT;
        if (score && d_pc0 && pc_stride < 3) return HIMO_ERR_INVALID_ARGUMENT;
    }
    if (workspace_bytes < himo_eval_workspace_bytes(n_frames, total_points, max_records) || !aligned16(d_workspace))
        return HIMO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    HIMO_HIP(hipMemsetAsync(d_counts, 0, 2 * sizeof(int64_t), s));
    if (total_points == 0) return HIMO_OK;

    const EvalLayout L = eval_layout(n_frames, total_points, max_records);
    char* ws = reinterpret_cast<char*>(d_workspace);
    EvalArgs a{};
    a.n_frames = n_frames; a.total = total_points; a.offsets = d_offsets;
    a.pc0 = d_pc0; a.pc_stride = pc_stride; a.gt = (const float*)d_gt; a.est = (const float*)d_est; a.lidar_dt = d_lidar_dt;
    a.category = d_category; a.instance = d_instance; a.eval_mask = d_eval_mask;
    for (int i = 0; i < 256; ++i) a.lut[i] = h_class_lut[i] & 3;
    a.sensor_dt = sensor_dt; a.mode = mode; a.direct_est_is_dis = (flags & HIMO_EVAL_DIRECT_EST_IS_DIS) != 0;
    a.block_counts = (int*)(ws + L.bc); a.frame_counts = (int*)(ws + L.fc);
    a.labels = (unsigned long long*)(ws + L.labels); a.orig = (int*)(ws + L.orig); a.spos = (int*)(ws + L.spos);
    a.seg_start = (int*)(ws + L.sstart); a.seg_len = (int*)(ws + L.slen);
    a.gt_ref = (double*)(ws + L.gt); a.est_ref = (double*)(ws + L.est);
    a.vel = (double*)(ws + L.vel); a.dis = (double*)(ws + L.dis); a.err = (double*)(ws + L.err);
    a.d12 = (double*)(ws + L.d12); a.d21 = (double*)(ws + L.d21);
    a.rec_start = (int*)(ws + L.recs); a.records = d_records; a.max_records = max_records; a.counts = d_counts;

    if (!score && !direct) {   // per-frame max(lidar_dt) + ego transforms, shared with the comp_dis path
        int st = launch_frame_prep(n_frames, total_points, d_offsets, d_pose0, d_pose1, flags, d_lidar_dt, ws + L.prep, s);
        if (st != HIMO_OK) return st;
        WorkspaceLayout w = carve(ws + L.prep, n_frames);
        a.keys = w.keys; a.xf = w.xf;
    }
    const int nblk = (int)((total_points + kSelBlock - 1) / kSelBlock);
    HIMO_HIP(hipMemsetAsync(a.frame_counts, 0, ((size_t)n_frames + 1) * 4, s));
    { ProfScope ps("select_count_kernel", s); hipLaunchKernelGGL(select_count_kernel, dim3(nblk), dim3(kSelThreads), 0, s, a); }
    HIMO_LAUNCH_CHECK("select_count_kernel");
    { ProfScope ps("select_scan_kernel", s);
      hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, s, a.block_counts, nblk, a.frame_counts, n_frames, d_counts); }
    HIMO_LAUNCH_CHECK("select_scan_kernel");
    { ProfScope ps("select_compact_kernel", s); hipLaunchKernelGGL(select_compact_kernel, dim3(nblk), dim3(kSelThreads), 0, s, a); }
    HIMO_LAUNCH_CHECK("select_compact_kernel");

    int64_t h_counts[2] = {0, 0};   // the one host round trip: later grids are sized by the selected count
    HIMO_HIP(hipMemcpyAsync(h_counts, d_counts, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIMO_HIP(hipStreamSynchronize(s));
    const int M = (int)h_counts[0];
    if (M == 0) return HIMO_OK;

    const int mblk = (M + 255) / 256;
    { ProfScope ps("rank_frames_kernel", s); hipLaunchKernelGGL(rank_frames_kernel, dim3(n_frames), dim3(kRankThreads), 0, s, a); }
    HIMO_LAUNCH_CHECK("rank_frames_kernel");
    { ProfScope ps("payload_kernel", s); hipLaunchKernelGGL(payload_kernel, dim3(mblk), dim3(256), 0, s, a, M); }
    HIMO_LAUNCH_CHECK("payload_kernel");
    int st = nn_search_ranges(M, M, a.gt_ref, a.est_ref, a.seg_start, a.seg_len, true, a.d12, nullptr, s);   // eval.py:56-57
    if (st != HIMO_OK) return st;
    st = nn_search_ranges(M, M, a.est_ref, a.gt_ref, a.seg_start, a.seg_len, true, a.d21, nullptr, s);       // eval.py:58-59
    if (st != HIMO_OK) return st;
    { ProfScope ps("seg_reduce_kernel", s); hipLaunchKernelGGL(seg_reduce_kernel, dim3(1024), dim3(256), 0, s, a); }
    HIMO_LAUNCH_CHECK("seg_reduce_kernel");
    return HIMO_OK;
}
