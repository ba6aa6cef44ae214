// nngrid.h -- internal interface of the exact grid nearest-neighbour search (nngrid.hip) for the translation units that
// run several searches over the same point sets (sslloss.hip): bin each set ONCE, then run any number of (query set, searched
// set) jobs in one launch.  Not part of the C ABI (include/himo_amd.h declares himo_nn_grid, the one-job form).
#pragma once
#include "himo_common.h"

namespace himo {

struct NnGrid {
    float x0, y0, inv_cell, cell;
    int gw, gh;
};

// one point set binned on the grid; every pointer is carved out of the caller's workspace by nng_carve
struct NngSet {
    const float* pts;        // [n][3] float32
    int n;
    int searched;            // 1: other sets look for neighbours IN this one (keeps the column-major copy as well)
    int* offset;             // [cells + 1] first sorted row of cell cy * gw + cx (row-major cell order)
    int* offset_t;           // [cells + 1] the same for cell cx * gh + cy (column-major order; searched sets only)
    int* cursor;             // [cells] fill cursors (scratch)
    int* cursor_t;
    int* cell_id;            // [n]
    float4* sorted;          // [n] (x, y, z, original row as int bits) in row-major cell order
    float4* sorted_t;        // [n] the same rows in column-major cell order
};

struct NngJob {              // nearest neighbour of every point of set q among the points of set r
    int q, r;
    float* dist2;            // [n_q] squared distance (+inf when set r is empty), in the ORIGINAL row order of set q
    int* idx;                // [n_q] original row of the neighbour in set r (-1 when empty); may be NULL
};

constexpr int kNngMaxSets = 4, kNngMaxJobs = 4;

size_t nng_workspace_bytes(int n_sets, int64_t n_max, int cells);
// lays `n_sets` sets out in the workspace (the integer arrays first, contiguous: one memset clears them)
void nng_carve(void* workspace, NngSet* sets, int n_sets, const float* const* pts, const int* n, const int* searched, int cells);
int nng_build(const NngSet* sets, int n_sets, const NnGrid& g, hipStream_t s);
int nng_query(const NngSet* sets, int n_sets, const NngJob* jobs, int n_jobs, const NnGrid& g, hipStream_t s);

}  // namespace himo
