// plan.hip -- run a prepared list of network operators from ONE call, optionally as a captured hipGraph.
//
// Why: with the convolutions at ~80 us a launch, the ~45 launches of a forward pass cost more host time (Python ->
// ctypes -> descriptor marshalling, ~20 us each) than some of the kernels take, and the GPU idled ~25 % of a frame.
// The backbone's operator list is static -- fixed buffers, shapes and weights -- so the host side builds it once
// (himo_amd/seflow/model.py) and replays it: one call, and with HIMO_OPS_GRAPH one hipGraphLaunch.
#include "himo_common.h"
#include <map>
#include <mutex>

using namespace himo;

namespace {
struct GraphEntry { hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; int n = 0; uint64_t hash = 0; bool failed = false; };

// FNV-1a over the list's bytes: a cached graph is replayed only for the exact contents it was captured from (the key
// is the list's address, which a freed and re-allocated list can share)
uint64_t ops_hash(const himo_op* ops, int n) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(ops);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(himo_op) * (size_t)n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
std::mutex g_mu;
std::map<const void*, GraphEntry> g_graphs;
hipStream_t g_capture_stream = nullptr;

int run_plain(const himo_op* ops, int n, void* stream) {
    for (int i = 0; i < n; ++i) {
        int st;
        if (ops[i].kind == HIMO_OP_CONV) st = himo_conv2d(&ops[i].conv, stream);
        else if (ops[i].kind == HIMO_OP_UPSAMPLE2X)
            st = himo_upsample2x_batch_ex(ops[i].up_n > 1 ? ops[i].up_n : 1, ops[i].up_x, ops[i].up_x_batch_stride, ops[i].up_x_pitch, ops[i].up_h,
                                          ops[i].up_w, ops[i].up_c, ops[i].up_y, ops[i].up_y_batch_stride, ops[i].up_y_pitch,
                                          ops[i].up_out_split, stream);
        else st = HIMO_ERR_INVALID_ARGUMENT;
        if (st != HIMO_OK) return st;
    }
    return HIMO_OK;
}
}  // namespace

extern "C" int himo_run_ops(const himo_op* h_ops, int n_ops, unsigned flags, void* stream) {
    if (n_ops < 0 || (n_ops > 0 && !h_ops)) return HIMO_ERR_INVALID_ARGUMENT;
    if (n_ops == 0) return HIMO_OK;
    // graphs only when asked for and the per-kernel profiler is off (its events must be recorded on a live stream)
    if (!(flags & HIMO_OPS_GRAPH) || prof_enabled()) return run_plain(h_ops, n_ops, stream);
    std::lock_guard<std::mutex> lk(g_mu);
    GraphEntry& e = g_graphs[h_ops];
    if (e.failed) return run_plain(h_ops, n_ops, stream);
    const uint64_t hash = ops_hash(h_ops, n_ops);
    if (!e.exec || e.n != n_ops || e.hash != hash) {
        // capture on a private stream (the caller's may be the legacy default stream, which cannot capture)
        if (!g_capture_stream && hipStreamCreateWithFlags(&g_capture_stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError(); e.failed = true; return run_plain(h_ops, n_ops, stream);
        }
        if (e.exec) { (void)hipGraphExecDestroy(e.exec); (void)hipGraphDestroy(e.graph); e.exec = nullptr; e.graph = nullptr; }
        bool ok = hipStreamBeginCapture(g_capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        int st = HIMO_OK;
        if (ok) {
            st = run_plain(h_ops, n_ops, g_capture_stream);
            ok = hipStreamEndCapture(g_capture_stream, &e.graph) == hipSuccess && st == HIMO_OK && e.graph;
        }
        if (ok) ok = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (e.graph) { (void)hipGraphDestroy(e.graph); e.graph = nullptr; }
            e.exec = nullptr; e.failed = true;
            if (st != HIMO_OK) return st;
            return run_plain(h_ops, n_ops, stream);
        }
        e.n = n_ops;
        e.hash = hash;
    }
    HIMO_HIP(hipGraphLaunch(e.exec, (hipStream_t)stream));
    return HIMO_OK;
}

// forget the graph captured for this operator list (call before changing the list's contents or freeing it)
extern "C" void himo_ops_release(const himo_op* h_ops) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_graphs.find(h_ops);
    if (it == g_graphs.end()) return;
    if (it->second.exec) { (void)hipGraphExecDestroy(it->second.exec); (void)hipGraphDestroy(it->second.graph); }
    g_graphs.erase(it);
}
