// abi.hip -- version / status / error-text entry points of libhimo_amd.so.
#include "himo_common.h"
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace himo {
static thread_local char g_hip_error[256] = "";
void set_hip_error(hipError_t e, const char* where) {
    snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}

struct ProfRec { const char* name; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;
static char g_prof_filter[64] = "";            // when non-empty: only kernels whose name contains it are timed
static std::vector<hipEvent_t> g_event_pool;   // events are recycled: creating two per launch costs host time
bool prof_enabled() { return g_prof_on; }
bool prof_wants(const char* name) { return g_prof_on && (!g_prof_filter[0] || strstr(name, g_prof_filter) != nullptr); }
hipEvent_t prof_event() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}
void prof_push(const char* name, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({name, a, b});
}
}  // namespace himo

extern "C" void himo_prof_enable(int on) { himo::g_prof_on = on != 0; }

// restrict the per-kernel timing to kernels whose name contains `substr` (NULL or "" = every kernel): two event
// records per launch are not free, so a throughput measurement times only the kernel it reports on
extern "C" void himo_prof_filter(const char* substr) {
    std::lock_guard<std::mutex> lk(himo::g_prof_mu);
    snprintf(himo::g_prof_filter, sizeof(himo::g_prof_filter), "%s", substr ? substr : "");
}

extern "C" void himo_prof_reset(void) {
    std::lock_guard<std::mutex> lk(himo::g_prof_mu);
    for (auto& r : himo::g_prof) { himo::g_event_pool.push_back(r.a); himo::g_event_pool.push_back(r.b); }
    himo::g_prof.clear();
}

// "name count total_ms min_ms max_ms\n" per kernel; waits for the recorded launches to finish.
extern "C" size_t himo_prof_summary(char* buf, size_t cap) {
    struct Acc { long n = 0; double tot = 0, mn = 1e30, mx = 0; };
    std::map<std::string, Acc> acc;
    {
        std::lock_guard<std::mutex> lk(himo::g_prof_mu);
        for (auto& r : himo::g_prof) {
            float ms = 0.f;
            if (hipEventSynchronize(r.b) != hipSuccess) continue;
            if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
            Acc& a = acc[r.name];
            a.n++; a.tot += ms; if (ms < a.mn) a.mn = ms; if (ms > a.mx) a.mx = ms;
        }
    }
    std::string out;
    char line[256];
    for (auto& kv : acc) {
        snprintf(line, sizeof(line), "%s %ld %.6f %.6f %.6f\n", kv.first.c_str(), kv.second.n, kv.second.tot, kv.second.mn, kv.second.mx);
        out += line;
    }
    if (buf && cap) { size_t n = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), n); buf[n] = 0; }
    return out.size() + 1;
}

extern "C" int himo_abi_version(void) { return HIMO_ABI_VERSION; }

// sizeof of the structs that cross the boundary by address, so a binding can check its mirror before the first call
extern "C" size_t himo_abi_sizeof(const char* struct_name) {
    if (!struct_name) return 0;
    if (!strcmp(struct_name, "himo_conv_desc")) return sizeof(himo_conv_desc);
    if (!strcmp(struct_name, "himo_head_sample")) return sizeof(himo_head_sample);
    if (!strcmp(struct_name, "himo_op")) return sizeof(himo_op);
    if (!strcmp(struct_name, "himo_sweep")) return sizeof(himo_sweep);
    if (!strcmp(struct_name, "himo_instance_record")) return sizeof(himo_instance_record);
    return 0;
}

extern "C" const char* himo_last_hip_error(void) { return himo::g_hip_error; }

extern "C" const char* himo_status_string(int status) {
    switch (status) {
        case HIMO_OK: return "ok";
        case HIMO_ERR_INVALID_ARGUMENT: return "invalid argument";
        case HIMO_ERR_EMPTY_FRAME: return "empty frame: max() arg is an empty sequence";
        case HIMO_ERR_WORKSPACE: return "workspace too small or misaligned";
        case HIMO_ERR_SINGULAR_POSE: return "Singular matrix";
        case HIMO_ERR_HIP: return "HIP runtime error";
        case HIMO_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}
