// Shared host-side helpers for the libhimo_amd.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/himo_amd.h"

namespace himo {

// last HIP error text for himo_last_hip_error(); one slot per thread
void set_hip_error(hipError_t e, const char* where);

inline int check_hip(hipError_t e, const char* where) {
    if (e == hipSuccess) return HIMO_OK;
    set_hip_error(e, where);
    return HIMO_ERR_HIP;
}

#define HIMO_HIP(call)                                                   \
    do {                                                                 \
        int _st = ::himo::check_hip((call), #call);                      \
        if (_st != HIMO_OK) return _st;                                  \
    } while (0)

#define HIMO_LAUNCH_CHECK(name)                                          \
    do {                                                                 \
        int _st = ::himo::check_hip(hipGetLastError(), name);            \
        if (_st != HIMO_OK) return _st;                                  \
    } while (0)

// Optional per-kernel timing with HIP events on the launch stream (himo_prof_* in the ABI).
// Off by default; when on, every ProfScope brackets one kernel launch with two events.
bool prof_enabled();
bool prof_wants(const char* name);             // enabled and the name passes himo_prof_filter
hipEvent_t prof_event();                       // from a recycled pool
void prof_push(const char* name, hipEvent_t a, hipEvent_t b);
struct ProfScope {
    const char* name; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(const char* n, hipStream_t st) : name(n), s(st), on(prof_wants(n)) {
        if (on) {
            a = prof_event(); b = prof_event();
            if (!a || !b) { on = false; return; }
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(b, s); prof_push(name, a, b); } }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// order-preserving map float -> uint32 so that an integer atomicMax is a float max
__host__ __device__ inline unsigned float_to_key(float v) {
    unsigned b = __builtin_bit_cast(unsigned, v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float key_to_float(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, b);
}

}  // namespace himo
