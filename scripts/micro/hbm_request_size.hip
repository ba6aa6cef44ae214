// Does the granularity of a read request matter to HBM throughput on MI355X?  Every variant reads the same 2 GiB buffer exactly once with
// global_load_dwordx4 (16 bytes per lane) and differs only in how a wave's 64 lanes are spread:
//   contiguous   lanes cover 1 KB contiguous                                             (comp_dis-like streaming)
//   line128      groups of 8 lanes cover one whole 128-byte line, the groups 256 bytes apart (every other line; a second pass takes the rest)
//   half64       groups of 4 lanes cover 64 bytes = HALF a 128-byte line, groups 128 bytes apart; the other halves are read by a later pass
//                (what a 16-channel slab of a 32-channel split-format pixel is: csrc/convsg.hip stride-2 / enc1.0 staging)
//   quarter32    groups of 2 lanes cover 32 bytes, four passes
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_req scripts/micro/hbm_request_size.hip && /tmp/hbm_req
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PIECE>      // bytes a lane group covers contiguously: 1024 (contiguous), 128, 64, 32
__global__ __launch_bounds__(256) void read_kernel(const unsigned char* __restrict__ x, size_t bytes, unsigned* sink) {
    constexpr int LPG = PIECE / 16;                  // lanes per group
    constexpr int PASSES = PIECE >= 1024 ? 1 : 128 * (PIECE < 128 ? 1 : 2) / PIECE;   // 128 -> 2 (every other line), 64 -> 2, 32 -> 4
    constexpr int STRIDE = PIECE >= 1024 ? 1024 : (PIECE == 128 ? 256 : 128);          // distance between the groups of one instruction
    const size_t span = (size_t)(64 / LPG) * STRIDE; // address range one wave instruction touches
    const size_t waves = bytes / span / 1;           // wave-instructions per pass ... each covers `span` of addresses with density PIECE/STRIDE
    const size_t gw = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64, nw = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63, grp = lane / LPG, sub = lane % LPG;
    unsigned acc = 0;
    for (int pass = 0; pass < PASSES; ++pass) {
        const size_t pass_off = PIECE >= 1024 ? 0 : (PIECE == 128 ? (size_t)pass * 128 : (size_t)pass * PIECE);
        for (size_t w = gw; w < waves; w += nw) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + w * span + (size_t)grp * STRIDE + pass_off + sub * 16);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int PIECE>
static void run(const char* name, const unsigned char* x, size_t bytes, unsigned* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(read_kernel<PIECE>, dim3(256 * 16), dim3(256), 0, 0, x, bytes, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    unsigned char* x; unsigned* sink;
    hipMalloc(&x, bytes); hipMalloc(&sink, 4);
    hipMemset(x, 1, bytes);
    run<1024>("contiguous", x, bytes, sink);
    run<128>("line128", x, bytes, sink);
    run<64>("half64", x, bytes, sink);
    run<32>("quarter32", x, bytes, sink);
    return 0;
}
