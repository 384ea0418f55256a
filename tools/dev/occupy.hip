// Developer tool: a kernel that holds `blocks` workgroups of `threads` threads for `microseconds` on a stream -- a stand-in for the
// collective of another rank's frame when a rank's frame pipeline is timed on one GPU (tools/stripe_efficiency.py,
// FVSRN_STRIPE_EMULATE_GATHER=blocks,threads,microseconds).  build: __graft_entry__.build() (through fv-srn_amd/csrc/hipcc_fixed.sh -> tools/dev/bin/liboccupy.so)
#include <hip/hip_runtime.h>

__global__ void occupy_kernel(long long ticks, unsigned* sink) {
    const long long t0 = wall_clock64();  // 100 MHz
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) *sink = x;
}

extern "C" int occupy(int blocks, int threads, int microseconds, void* stream) {
    static unsigned* sink = nullptr;
    if (!sink && hipMalloc(&sink, 4) != hipSuccess) return -1;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), 0, static_cast<hipStream_t>(stream), (long long)microseconds * 100, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// poison(): fills (nearly) the whole VGPR file of every SIMD with one bit pattern, so that a later kernel that reads a register it never
// wrote shows it (tools/dev/determinism.py --poison)
__global__ void __launch_bounds__(256, 2) poison_kernel(unsigned pattern, unsigned* sink) {
    unsigned r[240];
#pragma unroll
    for (int i = 0; i < 240; ++i) { r[i] = pattern; asm volatile("" : "+v"(r[i])); }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 240; ++i) { asm volatile("" : "+v"(r[i])); x ^= r[i] + i; }
    if (x == 0x12345678u) *sink = x;
}

extern "C" int poison(unsigned pattern, void* stream) {
    static unsigned* sink = nullptr;
    if (!sink && hipMalloc(&sink, 4) != hipSuccess) return -1;
    hipLaunchKernelGGL(poison_kernel, dim3(256 * 8), dim3(256), 0, static_cast<hipStream_t>(stream), pattern, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
