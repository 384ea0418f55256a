// Developer tool (hazard bisect, r04): "aggressor" kernels -- waves that share the SIMDs with a kernel under test and issue ONE class of
// instruction in a tight loop for a given time.  The latent-grid renderers are bit-identical from launch to launch with one wave per
// SIMD and not with two (profiles/r04/nondeterminism_r04.md); with the kernel under test held at one wave per SIMD and these as its
// neighbours, the class of the neighbour's instruction that disturbs it can be read off.  Each wave keeps ~72 VGPRs live so that two
// aggressor waves + ONE wave of the kernel under test (224 registers) fill a SIMD's 512 and a second victim wave does not fit.
// build: __graft_entry__.build() (fv-srn_amd/csrc/hipcc_fixed.sh + link -> tools/dev/bin/libaggressor.so)
#include <hip/hip_runtime.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(256) aggressor_kernel(long long ticks, unsigned* sink, const float* gmem) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = float(i);
    __syncthreads();
    unsigned pad[56];
#pragma unroll
    for (int i = 0; i < 56; ++i) { pad[i] = threadIdx.x + i; asm volatile("" : "+v"(pad[i])); }
    const long long t0 = wall_clock64();  // 100 MHz
    floatx16 acc = {0};
    floatx4 acc4 = {0, 0, 0, 0};
    half8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    float x = float(threadIdx.x) * 0.001f, y = 1.0f + x, z = 0.5f;
    float2 p = {x, y}, q = {y, z};
    unsigned u = threadIdx.x, v = u ^ 0x55u;
    unsigned long long m = 0;
    const float* lp = lds + (threadIdx.x & 63) * 4;
    while (wall_clock64() - t0 < ticks) {
        if constexpr (KIND == 0) { __builtin_amdgcn_s_sleep(8); }
        if constexpr (KIND == 1) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));) }
        if constexpr (KIND == 2) { REP8(asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b));) }
        if constexpr (KIND == 3) { REP8(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v));) }
        if constexpr (KIND == 4) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));) }
        if constexpr (KIND == 5) { REP8(asm volatile("v_cos_f32 %0, %0" : "+v"(x));) }
        if constexpr (KIND == 6) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));) }
        if constexpr (KIND == 7) { REP8(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(acc4) : "v"(unsigned(size_t(lp))));) }
        if constexpr (KIND == 8) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if constexpr (KIND == 9) { REP8(asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x) : "v"(u), "v"(v));) }
        if constexpr (KIND == 10) { REP8(asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(acc4) : "v"(gmem + (threadIdx.x & 63) * 4));) }
        if constexpr (KIND == 11) { REP8(asm volatile("v_cmp_lt_f32 %0, %1, %2\n s_and_b64 %0, %0, exec" : "=s"(m) : "v"(x), "v"(y));) }
        if constexpr (KIND == 12) { REP8(asm volatile("v_readlane_b32 %0, %1, 5\n s_nop 3\n v_writelane_b32 %1, %0, 7" : "=s"(v), "+v"(u));) }
        if constexpr (KIND == 13) { REP8(asm volatile("v_floor_f32 %0, %1\n v_med3_f32 %0, %0, %1, %2\n v_cvt_u32_f32 %3, %0\n v_mul_u32_u24 %3, %3, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(u));) }
        if constexpr (KIND == 14) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));) }
        if constexpr (KIND == 15) { REP8(asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(v));) }
    }
    unsigned r = u ^ v ^ __float_as_uint(x + y + z + p.x + p.y + acc[0] + acc[5] + acc4[0] + acc4[3]) ^ unsigned(m);
#pragma unroll
    for (int i = 0; i < 56; ++i) { asm volatile("" : "+v"(pad[i])); r ^= pad[i]; }
    if (r == 0xdeadbeefu) *sink = r;
}

extern "C" int aggressor_kinds() { return 16; }
extern "C" const char* aggressor_name(int kind) {
    static const char* n[] = {"s_sleep", "mfma 32x32x16 (C = D)", "mfma 16x16x32", "v_permlane32_swap", "v_pk_mul_f32", "v_cos_f32", "v_cvt_pk_f16_f32",
                              "ds_read_b128", "v_fma_f32", "v_dot2_f32_f16", "global_load_dwordx4", "v_cmp + s_and", "v_readlane / v_writelane",
                              "floor / med3 / cvt / mul24", "mfma 32x32x16 (C = 0)", "v_mov_b32"};
    return kind >= 0 && kind < 16 ? n[kind] : "?";
}
extern "C" int aggressor(int kind, int blocks, int microseconds, void* stream) {
    static unsigned* sink = nullptr;
    static float* gmem = nullptr;
    if (!sink && (hipMalloc(&sink, 4) != hipSuccess || hipMalloc(&gmem, 4096) != hipSuccess || hipMemset(gmem, 0, 4096) != hipSuccess)) return -1;
    const long long ticks = (long long)microseconds * 100;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define K(i) case i: hipLaunchKernelGGL(aggressor_kernel<i>, dim3(blocks), dim3(256), 0, s, ticks, sink, gmem); break;
    switch (kind) { K(0) K(1) K(2) K(3) K(4) K(5) K(6) K(7) K(8) K(9) K(10) K(11) K(12) K(13) K(14) K(15) default: return -3; }
#undef K
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
