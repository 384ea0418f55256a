// Microbenchmark: can one SIMD of gfx950 run MFMA (32x32x16 f16) from one wave and VALU from another at the same time?
// Each workgroup = 8 waves (2 per SIMD): even waves loop over MFMAs, odd waves loop over VALU ops (or idle / same).
// build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// ROLE_A / ROLE_B: 0 idle, 1 MFMA (VGPR accumulators, chains of 4), 2 VALU fma, 3 VALU cvt_pk, 4 MFMA AGPR accumulators,
// 5 v_cos
template <int ROLE>
__device__ __forceinline__ float work(int iters, int lane) {
    float r = 0;
    if constexpr (ROLE == 1) {
        half8_t a, b;
        for (int j = 0; j < 8; ++j) { a[j] = _Float16(0.01f * (lane + j)); b[j] = _Float16(0.02f * j); }
        floatx16 c0 = {0}, c1 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
        }
        for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
    } else if constexpr (ROLE == 4) {
        half8_t a, b;
        for (int j = 0; j < 8; ++j) { a[j] = _Float16(0.01f * (lane + j)); b[j] = _Float16(0.02f * j); }
        floatx16 c0 = {0}, c1 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
            }
        }
        for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
    } else if constexpr (ROLE == 2) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = float(lane + j);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], 1.0001f, 0.5f);   // 128 VALU per iteration
        }
        for (int j = 0; j < 8; ++j) r += x[j];
    } else if constexpr (ROLE == 3) {
        float x[16];
        for (int j = 0; j < 16; ++j) x[j] = float(lane + j);
        unsigned acc = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float2_t v = {x[2 * j], x[2 * j + 1]};
                    half2_t hh = __builtin_convertvector(v, half2_t);
                    unsigned u = __builtin_bit_cast(unsigned, hh);
                    asm volatile("" : "+v"(u));
                    acc ^= u;  // xor: 1 more VALU  => 16 VALU per j-loop... (8 cvt + 8 xor)
                }
            }
        }
        r = float(acc);
    } else if constexpr (ROLE == 5) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = 0.001f * float(lane + j);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_cosf(x[j]);   // 128 transcendental per iteration
        }
        for (int j = 0; j < 8; ++j) r += x[j];
    }
    return r;
}

template <int ROLE_A, int ROLE_B>
__global__ void __launch_bounds__(512) bench(float* out, int iters, long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // waves 0..3 land on SIMD 0..3, waves 4..7 again on SIMD 0..3 (round-robin placement)
    const long long c0 = clock64(), w0 = wall_clock64();
    float r;
    if (wave < 4) r = work<ROLE_A>(iters, lane);
    else r = work<ROLE_B>(iters, lane);
    const long long c1 = clock64(), w1 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) { clk[2 * wave] = c1 - c0; clk[2 * wave + 1] = w1 - w0; }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int ROLE_A, int ROLE_B>
void run(const char* name, float* out, long long* clk) {
    const int iters = 4000;
    bench<ROLE_A, ROLE_B><<<256, 512>>>(out, 10, clk);
    hipDeviceSynchronize();
    bench<ROLE_A, ROLE_B><<<256, 512>>>(out, iters, clk);
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = double(h[0]) / (double(h[1]) * 10.0);
    printf("%-44s wave A: %7.0f cycles/iter   wave B: %7.0f cycles/iter   (%.2f GHz)\n", name, double(h[0]) / iters, double(h[8]) / iters, ghz);
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&clk, 16 * 8);
    // per iteration: MFMA role = 16 MFMAs (512 cycles ideal); fma role = 128 VALU (512 cycles ideal); cos role = 128 trans
    run<1, 0>("A: 16 MFMA (VGPR acc)      B: idle", out, clk);
    run<4, 0>("A: 16 MFMA (AGPR acc)      B: idle", out, clk);
    run<2, 0>("A: 128 v_fma               B: idle", out, clk);
    run<3, 0>("A: 128 cvt_pk+128 xor      B: idle", out, clk);
    run<5, 0>("A: 128 v_cos               B: idle", out, clk);
    run<1, 1>("A: 16 MFMA (VGPR acc)      B: 16 MFMA", out, clk);
    run<2, 2>("A: 128 v_fma               B: 128 v_fma", out, clk);
    run<1, 2>("A: 16 MFMA (VGPR acc)      B: 128 v_fma", out, clk);
    run<4, 2>("A: 16 MFMA (AGPR acc)      B: 128 v_fma", out, clk);
    run<1, 3>("A: 16 MFMA (VGPR acc)      B: 128 cvt_pk+128 xor", out, clk);
    run<4, 3>("A: 16 MFMA (AGPR acc)      B: 128 cvt_pk+128 xor", out, clk);
    run<1, 5>("A: 16 MFMA (VGPR acc)      B: 128 v_cos", out, clk);
    run<2, 5>("A: 128 v_fma               B: 128 v_cos", out, clk);
    return 0;
}
