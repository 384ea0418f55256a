// Microbenchmark: issue model of one gfx950 SIMD for MFMA 32x32x16 f16 + VALU.
//  T1: one wave, loop { MFMA ; N independent v_fma } with 2 alternating accumulator chains  -> cycles per MFMA vs N
//  T2: same, one accumulator chain (every MFMA depends on the previous one)
//  T3: same as T1 but the VALU ops are v_cvt_pk_f16_f32 reading the OTHER chain's accumulator (the real data flow)
//  T4: two waves per SIMD: wave A = MFMAs only (1 or 2 chains), wave B = v_fma only: progress of B while A runs
// build: hipcc --offload-arch=gfx950 -O3 -o issue_model issue_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int N, int CHAINS, int KIND>
__global__ void __launch_bounds__(64) t1(float* out, int iters, long long* clk) {
    const int lane = threadIdx.x;
    half8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = _Float16(0.01f * (lane + j)); b[j] = _Float16(0.02f * j); }
    floatx16 c[2] = {{0}, {0}};
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = float(lane + j);
    unsigned sink = 0;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = CHAINS == 2 ? (k & 1) : 0;
            c[ch] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[ch], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if constexpr (KIND == 0) {
                    x[j % 16] = fmaf(x[j % 16], 1.0001f, 0.5f);
                } else {
                    const int o = CHAINS == 2 ? 1 - ch : 0;
                    float2_t v = {c[o][(2 * j) % 16], c[o][(2 * j + 1) % 16]};
                    half2_t hh = __builtin_convertvector(v, half2_t);
                    unsigned u = __builtin_bit_cast(unsigned, hh);
                    asm volatile("" : "+v"(u));
                    if (j == 0) sink ^= u;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64();
    if (blockIdx.x == 0 && lane == 0) clk[0] = c1 - c0;
    float r = float(sink);
    for (int j = 0; j < 16; ++j) r += c[0][j] + c[1][j] + x[j];
    out[blockIdx.x * 64 + lane] = r;
}

template <int N, int CHAINS, int KIND>
void run1(float* out, long long* clk) {
    const int iters = 2000;
    t1<N, CHAINS, KIND><<<1024, 64>>>(out, 10, clk);
    hipDeviceSynchronize();
    t1<N, CHAINS, KIND><<<1024, 64>>>(out, iters, clk);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("  %d chain(s), %2d %s per MFMA: %6.1f cycles per MFMA\n", CHAINS, N, KIND == 0 ? "v_fma   " : "v_cvt_pk", double(h) / iters / 8);
}

template <int CHAINS, int NOPS>
__global__ void __launch_bounds__(512) t4(float* out, int iters, long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r = 0;
    __shared__ int doneA;
    if (threadIdx.x == 0) doneA = 0;
    __syncthreads();
    if (wave < 4) {
        half8_t a, b;
        for (int j = 0; j < 8; ++j) { a[j] = _Float16(0.01f * (lane + j)); b[j] = _Float16(0.02f * j); }
        floatx16 c[2] = {{0}, {0}};
        const long long c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c[CHAINS == 2 ? (k & 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[CHAINS == 2 ? (k & 1) : 0], 0, 0, 0);
                if constexpr (NOPS > 0) asm volatile("s_nop %0" ::"n"(NOPS - 1));
            }
        }
        const long long c1 = clock64();
        if (lane == 0) { atomicAdd(&doneA, 1); if (blockIdx.x == 0 && wave == 0) clk[0] = c1 - c0; }
        for (int j = 0; j < 16; ++j) r += c[0][j] + c[1][j];
    } else {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = float(lane + j);
        long long n = 0;
        // count v_fma batches (64 each) completed while the MFMA waves are still running
        while (__builtin_amdgcn_readfirstlane(*(volatile int*)&doneA) < 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], 1.0001f, 0.5f);
            ++n;
        }
        if (blockIdx.x == 0 && wave == 4 && lane == 0) clk[1] = n;
        for (int j = 0; j < 8; ++j) r += x[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int CHAINS, int NOPS>
void run4(float* out, long long* clk) {
    const int iters = 2000;
    t4<CHAINS, NOPS><<<256, 512>>>(out, 10, clk);
    hipDeviceSynchronize();
    t4<CHAINS, NOPS><<<256, 512>>>(out, iters, clk);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("  wave A: %d chain(s), s_nop %2d after each MFMA: %6.1f cycles per MFMA; wave B issued %.1f v_fma per MFMA of A\n", CHAINS, NOPS,
           double(h[0]) / iters / 8, double(h[1]) * 64 / (double(iters) * 8));
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 1024 * 512 * 4);
    hipMalloc(&clk, 64);
    printf("T1/T2: one wave per SIMD, loop { MFMA ; N VALU }\n");
    run1<0, 2, 0>(out, clk); run1<2, 2, 0>(out, clk); run1<4, 2, 0>(out, clk); run1<6, 2, 0>(out, clk); run1<8, 2, 0>(out, clk);
    run1<10, 2, 0>(out, clk); run1<12, 2, 0>(out, clk); run1<16, 2, 0>(out, clk);
    run1<0, 1, 0>(out, clk); run1<4, 1, 0>(out, clk); run1<8, 1, 0>(out, clk); run1<12, 1, 0>(out, clk);
    printf("T3: VALU = v_cvt_pk_f16_f32 of the other chain's accumulator\n");
    run1<2, 2, 1>(out, clk); run1<4, 2, 1>(out, clk); run1<6, 2, 1>(out, clk); run1<8, 2, 1>(out, clk);
    printf("T4: two waves per SIMD, A = MFMA only, B = v_fma only\n");
    run4<2, 0>(out, clk); run4<1, 0>(out, clk); run4<2, 4>(out, clk); run4<2, 8>(out, clk); run4<2, 16>(out, clk);
    return 0;
}
