// Microbenchmark: how well do MFMA (32x32x16 f16) chains and the VALU convert work of the SRN layer loop overlap
// inside ONE wave on gfx950, for a blocked order (all MFMAs of a layer, then all converts) versus a software-pipelined
// order (tile 1's MFMAs interleaved with tile 0's converts and vice versa)?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void cvt_quarter(const floatx16& d, int q, half8_t& f0, half8_t& f1) {
    // q in 0..3: registers {2q,2q+1} -> f0 pair q, {8+2q, 9+2q} -> f1 pair q  (2 x v_cvt_pk_f16_f32 with clamp)
    float2_t v0 = {d[2 * q], d[2 * q + 1]}, v1 = {d[8 + 2 * q], d[8 + 2 * q + 1]};
    half2_t h0 = __builtin_convertvector(v0, half2_t), h1 = __builtin_convertvector(v1, half2_t);
    const half2_t z = {0, 0}, o = {1, 1};
    h0 = __builtin_elementwise_min(__builtin_elementwise_max(h0, z), o);
    h1 = __builtin_elementwise_min(__builtin_elementwise_max(h1, z), o);
    f0[2 * q] = h0[0]; f0[2 * q + 1] = h0[1];
    f1[2 * q] = h1[0]; f1[2 * q + 1] = h1[1];
}

template <int MT, int KS, int MODE, int EXTRA>
__global__ void __launch_bounds__(64) bench(float* out, const half8_t* w, int iters, long long* clk) {
    const int lane = threadIdx.x;
    half8_t a[MT * KS];
#pragma unroll
    for (int i = 0; i < MT * KS; ++i) a[i] = w[i * 64 + lane];
    floatx16 bias;
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = 0.01f * float(r + lane);
    half8_t xb[2][2 * MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2 * MT; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) xb[t][s][j] = _Float16(0.001f * float(lane + j + s + t));
    floatx16 acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = bias;
    float extra = float(lane);
    const long long c0 = clock64(), w0 = wall_clock64();

    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {  // blocked, k-major (both tiles share each A fragment)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m * KS + s], xb[0][s], s == 0 ? bias : acc[0][m], 0, 0, 0);
                    acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m * KS + s], xb[1][s], s == 0 ? bias : acc[1][m], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) cvt_quarter(acc[t][m], q, xb[t][2 * m], xb[t][2 * m + 1]);
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) extra = fmaf(extra, 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (MODE == 1) {  // software pipelined across the two tiles
            constexpr int NM = MT * KS, NV = MT * 4;
            // A: M(t0) || V(t1)
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int m = i / KS, s = i % KS;
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], xb[0][s], s == 0 ? bias : acc[0][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = (i * NV) / NM; j < ((i + 1) * NV) / NM; ++j) cvt_quarter(acc[1][j / 4], j % 4, xb[1][2 * (j / 4)], xb[1][2 * (j / 4) + 1]);
#pragma unroll
                for (int e = (i * EXTRA / 2) / NM; e < ((i + 1) * EXTRA / 2) / NM; ++e) extra = fmaf(extra, 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
            // B: M(t1) || V(t0)
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int m = i / KS, s = i % KS;
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], xb[1][s], s == 0 ? bias : acc[1][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = (i * NV) / NM; j < ((i + 1) * NV) / NM; ++j) cvt_quarter(acc[0][j / 4], j % 4, xb[0][2 * (j / 4)], xb[0][2 * (j / 4) + 1]);
#pragma unroll
                for (int e = (i * EXTRA / 2) / NM; e < ((i + 1) * EXTRA / 2) / NM; ++e) extra = fmaf(extra, 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (MODE == 2) {  // MFMAs only
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m * KS + s], xb[0][s], s == 0 ? bias : acc[0][m], 0, 0, 0);
                    acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m * KS + s], xb[1][s], s == 0 ? bias : acc[1][m], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        } else {  // VALU only
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) cvt_quarter(acc[t][m], q, xb[t][2 * m], xb[t][2 * m + 1]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][m][r] += float(xb[t][2 * m][r & 7]);  // keep a dependence (adds VALU)
                }
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) extra = fmaf(extra, 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && lane == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    float s = extra;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[t][m][r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 2 * MT; ++q) s += float(xb[t][q][0]);
    out[blockIdx.x * 64 + lane] = s;
}

template <int MT, int KS, int MODE, int EXTRA>
void run(const char* name, float* out, half8_t* w, int wavesPerSimd) {
    static long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    const int iters = 20000, blocks = 1024 * wavesPerSimd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MT, KS, MODE, EXTRA><<<blocks, 64>>>(out, w, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<MT, KS, MODE, EXTRA><<<blocks, 64>>>(out, w, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long hclk[2];
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = double(hclk[0]) / (double(hclk[1]) * 10.0);  // wall clock counts at 100 MHz
    const double nsPerIterPerSimd = double(ms) * 1e6 / iters / wavesPerSimd;
    const double cyc = nsPerIterPerSimd * ghz;
    printf("%-28s C=%3d waves/SIMD=%d extraVALU=%2d  %7.1f ns = %6.0f cycles per layer-step @ %.2f GHz (MFMA-only bound %d cycles) -> MFMA pipe %.0f%%\n", name, 16 * KS,
           wavesPerSimd, EXTRA, nsPerIterPerSimd, cyc, ghz, 2 * MT * KS * 32, 100.0 * 2 * MT * KS * 32 / cyc);
}

int main() {
    float* out; half8_t* w;
    hipMalloc(&out, 1024 * 8 * 64 * 4);
    hipMalloc(&w, 64 * 64 * 16);
    std::vector<_Float16> hw(64 * 64 * 8);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = _Float16(0.01f * float(int(i % 17) - 8));
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 3, 4}) {
        if (wps == 1) {
            run<1, 2, 2, 0>("mfma only", out, w, wps);
            run<1, 2, 3, 0>("valu only (cvt + 32 adds)", out, w, wps);
        }
        run<1, 2, 0, 0>("blocked", out, w, wps);
        run<1, 2, 1, 0>("pipelined", out, w, wps);
        run<1, 2, 0, 16>("blocked", out, w, wps);
        run<1, 2, 1, 16>("pipelined", out, w, wps);
        if (wps == 1) run<2, 4, 2, 0>("mfma only", out, w, wps);
        run<2, 4, 0, 0>("blocked", out, w, wps);
        run<2, 4, 1, 0>("pipelined", out, w, wps);
        run<2, 4, 0, 32>("blocked", out, w, wps);
        run<2, 4, 1, 32>("pipelined", out, w, wps);
    }
    return 0;
}
