// Round-2 microbenchmark: does the MFMA shape change what the chip delivers on RANDOM data (MI355X_MICROARCH.md, DVFS give-back item 7)?
// A 32x4-like layer chain on 64 samples per wave, weights / biases in registers, activations fed back through a clamped convert
// (the ReLU01 image), plus the 62-instruction vector phase of the headline step as packed-fp32 filler:
//   SHAPE 0: v_mfma_f32_32x32x16_f16, 4 per layer (2 sample tiles x 2 K steps)       -- what render_small_kernel issues
//   SHAPE 1: v_mfma_f32_16x16x32_f16, 8 per layer (4 sample tiles x 2 row tiles)     -- same FLOPs, same converts
// Reports wall time per wave step, the in-kernel clock (s_memtime / s_memrealtime) and samples/s.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned pack_clamped(float a, float b) {
    const float2_t v = {a, b};
    const half2_t z = {0, 0}, o = {1, 1};
    const half2_t h = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_convertvector(v, half2_t), z), o);
    return __builtin_bit_cast(unsigned, h);
}

template <int SHAPE, int FILL>
__global__ void __launch_bounds__(256, 2) chain(const half8_t* __restrict__ w, const float* __restrict__ b, const half8_t* __restrict__ x, float* out,
                                                int iters, long long* clk) {
    const int lane = threadIdx.x & 63;
    half8_t W[3][2];
    for (int l = 0; l < 3; ++l)
        for (int s = 0; s < 2; ++s) W[l][s] = w[(l * 2 + s) * 64 + lane];
    float st[32];
    for (int i = 0; i < 32; ++i) st[i] = b[(i * 64 + lane) & 1023];
    float4_t B4[3][2];
    floatx16 B16[3];
    for (int l = 0; l < 3; ++l) {
        for (int i = 0; i < 16; ++i) B16[l][i] = b[(l * 16 + i) * 64 + lane] * 0.25f + 0.2f;
        for (int r = 0; r < 2; ++r)
            for (int i = 0; i < 4; ++i) B4[l][r][i] = b[((l * 2 + r) * 4 + i) * 64 + lane] * 0.25f + 0.2f;
    }
    unsigned X[16];  // 64 samples x 32 channels of fp16 = 16 registers per lane in either layout
    for (int i = 0; i < 4; ++i) {
        const half8_t v = x[i * 64 + lane];
        for (int j = 0; j < 4; ++j) X[4 * i + j] = __builtin_bit_cast(unsigned, half2_t{v[2 * j], v[2 * j + 1]});
    }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (FILL) {  // vector phase: 30 packed fp32 rotations + 16 converts (their results join the activations below)
#pragma unroll
            for (int i = 0; i < 30; i += 2) {
                const float2_t a = {st[i], st[i + 1]}, r = {0.99985f, 0.0175f};
                const float2_t n = a * r[0] + float2_t{-a[1], a[0]} * r[1];
                st[i] = n[0]; st[i + 1] = n[1];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) X[i] ^= pack_clamped(st[(2 * i) & 31], st[(2 * i + 1) & 31]) & 0x00010001u;
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (SHAPE == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    half8_t x0, x1;
                    x0 = __builtin_bit_cast(half8_t, uint4{X[8 * t], X[8 * t + 1], X[8 * t + 2], X[8 * t + 3]});
                    x1 = __builtin_bit_cast(half8_t, uint4{X[8 * t + 4], X[8 * t + 5], X[8 * t + 6], X[8 * t + 7]});
                    floatx16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[l][0], x0, B16[l], 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[l][1], x1, acc, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 8; ++q) X[8 * t + q] = pack_clamped(acc[2 * q], acc[2 * q + 1]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half8_t xj = __builtin_bit_cast(half8_t, uint4{X[4 * j], X[4 * j + 1], X[4 * j + 2], X[4 * j + 3]});
                    const float4_t a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[l][0], xj, B4[l][0], 0, 0, 0);
                    const float4_t a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[l][1], xj, B4[l][1], 0, 0, 0);
                    X[4 * j] = pack_clamped(a0[0], a0[1]); X[4 * j + 1] = pack_clamped(a0[2], a0[3]);
                    X[4 * j + 2] = pack_clamped(a1[0], a1[1]); X[4 * j + 3] = pack_clamped(a1[2], a1[3]);
                }
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    unsigned acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= X[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = float(acc & 0xffff) + st[3];
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// The wave step of the 32x4 + 16-channel latent grid kernel (render_small_kernel<.., SGRID = 1>) as a program on random data: 2 phase MFMAs,
// 28 v_cos + 16 converts (direct Fourier features), taps (45 fp32 instructions + 8 lane swaps), 16 x 16-byte gathers from a 128 KB grid
// (L2-resident, random cells per lane), 64 v_dot2 + 8 converts, the 14-MFMA layer chain (12 + the latent K step of both tiles) with 48 converts.
// No ray bookkeeping, no transfer function, no blending: a floor for that formulation, not a kernel.
__global__ void __launch_bounds__(256, 2) grid_step(const half8_t* __restrict__ w, const float* __restrict__ b, const half8_t* __restrict__ x,
                                                    const uint4* __restrict__ grid, float* out, int iters, long long* clk) {
    const int lane = threadIdx.x & 63;
    half8_t W[3][2], WP, WG;
    for (int l = 0; l < 3; ++l)
        for (int s = 0; s < 2; ++s) W[l][s] = w[(l * 2 + s) * 64 + lane];
    WP = w[lane]; WG = w[64 + lane];
    floatx16 B16[3];
    for (int l = 0; l < 3; ++l)
        for (int i = 0; i < 16; ++i) B16[l][i] = b[(l * 16 + i) * 64 + lane] * 0.25f + 0.2f;
    // an 8x8 pixel tile of nearly parallel rays (neighbouring lanes share grid cells like in a frame), step 1/512
    float px = 0.31f + (lane & 7) * 1e-3f + blockIdx.x * 1.3e-3f, py = 0.42f + (lane >> 3) * 1e-3f, pz = 0.05f + b[128 + lane] * 1e-4f;
    const float dx = 0.0007f + b[192 + lane] * 1e-6f, dy = 0.0004f + b[256 + lane] * 1e-6f, dz = 0.0018f;
    unsigned X[16];
    for (int i = 0; i < 16; ++i) X[i] = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        px += dx; py += dy; pz += dz;
        px -= floorf(px); py -= floorf(py); pz -= floorf(pz);
        // taps (the arithmetic of grid_tap: 16^3 grid, 64-byte records)
        const float fx = fmaf(px, 16.f, -0.5f), fy = fmaf(py, 16.f, -0.5f), fz = fmaf(pz, 16.f, -0.5f);
        const float x0 = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
        const float wx = fx - x0, wy = fy - y0f, wz = fz - z0f;
        const float xi = __builtin_amdgcn_fmed3f(x0 + 1.f, 0.f, 16.f);
        const float y0 = __builtin_amdgcn_fmed3f(y0f, 0.f, 15.f), y1 = __builtin_amdgcn_fmed3f(y0f + 1.f, 0.f, 15.f);
        const float z0 = __builtin_amdgcn_fmed3f(z0f, 0.f, 15.f), z1 = __builtin_amdgcn_fmed3f(z0f + 1.f, 0.f, 15.f);
        unsigned off[4];
        off[0] = __umul24(unsigned(fmaf(fmaf(z0, 16.f, y0), 17.f, xi)), 4u);
        off[1] = __umul24(unsigned(fmaf(fmaf(z0, 16.f, y1), 17.f, xi)), 4u);
        off[2] = __umul24(unsigned(fmaf(fmaf(z1, 16.f, y0), 17.f, xi)), 4u);
        off[3] = __umul24(unsigned(fmaf(fmaf(z1, 16.f, y1), 17.f, xi)), 4u);
        const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
        const float w4[4] = {uz * uy, uz * wy, wz * uy, wz * wy};
        unsigned wt[4];
        for (int k = 0; k < 4; ++k) wt[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{w4[k] * ux, w4[k] * wx}, half2_t));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned o_[4], w_[4];
            for (int k = 0; k < 4; ++k) {
                auto a = __builtin_amdgcn_permlane32_swap(off[k], off[k], false, false);
                auto c = __builtin_amdgcn_permlane32_swap(wt[k], wt[k], false, false);
                o_[k] = t ? a[1] : a[0]; w_[k] = t ? c[1] : c[0];
            }
            uint4 r[4][2];
            for (int k = 0; k < 4; ++k) { r[k][0] = grid[(o_[k] & 0x1fff) * 1 + 0]; r[k][1] = grid[(o_[k] & 0x1fff) + 1]; }
            // direct Fourier features of this tile: phase MFMA, 14 cos (+ 2 pass-through), 8 converts
            const half8_t bp = __builtin_bit_cast(half8_t, uint4{__builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{px, px}, half2_t)),
                                                                __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{py, py}, half2_t)),
                                                                __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{pz, pz}, half2_t)), 0x3c003c00u});
            floatx16 d = {0};
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(WP, bp, d, 0, 0, 0);
            for (int i = 2; i < 16; ++i) d[i] = __builtin_amdgcn_cosf(d[i]);
            for (int q = 0; q < 8; ++q) X[8 * t + q] = __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{d[2 * q], d[2 * q + 1]}, half2_t));
            // 32 dot products + 4 converts
            float acc[8];
            for (int j = 0; j < 4; ++j) {
                const unsigned u0[4] = {r[0][0].x, r[0][0].y, r[0][0].z, r[0][0].w};
                (void)u0;
            }
            for (int k = 0; k < 4; ++k) {
                const unsigned a0[4] = {r[k][0].x, r[k][0].y, r[k][0].z, r[k][0].w}, a1[4] = {r[k][1].x, r[k][1].y, r[k][1].z, r[k][1].w};
                const half2_t wh = __builtin_bit_cast(half2_t, w_[k]);
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a0[j]), wh, k ? acc[j] : 0.f, false);
                    acc[4 + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a1[j]), wh, k ? acc[4 + j] : 0.f, false);
                }
            }
            uint4 gfu;
            gfu.x = __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{acc[0], acc[1]}, half2_t));
            gfu.y = __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{acc[2], acc[3]}, half2_t));
            gfu.z = __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{acc[4], acc[5]}, half2_t));
            gfu.w = __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{acc[6], acc[7]}, half2_t));
            // first layer of this tile with the latent K step, then the two hidden layers
            half8_t x0h = __builtin_bit_cast(half8_t, uint4{X[8 * t], X[8 * t + 1], X[8 * t + 2], X[8 * t + 3]});
            half8_t x1h = __builtin_bit_cast(half8_t, uint4{X[8 * t + 4], X[8 * t + 5], X[8 * t + 6], X[8 * t + 7]});
            floatx16 a = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[0][0], x0h, B16[0], 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[0][1], x1h, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(WG, __builtin_bit_cast(half8_t, gfu), a, 0, 0, 0);
            for (int q = 0; q < 8; ++q) X[8 * t + q] = pack_clamped(a[2 * q], a[2 * q + 1]);
#pragma unroll
            for (int l = 1; l < 3; ++l) {
                x0h = __builtin_bit_cast(half8_t, uint4{X[8 * t], X[8 * t + 1], X[8 * t + 2], X[8 * t + 3]});
                x1h = __builtin_bit_cast(half8_t, uint4{X[8 * t + 4], X[8 * t + 5], X[8 * t + 6], X[8 * t + 7]});
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[l][0], x0h, B16[l], 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[l][1], x1h, a, 0, 0, 0);
                for (int q = 0; q < 8; ++q) X[8 * t + q] = pack_clamped(a[2 * q], a[2 * q + 1]);
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    unsigned acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= X[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = float(acc & 0xffff) + px;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int SHAPE, int FILL>
void run(const char* name, const half8_t* w, const float* b, const half8_t* x, float* out, long long* clk, int blocks) {
    const int iters = 200000;
    chain<SHAPE, FILL><<<blocks, 256>>>(w, b, x, out, 20000, clk);
    hipDeviceSynchronize();
    // ~2 s of back-to-back launches first (DVFS give-back item 6), then the timed one
    for (int i = 0; i < 8; ++i) chain<SHAPE, FILL><<<blocks, 256>>>(w, b, x, out, iters, clk);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    chain<SHAPE, FILL><<<blocks, 256>>>(w, b, x, out, iters, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < blocks; ++i) ghz.push_back(double(h[2 * i]) / double(h[2 * i + 1]) * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double steps = double(iters) * blocks * 4;  // wave steps
    printf("%-44s %8.2f ms  %7.1f ns / wave step / SIMD  clock %.3f GHz  cycles / wave step / SIMD %7.1f  %6.1f Gsamples/s\n", name, ms,
           ms * 1e6 / (double(iters) * blocks * 4 / 1024.0), ghz[ghz.size() / 2], ms * 1e6 * ghz[ghz.size() / 2] / (double(iters) * blocks * 4 / 1024.0),
           steps * 64 / ms / 1e6);
}

int main() {
    const int blocks = 512;  // 256 CUs x 2 workgroups of 4 waves: 2 waves per SIMD
    std::vector<_Float16> hw(6 * 64 * 8), hx(4 * 64 * 8);
    std::vector<float> hb(64 * 64);
    srand(7);
    for (auto& v : hw) v = _Float16((rand() / float(RAND_MAX) - 0.5f) * 0.7f);
    for (auto& v : hx) v = _Float16(rand() / float(RAND_MAX));
    for (auto& v : hb) v = rand() / float(RAND_MAX) - 0.5f;
    half8_t *w, *x; float *b, *out; long long* clk;
    hipMalloc(&w, hw.size() * 2); hipMalloc(&x, hx.size() * 2); hipMalloc(&b, hb.size() * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    run<0, 0>("32x32x16 chain, no vector phase", w, b, x, out, clk, blocks);
    run<1, 0>("16x16x32 chain, no vector phase", w, b, x, out, clk, blocks);
    run<0, 1>("32x32x16 chain + vector phase", w, b, x, out, clk, blocks);
    run<1, 1>("16x16x32 chain + vector phase", w, b, x, out, clk, blocks);
    run<0, 1>("32x32x16 chain + vector phase (again)", w, b, x, out, clk, blocks);
    run<1, 1>("16x16x32 chain + vector phase (again)", w, b, x, out, clk, blocks);
    {   // latent-grid step: 128 KB of random fp16 records
        std::vector<_Float16> hg(8192 * 8 + 16);
        for (auto& v : hg) v = _Float16((rand() / float(RAND_MAX) - 0.5f) * 0.6f);
        uint4* g;
        hipMalloc(&g, hg.size() * 2);
        hipMemcpy(g, hg.data(), hg.size() * 2, hipMemcpyHostToDevice);
        const int iters = 60000;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 6; ++i) grid_step<<<blocks, 256>>>(w, b, x, g, out, iters, clk);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            grid_step<<<blocks, 256>>>(w, b, x, g, out, iters, clk);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(2 * blocks);
            hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            std::vector<double> ghz;
            for (int i = 0; i < blocks; ++i) ghz.push_back(double(h[2 * i]) / double(h[2 * i + 1]) * 0.1);
            std::sort(ghz.begin(), ghz.end());
            const double wsteps = double(iters) * blocks * 4;
            printf("%-44s %8.2f ms  %7.1f ns / wave step / SIMD  clock %.3f GHz  cycles / wave step / SIMD %7.1f  %6.1f Gsamples/s\n",
                   "32x4 + 16-ch grid step (2 waves / SIMD)", ms, ms * 1e6 / (wsteps / 1024.0), ghz[ghz.size() / 2],
                   ms * 1e6 * ghz[ghz.size() / 2] / (wsteps / 1024.0), wsteps * 64 / ms / 1e6);
        }
    }
    return 0;
}
