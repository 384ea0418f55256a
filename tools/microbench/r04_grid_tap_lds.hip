// Round-4 microbenchmark (VERDICT r03 task 5, kill criterion): the latent-grid fetch of BASELINE configs[2] (16^3 x 16 channels) in two forms.
//   (a) shipped: fp16 x-pair records in global memory (L2-resident, 16 x 16 x 17 records x 64 B = 278 KB), per lane and tile 8 x 16-byte gathers
//       + 32 v_dot2_f32_f16 (4 records x 8 channels) + 4 v_cvt_pk_f16_f32
//   (b) proposed: u8 x-pair records (BYTE_LINEAR, affine decode folded into the first layer) in LDS (139 KB per workgroup of 8 waves), per lane
//       and tile 4 x ds_read_b128 (8 channels x 2 bytes of 4 records... 16 B each) + byte permutes + v_dot4_u32_u8 with 8-bit weights + converts
// Both kernels run the same number of taps per lane on random record indices (one tap = the work of one lane for one tile and one 16-channel
// chunk), at two waves per SIMD, and accumulate so that nothing is optimised away.  Output: ns per wave tap-pair (two tiles = one wave step).
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/r04_grid_tap_lds.hip -o tools/microbench/bin/r04_grid_tap_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr int kRecords = 16 * 16 * 17;

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

// Both kernels are software-pipelined like the renderer (GridPre, srn_device.hpp): the records of the NEXT tile are in flight while this tile
// is reduced, so the loop measures issue + memory throughput, not the latency of one gather.
template <int SPREAD>
__device__ __forceinline__ unsigned tap_record(int it, int tile, int lane, unsigned& seed) {
    // four records of neighbouring rows (rows y, y+1 of planes z, z+1).  COHERENT: the 64 rays of an 8 x 8 pixel tile span ~0.2 cells of a
    // 16^3 grid at 1024^2 pixels, so all lanes of a wave tap the same cell except a few in the x neighbour (SPREAD = 16: one lane in 16);
    // SPREAD = 0: every lane its own random cell (the worst case, not what the renderer sees)
    // ... and a ray spends ~10 steps of 1/512 in one cell of a 16^3 grid: the cell changes every tenth iteration (L1 hits in between)
    const unsigned cell = unsigned((it / 10) * 2 + tile) * 2654435761u;
    return SPREAD ? ((cell >> 8) + (((lane * 13 + it) % (SPREAD ? SPREAD : 1)) == 0 ? 1u : 0u)) % (kRecords - 17 * 17 - 18) : lcg(seed) % (kRecords - 17 * 17 - 18);
}

// (a) global fp16 records: record r = 16 channels x {v(x0), v(x1)} fp16 = 64 B; lane half h reads bytes [32 h, +32)
template <int SPREAD>
__global__ void __launch_bounds__(256, 2) tap_global(const char* __restrict__ grid, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    unsigned seed = blockIdx.x * 256 + threadIdx.x + 1;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint4_t v[4][2], nv[4][2];
    auto load = [&](int it, int tile, uint4_t (&dst)[4][2]) {
        const unsigned r0 = tap_record<SPREAD>(it, tile, lane, seed);
        const unsigned off[4] = {r0 * 64u, (r0 + 17u) * 64u, (r0 + 17u * 16u) * 64u, (r0 + 17u * 16u + 17u) * 64u};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4_t* p = reinterpret_cast<const uint4_t*>(grid + (off[k] + 32u * h));
            dst[k][0] = p[0]; dst[k][1] = p[1];
        }
    };
    load(0, 0, nv);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k][0] = nv[k][0]; v[k][1] = nv[k][1]; }
            load(tile ? it + 1 : it, tile ^ 1, nv);
            const unsigned wbits = 0x34003800u + (seed >> 20);  // some fp16 weight pair
            const half2_t w = __builtin_bit_cast(half2_t, wbits);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned u0 = v[k][0][j], u1 = v[k][1][j];
                    acc[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u0), w, acc[j], false);
                    acc[4 + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u1), w, acc[4 + j], false);
                }
            // 4 x v_cvt_pk_f16_f32 (the B fragment) -- folded back so that the converts stay
            float folded = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef float float2_t __attribute__((ext_vector_type(2)));
                const half2_t hh = __builtin_convertvector(float2_t{acc[2 * j], acc[2 * j + 1]}, half2_t);
                folded += float(hh[0]) * 1e-3f + float(hh[1]) * 1e-3f;
            }
            acc[0] = folded;
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j];
    if (s == 123.456f) sink[0] = s;
}

// (b) LDS u8 records: record r = 16 channels x {v(x0), v(x1)} u8 = 32 B; lane half h reads bytes [16 h, +16) = 8 channels
template <int SPREAD>
__global__ void __launch_bounds__(512, 1) tap_lds(const char* __restrict__ grid8, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < kRecords * 32 / 16; i += blockDim.x) reinterpret_cast<uint4_t*>(lds)[i] = reinterpret_cast<const uint4_t*>(grid8)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5;
    unsigned seed = blockIdx.x * 512 + threadIdx.x + 1;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint4_t v[4], nv[4];
    auto load = [&](int it, int tile, uint4_t (&dst)[4]) {
        const unsigned r0 = tap_record<SPREAD>(it, tile, lane, seed);
        const unsigned off[4] = {r0 * 32u, (r0 + 17u) * 32u, (r0 + 17u * 16u) * 32u, (r0 + 17u * 16u + 17u) * 32u};
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const uint4_t*>(lds + (off[k] + 16u * h));
    };
    load(0, 0, nv);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = nv[k];
            load(tile ? it + 1 : it, tile ^ 1, nv);
            // 8-bit weights of the four records: {w_k (1 - wx), w_k wx} per record -> two dwords {w0lo, w0hi, w1lo, w1hi}, {w2lo, w2hi, w3lo, w3hi}
            const unsigned w01 = 0x40302010u + (seed >> 24), w23 = 0x10203040u + (seed >> 25);
            // channel c of record k sits in bytes (2c, 2c+1) of v[k]: dword j holds channels 2j, 2j+1.  For each channel gather the pairs of
            // records (0,1) and (2,3) into one dword each (v_perm_b32) and take two v_dot4_u32_u8
            unsigned isum[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned a0 = __builtin_amdgcn_perm(v[1][j], v[0][j], 0x05040100u);  // {r0.c0.x0, r0.c0.x1, r1.c0.x0, r1.c0.x1}
                const unsigned a1 = __builtin_amdgcn_perm(v[1][j], v[0][j], 0x07060302u);  // the same for channel 2j + 1
                const unsigned b0 = __builtin_amdgcn_perm(v[3][j], v[2][j], 0x05040100u);
                const unsigned b1 = __builtin_amdgcn_perm(v[3][j], v[2][j], 0x07060302u);
                isum[2 * j] = __builtin_amdgcn_udot4(b0, w23, __builtin_amdgcn_udot4(a0, w01, 0u, false), false);
                isum[2 * j + 1] = __builtin_amdgcn_udot4(b1, w23, __builtin_amdgcn_udot4(a1, w01, 0u, false), false);
            }
            float folded = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef float float2_t __attribute__((ext_vector_type(2)));
                // (the 2^-16 scale of 8-bit data x 8-bit weights folds into the layer's weights: convert only)
                const half2_t hh = __builtin_convertvector(float2_t{float(isum[2 * j]), float(isum[2 * j + 1])}, half2_t);
                folded += float(hh[0]) * 1e-3f + float(hh[1]) * 1e-3f;
            }
            acc[0] += folded;
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j];
    if (s == 123.456f) sink[0] = s;
}

int main() {
    std::vector<unsigned char> h(kRecords * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    for (size_t i = 1; i < h.size(); i += 2) h[i] = 0x30 + (h[i] & 7);  // fp16 high bytes: finite values
    char *g, *g8; float* sink;
    hipMalloc(&g, h.size()); hipMalloc(&g8, kRecords * 32); hipMalloc(&sink, 4);
    hipMemcpy(g, h.data(), h.size(), hipMemcpyHostToDevice);
    hipMemcpy(g8, h.data(), kRecords * 32, hipMemcpyHostToDevice);
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(tap_lds<16>), hipFuncAttributeMaxDynamicSharedMemorySize, kRecords * 32);
    hipFuncSetAttribute(reinterpret_cast<const void*>(tap_lds<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kRecords * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int coherent = 1; coherent >= 0; --coherent) {
            float msA = 0, msB = 0;
            hipEventRecord(e0);
            if (coherent) hipLaunchKernelGGL(tap_global<16>, dim3(256 * 2), dim3(256), 0, 0, g, sink, iters);  // 8 waves per CU = 2 per SIMD
            else hipLaunchKernelGGL(tap_global<0>, dim3(256 * 2), dim3(256), 0, 0, g, sink, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msA, e0, e1);
            hipEventRecord(e0);
            if (coherent) hipLaunchKernelGGL(tap_lds<16>, dim3(256), dim3(512), kRecords * 32, 0, g8, sink, iters);  // one workgroup of 8 waves per CU
            else hipLaunchKernelGGL(tap_lds<0>, dim3(256), dim3(512), kRecords * 32, 0, g8, sink, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msB, e0, e1);
            // every SIMD runs 2 waves x iters wave steps (two tiles each): time per wave step and SIMD
            printf("round %d, %s taps: (a) global fp16 records + v_dot2: %.3f ms = %.0f ns per wave step and SIMD | (b) LDS u8 records + v_perm + v_dot4: %.3f ms = %.0f ns  -> (b) / (a) = %.3f\n",
                   rep, coherent ? "coherent" : "random  ", msA, msA * 1e6 / (2.0 * iters), msB, msB * 1e6 / (2.0 * iters), msB / msA);
        }
    return 0;
}
