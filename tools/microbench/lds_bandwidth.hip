// Microbenchmark: LDS read throughput per CU on gfx950 for the two access patterns of the SRN kernels:
//   frag : ds_read_b128 at base + 16*lane (1 KiB per wave instruction, conflict-free)
//   bcast: ds_read_b128 at base + 16*(lane>>5) (two distinct addresses per wave: the bias blocks)
// build: hipcc --offload-arch=gfx950 -O3 -o lds_bandwidth lds_bandwidth.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL>
__global__ void __launch_bounds__(1024) bench(unsigned* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char* p = lds + (MODE == 0 ? 16 * lane : 16 * (lane >> 5));
    uint4_t acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        uint4_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = *reinterpret_cast<const uint4_t*>(p + ((it * UNROLL + u) & 31) * 1024);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <int MODE>
void run(const char* name, unsigned* out, int wavesPerCU) {
    const int iters = 4000, U = 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // one workgroup per CU (64 KiB of LDS keeps a second one away... use 2 blocks max): blockDim = 64 * wavesPerCU
    const int blocks = 256, threads = 64 * wavesPerCU;
    bench<MODE, U><<<blocks, threads, 96 * 1024>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<MODE, U><<<blocks, threads, 96 * 1024>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instrPerCU = double(iters) * U * wavesPerCU;
    const double nsPerInstr = double(ms) * 1e6 / instrPerCU;
    printf("%-6s waves/CU=%2d  %6.2f ns per ds_read_b128 per CU  = %6.1f B/ns/CU  (%.1f cycles @2.4GHz, %.1f B/clk)\n", name, wavesPerCU, nsPerInstr,
           1024.0 / nsPerInstr, nsPerInstr * 2.4, 1024.0 / (nsPerInstr * 2.4));
}

int main() {
    unsigned* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int w : {1, 2, 4, 8, 16}) run<0>("frag", out, w);
    for (int w : {1, 2, 4, 8, 16}) run<1>("bcast", out, w);
    return 0;
}
