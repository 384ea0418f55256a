// Developer tool: does grid_tap (srn_device.hpp) alone reproduce the launch-to-launch differences of profiles/r03/nondeterminism_r03.md?
// The function is compiled twice into one kernel -- without the spacing (FVSRN_TAP_NOPS = 0) and with all of it (127) -- and every lane
// compares the two taps of the same position over many steps of a ray.
// build: tools/dev/build_tap_repro.sh
#include <hip/hip_runtime.h>
#include <cstdio>

#define FVSRN_TAP_NOPS 0
#include "srn_device.hpp"
#include "grid_tap_spaced.hpp"

__global__ void __launch_bounds__(256, 2) repro(fvsrn::NetParams P, fvsrn::NetParams Q, unsigned* bad, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float px = 0.11f + 0.0007f * (i & 63), py = 0.23f + 0.0011f * ((i >> 6) & 63), pz = 0.05f;
    const float dx = 0.00071f, dy = 0.00043f, dz = 0.00183f;
    unsigned mism = 0, sink = 0;
    for (int it = 0; it < iters; ++it) {
        px += dx; py += dy; pz += dz;
        if (px > 1.f) px -= 1.f; if (py > 1.f) py -= 1.f; if (pz > 1.f) pz -= 1.f;
        const fvsrn::GridTap a = fvsrn::grid_tap(P, px, py, pz);
        const fvsrn::GridTap b = fvsrn::grid_tap_spaced(Q, px, py, pz);
        bool d = false;
        for (int k = 0; k < 4; ++k) d = d || a.off[k] != b.off[k] || a.w[k] != b.w[k];
        mism += d ? 1u : 0u;
        sink ^= a.off[0] + b.w[3];
    }
    if (mism) atomicAdd(bad, mism);
    if (sink == 0x12345u) atomicAdd(bad + 1, 1u);
}

int main() {
    fvsrn::NetParams P{};
    P.gridX = P.gridY = P.gridZ = 32; P.gridC = 16; P.gridXf = P.gridYf = P.gridZf = 32.f; P.gridK = 1;
    fvsrn::NetParams Q{};
    Q.gridX = Q.gridY = Q.gridZ = 32; Q.gridC = 16; Q.gridXf = Q.gridYf = Q.gridZf = 32.f; Q.gridK = 1;
    unsigned* bad;
    hipMalloc(&bad, 8);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(repro, dim3(256 * 8), dim3(256), 0, 0, P, Q, bad, 4000);
        unsigned b[2] = {0, 0};
        hipMemcpy(b, bad, 8, hipMemcpyDeviceToHost);
        printf("launch %d: %u of %llu taps differ between the unspaced and the spaced grid_tap\n", rep, b[0], 256ull * 8 * 256 * 4000);
    }
    return 0;
}
