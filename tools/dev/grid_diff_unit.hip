// Developer unit check (GPU box): the piecewise-exact central differences of the latent grid (srn_gradient.hpp,
// grid_value_and_differences8) against six plain fetches (grid_values8) on a random grid of x-pair records.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fv-srn_amd/csrc tools/dev/grid_diff_unit.hip -o tools/dev/bin/grid_diff_unit
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "srn_gradient.hpp"

using namespace fvsrn;

__global__ void check_kernel(NetParams P, const float* pos, int n, float step, float* outFast, float* outRef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    float val[8], d[3][8];
    const GridDiffTap t = grid_diff_tap(P, px, py, pz, step);
    grid_value_and_differences8(P, t, 0, 0, val, d[0], d[1], d[2]);
    for (int j = 0; j < 8; ++j) {
        outFast[32 * i + j] = val[j];
        outFast[32 * i + 8 + j] = d[0][j];
        outFast[32 * i + 16 + j] = d[1][j];
        outFast[32 * i + 24 + j] = d[2][j];
    }
    if (i < 4) printf("sample %d pos %f %f %f  cy %f %f %f cz %f %f %f off8 %u off9 %u off10 %u off11 %u uy %f wy %f uz %f wz %f\n", i, px, py, pz, t.cy[0], t.cy[1], t.cy[2], t.cz[0], t.cz[1], t.cz[2],
                      t.off[8], t.off[9], t.off[10], t.off[11], t.uy, t.wy, t.uz, t.wz);
    float v[8], hi[8], lo[8];
    grid_values8<1>(P, grid_tap(P, px, py, pz), 0, 0, v);
    for (int j = 0; j < 8; ++j) outRef[32 * i + j] = v[j];
    for (int a = 0; a < 3; ++a) {
        grid_values8<1>(P, grid_tap(P, px + (a == 0 ? step : 0.f), py + (a == 1 ? step : 0.f), pz + (a == 2 ? step : 0.f)), 0, 0, hi);
        grid_values8<1>(P, grid_tap(P, px - (a == 0 ? step : 0.f), py - (a == 1 ? step : 0.f), pz - (a == 2 ? step : 0.f)), 0, 0, lo);
        for (int j = 0; j < 8; ++j) outRef[32 * i + 8 + 8 * a + j] = (hi[j] - lo[j]) * (0.5f / step);
    }
}

int main() {
    const int X = 8, Y = 6, Z = 5, G = 16, n = 1 << 16;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-0.1f, 1.1f);
    std::vector<float> vol(size_t(Z) * Y * X * G);
    for (auto& v : vol) v = nd(rng);
    std::vector<__half> rec(size_t(Z) * Y * (X + 1) * G * 2);
    for (int z = 0; z < Z; ++z) for (int y = 0; y < Y; ++y) for (int xi = 0; xi <= X; ++xi) for (int c = 0; c < G; ++c) for (int p = 0; p < 2; ++p) {
        const int xs = std::min(std::max(xi - 1 + p, 0), X - 1);
        rec[(((size_t(z) * Y + y) * (X + 1) + xi) * G + c) * 2 + p] = __float2half(vol[((size_t(z) * Y + y) * X + xs) * G + c]);
    }
    std::vector<float> pos(3 * n);
    for (auto& v : pos) v = ud(rng);
    void *dRec, *dPos, *dFast, *dRef;
    hipMalloc(&dRec, rec.size() * 2); hipMemcpy(dRec, rec.data(), rec.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&dPos, pos.size() * 4); hipMemcpy(dPos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dFast, size_t(n) * 32 * 4); hipMalloc(&dRef, size_t(n) * 32 * 4);
    NetParams P{};
    P.grid = dRec; P.gridX = X; P.gridY = Y; P.gridZ = Z; P.gridC = G; P.gridXf = X; P.gridYf = Y; P.gridZf = Z; P.gridK = 1;
    for (float scale : {4.f, 2.f, 10.f}) {
        const float step = 1.f / (X * scale);
        hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, 0, P, (const float*)dPos, n, step, (float*)dFast, (float*)dRef);
        std::vector<float> a(size_t(n) * 32), b(size_t(n) * 32);
        hipMemcpy(a.data(), dFast, a.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), dRef, b.size() * 4, hipMemcpyDeviceToHost);
        double worst[4] = {0, 0, 0, 0}, mag[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) for (int q = 0; q < 4; ++q) for (int j = 0; j < 8; ++j) {
            const double e = std::fabs(double(a[32 * i + 8 * q + j]) - b[32 * i + 8 * q + j]);
            if (!(e <= worst[q])) worst[q] = e;
            mag[q] = std::max(mag[q], std::fabs(double(b[32 * i + 8 * q + j])));
        }
        for (int i = 0; i < 4; ++i) {
            const double fy = pos[3 * i + 1] * Y - 0.5, fz = pos[3 * i + 2] * Z - 0.5, fx = pos[3 * i] * X - 0.5;
            std::printf("host sample %d texel %f %f %f  fast dY[0] %f ref %f   fast dZ[0] %f ref %f\n", i, fx, fy, fz, a[32 * i + 16], b[32 * i + 16], a[32 * i + 24], b[32 * i + 24]);
        }
        int shown = 0;
        for (int i = 0; i < n && shown < 6; ++i) for (int q = 0; q < 4; ++q) for (int j = 0; j < 8; ++j) {
            const double e = std::fabs(double(a[32 * i + 8 * q + j]) - b[32 * i + 8 * q + j]);
            if (!(e < 0.1) && shown < 6) { ++shown; std::printf("bad sample %d quantity %d channel %d texel %f %f %f fast %f ref %f\n", i, q, j, pos[3 * i] * X - 0.5, pos[3 * i + 1] * Y - 0.5, pos[3 * i + 2] * Z - 0.5, a[32 * i + 8 * q + j], b[32 * i + 8 * q + j]); }
        }
        std::printf("step 1/(%d * %g): max |fast - six fetches|  value %.3g (of %.3g)  d/dx %.3g (of %.3g)  d/dy %.3g (of %.3g)  d/dz %.3g (of %.3g)\n", X, scale,
                    worst[0], mag[0], worst[1], mag[1], worst[2], mag[2], worst[3], mag[3]);
    }
    return 0;
}
