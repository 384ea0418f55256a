// Dense grid volumes: device-side restatement of kernel::VolumeInterpolationGrid (reference
// renderer/renderer_volume_grid.cuh:89-232) -- nearest / trilinear / tricubic sampling of one scalar feature, in the
// reference's two addressing conventions (tensor accessor, CUDA texture).
#pragma once
#include <hip/hip_runtime.h>

#include "device_params.hpp"

namespace fvsrn {

struct VolumeParams {
    const float* data;        // fp32 voxels in HBM, in 4x4x4 bricks (256 bytes = two cache lines): brick (x>>2, y>>2, z>>2) in
                              // x-fastest order, voxel (x&3) + 4 (y&3) + 16 (z&3) inside -- a trilinear footprint touches one brick
                              // most of the time whatever the direction of the ray (an x-fastest array is fast only along x:
                              // 2.0 .. 5.3 ms per 1024^2 frame of a 512^3 volume depending on the view, r01)
    int res[3];               // X, Y, Z
    int bricks[2];            // bricks along x and y
    float boxMin[3], boxSize[3];
    int interpolation;        // fvsrn_volume_interpolation
    int source;               // fvsrn_volume_source
    int newBehavior;          // grid_resolution_new_behavior: world -> object scale = resolution instead of resolution - 1
    int provideNormals;       // volumeShouldProvideNormals (shading BRDF, normal channel): central-difference gradients
};

__device__ __forceinline__ int vol_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float vol_fetch(const VolumeParams& V, int x, int y, int z) {
    x = vol_clampi(x, 0, V.res[0] - 1); y = vol_clampi(y, 0, V.res[1] - 1); z = vol_clampi(z, 0, V.res[2] - 1);
    const size_t brick = (size_t(z >> 2) * size_t(V.bricks[1]) + size_t(y >> 2)) * size_t(V.bricks[0]) + size_t(x >> 2);
    return V.data[brick * 64 + size_t(((z & 3) << 4) | ((y & 3) << 2) | (x & 3))];
}
__device__ __forceinline__ float vol_lerp(float a, float b, float t) { return a + t * (b - a); }  // helper_math.cuh lerp

// sampleNearest(make_int3(round(posObject))) (:88-101,189): tensor branch clamps the index; a point-filtered texture
// fetch at integer coordinates with clamp addressing returns the same texel
__device__ __forceinline__ float vol_sample_nearest(const VolumeParams& V, float x, float y, float z) {
    return vol_fetch(V, int(roundf(x)), int(roundf(y)), int(roundf(z)));
}

// sampleLinear (:102-139)
__device__ __forceinline__ float vol_sample_linear(const VolumeParams& V, float x, float y, float z) {
    if (V.source == 1) {  // tensor branch: nodes at integer coordinates, ipos = make_int3(posObject) truncates
        const int ix = int(x), iy = int(y), iz = int(z);
        const float fx = x - float(ix), fy = y - float(iy), fz = z - float(iz);
        const float d000 = vol_fetch(V, ix, iy, iz), d001 = vol_fetch(V, ix, iy, iz + 1);
        const float d010 = vol_fetch(V, ix, iy + 1, iz), d011 = vol_fetch(V, ix, iy + 1, iz + 1);
        const float d100 = vol_fetch(V, ix + 1, iy, iz), d101 = vol_fetch(V, ix + 1, iy, iz + 1);
        const float d110 = vol_fetch(V, ix + 1, iy + 1, iz), d111 = vol_fetch(V, ix + 1, iy + 1, iz + 1);
        return vol_lerp(vol_lerp(vol_lerp(d000, d100, fx), vol_lerp(d010, d110, fx), fy),
                        vol_lerp(vol_lerp(d001, d101, fx), vol_lerp(d011, d111, fx), fy), fz);
    }
    // tex3D, cudaFilterModeLinear, un-normalised coordinates, clamp addressing (CUDA programming guide, "Linear
    // Filtering"): xB = x - 0.5, i = floor(xB), alpha = frac(xB) in 1.8 fixed point
    const float xb = x - 0.5f, yb = y - 0.5f, zb = z - 0.5f;
    const float fxi = floorf(xb), fyi = floorf(yb), fzi = floorf(zb);
    const float a = rintf((xb - fxi) * 256.f) * (1.f / 256.f), b = rintf((yb - fyi) * 256.f) * (1.f / 256.f),
                c = rintf((zb - fzi) * 256.f) * (1.f / 256.f);
    const int i = int(fxi), j = int(fyi), k = int(fzi);
    return (1 - a) * (1 - b) * (1 - c) * vol_fetch(V, i, j, k) + a * (1 - b) * (1 - c) * vol_fetch(V, i + 1, j, k) +
           (1 - a) * b * (1 - c) * vol_fetch(V, i, j + 1, k) + a * b * (1 - c) * vol_fetch(V, i + 1, j + 1, k) +
           (1 - a) * (1 - b) * c * vol_fetch(V, i, j, k + 1) + a * (1 - b) * c * vol_fetch(V, i + 1, j, k + 1) +
           (1 - a) * b * c * vol_fetch(V, i, j + 1, k + 1) + a * b * c * vol_fetch(V, i + 1, j + 1, k + 1);
}

// bspline_weights + sampleCubic (:141-186): eight linear fetches
__device__ __forceinline__ void vol_bspline(float f, float& w0, float& w1, float& w2, float& w3) {
    const float one_frac = 1.0f - f, squared = f * f, one_sqd = one_frac * one_frac;
    w0 = 1.0f / 6.0f * one_sqd * one_frac;
    w1 = 2.0f / 3.0f - 0.5f * squared * (2.0f - f);
    w2 = 2.0f / 3.0f - 0.5f * one_sqd * (2.0f - one_frac);
    w3 = 1.0f / 6.0f * squared * f;
}
__device__ __forceinline__ float vol_sample_cubic(const VolumeParams& V, float x, float y, float z) {
    const float c[3] = {x - 0.5f, y - 0.5f, z - 0.5f};
    float g0[3], g1[3], h0[3], h1[3];
    for (int d = 0; d < 3; ++d) {
        const float index = floorf(c[d]), fraction = c[d] - index;
        float w0, w1, w2, w3;
        vol_bspline(fraction, w0, w1, w2, w3);
        g0[d] = w0 + w1;
        g1[d] = w2 + w3;
        h0[d] = (w1 / g0[d]) - 0.5f + index;
        h1[d] = (w3 / g1[d]) + 1.5f + index;
    }
    // eight linear fetches, weighted along x, then y, then z (the order of the reference's sums)
    float alongZ[2];
    for (int kz = 0; kz < 2; ++kz) {
        const float z = kz ? h1[2] : h0[2];
        float alongY[2];
        for (int ky = 0; ky < 2; ++ky) {
            const float y = ky ? h1[1] : h0[1];
            alongY[ky] = g0[0] * vol_sample_linear(V, h0[0], y, z) + g1[0] * vol_sample_linear(V, h1[0], y, z);
        }
        alongZ[kz] = g0[1] * alongY[0] + g1[1] * alongY[1];
    }
    return g0[2] * alongZ[0] + g1[2] * alongZ[1];
}

// eval (:193-232): world position -> object coordinates [0, res-1] (old behaviour) or [0, res] (new), then sample()
__device__ __forceinline__ void vol_to_object(const VolumeParams& V, float wx, float wy, float wz, float p[3]) {
    const float w[3] = {wx, wy, wz};
    for (int d = 0; d < 3; ++d) {
        const float scale = float(V.newBehavior ? V.res[d] : V.res[d] - 1);
        p[d] = (w[d] - V.boxMin[d]) / V.boxSize[d] * scale;
    }
}
__device__ __forceinline__ float vol_sample(const VolumeParams& V, float x, float y, float z) {
    if (V.interpolation == 0) return vol_sample_nearest(V, x, y, z);
    if (V.interpolation == 1) return vol_sample_linear(V, x, y, z);
    return vol_sample_cubic(V, x, y, z);
}
__device__ __forceinline__ float vol_eval(const VolumeParams& V, float wx, float wy, float wz) {
    float p[3];
    vol_to_object(V, wx, wy, wz, p);
    return vol_sample(V, p[0], p[1], p[2]);
}
// evalNormalImpl (:234-283): central differences one voxel to either side (normalStep = 1), scaled by 0.5 / voxelSize with
// voxelSize = boxSize / (resolution - 1 | resolution) (volume_interpolation_grid.cpp:1097-1104)
__device__ __forceinline__ void vol_normal(const VolumeParams& V, float wx, float wy, float wz, float n[3]) {
    float p[3];
    vol_to_object(V, wx, wy, wz, p);
    for (int d = 0; d < 3; ++d) {
        const float voxel = V.boxSize[d] / float(V.newBehavior ? V.res[d] : V.res[d] - 1);
        const float scale = 0.5f / voxel;
        float a[3] = {p[0], p[1], p[2]}, b[3] = {p[0], p[1], p[2]};
        a[d] += 1.f;
        b[d] -= 1.f;
        n[d] = scale * (vol_sample(V, a[0], a[1], a[2]) - vol_sample(V, b[0], b[1], b[2]));
    }
}

hipError_t launch_volume_evaluate(const VolumeParams& V, const float* pos, size_t n, float* out, hipStream_t s);
hipError_t launch_volume_render(const VolumeParams& V, const SceneParams& S, float* out, unsigned long long* stats, size_t tfFloats,
                                hipStream_t s);

}  // namespace fvsrn
