// DVR of a dense grid volume: ImageEvaluatorSimpleKernel + RayEvaluationSteppingDvr with kernel::VolumeInterpolationGrid as
// the volume (reference renderer_image_evaluator_simple.cuh:36-127, renderer_ray_evaluation_stepping_dvr.cuh:48-157,
// renderer_volume_grid.cuh).  One thread per pixel, 8x8 pixel tiles per wave (neighbouring rays read neighbouring voxels);
// the kernel is bound by the voxel gathers (8 per trilinear sample, L2/HBM), not by arithmetic.
#include "kernels.hpp"
#include "grid_volume.hpp"

namespace fvsrn {

__global__ __launch_bounds__(256) void volume_evaluate_kernel(VolumeParams V, const float* __restrict__ pos, size_t n,
                                                             float* __restrict__ out) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        out[i] = vol_eval(V, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
}

// SHADE: normals from the grid, shading BRDF, pre-integrated TFs (compiled out of the plain renderer: -11 % otherwise, r01)
template <bool SHADE>
__global__ __launch_bounds__(256) void volume_render_kernel(VolumeParams V, SceneParams S, float* __restrict__ out,
                                                           unsigned long long* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float tfLds[];
    {
        const int cols = S.tfKind == FVSRN_TF_GAUSSIAN ? 6 : (S.tfKind == FVSRN_TF_PIECEWISE ? 5 : (S.tfKind == FVSRN_TF_TEXTURE ? 4 : 0));
        for (int i = threadIdx.x; i < cols * S.tfRows; i += int(blockDim.x)) tfLds[i] = S.tfTable[i];
        __syncthreads();
    }
    // 256 threads = 4 waves, each an 8x8 pixel tile of a 16x16 block
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tilesX = (S.width + 15) >> 4;
    const int bx = blockIdx.x % tilesX, by = blockIdx.x / tilesX;
    // depth segments like render_body (kernels.hpp): blockIdx.y = segment of the ray's step range, partial results composited
    // front to back by composite_kernel; a small image (BASELINE configs[0]: 256^2 = one wave per SIMD) is bound by the latency of
    // its serial gathers otherwise
    const int K = S.segments, seg = int(blockIdx.y);
    const int x = bx * 16 + (wave & 1) * 8 + (lane & 7), y = by * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inImage = x < S.width && y < S.height;

    const float ndcx = 2.f * (float(x) + 0.5f) / float(S.width) - 1.f;
    const float ndcy = 2.f * (float(y) + 0.5f) / float(S.height) - 1.f;
    float dx = S.front[0] + ndcx * S.tanFovX * S.right[0] + ndcy * S.tanFovY * S.up[0];
    float dy = S.front[1] + ndcx * S.tanFovX * S.right[1] + ndcy * S.tanFovY * S.up[1];
    float dz = S.front[2] + ndcx * S.tanFovX * S.right[2] + ndcy * S.tanFovY * S.up[2];
    const float invLen = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= invLen; dy *= invLen; dz *= invLen;
    const float ox = S.eye[0], oy = S.eye[1], oz = S.eye[2];
    float tmin, tmax;
    {
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
        const float t1 = (V.boxMin[0] - ox) * ix, t2 = (V.boxMin[0] + V.boxSize[0] - ox) * ix;
        const float t3 = (V.boxMin[1] - oy) * iy, t4 = (V.boxMin[1] + V.boxSize[1] - oy) * iy;
        const float t5 = (V.boxMin[2] - oz) * iz, t6 = (V.boxMin[2] + V.boxSize[2] - oz) * iz;
        tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    }
    tmin = fmaxf(tmin, 0.f);
    if (!inImage) tmax = -1.f;

    float cr = 0, cg = 0, cb = 0, ca = 0, depth = 0, nx = 0, ny = 0, nz = 0;
    float previousDensity = -1.f;  // pre-integrated transfer functions, stepping_dvr.cuh:81
    unsigned count = 0;
    int i0 = 0, i1 = 0x7fffffff;
    if (K > 1) {
        const float span = tmax - tmin;
        const int n = span >= 0.f ? int(span / S.stepsize) + 1 : 0;
        i0 = (n * seg) / K;
        if (seg + 1 < K) i1 = (n * (seg + 1)) / K;
    }
    for (int i = i0;; ++i) {
        const float t = tmin + float(i) * S.stepsize;
        const bool valid = (t <= tmax) && (i < i1) && (!S.earlyOut || ca < S.alphaEarlyOut);
        if (!valid) break;  // per-lane view of the reference's warp-synchronous loop: an invalid lane never blends again
        ++count;
        const float wx = ox + dx * t, wy = oy + dy * t, wz = oz + dz * t;
        const float value = vol_eval(V, wx, wy, wz);
        const float density2 = (value - S.densityMin) * S.divDensityRange;
        const float prev = previousDensity;
        previousDensity = density2;  // stepping_dvr.cuh:135
        if (value >= S.densityMin) {  // :110-135
            float g[3] = {0.f, 0.f, 0.f};
            if (SHADE && V.provideNormals) vol_normal(V, wx, wy, wz, g);  // :122-128 (requireNormal)
            float4_t color = SHADE && S.tfPreintegration != FVSRN_PREINTEGRATE_NONE ? tf_eval_preintegrated(S, tfLds, fminf(fmaxf(density2, 0.f), 1.f), prev)
                                                                                  : tf_eval(S, tfLds, density2, sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), prev);
            if (SHADE && color[3] > 0.f && (S.brdfMagnitudeScaling | S.brdfPhong)) {  // BRDFLambert::eval, renderer_brdf_lambert.cuh:56-103
                const float g2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
                if (S.brdfMagnitudeScaling) color[3] *= 1.f - __expf(-S.brdfMagScale * g2);
                if (S.brdfPhong) {
                    const float gradientNorm = rsqrtf(g2);
                    float nX = g[0], nY = g[1], nZ = g[2];
                    if (g2 >= 1e-8f) { nX *= gradientNorm; nY *= gradientNorm; nZ *= gradientNorm; }
                    float lx, ly, lz;
                    if (S.brdfLightType == FVSRN_LIGHT_DIRECTIONAL) { lx = -S.brdfLight[0]; ly = -S.brdfLight[1]; lz = -S.brdfLight[2]; }
                    else { lx = S.brdfLight[0] - wx; ly = S.brdfLight[1] - wy; lz = S.brdfLight[2] - wz; }
                    const float il = rsqrtf(lx * lx + ly * ly + lz * lz);
                    lx *= il; ly *= il; lz *= il;
                    const float lo = S.brdfMagCenter - S.brdfMagRadius, hi = S.brdfMagCenter + S.brdfMagRadius;
                    const float ys = fminf(fmaxf((gradientNorm - lo) / (hi - lo), 0.f), 1.f);
                    const float phongStrength = ys * ys * (3.f - 2.f * ys);
                    const float ambientStrength = 1.f + phongStrength * (S.brdfAmbient - 1.f);
                    const float nl = nX * lx + nY * ly + nZ * lz;
                    const float rx = lx - 2.f * nX * nl, ry = ly - 2.f * nY * nl, rz = lz - 2.f * nZ * nl;
                    const float e = float(S.brdfSpecularExponent);
                    const float spec = (e + 2.f) * 0.159155f * powf(fmaxf(0.f, dx * rx + dy * ry + dz * rz), e);
                    for (int c = 0; c < 3; ++c)
                        color[c] = ambientStrength * color[c] + (1.f - ambientStrength) * (fabsf(nl) * color[c] + S.brdfSpecular * spec);
                }
            }
            if (color[3] > 0.f) {  // Blending::eval
                const float a = S.blendMode == FVSRN_BLEND_BEER_LAMBERT ? 1.f - __expf(-color[3]) : fminf(1.f, color[3]);
                const float w = (1.f - ca) * a;
                cr += w * color[0]; cg += w * color[1]; cb += w * color[2];
                depth += w * t;
                ca += w;
                if (SHADE && V.provideNormals) {
                    const float l2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
                    if (l2 >= 1e-8f) { const float il = rsqrtf(l2); g[0] *= il; g[1] *= il; g[2] *= il; }  // safeNormalize
                    nx += w * g[0]; ny += w * g[1]; nz += w * g[2];
                }
            }
        }
    }
    if (inImage && K > 1) {  // raw accumulators of this segment
        const size_t plane = size_t(S.width) * S.height;
        float* p = S.partial + size_t(seg) * 8 * plane + size_t(y) * S.width + x;
        p[0] = cr; p[plane] = cg; p[2 * plane] = cb; p[3 * plane] = ca;
        p[4 * plane] = nx; p[5 * plane] = ny; p[6 * plane] = nz; p[7 * plane] = depth;
    } else if (inImage) {
        const size_t plane = size_t(S.width) * S.height, o = size_t(y) * S.width + x;
        out[o] = cr; out[plane + o] = cg; out[2 * plane + o] = cb; out[3 * plane + o] = ca;
        out[4 * plane + o] = nx * ca; out[5 * plane + o] = ny * ca; out[6 * plane + o] = nz * ca;
        out[7 * plane + o] = depth * ca / ca;
    }
    if (stats) {
        unsigned sum = count, mx = count;
        for (int off = 32; off > 0; off >>= 1) {
            sum += __shfl_down(sum, off);
            const unsigned other = __shfl_down(mx, off);
            mx = other > mx ? other : mx;
        }
        if (lane == 0) {
            atomicAdd(&stats[0], (unsigned long long)sum);
            atomicAdd(&stats[1], (unsigned long long)mx * 64ull);
        }
    }
}

hipError_t launch_volume_evaluate(const VolumeParams& V, const float* pos, size_t n, float* out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned grid = unsigned(std::min<size_t>((n + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(volume_evaluate_kernel, dim3(grid), dim3(256), 0, s, V, pos, n, out);
    return hipGetLastError();
}

hipError_t launch_volume_render(const VolumeParams& V, const SceneParams& S, float* out, unsigned long long* stats, size_t tfFloats,
                                hipStream_t s) {
    const dim3 grid(unsigned(((S.width + 15) / 16) * ((S.height + 15) / 16)), unsigned(std::max(S.segments, 1)));
    const bool shade = V.provideNormals || S.brdfMagnitudeScaling || S.brdfPhong || S.tfPreintegration != FVSRN_PREINTEGRATE_NONE ||
                       S.tfGaussianMode != FVSRN_TF_GAUSSIAN_PLAIN;
    if (shade) hipLaunchKernelGGL(volume_render_kernel<true>, grid, dim3(256), tfFloats * 4, s, V, S, out, stats);
    else hipLaunchKernelGGL(volume_render_kernel<false>, grid, dim3(256), tfFloats * 4, s, V, S, out, stats);
    return hipGetLastError();
}

}  // namespace fvsrn
