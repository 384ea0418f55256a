#include "launch.hpp"
#include "launch_host.hpp"

#include "kernels.hpp"

#include <algorithm>

namespace fvsrn {

// One thread per fp16 output element.  Time channels: lerp(decodeA(rawA), decodeA(rawB), frac) -- key frame B is
// decoded with A's offset/scale exactly like the reference (renderer_volume_tensorcores.cuh:586-591); ensemble
// channels: decode only (volume_interpolation_network.cpp:1332-1350).
__global__ void grid_blend_kernel(BlendParams p) {
    const int G = p.Gt + p.Ge;
    const unsigned long long n = p.records * (unsigned long long)(2 * G);
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long rec = i / (unsigned)(2 * G);
        const int w = int(i % (unsigned)(2 * G));
        const int c = w >> 1, pp = w & 1;
        float v;
        if (p.enc == FVSRN_GRID_BYTE_GAUSSIAN) {  // no decode, no blend: byte values of A and B for the render kernel
            float a, b;
            if (c < p.Gt) {
                const unsigned long long idx = rec * (unsigned)(2 * p.Gt) + (unsigned)(2 * c + pp);
                a = float(static_cast<const unsigned char*>(p.timeLo)[idx]);
                b = float(static_cast<const unsigned char*>(p.timeHi)[idx]);
            } else {
                const int ce = c - p.Gt;
                const unsigned long long per = p.records * (unsigned long long)(2 * p.Ge);
                a = b = float(static_cast<const unsigned char*>(p.ensData)[p.ens * per + rec * (unsigned)(2 * p.Ge) + (unsigned)(2 * ce + pp)]);
            }
            static_cast<_Float16*>(p.out)[i] = _Float16(a);
            static_cast<_Float16*>(p.outB)[i] = _Float16(b);
            continue;
        }
        if (c < p.Gt) {
            const unsigned long long idx = rec * (unsigned)(2 * p.Gt) + (unsigned)(2 * c + pp);
            float a, b;
            if (p.enc == FVSRN_GRID_FLOAT) {
                a = static_cast<const float*>(p.timeLo)[idx];
                b = static_cast<const float*>(p.timeHi)[idx];
            } else {  // key frame B is decoded with A's coefficients (renderer_volume_tensorcores.cuh:586-587)
                const float off = p.timeOffset[p.lo * p.Gt + c], sc = p.timeScale[p.lo * p.Gt + c];
                a = off + (static_cast<const unsigned char*>(p.timeLo)[idx] / 255.0f) * sc;
                b = off + (static_cast<const unsigned char*>(p.timeHi)[idx] / 255.0f) * sc;
            }
            v = a + p.frac * (b - a);
        } else {
            const int ce = c - p.Gt;
            const unsigned long long per = p.records * (unsigned long long)(2 * p.Ge);
            const unsigned long long idx = rec * (unsigned)(2 * p.Ge) + (unsigned)(2 * ce + pp);
            if (p.enc == FVSRN_GRID_FLOAT)
                v = static_cast<const float*>(p.ensData)[p.ens * per + idx];
            else
                v = p.ensOffset[p.ens * p.Ge + ce] +
                    (static_cast<const unsigned char*>(p.ensData)[p.ens * per + idx] / 255.0f) * p.ensScale[p.ens * p.Ge + ce];
        }
        static_cast<_Float16*>(p.out)[i] = _Float16(v);
    }
}

// ---- IImageEvaluator::ExtractColor (renderer/iimage_evaluator.cpp:26-135) --------------------------------------------
// min / max of the depth plane with the NaN semantics of torch's min()/max() (any NaN -> NaN): per-block reduction,
// then one atomic per block on an order-preserving integer image of the float; a NaN poisons both results.
__device__ __forceinline__ unsigned orderedBits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fromOrderedBits(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// minmaxBits: {min as ordered bits (init 0xffffffff), max (init 0), nan flag}
__global__ void depth_minmax_kernel(const float* __restrict__ depth, unsigned long long n, unsigned* __restrict__ minmaxBits) {
    unsigned mn = 0xffffffffu, mx = 0u, nan = 0u;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float v = depth[i];
        if (v != v) nan = 1u;
        else { const unsigned k = orderedBits(v); mn = min(mn, k); mx = max(mx, k); }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, unsigned(__shfl_xor(int(mn), off)));
        mx = max(mx, unsigned(__shfl_xor(int(mx), off)));
        nan |= unsigned(__shfl_xor(int(nan), off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&minmaxBits[0], mn);
        atomicMax(&minmaxBits[1], mx);
        if (nan) atomicOr(&minmaxBits[2], 1u);
    }
}

__device__ __forceinline__ float tonemap_channel(float v, float maxExposure) {  // iimage_evaluator_cuda.cu:144-165
    v /= maxExposure;
    v = (v * (2.51f * v + 0.03f)) / (v * (2.43f * v + 0.59f) + 0.14f);  // ACES filmic curve
    v = fminf(fmaxf(v, 0.f), 1.f);
    return powf(v, 1.0f / 2.4f);
}
__device__ __forceinline__ unsigned rgba_to_int(float r, float g, float b, float a) {  // renderer_utils.cuh:48-57
    r = fminf(fmaxf(r * 255.f, 0.f), 255.f);
    g = fminf(fmaxf(g * 255.f, 0.f), 255.f);
    b = fminf(fmaxf(b * 255.f, 0.f), 255.f);
    a = fminf(fmaxf(a * 255.f, 0.f), 255.f);
    return (unsigned(a) << 24) | (unsigned(b) << 16) | (unsigned(g) << 8) | unsigned(r);
}

// {min, max, nan flag} as ordered bits -> three floats {-min, max, nan flag}: a form that merges by an element-wise MAXIMUM, i.e. by one all-reduce when the
// image is spread over the ranks of a multi-GPU frame (fvsrn_depth_range).  No finite depth at all: {-inf, -inf} like an empty maximum.
__global__ void depth_range_kernel(const unsigned* __restrict__ minmaxBits, float* __restrict__ range3) {
    const bool any = minmaxBits[0] <= minmaxBits[1];
    range3[0] = any ? -fromOrderedBits(minmaxBits[0]) : -__builtin_inff();
    range3[1] = any ? fromOrderedBits(minmaxBits[1]) : -__builtin_inff();
    range3[2] = minmaxBits[2] ? 1.f : 0.f;
}

__global__ void extract_color_kernel(ExtractParams p, const unsigned* __restrict__ minmaxBits) {
    const unsigned long long n = p.pixels;
    float scaleRGB = 1.f, offsetRGB = 0.f, scaleA = 1.f, offsetA = 0.f;
    int c0 = 0, c1 = 1, c2 = 2, ca = 3;
    switch (p.mode) {  // iimage_evaluator.cpp:56-113
        case FVSRN_CHANNEL_DEPTH: {
            float mn, mx;
            if (p.range3) {  // the caller's range (merged over the ranks of a multi-GPU frame): {-min, max, nan flag}
                if (p.range3[2] != 0.f) { mn = mx = __uint_as_float(0x7fc00000u); }
                else { mn = -p.range3[0]; mx = p.range3[1]; }
            } else if (minmaxBits[2]) { mn = mx = __uint_as_float(0x7fc00000u); }
            else { mn = fromOrderedBits(minmaxBits[0]); mx = fromOrderedBits(minmaxBits[1]); }
            c0 = c1 = c2 = 7;
            scaleRGB = 1.f / (mx - mn);
            offsetRGB = -mn / (mx - mn);
            scaleA = 0.f; offsetA = 1.f;
        } break;
        case FVSRN_CHANNEL_MASK: c0 = c1 = c2 = 3; scaleA = 0.f; offsetA = 1.f; break;
        case FVSRN_CHANNEL_NORMAL: c0 = 4; c1 = 5; c2 = 6; scaleRGB = 0.5f; offsetRGB = 0.5f; break;
        default: break;
    }
    const bool tm = p.mode == FVSRN_CHANNEL_COLOR && p.tonemap;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        float r = p.raw[c0 * n + i], g = p.raw[c1 * n + i], b = p.raw[c2 * n + i], a = p.raw[ca * n + i];
        if (tm) {
            r = tonemap_channel(r, p.maxExposure); g = tonemap_channel(g, p.maxExposure); b = tonemap_channel(b, p.maxExposure);
        } else {
            r = r * scaleRGB + offsetRGB; g = g * scaleRGB + offsetRGB; b = b * scaleRGB + offsetRGB;
            a = a * scaleA + offsetA;
        }
        if (p.out4) { p.out4[i] = r; p.out4[n + i] = g; p.out4[2 * n + i] = b; p.out4[3 * n + i] = a; }
        else p.out8[i] = rgba_to_int(r, g, b, a);
    }
}

hipError_t launch_extract_color(const ExtractParams& p, hipStream_t s) {
    unsigned* bits = reinterpret_cast<unsigned*>(p.minmax);
    const unsigned grid = unsigned(std::min<unsigned long long>((p.pixels + 255) / 256, 4096ull));
    if (p.mode == FVSRN_CHANNEL_DEPTH && !p.range3) {
        const unsigned init[3] = {0xffffffffu, 0u, 0u};
        hipError_t e = hipMemcpyAsync(bits, init, sizeof(init), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(depth_minmax_kernel, dim3(grid), dim3(256), 0, s, p.raw + 7 * p.pixels, p.pixels, bits);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(extract_color_kernel, dim3(grid), dim3(256), 0, s, p, bits);
    return hipGetLastError();
}

hipError_t launch_depth_range(const float* depth, unsigned long long pixels, float* scratchBits, float* range3, hipStream_t s) {
    unsigned* bits = reinterpret_cast<unsigned*>(scratchBits);
    const unsigned grid = unsigned(std::min<unsigned long long>((pixels + 255) / 256, 4096ull));
    const unsigned init[3] = {0xffffffffu, 0u, 0u};
    hipError_t e = hipMemcpyAsync(bits, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(depth_minmax_kernel, dim3(grid), dim3(256), 0, s, depth, pixels, bits);
    hipLaunchKernelGGL(depth_range_kernel, dim3(1), dim3(1), 0, s, bits, range3);
    return hipGetLastError();
}

// ---- composite of depth segments ---------------------------------------------------------------------------------------
// Segment s holds sum_j w_j c_j with weights that start from alpha 0; the weight of its samples in the whole ray is
// (1 - A_before) times that (Blending::eval, renderer_blending.cuh:35-51), so the segments blend like single samples.
// Rows outside [y0, y1) of a non-compact image are not touched (the render kernel does not write them either).
__global__ void composite_kernel(const float* __restrict__ partial, float* __restrict__ out, int K, unsigned long long plane, int width,
                                 int rowBegin, int rowEnd) {
    const unsigned long long first = (unsigned long long)rowBegin * width, last = (unsigned long long)rowEnd * width;
    for (unsigned long long o = first + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; o < last;
         o += (unsigned long long)gridDim.x * blockDim.x) {
        float c[3] = {0, 0, 0}, n[3] = {0, 0, 0}, depth = 0, A = 0;
        for (int s = 0; s < K; ++s) {
            const float* p = partial + (unsigned long long)s * 8 * plane + o;
            const float w = 1.f - A;
            c[0] += w * p[0]; c[1] += w * p[plane]; c[2] += w * p[2 * plane];
            n[0] += w * p[4 * plane]; n[1] += w * p[5 * plane]; n[2] += w * p[6 * plane];
            depth += w * p[7 * plane];
            A += w * p[3 * plane];
        }
        out[o] = c[0]; out[plane + o] = c[1]; out[2 * plane + o] = c[2]; out[3 * plane + o] = A;
        out[4 * plane + o] = n[0] * A; out[5 * plane + o] = n[1] * A; out[6 * plane + o] = n[2] * A;
        out[7 * plane + o] = depth * A / A;
    }
}

hipError_t launch_composite(const float* partial, float* out, int segments, unsigned long long plane, const SceneParams& S,
                            hipStream_t s) {
    const int rowBegin = S.compact ? 0 : S.y0, rowEnd = S.compact ? S.numLocalRows : S.y1;
    const unsigned long long n = (unsigned long long)(rowEnd - rowBegin) * S.width;
    const unsigned grid = unsigned(std::min<unsigned long long>((n + 255) / 256, 8192ull));
    hipLaunchKernelGGL(composite_kernel, dim3(std::max(grid, 1u)), dim3(256), 0, s, partial, out, segments, plane, S.width, rowBegin, rowEnd);
    return hipGetLastError();
}

// ---- pre-integration tables of a Texture TF ------------------------------------------------------------------------------
// tex1D with linear filtering, normalized coordinates and clamp addressing on an [R][4] table
__device__ __forceinline__ float4 tf_tex1d(const float* __restrict__ tex, int R, float x) {
    const float d = x * R - 0.5f;
    const int di = int(floorf(d));
    const float f = d - di;
    const float* a = tex + 4 * min(max(di, 0), R - 1);
    const float* b = tex + 4 * min(max(di + 1, 0), R - 1);
    return make_float4(a[0] + f * (b[0] - a[0]), a[1] + f * (b[1] - a[1]), a[2] + f * (b[2] - a[2]), a[3] + f * (b[3] - a[3]));
}

__global__ void tf_preintegrate_1d_kernel(const float* __restrict__ tex, float* __restrict__ out, int R) {  // :9-36, serial like the reference
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float4 integral = make_float4(0, 0, 0, 0);
    float lastDensity = 0.f;
    float4 last = tf_tex1d(tex, R, lastDensity);
    for (int i = 0; i < R; ++i) {
        const float cur = (float(i) + 0.5f) / float(R);
        const float4 v = tf_tex1d(tex, R, cur);
        const float w = cur - lastDensity;
        integral.x += w * 0.5f * (last.x * last.w + v.x * v.w);
        integral.y += w * 0.5f * (last.y * last.w + v.y * v.w);
        integral.z += w * 0.5f * (last.z * last.w + v.z * v.w);
        integral.w += w * 0.5f * (last.w + v.w);
        out[4 * i + 0] = integral.x; out[4 * i + 1] = integral.y; out[4 * i + 2] = integral.z; out[4 * i + 3] = integral.w;
        last = v;
        lastDensity = cur;
    }
}

__global__ void tf_preintegrate_2d_kernel(const float* __restrict__ tex, float* __restrict__ out, int R, float stepsize, int N) {  // :50-79
    const int istart = blockIdx.x * blockDim.x + threadIdx.x, iend = blockIdx.y;
    if (istart >= R || iend >= R) return;
    const float dstart = (float(istart) + 0.5f) / float(R), dend = (float(iend) + 0.5f) / float(R);
    float r = 0, g = 0, b = 0, alphaSum = 0;
    const float h = 1.0f / float(N);
    for (int i = 1; i <= N; ++i) {  // Riemann sum along the density segment
        const float omega = i * h;
        const float4 v = tf_tex1d(tex, R, (1 - omega) * dstart + omega * dend);
        alphaSum += v.w * h * stepsize;
        const float k = h * v.w * stepsize * expf(-alphaSum);
        r += k * v.x; g += k * v.y; b += k * v.z;
    }
    float* o = out + 4 * (size_t(iend) * R + istart);  // surf2Dwrite(x = istart, y = iend)
    o[0] = r; o[1] = g; o[2] = b; o[3] = 1 - expf(-alphaSum);
}

hipError_t launch_tf_preintegration(const float* tex, float* out, int R, int mode, float stepsize, int quadratureSteps, hipStream_t s) {
    if (mode == 1) hipLaunchKernelGGL(tf_preintegrate_1d_kernel, dim3(1), dim3(64), 0, s, tex, out, R);
    else hipLaunchKernelGGL(tf_preintegrate_2d_kernel, dim3((R + 63) / 64, R), dim3(64), 0, s, tex, out, R, stepsize, quadratureSteps);
    return hipGetLastError();
}

// ---- tensor APIs of the camera and the transfer functions ------------------------------------------------------------------
// CameraGenerateRayKernel (renderer_camera_kernels.cuh:12-43) with CameraReferenceFrame::eval (renderer_camera.cuh:33-52)
__global__ void generate_rays_kernel(SceneParams S, float* __restrict__ rayStart, float* __restrict__ rayDir) {
    const size_t n = size_t(S.width) * S.height;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int x = int(i % size_t(S.width)), y = int(i / size_t(S.width));
        const float ndcx = 2.f * (float(x) + 0.5f) / float(S.width) - 1.f;
        const float ndcy = 2.f * (float(y) + 0.5f) / float(S.height) - 1.f;
        float dx = S.front[0] + ndcx * S.tanFovX * S.right[0] + ndcy * S.tanFovY * S.up[0];
        float dy = S.front[1] + ndcx * S.tanFovX * S.right[1] + ndcy * S.tanFovY * S.up[1];
        float dz = S.front[2] + ndcx * S.tanFovX * S.right[2] + ndcy * S.tanFovY * S.up[2];
        const float invLen = rsqrtf(dx * dx + dy * dy + dz * dz);
        rayStart[3 * i + 0] = S.eye[0]; rayStart[3 * i + 1] = S.eye[1]; rayStart[3 * i + 2] = S.eye[2];
        rayDir[3 * i + 0] = dx * invLen; rayDir[3 * i + 1] = dy * invLen; rayDir[3 * i + 2] = dz * invLen;
    }
}
hipError_t launch_generate_rays(const SceneParams& S, float* rayStart, float* rayDir, hipStream_t s) {
    const size_t n = size_t(S.width) * S.height;
    hipLaunchKernelGGL(generate_rays_kernel, dim3(unsigned(std::min<size_t>((n + 255) / 256, 8192))), dim3(256), 0, s, S, rayStart, rayDir);
    return hipGetLastError();
}

// EvaluateTF / EvaluateTFWithPrevious (renderer_tf_kernels.cuh:11-70); S.stepsize is the step size the TF multiplies in
__global__ void evaluate_tf_kernel(SceneParams S, const float* __restrict__ density, const float* __restrict__ previous, size_t n,
                                   float* __restrict__ colors) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const float d = density[i];
        float4_t c = {0, 0, 0, 0};
        if (d >= S.densityMin) {
            const float d2 = (d - S.densityMin) * S.divDensityRange;
            if (S.tfPreintegration != FVSRN_PREINTEGRATE_NONE) {
                const float p = previous ? previous[i] : -1.f;
                const float p2 = p >= 0.f ? (p - S.densityMin) * S.divDensityRange : -1.f;
                c = tf_eval_preintegrated(S, S.tfTable, fminf(fmaxf(d2, 0.f), 1.f), p2);
            } else {
                // (no gradient on this entry: the reference passes a zero normal, renderer_tf_kernels.cuh:30,61)
                const float p = previous ? previous[i] : -1.f;
                c = tf_eval(S, S.tfTable, d2, 0.f, p >= 0.f ? (p - S.densityMin) * S.divDensityRange : -1.f);
            }
        }
        colors[4 * i + 0] = c[0]; colors[4 * i + 1] = c[1]; colors[4 * i + 2] = c[2]; colors[4 * i + 3] = c[3];
    }
}
hipError_t launch_evaluate_tf(const SceneParams& S, const float* density, const float* previous, size_t n, float* colors, hipStream_t s) {
    hipLaunchKernelGGL(evaluate_tf_kernel, dim3(unsigned(std::min<size_t>((n + 255) / 256, 8192))), dim3(256), 0, s, S, density, previous, n, colors);
    return hipGetLastError();
}

// One thread per (cell, M tile, row) over the grid extended by one ghost cell per side (cell e spans the nodes e - 1, e per axis, clamped to the grid: the
// ghost nodes repeat the boundary = clamp-to-edge).  V[corner] = sum over the latent channels of (first-layer latent column of the row) x (grid value at the
// corner); the entry holds the coefficients of the interpolant in cell-centred coordinates, slot 4 c + 2 b + a = the coefficient of x^a y^b z^c:
//     coefficient = sum over the corners of V[corner] s_x^a s_y^b s_z^c / 2^(3 - a - b - c),   s = -1 / +1 for the lower / upper node  (cell_tap, srn_device.hpp)
// v(x) is the first half of the x-pair record x + 1; the fragment of (g, m) holds row r's weights of channels 16 g + 8 h .. + 7 in lane 32 h + r.
__global__ void grid_cell_table_kernel(CellTableParams p) {
    const unsigned cx = unsigned(p.X + 1), cy = unsigned(p.Y + 1), cz = unsigned(p.Z + 1);
    const unsigned long long total = (unsigned long long)cx * cy * cz * unsigned(p.MT) * 32ull;
    const int KG = p.G / 16;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned r = unsigned(i & 31);
        const unsigned m = unsigned((i >> 5) % unsigned(p.MT));
        const unsigned long long cell = (i >> 5) / unsigned(p.MT);
        const int ex = int(cell % cx), ey = int((cell / cx) % cy), ez = int(cell / (cx * (unsigned long long)cy));
        float V[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // corner 4 dz + 2 dy + dx
        for (int g = 0; g < KG; ++g)
            for (int h = 0; h < 2; ++h) {
                const _Float16* w = reinterpret_cast<const _Float16*>(static_cast<const char*>(p.latentFrags) + (size_t(g) * p.MT + m) * kFragBytes +
                                                                      (32 * h + r) * 16);
                for (int k = 0; k < 8; ++k) {
                    const int nx = min(max(ex - 1 + (k & 1), 0), p.X - 1), ny = min(max(ey - 1 + ((k >> 1) & 1), 0), p.Y - 1), nz = min(max(ez - 1 + (k >> 2), 0), p.Z - 1);
                    const unsigned long long rec = ((unsigned long long)nz * unsigned(p.Y) + unsigned(ny)) * unsigned(p.X + 1) + unsigned(nx + 1);
                    const _Float16* v = reinterpret_cast<const _Float16*>(p.grid) + (rec * unsigned(p.G) + unsigned(16 * g + 8 * h)) * 2;
                    for (int j = 0; j < 8; ++j) V[k] = fmaf(float(w[j]), float(v[2 * j]), V[k]);
                }
            }
        _Float16* o = static_cast<_Float16*>(p.out) + i * 8;
        for (int s = 0; s < 8; ++s) {  // slot s: exponents a = s & 1, b = (s >> 1) & 1, c = s >> 2
            float acc = 0.f;
            for (int k = 0; k < 8; ++k) {
                const int flips = __popc(unsigned(~k & s));  // lower nodes (bit 0 of the corner) along the axes the monomial contains
                acc += (flips & 1) ? -V[k] : V[k];
            }
            o[s] = _Float16(acc * (1.0f / float(1 << (3 - __popc(unsigned(s))))));
        }
    }
}

// The corner form (r04 / r05; CellTableParams::corners): out[cell][m][row][corner (dz, dy, dx)] over the (X - 1)(Y - 1)(Z - 1) cells of the grid itself -- the table
// of the shaded kernels (cell_tap_corners, srn_device.hpp).  Records: the pair {v(x0), v(x0 + 1)} of channel c sits at dword c of record (z, y, x0 + 1).
__global__ void grid_cell_table_corners_kernel(CellTableParams p) {
    const unsigned cx = unsigned(p.X - 1), cy = unsigned(p.Y - 1), cz = unsigned(p.Z - 1);
    const unsigned long long total = (unsigned long long)cx * cy * cz * unsigned(p.MT) * 32ull;
    const int KG = p.G / 16;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned r = unsigned(i & 31);
        const unsigned m = unsigned((i >> 5) % unsigned(p.MT));
        const unsigned long long cell = (i >> 5) / unsigned(p.MT);
        const unsigned x0 = unsigned(cell % cx), y0 = unsigned((cell / cx) % cy), z0 = unsigned(cell / (cx * (unsigned long long)cy));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < KG; ++g)
            for (int h = 0; h < 2; ++h) {
                const _Float16* w = reinterpret_cast<const _Float16*>(static_cast<const char*>(p.latentFrags) + (size_t(g) * p.MT + m) * kFragBytes +
                                                                      (32 * h + r) * 16);
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long rec = ((unsigned long long)(z0 + (k >> 1)) * unsigned(p.Y) + (y0 + (k & 1))) * unsigned(p.X + 1) + (x0 + 1);
                    const _Float16* v = reinterpret_cast<const _Float16*>(p.grid) + (rec * unsigned(p.G) + unsigned(16 * g + 8 * h)) * 2;
                    for (int j = 0; j < 8; ++j) {
                        acc[2 * k] = fmaf(float(w[j]), float(v[2 * j]), acc[2 * k]);
                        acc[2 * k + 1] = fmaf(float(w[j]), float(v[2 * j + 1]), acc[2 * k + 1]);
                    }
                }
            }
        _Float16* o = static_cast<_Float16*>(p.out) + i * 8;
        for (int j = 0; j < 8; ++j) o[j] = _Float16(acc[j]);
    }
}


hipError_t launch_grid_cell_table(const CellTableParams& p, hipStream_t s) {
    const unsigned long long total = (p.corners ? (unsigned long long)(p.X - 1) * (p.Y - 1) * (p.Z - 1) : (unsigned long long)(p.X + 1) * (p.Y + 1) * (p.Z + 1)) * unsigned(p.MT) * 32ull;
    if (total == 0) return hipSuccess;
    const unsigned blocks = unsigned(std::min<unsigned long long>((total + 255) / 256, 16384ull));
    if (p.corners) hipLaunchKernelGGL(grid_cell_table_corners_kernel, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(grid_cell_table_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_grid_blend(const BlendParams& p, hipStream_t s) {
    const unsigned long long n = p.records * (unsigned long long)(2 * (p.Gt + p.Ge));
    if (n == 0) return hipSuccess;
    const unsigned blocks = unsigned(std::min<unsigned long long>((n + 255) / 256, 4096ull));
    hipLaunchKernelGGL(grid_blend_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

#define FVSRN_DISPATCH_CD(expr_prefix, ...)          \
    switch (k.CD) {                                  \
        case 1: return expr_prefix<1>(__VA_ARGS__);  \
        case 2: return expr_prefix<2>(__VA_ARGS__);  \
        case 3: return expr_prefix<3>(__VA_ARGS__);  \
        case 4: return expr_prefix<4>(__VA_ARGS__);  \
        case 5: return expr_prefix<5>(__VA_ARGS__);  \
        case 6: return expr_prefix<6>(__VA_ARGS__);  \
        case 7: return expr_prefix<7>(__VA_ARGS__);  \
        case 8: return expr_prefix<8>(__VA_ARGS__);  \
        default: break;                              \
    }

bool kernel_info(const VariantKey& k, KernelInfo* info) {
    FVSRN_DISPATCH_CD(kernel_info_cd, k, info)
    return false;
}
hipError_t launch_eval(const VariantKey& k, const EvalArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_eval_cd, k, a, gridDim, blockDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}
hipError_t launch_eval_gradient(const VariantKey& k, const EvalArgs& a, float gridStep, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_eval_gradient_cd, k, a, gridStep, gridDim, blockDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}
const void* render_cells_fn(const VariantKey& k) {
    FVSRN_DISPATCH_CD(render_cells_fn_cd, k)
    return nullptr;
}
hipError_t launch_render_cells(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_render_cells_cd, k, a, gridDim, blockDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}
const void* render_shaded_cells_fn(const VariantKey& k) {
    FVSRN_DISPATCH_CD(render_shaded_cells_fn_cd, k)
    return nullptr;
}
hipError_t launch_render_shaded_cells(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_render_shaded_cells_cd, k, a, gridDim, blockDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}
const void* render_stripe_fn(const VariantKey& k) {
    if (k.grid == 0) return nullptr;
    if (k.CD == 3) return render_fn_cd<3>(k);
    if (k.CD == 4) return render_fn_cd<4>(k);
    return nullptr;
}
hipError_t launch_render_stripe(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    if (k.grid != 0 && k.CD == 3) return launch_render_plain_cd<3>(k, a, gridDim, blockDim, ldsBytes, s);
    if (k.grid != 0 && k.CD == 4) return launch_render_plain_cd<4>(k, a, gridDim, blockDim, ldsBytes, s);
    return hipErrorInvalidDeviceFunction;
}
const void* render_adjoint_fn(const VariantKey& k) {
    if (k.CD == 2) return render_adjoint_fn_cd<2>(k);
    if (k.CD == 3) return render_adjoint_fn_cd<3>(k);
    if (k.CD == 4) return render_adjoint_fn_cd<4>(k);
    return nullptr;
}
hipError_t launch_render_adjoint(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    if (k.CD == 2) return launch_render_adjoint_cd<2>(k, a, gridDim, blockDim, ldsBytes, s);
    if (k.CD == 3) return launch_render_adjoint_cd<3>(k, a, gridDim, blockDim, ldsBytes, s);
    if (k.CD == 4) return launch_render_adjoint_cd<4>(k, a, gridDim, blockDim, ldsBytes, s);
    return hipErrorInvalidDeviceFunction;
}
hipError_t launch_render(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_render_cd, k, a, gridDim, blockDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}

// fvsrn_probe_stream_concurrency: one wave that spins for `ticks` of the 100 MHz wall clock
__global__ void spin_kernel(long long ticks, unsigned* sink) {
    const long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu && sink) *sink = x;
}
hipError_t launch_spin(long long ticks, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, static_cast<unsigned*>(nullptr));
    return hipGetLastError();
}

}  // namespace fvsrn
