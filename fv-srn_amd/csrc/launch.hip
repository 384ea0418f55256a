#include "launch.hpp"

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
                const unsigned long long per = p.records * (unsigned long long)(2 * p.Gt);
                const unsigned long long idx = rec * (unsigned)(2 * p.Gt) + (unsigned)(2 * c + pp);
                a = float(static_cast<const unsigned char*>(p.timeData)[p.lo * per + idx]);
                b = float(static_cast<const unsigned char*>(p.timeData)[p.hi * per + idx]);
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
            const unsigned long long per = p.records * (unsigned long long)(2 * p.Gt);
            const unsigned long long idx = rec * (unsigned)(2 * p.Gt) + (unsigned)(2 * c + pp);
            float a, b;
            if (p.enc == FVSRN_GRID_FLOAT) {
                a = static_cast<const float*>(p.timeData)[p.lo * per + idx];
                b = static_cast<const float*>(p.timeData)[p.hi * per + idx];
            } else {
                const float off = p.timeOffset[p.lo * p.Gt + c], sc = p.timeScale[p.lo * p.Gt + c];
                a = off + (static_cast<const unsigned char*>(p.timeData)[p.lo * per + idx] / 255.0f) * sc;
                b = off + (static_cast<const unsigned char*>(p.timeData)[p.hi * per + idx] / 255.0f) * sc;
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

hipError_t launch_grid_blend(const BlendParams& p, hipStream_t s) {
    const unsigned long long n = p.records * (unsigned long long)(2 * (p.Gt + p.Ge));
    if (n == 0) return hipSuccess;
    const unsigned blocks = unsigned(std::min<unsigned long long>((n + 255) / 256, 4096ull));
    hipLaunchKernelGGL(grid_blend_kernel, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

#define FVSRN_DISPATCH_CD(expr_prefix, ...)          \
    switch (k.CD) {                                  \
        case 2: return expr_prefix<2>(__VA_ARGS__);  \
        case 3: return expr_prefix<3>(__VA_ARGS__);  \
        case 4: return expr_prefix<4>(__VA_ARGS__);  \
        case 6: return expr_prefix<6>(__VA_ARGS__);  \
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
hipError_t launch_render(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_render_cd, k, a, gridDim, blockDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}

}  // namespace fvsrn
