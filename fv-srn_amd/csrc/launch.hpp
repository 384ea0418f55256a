// Variant dispatch: one translation unit per hidden width (CD = C/16) instantiates the kernels for
// every activation x {grid, no grid}; the host picks a variant from NetworkConfig.  This replaces
// the reference's run-time NVRTC specialisation (renderer/kernel_loader.cpp:197-273, the 13 #defines
// of SceneNetwork::getDefines) by a bounded ahead-of-time matrix.
#pragma once
#include <hip/hip_runtime.h>

#include "device_params.hpp"

namespace fvsrn {

struct EvalArgs {
    NetParams P;
    const float* pos;
    const float* dir;
    size_t n;
    float* out;
    int outChannels;
};
struct RenderArgs {
    NetParams P;
    SceneParams S;
    float* out;
    unsigned long long* stats;
    bool shaded;  // render_shaded_kernel: finite-difference normals / shading BRDF
};
struct VariantKey {
    int CD;    // hidden channels / 16
    int act;   // ACT_*
    int grid;  // 0 none, 1 decoded/blended working grid (FLOAT, BYTE_LINEAR), 2 BYTE_GAUSSIAN
    bool dir;
};
struct KernelInfo {
    const void* evalFn = nullptr;
    const void* renderFn = nullptr;
    const void* renderShadedFn = nullptr;
    const char* evalName = "";
    const char* renderName = "";
};

// implemented per CD in kernels_cd*_eval.hip; returns false if the variant is not compiled in
template <int CD> bool kernel_info_cd(const VariantKey& k, KernelInfo* info);
template <int CD> hipError_t launch_eval_cd(const VariantKey& k, const EvalArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> hipError_t launch_render_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
// the render kernels live in their own translation units (kernels_cd*_render.hip / _shaded.hip)
template <int CD> const void* render_fn_cd(const VariantKey& k);
template <int CD> const void* render_shaded_fn_cd(const VariantKey& k);
template <int CD> hipError_t launch_render_plain_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> hipError_t launch_render_shaded_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> const void* render_cells_fn_cd(const VariantKey& k);
template <int CD> const void* render_shaded_cells_fn_cd(const VariantKey& k);
template <int CD> hipError_t launch_render_shaded_cells_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> hipError_t launch_render_cells_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
// render_adjoint_kernel (kernels_cd{2,3,4}_adjoint.hip): the adjoint gradient mode up to 64 channels; other widths: nullptr / hipErrorInvalidDeviceFunction
template <int CD> const void* render_adjoint_fn_cd(const VariantKey& k);
template <int CD> hipError_t launch_render_adjoint_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> hipError_t launch_eval_gradient_cd(const VariantKey& k, const EvalArgs& a, float gridStep, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);

// (the small host-launched kernels of launch.hip -- key-frame blend, cell table, ExtractColor, depth-segment composite, TF pre-integration, ray and TF tensor
// APIs -- are declared in launch_host.hpp: the kernel translation units do not see them, so a change there rebuilds launch.hip and the host objects only)

// render_small_kernel (kernels_small_render.hip): nullptr / hipErrorInvalidDeviceFunction if that variant is not compiled in
// tail: 4 / 5 Identity / Texture TF with Beer-Lambert blending, 1 the same with Alpha blending, 2 Piecewise/Gaussian TF, 3 colour network; grid: 0 Fourier-only, 1 one decoded 16-channel latent chunk
// (grid 2: the latent grid through the cell table, kernels_small_cells.hip)
const void* render_small_fn(int act, bool dir, int numLayers, int tail, int grid = 0);
const void* render_small_cells_fn(int act, bool dir, int numLayers, int tail);
// the variants without the feature rotation (FVSRN_OPT_FOURIER_RESYNC = 1; grid 0 or 2), kernels_small_exact.hip
const void* render_small_exact_fn(int act, bool dir, int numLayers, int tail, int grid);
hipError_t launch_render_small_exact(int act, bool dir, int numLayers, int tail, int grid, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
hipError_t launch_render_small_cells(int act, bool dir, int numLayers, int tail, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
hipError_t launch_render_small(int act, bool dir, int numLayers, int tail, int grid, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);

// evaluate_small_kernel: hipErrorInvalidDeviceFunction if that variant is not compiled in
hipError_t launch_eval_small(int act, bool dir, int numLayers, int grid, const EvalArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);

bool kernel_info(const VariantKey& k, KernelInfo* info);
hipError_t launch_eval(const VariantKey& k, const EvalArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
// the 48- / 64-wide latent-grid variants of render_kernel (fragment-major layer order, kernels.hpp render_layer_schedule): the gather path of those widths; nullptr elsewhere
const void* render_stripe_fn(const VariantKey& k);
// render_cells_kernel: render_kernel with the decoded latent grid through the cell table (NetParams::cellTable); nullptr if not compiled in
const void* render_cells_fn(const VariantKey& k);
// render_shaded_cells_kernel: render_shaded_kernel with the grid through the cell table of the PLAIN weight image (finite differences, BRDF, pre-integration)
const void* render_shaded_cells_fn(const VariantKey& k);
hipError_t launch_render_shaded_cells(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
hipError_t launch_render_cells(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
const void* render_adjoint_fn(const VariantKey& k);
hipError_t launch_render_adjoint(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
hipError_t launch_render_stripe(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
// evaluate_gradient_kernel: a.out is (n,4) = value, analytic gradient; gridStep: central-difference step of the latent grid (unit box)
hipError_t launch_eval_gradient(const VariantKey& k, const EvalArgs& a, float gridStep, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
hipError_t launch_render(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);

}  // namespace fvsrn
