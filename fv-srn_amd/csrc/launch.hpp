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
// render_stripe_kernel (kernels_cd{3,4}_stripe.hip: 48 / 64 wide with a latent grid); other widths: nullptr / hipErrorInvalidDeviceFunction
template <int CD> const void* render_stripe_fn_cd(const VariantKey& k);
template <int CD> const void* render_cells_fn_cd(const VariantKey& k);
template <int CD> const void* render_shaded_cells_fn_cd(const VariantKey& k);
template <int CD> hipError_t launch_render_shaded_cells_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> hipError_t launch_render_cells_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> hipError_t launch_render_stripe_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
// render_adjoint_kernel (kernels_cd{2,3,4}_adjoint.hip): the adjoint gradient mode up to 64 channels; other widths: nullptr / hipErrorInvalidDeviceFunction
template <int CD> const void* render_adjoint_fn_cd(const VariantKey& k);
template <int CD> hipError_t launch_render_adjoint_cd(const VariantKey& k, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);
template <int CD> hipError_t launch_eval_gradient_cd(const VariantKey& k, const EvalArgs& a, float gridStep, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s);

// decode + A/B time blend of the resident latent key frames into the fp16 working grid (pack.cpp, packLatentGrid)
struct BlendParams {
    const void* timeLo;    // time key frames lo / hi of this blend: [records][Gt][2] fp32 | uint8 each (api.cpp, KeyframeStore)
    const void* timeHi;
    const void* ensData;   // [ensNum][records][Ge][2]
    const float* timeOffset; const float* timeScale;  // [timeNum][Gt]
    const float* ensOffset; const float* ensScale;    // [ensNum][Ge]
    void* out;             // fp16 [records][Gt+Ge][2]
    void* outB;            // BYTE_GAUSSIAN only: raw bytes of key frame B (out = key frame A)
    unsigned long long records;
    int enc, Gt, Ge, lo, hi, ens;
    float frac;
};
hipError_t launch_grid_blend(const BlendParams& p, hipStream_t s);

// Cell table of a working grid (NetParams::cellTable, device_params.hpp): out[cell][m][row][corner] = sum over the latent channels of
// (first-layer latent column of row 32 m + row) x (grid value at the cell's corner), fp32 sums rounded to fp16
struct CellTableParams {
    const void* grid;        // working grid, fp16 x-pair records [Z][Y][X+1][G][2]
    const void* latentFrags; // the weight image's latent K-step fragments of layer 0, [g][m] x 1 KiB (pack.cpp)
    void* out;               // fp16 [cells][MT][32][8]
    int X, Y, Z, G;          // grid resolution, latent channels (a multiple of 16)
    int MT;                  // M tiles of the network
};
hipError_t launch_grid_cell_table(const CellTableParams& p, hipStream_t s);

// IImageEvaluator::ExtractColor: raw (8,H,W) -> planar fp32 (4,H,W) or packed RGBA8; d_minmax: 2 floats of scratch
struct ExtractParams {
    const float* raw;
    float* out4;          // or
    unsigned int* out8;
    float* minmax;        // device scratch: {min, max} of the depth channel (DEPTH mode)
    unsigned long long pixels;
    int mode, tonemap;
    float maxExposure;
};
hipError_t launch_extract_color(const ExtractParams& p, hipStream_t s);

// front-to-back composite of the depth segments of a render (kernels.hpp): partial [K][8][plane] raw accumulators ->
// out [8][plane] in the layout of ImageEvaluatorSimpleKernel (normal and depth finished like :100-124)
hipError_t launch_spin(long long ticks, hipStream_t s);  // one wave spinning for `ticks` x 10 ns (fvsrn_probe_stream_concurrency)
hipError_t launch_composite(const float* partial, float* out, int segments, unsigned long long plane, const SceneParams& S,
                            hipStream_t s);

// pre-integration tables of a Texture TF (transfer_function_texture_cuda.cu:9-90): tex = device [R][4] texels,
// mode 1 -> out [R][4], mode 2 -> out [R][R][4] (row = current density index, column = previous density index)
hipError_t launch_tf_preintegration(const float* tex, float* out, int R, int mode, float stepsize, int quadratureSteps, hipStream_t s);

// ICamera::generateRays / ITransferFunction::evaluate tensor APIs (launch.hip)
hipError_t launch_generate_rays(const SceneParams& S, float* rayStart, float* rayDir, hipStream_t s);
hipError_t launch_evaluate_tf(const SceneParams& S, const float* density, const float* previous, size_t n, float* colors, hipStream_t s);

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
