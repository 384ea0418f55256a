// Host-launched helper kernels of launch.hip (key-frame blend, cell table, ExtractColor, depth-segment composite, TF pre-integration, ray / TF tensor APIs).
// Separate from launch.hpp (the variant dispatch of the render / evaluate kernels) so that the 55 kernel translation units do not depend on it.
#pragma once
#include <hip/hip_runtime.h>

#include "device_params.hpp"

namespace fvsrn {

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
    void* out;               // fp16 [cells][MT][32][8]  (cells: (X+1)(Y+1)(Z+1), corners: (X-1)(Y-1)(Z-1))
    int X, Y, Z, G;          // grid resolution, latent channels (a multiple of 16)
    int MT;                  // M tiles of the network
    int corners;             // 0: monomial coefficients over the ghost-extended cells (cell_tap); 1: corner vectors over the grid's own cells (cell_tap_corners: shaded kernels)
};
hipError_t launch_grid_cell_table(const CellTableParams& p, hipStream_t s);

// IImageEvaluator::ExtractColor: raw (8,H,W) -> planar fp32 (4,H,W) or packed RGBA8; d_minmax: 2 floats of scratch
struct ExtractParams {
    const float* raw;
    float* out4;          // or
    unsigned int* out8;
    float* minmax;        // device scratch: {min, max} of the depth channel (DEPTH mode)
    const float* range3;  // optional, DEPTH mode: the depth range to use instead of this image's own, {-min, max, nan flag} on the device (fvsrn_depth_range)
    unsigned long long pixels;
    int mode, tonemap;
    float maxExposure;
};
hipError_t launch_extract_color(const ExtractParams& p, hipStream_t s);
// depth range of an image part in the mergeable form {-min, max, nan flag} (three floats at range3; scratchBits: 16 bytes of device scratch)
hipError_t launch_depth_range(const float* depth, unsigned long long pixels, float* scratchBits, float* range3, hipStream_t s);

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

}  // namespace fvsrn
