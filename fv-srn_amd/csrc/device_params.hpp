// PODs shared by host code and the HIP kernels (passed by value as kernel arguments).
#pragma once
#include <cstdint>

#include "../../include/fvsrn.h"

namespace fvsrn {

// MFMA tiling of one SRN (v_mfma_f32_32x32x16_f16):
//   M tile = 32 output channels, K step = 16 input channels, N tile = 32 samples.
// LDS image (built by pack.cpp, copied global->LDS once per workgroup):
//   A fragments: 1 KiB each = 64 lanes x 8 halfs, lane-linear (ds_read_b128 at base+16*lane)
//     [phase: MT] [layer 0: MT x KS (Fourier part, [m][s]) + G/16 x MT (latent steps, [g][m])] [last: KS]
//     [layer l=1..NL-1: MT x KS, [m][s]]   (the last layer sits before the hidden ones so that "fragment i of the
//     next layer" is always readable, see srn_forward_pipelined)
//   biases: fp32, natural channel order, 32*MT per C->C layer, 32*MT (first 8 rows used) for the last layer
constexpr int kFragBytes = 1024;
constexpr int kFourierResync = 64;  // default period of the exact re-derivation of rotated Fourier features (SceneParams::resyncMask)

// Developer build (-DFVSRN_PROF_SECTIONS, tools/section_profile.py): cycle counter marks inside the render loop; the sums
// go to stats[2 + k].  Marks wait for outstanding LDS reads (s_memtime is a scalar memory read), i.e. they perturb the
// schedule they measure: a coarse split of the wave step, not a profile of the shipped kernel.
#ifdef FVSRN_PROF_SECTIONS
#define FVSRN_MARK(P, k)                                                  \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        const unsigned long long now_ = __builtin_readcyclecounter();     \
        (P).prof[k] += now_ - (P).profLast;                               \
        (P).profLast = now_;                                              \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#else
#define FVSRN_MARK(P, k) do { } while (0)
#endif
constexpr int kProfSections = 8;

struct NetParams {
#ifdef FVSRN_PROF_SECTIONS
    mutable unsigned long long prof[kProfSections];
    mutable unsigned long long profLast;
#endif
    const void* ldsImage;  // device pointer
    int evalHalfIO;        // fvsrn_evaluate_points_half: positions / directions / values of evaluate_points are fp16 (load_eval_point, write_eval_outputs)
    int ldsBytes;          // multiple of 16
    int numLayers;         // NL: number of C->C Linear layers, the first one consumes Fourier(+grid) features; 0 only with noFourier
    int noFourier;         // the "phase" fragments are the scalar first layer (3|6 -> C, bias in the constant slot): activation instead of cos
    int gridK;             // latent grid channels / 16 (0 = none)
    int outputMode;        // fvsrn_output_mode
    int offPhase, offLayer0, offHidden, offLast, offBias;  // byte offsets into the LDS image
    int fourierNeedsFract;      // |phase| may exceed the v_cos_f32 domain of 256 revolutions for positions inside the box
    int fourierNeedsFractEval;  // ... for positions up to 4 box sizes away (evaluate_points takes arbitrary positions)
    int fourierNeedsFractPlain; // ... for positions inside the box only (unshaded renders: fvsrn_render copies it into fourierNeedsFract)
    int fourierClampPos;    // unshaded renders: the phase range is the whole +-256 revolution domain (a 2^9 ladder); sample positions are
                            // clamped to the unit box (they leave it by the rounding of o + t d only) instead of a v_fract per phase
    int timeSlotOffset;     // networks that take the time as an input: byte offset of the fp16 time entry inside the LDS image (-1: none) ...
    unsigned timeSlotBits;  // ... and its value for this launch: the kernel patches its LDS copy, the device images are never written
    int bias0Folded;        // the first layer's bias sits in its weight column of input channel 3, which carries the constant 1 (pack.cpp)
    int reluClamp;          // ldsImage is the [0,1]-scaled ReLU image: convert+ReLU is one clamped v_cvt_pk_f16_f32
    float actA, actB;       // activation constants, see act() in srn_device.hpp
    float boxMin[3];
    float boxSize[3];
    float invBoxSize[3];
    // latent grid working copy (x-pair records, see pack.cpp): fp16 [Z][Y][X+1][G][2].
    //  FLOAT / BYTE_LINEAR: decoded and time-blended values in `grid`.
    //  BYTE_GAUSSIAN: the decode is non-linear and sits between the spatial and the time interpolation
    //  (renderer_volume_tensorcores.cuh:581-591), so `grid` / `gridB` hold the raw byte values (0..255) of key
    //  frames A / B and the kernel decodes: mean + std * sqrt2 * erfinv((2-1e-4)(x-0.5)), then lerps with gridFrac.
    const void* grid;
    const void* gridB;
    const float* gridMeanTime; const float* gridStdTime;  // [Gt] of key frame A (the reference uses A's for B too)
    const float* gridMeanEns; const float* gridStdEns;    // [Ge]
    float gridFrac;
    int gridEncoding, gridTimeChannels;
    int gridX, gridY, gridZ, gridC;
    float gridXf, gridYf, gridZf;  // the same as floats (grid_tap works in fp32: fewer than 2^24 records, pack.cpp)
    // Cell table of the working grid (r04; r06: monomial coefficients and ghost cells; grid_cell_table_kernel in launch.hip; cell_tap / cell_prepare /
    // cells_accumulate in srn_device.hpp): for every cell of the grid extended by one ghost cell per side (cell e spans the nodes e - 1, e per axis, the
    // ghost nodes repeat the boundary = clamp-to-edge), the eight coefficient vectors of the cell's interpolant in cell-centred coordinates times the first
    // layer's latent columns, fp16 [cell][m][row 0..31][slot 4 c + 2 b + a = coefficient of x^a y^b z^c] -- 512 bytes per cell and M tile, which IS the A
    // fragment of an MFMA K step whose B operand holds the eight monomials of a sample.  null: none (BYTE_GAUSSIAN grids, tables above the size cap).
    const void* cellTable;
    unsigned cellStride;  // bytes per cell: 512 * MT
    unsigned cellCount;   // (X + 1)(Y + 1)(Z + 1)
};

constexpr int kMaxFramesPerLaunch = 8;
struct SceneParams {
    // camera (renderer_camera.cuh:33-52), front = cross(up,right) and tan(fov/2) precomputed on the host
    float eye[3], right[3], up[3], front[3];
    float tanFovX, tanFovY;
    // DVR (renderer_ray_evaluation_stepping_dvr.cuh:22-30)
    float stepsize, alphaEarlyOut, densityMin, divDensityRange;
    int earlyOut, blendMode;
    // normals: 0 = from the network if it predicts them, 1 = central differences (6 extra network evaluations)
    int gradientMode;
    float fdStep;  // world units
    float gridDiffStep;  // adjoint mode: central-difference step of the latent grid, unit-box coordinates (1 / (resolution * 4))
    // BRDFLambert (renderer_brdf_lambert.cuh:19-103)
    int brdfMagnitudeScaling, brdfPhong, brdfLightType, brdfSpecularExponent;
    float brdfMagScale, brdfAmbient, brdfSpecular, brdfMagCenter, brdfMagRadius;
    float brdfLight[3];
    // TF
    int tfKind, tfRows;
    float tfRowsF;  // float(tfRows): a scalar operand of the texture lookup instead of a per-lane convert
    float tfScaleAbsorption, tfScaleEmission;
    float tfAbsorptionStep;  // tfScaleAbsorption * stepsize (Identity TF, straight-line tail)
    float tfAbsorptionStepLog2e, densityBias;  // TAIL_SCALAR_IDENTITY: -tfAbsorptionStep * log2(e); -densityMin * divDensityRange
    float stepLog2e;                           // TAIL_SCALAR_TEXTURE: -stepsize * log2(e)
    const float* tfTable;  // device pointer
    int tfLdsFloats;       // floats of the TF table in LDS (behind the network image); render_small_kernel<.., SGRID = 1> keeps its waves' rotation state behind it
    int tfGaussianMode;        // fvsrn_tf_gaussian_mode (scale sigma with |gradient| / piecewise analytic integration)
    int tfOpacityNonNegative;  // Texture TF: no table entry has a negative opacity (the straight-line tail leaves out the reference's `opacity > 0` test)
    int tfPreintegration;          // 0 none, 1: tfPreintegrated = [R][4] running integral, 2: [R][R][4] (previous, current density)
    const float* tfPreintegrated;  // device pointer (global memory: 1 MiB in 2D mode), R = tfRows
    // image: the launch covers `numLocalRows` rows; local row l is image row
    //   y = y0 + ((l / stripeRows) * stripeWorld + stripeRank) * stripeRows + l % stripeRows   (y < y1)
    // (stripeWorld == 1: the contiguous range [y0,y1)); compact != 0 writes a [8][numLocalRows][width] image
    int width, height, y0, y1;
    int numLocalRows, stripeRows, stripeRank, stripeWorld, compact;
    const int* tileOrder;  // device pointer: permutation of the 8x8 pixel tiles of this launch, or null
    // persistent-wave scheduling (kernels.hpp): counter of this launch (starts at 0) and the one to reset for the next
    // launch; null = every wave renders the slots {w, w + totalWaves, ...}
    int* tileCounter;
    int* tileCounterNext;
    // > 0: a wave retires after this many work units (all of them taken from the counter) and the launch holds more
    // workgroups than fit on the chip at once -- used for the stripes of a multi-GPU frame, see api.cpp
    int unitQuota;
    // depth segments (kernels.hpp): every ray is cut into `segments` consecutive step ranges rendered by different waves
    // into `partial` ([segments][8][rows][width] raw accumulators), composited front to back by composite_kernel
    int segments;
    float* partial;
    int resyncMask;  // feature rotation (srn_device.hpp): exact features every resyncMask + 1 steps (a power of two)
    // Several frames in ONE launch (r05, fvsrn_render_stripes_batch): `frames` camera poses of the same scene; a work unit is (frame, tile[, segment]),
    // frame f takes its camera from cams[f] = { eye, right, up, front } and writes its image at out + f * 8 * plane.  A rank's share of a multi-GPU frame
    // is a small launch (2 048 tiles for 2 048 wave slots at world 8): its ramp-up, its longest tile and the gap to the next launch cost a fifth of
    // the 0.26 ms; eight poses in one launch are a whole frame's worth of work units handed out by the same device counter.  frames <= 1: one frame, camera
    // cams[0] (render_body reads the camera from cams[] only; eye / right / up / front above serve the other kernels and the host).
    int frames;
    float cams[kMaxFramesPerLaunch][12];
};

}  // namespace fvsrn
