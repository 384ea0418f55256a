// Kernel templates: point evaluation and the fused DVR ray-marching renderer.
#pragma once
#include "srn_device.hpp"
#include "srn_gradient.hpp"

namespace fvsrn {

#ifndef FVSRN_WAVES_PER_EU
#define FVSRN_WAVES_PER_EU 2
#endif
#ifndef FVSRN_WAVES_PER_EU_CD2
#define FVSRN_WAVES_PER_EU_CD2 3
#endif
// Workgroups are 1..4 waves (blockDim.x = 64..256, chosen by the host from the LDS footprint of the network): the
// waves of a workgroup share one LDS copy of the network, and a workgroup's resources are only recycled when its
// slowest wave (= longest ray of 4 pixel tiles) is done, so small networks run one wave per workgroup.
constexpr int kBlockThreads = 256;  // upper bound (launch bounds)
// register budget: 512 / waves.  32-wide Fourier-only kernels (input features + their rotation in registers) fit 168.
#ifndef FVSRN_WAVES_PER_EU_CD2_GRID
#define FVSRN_WAVES_PER_EU_CD2_GRID 3
#endif
// 112 / 128 channels: TWO waves per SIMD since r05.  r04 gave these widths the 512-register budget of one wave because their gather kernels spilled; measured r05
// (tools/dev/wide_ab.py, 1024^2 x 512, 16-channel 32^3 grid: profiles/r05/wide_two_waves_ab_r05.txt) the second wave is worth far more than the 63 - 82 spilled registers
// cost -- 128x2 through the cell table 18.7 -> 11.7 ms per frame, on the gather path 19.5 -> 12.6, BYTE_GAUSSIAN 29.6 -> 20.7, Fourier-only 11.7 -> 10.1, evaluate_points
// 0.300 -> 0.206 ms per 2^22 points; 112x2 16.7 -> 10.4 ms; 128x3 28.2 -> 24.1 (its BYTE_GAUSSIAN variants +-4 %).  The shaded kernels keep one wave (> 1000 spilled registers at 256).
#ifndef FVSRN_WAVES_PER_EU_WIDE
#define FVSRN_WAVES_PER_EU_WIDE 2
#endif
// ... and the shaded renderers at 80 / 112 / 128 channels (r04: one wave, "> 1000 spilled registers at 256").  Measured r05 (tools/dev/shaded_wide_ab.py, finite
// differences + Phong, 512^2 x 256, one -> two waves): 80x3 + grid 18.2 -> 12.3 ms through the cell table, 19.3 -> 11.9 gathering, 5.0 -> 3.4 Fourier-only; 128x2 12.4 -> 10.3
// gathering, 8.4 -> 6.6 Fourier-only, 12.5 -> 12.8 through the cell table (profiles/r05/shaded_wide_two_waves_ab_r05.txt).
#ifndef FVSRN_WAVES_PER_EU_SHADED_WIDE
#define FVSRN_WAVES_PER_EU_SHADED_WIDE 2
#endif
constexpr int min_waves_per_simd(int CD, int GRID) {
    // 32 wide: Fourier-only and decoded-grid kernels fit the 168 registers of 3 waves per SIMD (r02, 32x4 + 16^3 grid: gathers hoisted
    // in front of the Fourier work, GridPre: 2 waves 66.8, 3 waves 69.5 Gsamples/s; without the hoist 68.5); BYTE_GAUSSIAN would spill
    // (112 / 128 channels: FVSRN_WAVES_PER_EU_WIDE above)
    return CD >= 7 ? FVSRN_WAVES_PER_EU_WIDE : (CD == 2 ? (GRID == 0 ? FVSRN_WAVES_PER_EU_CD2 : (GRID == 1 ? FVSRN_WAVES_PER_EU_CD2_GRID : FVSRN_WAVES_PER_EU)) : FVSRN_WAVES_PER_EU);
}
// __launch_bounds__(256, 2): at most 256 registers per lane, which also makes hipcc use the VGPR form of the MFMA
// (accumulators in AGPRs cost one v_accvgpr_read per value before the VALU can touch them: +32 VALU per layer).
// C = 128 needs more than 256 registers and takes the 512-register budget instead.

__device__ __forceinline__ void load_network_to_lds(const NetParams& P, char* lds) {
    const uint4_t* src = reinterpret_cast<const uint4_t*>(P.ldsImage);
    uint4_t* dst = reinterpret_cast<uint4_t*>(lds);
    const int n = P.ldsBytes >> 4;
    for (int i = threadIdx.x; i < n; i += int(blockDim.x)) dst[i] = src[i];
    if (P.timeSlotOffset >= 0) {  // the time input of this launch (api.cpp, syncTime): the 16-byte chunk that holds it is copied by thread (offset / 16) % blockDim
        __syncthreads();
        if (threadIdx.x == 0) *reinterpret_cast<unsigned short*>(lds + P.timeSlotOffset) = static_cast<unsigned short>(P.timeSlotBits);
    }
    __syncthreads();
}

// positions (and directions) of batch b for this lane; lanes beyond n read point 0.  The batch base is wave-uniform (scalar address
// arithmetic), the lane adds a 32-bit offset: no 64-bit vector arithmetic per point (r02: 18 v_lshl_add_u64 + 6 v_mad_u64_u32 per batch).
// wave index inside the workgroup in a SCALAR register: threadIdx.x >> 6 is wave-uniform, which the compiler cannot see -- without this the
// whole batch bookkeeping of evaluate_points (batch index, base addresses, loop bound) is 64-bit vector arithmetic under exec masks
// (r04: 5 v_lshl_add_u64, 3 v_cmp_*_u64 and 4 exec-mask regions per batch of evaluate_small_kernel)
__device__ __forceinline__ unsigned wave_in_block() { return unsigned(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))); }

// (r05's A/B of the run-time branches below -- fp16 I/O and the unclamped pass compiled out, profiles/r05/evaluate_ab_r05.txt -- used two build switches that are gone: a
// build without them would have returned wrong values for fp16 tensors and for points outside the box)
__device__ __forceinline__ bool eval_half_io(const NetParams& P) { return P.evalHalfIO != 0; }

template <bool HAS_DIR>
__device__ __forceinline__ void load_eval_point(const NetParams& P, const float* __restrict__ pos, const float* __restrict__ dir, size_t n, size_t b, int lane,
                                                float (&p)[3], float (&d)[3]) {
    const size_t first = b * 64;                        // wave-uniform (scalar registers: b comes from wave_in_block())
    const unsigned cnt = unsigned(n - first < 64 ? n - first : 64);
    const unsigned j = unsigned(lane) < cnt ? unsigned(lane) : 0u;
    if (eval_half_io(P)) {  // wave-uniform: fvsrn_evaluate_points_half -- (n,3) fp16 positions / directions, 6 bytes per point
        const unsigned off = 6u * j;
        const char* pb = reinterpret_cast<const char*>(pos) + 6 * first;
        p[0] = float(*reinterpret_cast<const _Float16*>(pb + off)); p[1] = float(*reinterpret_cast<const _Float16*>(pb + (off + 2u)));
        p[2] = float(*reinterpret_cast<const _Float16*>(pb + (off + 4u)));
        if constexpr (HAS_DIR) {
            const char* db = reinterpret_cast<const char*>(dir) + 6 * first;
            d[0] = float(*reinterpret_cast<const _Float16*>(db + off)); d[1] = float(*reinterpret_cast<const _Float16*>(db + (off + 2u)));
            d[2] = float(*reinterpret_cast<const _Float16*>(db + (off + 4u)));
        } else { d[0] = d[1] = d[2] = 0.f; }
        return;
    }
    const unsigned off = 12u * j;                        // 32-bit byte offset of the lane: SGPR base + VGPR offset loads
    // (one global_load_dwordx3 per point: r02 - r04 issued three dword loads with three offset registers)
    typedef float float3a4_t __attribute__((ext_vector_type(3), aligned(4)));
    const char* pb = reinterpret_cast<const char*>(pos + 3 * first);
    const float3a4_t pv = *reinterpret_cast<const float3a4_t*>(pb + off);
    p[0] = pv[0]; p[1] = pv[1]; p[2] = pv[2];
    if constexpr (HAS_DIR) {
        const char* db = reinterpret_cast<const char*>(dir + 3 * first);
        const float3a4_t dv = *reinterpret_cast<const float3a4_t*>(db + off);
        d[0] = dv[0]; d[1] = dv[1]; d[2] = dv[2];
    } else { d[0] = d[1] = d[2] = 0.f; }
}

// evaluate_points of a ReLU network runs the weight image whose activations are scaled into [0,1] (pack.cpp: convert + ReLU is one clamped
// v_cvt_pk_f16_f32; the plain ReLU costs 48 v_pk_max_f16 per batch of a 32x4 network on top of the converts).  Its bound holds for positions
// inside the unit box and directions inside [-1,1]^3 -- what a caller evaluates as a rule, but not a promise of this entry point.  So the
// ACT_RELU01 kernels check the 64 points of every batch (wave-uniform result) and run a batch with a point outside through the SAME image with
// the unclamped activation (convert, then v_pk_max_f16 with 0): the scales are powers of two, so relu(x) 2^-e is what the clamped convert
// returns wherever the bound holds and the exact scaled activation where it does not, and the last layer takes the scale out again; the
// phases of such a batch go through v_fract.  (r03 - r04 handed these batches to a second launch with the plain image through a per-call
// list: a memset and a launch per call, ~9 us, which made the scaled image lose below 2^23 points.)
template <bool HAS_DIR>
__device__ __forceinline__ bool eval_batch_outside(float px, float py, float pz, float dx, float dy, float dz) {
    // inside the unit box <=> max |p - 1/2| <= 1/2 (a NaN fails the comparison: outside): three subtractions, one v_max3_f32 with |.|, one compare
    // (v_max3_f32 drops NaN operands: a NaN coordinate is caught by the unordered compare of the sums; `|` instead of `||`: no exec-mask regions)
    bool outside = bool(int(!(__builtin_fmaxf(__builtin_fmaxf(fabsf(px - 0.5f), fabsf(py - 0.5f)), fabsf(pz - 0.5f)) <= 0.5f)) | int(__builtin_isunordered(px + py, pz)));
    if constexpr (HAS_DIR)
        outside = bool(int(outside) | int(!(__builtin_fmaxf(__builtin_fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) <= 1.f)) | int(__builtin_isunordered(dx + dy, dz)));
    return __builtin_expect(__builtin_amdgcn_ballot_w64(outside) != 0, 0);  // wave-uniform; the rare case
}
// Fourier mode of such a batch: v_fract in front of the cosines (FM_FIRST_LAYER networks have no phases)
constexpr int eval_outside_fmode(int FMODE) { return FMODE == FM_FIRST_LAYER ? FM_FIRST_LAYER : FM_FRACT_COS; }

// output parametrization of IVolumeInterpolation::evaluate for point i (renderer_volume_tensorcores.cuh:1054-1158)
template <class T>
__device__ __forceinline__ void write_eval_outputs_as(const NetParams& P, const float4_t& o, T* __restrict__ out, unsigned i, int outChannels) {
    switch (P.outputMode) {
        case FVSRN_OUT_DENSITY:
        case FVSRN_OUT_DENSITY_GRADIENT:
        case FVSRN_OUT_DENSITY_CURVATURE:
            out[i * outChannels] = T(sigmoid_f(o[0]));
            break;
        case FVSRN_OUT_RGBO:
            out[i * 4 + 0] = T(sigmoid_f(o[0]));
            out[i * 4 + 1] = T(sigmoid_f(o[1]));
            out[i * 4 + 2] = T(sigmoid_f(o[2]));
            out[i * 4 + 3] = T(softplus_f(o[3]));
            break;
        case FVSRN_OUT_RGBO_DIRECT:
            out[i * 4 + 0] = T(fminf(fmaxf(o[0], 0.f), 1.f));
            out[i * 4 + 1] = T(fminf(fmaxf(o[1], 0.f), 1.f));
            out[i * 4 + 2] = T(fminf(fmaxf(o[2], 0.f), 1.f));
            out[i * 4 + 3] = T(fmaxf(o[3], 0.f));
            break;
        default:  // density:direct and the direct gradient modes: un-clamped
            out[i * outChannels] = T(o[0]);
            break;
    }
    if (outChannels == 4 && P.outputMode >= FVSRN_OUT_DENSITY_GRADIENT) {  // FVSRN_EVAL_WITH_PREDICTED_GRADIENT
        const bool cubic = P.outputMode == FVSRN_OUT_DENSITY_GRADIENT_CUBIC;  // evalNormal :1166-1183
        for (int k = 1; k < 4; ++k) out[i * 4 + k] = T(cubic ? o[k] * o[k] * o[k] : o[k]);
    }
}
// `out`: the first value of batch b (fp32, or fp16 behind fvsrn_evaluate_points_half: wave-uniform branch)
__device__ __forceinline__ void write_eval_outputs(const NetParams& P, const float4_t& o, float* __restrict__ out, unsigned i, int outChannels) {
    if (eval_half_io(P)) write_eval_outputs_as(P, o, reinterpret_cast<_Float16*>(out), i, outChannels);
    else write_eval_outputs_as(P, o, out, i, outChannels);
}
// first output value of batch b
__device__ __forceinline__ float* eval_out_of_batch(const NetParams& P, float* __restrict__ out, size_t b, int outChannels) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(out) + b * 64 * size_t(outChannels) * (eval_half_io(P) ? 2 : 4));
}

// positions (directions) of batch b as loaded: three fp32 values, or (HIO) three fp16 values in the low halves -- converted where they are
// used, so that nothing waits for the load where it is issued
template <bool HAS_DIR, bool HIO>
__device__ __forceinline__ void load_eval_raw(const float* __restrict__ pos, const float* __restrict__ dir, size_t n, size_t b, int lane,
                                              unsigned (&p)[3], unsigned (&d)[3]) {
    const size_t first = b * 64;
    const unsigned cnt = unsigned(n - first < 64 ? n - first : 64);
    const unsigned j = unsigned(lane) < cnt ? unsigned(lane) : 0u;
    if constexpr (HIO) {
        const unsigned off = 6u * j;
        const char* pb = reinterpret_cast<const char*>(pos) + 6 * first;
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = *reinterpret_cast<const unsigned short*>(pb + off + 2 * c);
        if constexpr (HAS_DIR) {
            const char* db = reinterpret_cast<const char*>(dir) + 6 * first;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = *reinterpret_cast<const unsigned short*>(db + off + 2 * c);
        }
    } else {
        typedef unsigned uint3a4_t __attribute__((ext_vector_type(3), aligned(4)));
        const unsigned off = 12u * j;
        const uint3a4_t pv = *reinterpret_cast<const uint3a4_t*>(reinterpret_cast<const char*>(pos + 3 * first) + off);
        p[0] = pv[0]; p[1] = pv[1]; p[2] = pv[2];
        if constexpr (HAS_DIR) {
            const uint3a4_t dv = *reinterpret_cast<const uint3a4_t*>(reinterpret_cast<const char*>(dir + 3 * first) + off);
            d[0] = dv[0]; d[1] = dv[1]; d[2] = dv[2];
        }
    }
}
template <bool HIO>
__device__ __forceinline__ float eval_raw_value(unsigned u) {
    if constexpr (HIO) return float(__builtin_bit_cast(_Float16, static_cast<unsigned short>(u)));
    else return __uint_as_float(u);
}

// ------------------------------------------------------------------------------------------------
// EvaluateNoBatches (reference renderer/renderer_volume_kernels1.cuh:15): positions -> network value
// ------------------------------------------------------------------------------------------------
template <int CD, int ACT, int GRID, bool HAS_DIR, int FMODE>
__device__ __forceinline__ void evaluate_body(const NetParams& P, const char* lds, const float* __restrict__ pos,
                                              const float* __restrict__ dir, size_t n, float* __restrict__ out, int outChannels) {
    const int lane = lane_id();
    const size_t wavesPerBlock = blockDim.x >> 6;
    const size_t wave = size_t(blockIdx.x) * wavesPerBlock + wave_in_block();
    const size_t numWaves = size_t(gridDim.x) * wavesPerBlock;
    const size_t batches = (n + 63) / 64;
    for (size_t b = wave; b < batches; b += numWaves) {  // wave-uniform trip count (scalar loop control): EXEC stays full
        const bool valid = unsigned(lane) < unsigned(n - b * 64 < 64 ? n - b * 64 : 64);
        // (the next batch is not fetched ahead here.  r03: six more live registers for the wide kernels, +3 % at 32 wide; r05, with one dwordx3 per
        // point and three registers: 48x5, 64x6 and 64x6 + grid the same within 0.5 %, two waves per SIMD hide the fetch)
        float np_[3], nd_[3];
        load_eval_point<HAS_DIR>(P, pos, dir, n, b, lane, np_, nd_);
        float px = np_[0], py = np_[1], pz = np_[2];
        const float dx = nd_[0], dy = nd_[1], dz = nd_[2];
        // renderer_volume_tensorcores.cuh:744-746
        px = (px - P.boxMin[0]) * P.invBoxSize[0];
        py = (py - P.boxMin[1]) * P.invBoxSize[1];
        pz = (pz - P.boxMin[2]) * P.invBoxSize[2];
        float4_t o;
        if constexpr (ACT == ACT_RELU01 && GRID == 2) {
            // BYTE_GAUSSIAN grids (decode in the kernel): two copies of the forward pass cost 110 - 170 spilled registers in both; evaluate_points takes
            // the plain image for these networks (api.cpp), and this variant is the unclamped pass for every batch -- correct for any position
            o = srn_forward<CD, ACT_RELU, GRID, HAS_DIR, eval_outside_fmode(FMODE)>(P, lds, px, py, pz, dx, dy, dz);
        } else if constexpr (ACT == ACT_RELU01) {
            if (eval_batch_outside<HAS_DIR>(px, py, pz, dx, dy, dz)) o = srn_forward<CD, ACT_RELU, GRID, HAS_DIR, eval_outside_fmode(FMODE)>(P, lds, px, py, pz, dx, dy, dz);
            else o = srn_forward<CD, ACT, GRID, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz);
        } else {
            o = srn_forward<CD, ACT, GRID, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz);
        }
        if (valid) write_eval_outputs(P, o, eval_out_of_batch(P, out, b, outChannels), unsigned(lane), outChannels);  // (wave-uniform base, 32-bit lane offset)
    }
}

// FMODE (srn_device.hpp: Fourier phases need a v_fract first / no Fourier features at all) is a property of the network:
// resolved by ONE wave-uniform branch around the whole body instead of one per sample, which keeps the sample loop a
// single schedulable region.
template <int CD, int ACT, int GRID, bool HAS_DIR>
__global__ __launch_bounds__(kBlockThreads, min_waves_per_simd(CD, GRID)) void evaluate_kernel(NetParams P, const float* __restrict__ pos,
                                                                 const float* __restrict__ dir, size_t n,
                                                                 float* __restrict__ out, int outChannels) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    load_network_to_lds(P, lds);
    if constexpr (GRID == 0) {  // a latent grid needs Fourier features (SceneNetwork::valid)
        if (P.noFourier) return evaluate_body<CD, ACT, GRID, HAS_DIR, FM_FIRST_LAYER>(P, lds, pos, dir, n, out, outChannels);
    }
    // the batches the [0,1]-scaled ReLU image takes on its fast path hold positions inside the unit box only (evaluate_body): the renderer's
    // bound on the phases holds for them, not the one for arbitrary positions (r03: no v_fract for the 2^9 ladder of the 64-wide bench
    // network, like in the renderer since r02)
    const int needsFract = ACT == ACT_RELU01 ? P.fourierNeedsFractPlain : P.fourierNeedsFractEval;
    if (needsFract) evaluate_body<CD, ACT, GRID, HAS_DIR, FM_FRACT_COS>(P, lds, pos, dir, n, out, outChannels);
    else evaluate_body<CD, ACT, GRID, HAS_DIR, FM_COS>(P, lds, pos, dir, n, out, outChannels);
}

// ------------------------------------------------------------------------------------------------
// EvaluateNoBatchesWithGradient (reference volume_interpolation.cpp:128-243) for GRADIENT_MODE_ADJOINT_METHOD: value + the analytic
// gradient w.r.t. the normalized position (evalNormal, renderer_volume_tensorcores.cuh:1198-1540; forward mode here, srn_gradient.hpp).
// out: (n,4) = value (output parametrization of evaluate), gradient
// ------------------------------------------------------------------------------------------------
template <int CD, int ACT, int GRID, bool HAS_DIR, int FMODE>
__device__ __forceinline__ void evaluate_gradient_body(const NetParams& P, const char* lds, const float* __restrict__ pos,
                                                       const float* __restrict__ dir, size_t n, float* __restrict__ out, float gridStep) {
    const int lane = lane_id();
    const size_t wavesPerBlock = blockDim.x >> 6;
    const size_t wave = size_t(blockIdx.x) * wavesPerBlock + wave_in_block();
    const size_t numWaves = size_t(gridDim.x) * wavesPerBlock;
    const size_t batches = (n + 63) / 64;
    for (size_t b = wave; b < batches; b += numWaves) {
        const size_t i = b * 64 + lane;
        float np_[3], nd_[3];
        load_eval_point<HAS_DIR>(P, pos, dir, n, b, lane, np_, nd_);
        const float px = (np_[0] - P.boxMin[0]) * P.invBoxSize[0];
        const float py = (np_[1] - P.boxMin[1]) * P.invBoxSize[1];
        const float pz = (np_[2] - P.boxMin[2]) * P.invBoxSize[2];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        float4_t o = srn_forward_gradient<CD, ACT, GRID, HAS_DIR, FMODE>(P, lds, px, py, pz, nd_[0], nd_[1], nd_[2], gridStep, gx, gy, gz);
        if (i < n) {
            o[1] = o[2] = o[3] = 0.f;
            float v[4];
            // (value through the same switch as evaluate; a scalar network's parametrization only touches channel 0)
            write_eval_outputs_as(P, o, v, 0u, 1);
            *reinterpret_cast<float4_t*>(out + 4 * i) = float4_t{v[0], gx, gy, gz};
        }
    }
}

// (one wave per SIMD = the 512-register budget: the four column sets spill nothing; measured r03, 2^22 points: 32x4 + grid 0.61 -> 0.56 ms,
// 64x6 + grid 1.80 -> 1.25 ms against the 256-register build)
template <int CD, int ACT, int GRID, bool HAS_DIR>
__global__ __launch_bounds__(kBlockThreads, 1) void evaluate_gradient_kernel(NetParams P, const float* __restrict__ pos,
                                                                 const float* __restrict__ dir, size_t n, float* __restrict__ out, float gridStep) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    load_network_to_lds(P, lds);
    if constexpr (GRID == 0) {
        if (P.noFourier) return evaluate_gradient_body<CD, ACT, GRID, HAS_DIR, FM_FIRST_LAYER>(P, lds, pos, dir, n, out, gridStep);
    }
    if (P.fourierNeedsFractEval) evaluate_gradient_body<CD, ACT, GRID, HAS_DIR, FM_FRACT_COS>(P, lds, pos, dir, n, out, gridStep);
    else evaluate_gradient_body<CD, ACT, GRID, HAS_DIR, FM_COS>(P, lds, pos, dir, n, out, gridStep);
}

// evaluate_kernel for 32-wide Fourier-only networks with NLC <= 3 C->C layers: weights and biases in registers (ResidentNet,
// srn_device.hpp), 2 waves per SIMD; the phase fragments of the Fourier stage are the only LDS reads of a batch
// EGRID = 1: with one decoded 16-channel latent chunk (srn_forward_resident_grid)
template <int ACT, bool HAS_DIR, int NLC, int EGRID, int FMODE>
__device__ __forceinline__ float4_t evaluate_small_forward(const NetParams& P, const ResidentNet<NLC, EGRID>& R, const char* lds, float px, float py, float pz,
                                                           float dx, float dy, float dz) {
    if constexpr (EGRID == 1) {
        return srn_forward_resident_grid<ACT, HAS_DIR, NLC, true, FMODE>(P, R, px, py, pz, dx, dy, dz);
    } else {
        half8_t xb[2][2];
        return srn_layers_resident<ACT, NLC, 0, true>(
#ifndef FVSRN_EVAL_PHASES_AHEAD
#define FVSRN_EVAL_PHASES_AHEAD 1
#endif
            P, R, xb, [&]() {
                if constexpr (FVSRN_EVAL_PHASES_AHEAD != 0) fourier_fragments_ahead<HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz, xb);
                else fourier_fragments<2, ACT, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz, xb);
            }, [](int) {});
    }
}

// HIO: fvsrn_evaluate_points_half (fp16 positions / directions / values) as a compile-time case of this kernel
template <int ACT, bool HAS_DIR, int NLC, int EGRID = 0, bool HIO = false>
__global__ __launch_bounds__(kBlockThreads, 2) void evaluate_small_kernel(NetParams P, const float* __restrict__ pos, const float* __restrict__ dir,
                                                                        size_t n, float* __restrict__ out, int outChannels) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    load_network_to_lds(P, lds);
    ResidentNet<NLC, EGRID> R;
    load_resident(P, lds, R);
    const int lane = lane_id();
    const size_t wavesPerBlock = blockDim.x >> 6;
    const size_t wave = size_t(blockIdx.x) * wavesPerBlock + wave_in_block();
    const size_t numWaves = size_t(gridDim.x) * wavesPerBlock;
    const size_t batches = (n + 63) / 64;
    if (wave >= batches) return;
    // Two batches of positions in flight per wave (r03): at 90 G points/s the kernel streams 1.4 TB/s of positions and values; with one
    // batch ahead a CU had 12 waves x 768 B = 9 KB of loads outstanding, which at ~2 us of HBM latency under load is what bounds it.
    // r05: two named buffers, the loop unrolled by two, every load unconditional (beyond the end: the last batch again), one dwordx3 per point,
    // its values converted where they are used.  The r03 form rotated one register set into the other at the top of every iteration and
    // loaded under a condition: the wave waited at the loop head for the loads issued ONE iteration earlier -- and, the vector memory counter
    // being in order, for the store of the previous batch behind them (s_waitcnt vmcnt(0)).  Now the store stays in flight (vmcnt(1) / (2); the
    // first pair of batches peeled out of the loop did not get the compiler to leave more in flight: its copies of the loop-carried buffers
    // at the latch wait for everything again): 2^24 points 128.3 -> 131.6 G points/s on one box.
    unsigned pa[3], da[3] = {0, 0, 0}, pb[3], db[3] = {0, 0, 0};
    const size_t lastBatch = batches - 1;
    auto clampB = [&](size_t b) { return b < lastBatch ? b : lastBatch; };  // (scalar)
    load_eval_raw<HAS_DIR, HIO>(pos, dir, n, wave, lane, pa, da);
    load_eval_raw<HAS_DIR, HIO>(pos, dir, n, clampB(wave + numWaves), lane, pb, db);
    auto batch = [&](size_t b, unsigned (&pp)[3], unsigned (&dd)[3]) {
        const bool valid = unsigned(lane) < unsigned(n - b * 64 < 64 ? n - b * 64 : 64);
        float px = eval_raw_value<HIO>(pp[0]), py = eval_raw_value<HIO>(pp[1]), pz = eval_raw_value<HIO>(pp[2]);
        const float dx = HAS_DIR ? eval_raw_value<HIO>(dd[0]) : 0.f, dy = HAS_DIR ? eval_raw_value<HIO>(dd[1]) : 0.f, dz = HAS_DIR ? eval_raw_value<HIO>(dd[2]) : 0.f;
        __builtin_amdgcn_sched_barrier(0);  // (the values above are read before their registers are loaded again)
        load_eval_raw<HAS_DIR, HIO>(pos, dir, n, clampB(b + 2 * numWaves), lane, pp, dd);
        px = (px - P.boxMin[0]) * P.invBoxSize[0];
        py = (py - P.boxMin[1]) * P.invBoxSize[1];
        pz = (pz - P.boxMin[2]) * P.invBoxSize[2];
        float4_t o;
        if constexpr (ACT == ACT_RELU01) {  // (see eval_batch_outside)
            if (eval_batch_outside<HAS_DIR>(px, py, pz, dx, dy, dz)) o = evaluate_small_forward<ACT_RELU, HAS_DIR, NLC, EGRID, FM_FRACT_COS>(P, R, lds, px, py, pz, dx, dy, dz);
            else o = evaluate_small_forward<ACT, HAS_DIR, NLC, EGRID, FM_COS>(P, R, lds, px, py, pz, dx, dy, dz);
        } else {
            o = evaluate_small_forward<ACT, HAS_DIR, NLC, EGRID, FM_COS>(P, R, lds, px, py, pz, dx, dy, dz);
        }
        if (valid) {
            if constexpr (HIO) write_eval_outputs_as(P, o, reinterpret_cast<_Float16*>(out) + b * 64 * size_t(outChannels), unsigned(lane), outChannels);
            else write_eval_outputs_as(P, o, out + b * 64 * size_t(outChannels), unsigned(lane), outChannels);
        }
    };
    for (size_t b = wave; b < batches; b += 2 * numWaves) {
        batch(b, pa, da);
        if (b + numWaves < batches) batch(b + numWaves, pb, db);
    }
}

// ------------------------------------------------------------------------------------------------
// transfer functions (reference renderer/renderer_tf_*.cuh); density already mapped by the DVR loop
// ------------------------------------------------------------------------------------------------
// tex1D / tex2D with linear filtering, normalized coordinates, clamp addressing on the pre-integration tables
__device__ __forceinline__ float4_t preint_fetch1(const float* __restrict__ t, int R, float x) {
    const float d = x * R - 0.5f;
    const int di = int(floorf(d));
    const float f = d - di;
    const float4_t a = *reinterpret_cast<const float4_t*>(t + 4 * min(max(di, 0), R - 1));
    const float4_t b = *reinterpret_cast<const float4_t*>(t + 4 * min(max(di + 1, 0), R - 1));
    return a + f * (b - a);
}
__device__ __forceinline__ float4_t preint_fetch2(const float* __restrict__ t, int R, float x, float y) {
    const float dx = x * R - 0.5f, dy = y * R - 0.5f;
    const int ix = int(floorf(dx)), iy = int(floorf(dy));
    const float fx = dx - ix, fy = dy - iy;
    const int x0 = min(max(ix, 0), R - 1), x1 = min(max(ix + 1, 0), R - 1), y0 = min(max(iy, 0), R - 1), y1 = min(max(iy + 1, 0), R - 1);
    const float4_t a = *reinterpret_cast<const float4_t*>(t + 4 * (size_t(y0) * R + x0));
    const float4_t b = *reinterpret_cast<const float4_t*>(t + 4 * (size_t(y0) * R + x1));
    const float4_t c = *reinterpret_cast<const float4_t*>(t + 4 * (size_t(y1) * R + x0));
    const float4_t d = *reinterpret_cast<const float4_t*>(t + 4 * (size_t(y1) * R + x1));
    const float4_t lo = a + fx * (b - a), hi = c + fx * (d - c);
    return lo + fy * (hi - lo);
}

// TransferFunctionTexture::eval with pre-integration (renderer_tf_texture.cuh:55-93); density already clamped
__device__ __forceinline__ float4_t tf_eval_preintegrated(const SceneParams& S, const float* __restrict__ tfLds, float density,
                                                          float previousDensity) {
    if (previousDensity < 0.f) previousDensity = density;
    float4_t rgba;
    if (S.tfPreintegration == FVSRN_PREINTEGRATE_1D) {
        if (fabsf(previousDensity - density) < 1e-3f) {  // fallback for constant density
            rgba = preint_fetch1(tfLds, S.tfRows, density);
            rgba[3] *= S.stepsize;
        } else {
            const float4_t f = preint_fetch1(S.tfPreintegrated, S.tfRows, previousDensity), b = preint_fetch1(S.tfPreintegrated, S.tfRows, density);
            const float inv = 1.0f / (density - previousDensity);
            rgba = float4_t{S.stepsize * (b[0] - f[0]) * inv, S.stepsize * (b[1] - f[1]) * inv, S.stepsize * (b[2] - f[2]) * inv,
                            1.f - __expf(-S.stepsize * (b[3] - f[3]) * inv)};
            if (rgba[3] > 1e-5f) { rgba[0] /= rgba[3]; rgba[1] /= rgba[3]; rgba[2] /= rgba[3]; }  // premultiplication
        }
    } else {
        rgba = preint_fetch2(S.tfPreintegrated, S.tfRows, previousDensity, density);
        if (rgba[3] > 1e-5f) { rgba[0] /= rgba[3]; rgba[1] /= rgba[3]; rgba[2] /= rgba[3]; }
    }
    return rgba;
}

// gradLen / previousDensity: only the Gaussian TF variants read them (SceneParams::tfGaussianMode): |gradient| of the sample for
// TRANSFER_FUNCTION_GAUSSIAN__SCALE_WITH_GRADIENT, the un-clamped mapped density of the previous sample of the ray (-1: none) for
// TRANSFER_FUNCTION_GAUSSIAN__ANALYTIC (renderer_tf_gaussian.cuh:55-73)
__device__ __forceinline__ float4_t tf_eval(const SceneParams& S, const float* __restrict__ tfLds, float density, float gradLen = 0.f,
                                            float previousDensity = -1.f) {
    density = fminf(fmaxf(density, 0.f), 1.f);
    float4_t c = {0, 0, 0, 0};
    switch (S.tfKind) {
        case FVSRN_TF_IDENTITY: {  // renderer_tf_identity.cuh:36-54
            const float e = density * S.tfScaleEmission;
            c = float4_t{e, e, e, density * S.tfScaleAbsorption * S.stepsize};
        } break;
        case FVSRN_TF_GAUSSIAN: {  // renderer_tf_gaussian.cuh:43-86
            // (wave-uniform mode; the plain form keeps the fast exponential it had, the two variants are restated with erff / expf)
            const float sigmaScale = S.tfGaussianMode == FVSRN_TF_GAUSSIAN_SCALE_WITH_GRADIENT ? fmaxf(1e-5f, gradLen * 0.1f) : 1.f;
            const bool segment = S.tfGaussianMode == FVSRN_TF_GAUSSIAN_ANALYTIC && !(previousDensity < 0.f || previousDensity == density);
            for (int i = 0; i < S.tfRows; ++i) {
                const float* r = tfLds + 6 * i;
                const float mu = r[4], sigma = r[5] * sigmaScale;
                float ni;
                if (segment) {  // piecewise analytic integration over [previousDensity, density], constant colour per segment
                    constexpr float kSqrtPi2 = 0.8862269254527580136f;  // sqrt(pi) / 2
                    ni = kSqrtPi2 / (previousDensity - density) * sigma * (erff((previousDensity - mu) / sigma) + erff((mu - density) / sigma));
                } else {
                    const float t = density - mu;
                    ni = __expf(-t * t / (sigma * sigma));
                }
                c[0] += r[0] * ni; c[1] += r[1] * ni; c[2] += r[2] * ni; c[3] += r[3] * ni;
            }
            c[3] *= S.stepsize;
        } break;
        case FVSRN_TF_PIECEWISE: {  // renderer_tf_piecewise.cuh:29-62
            int i;
            for (i = 0; i < S.tfRows - 2; ++i)
                if (tfLds[5 * (i + 1) + 4] > density) break;
            const float* a = tfLds + 5 * i;
            const float* b = tfLds + 5 * (i + 1);
            const float d = fminf(fmaxf(density, a[4]), b[4]);
            const float f = (d - a[4]) / (b[4] - a[4]);
            c = float4_t{a[0] + f * (b[0] - a[0]), a[1] + f * (b[1] - a[1]), a[2] + f * (b[2] - a[2]),
                         (a[3] + f * (b[3] - a[3])) * S.stepsize};
        } break;
        case FVSRN_TF_TEXTURE: {  // renderer_tf_texture.cuh:46-55 (tensor mode)
            const int R = S.tfRows;
            const float d = density * R - 0.5f;
            const int di = int(floorf(d));
            const float df = d - di;
            const float* a = tfLds + 4 * min(max(di, 0), R - 1);
            const float* b = tfLds + 4 * min(max(di + 1, 0), R - 1);
            c = float4_t{a[0] + df * (b[0] - a[0]), a[1] + df * (b[1] - a[1]), a[2] + df * (b[2] - a[2]),
                         (a[3] + df * (b[3] - a[3])) * S.stepsize};
        } break;
        default: break;
    }
    return c;
}

// ------------------------------------------------------------------------------------------------
// ImageEvaluatorSimpleKernel + CameraReferenceFrame + RayEvaluationSteppingDvr + Blending, fused.
//   reference: renderer_image_evaluator_simple.cuh:36-127, renderer_camera.cuh:33-52,
//              renderer_utils.cuh:91-105, renderer_ray_evaluation_stepping_dvr.cuh:48-157,
//              renderer_blending.cuh:35-51
// One wave = one 8x8 pixel tile (the reference: 32 consecutive x of one row; per-pixel results do not
// depend on the grouping because blending is guarded by isValid).  Stepping is wave-synchronous like
// the reference's __any_sync loop: all 64 lanes evaluate the network until no lane is valid.
// ------------------------------------------------------------------------------------------------
// SHADED: finite-difference normals and the BRDF (magnitude scaling, Phong).  A separate kernel (render_shaded_kernel): the
// extra network evaluation inside the step loop needs so many registers that it would cost the plain renderer a wave.
// TAIL: what follows the network inside the step loop.  TAIL_GENERIC handles every output mode / transfer function /
// blend mode / early-out setting through wave-uniform branches (~20 scalar branches, exec-mask regions and scalar spills
// per step: ~1000 cycles of latency for a lone wave, r01 tools/section_profile.py, more than the network itself).
// TAIL_SCALAR_TABLE -- a scalar density network (density | density:direct) with an Identity or Texture transfer function, either
// blend mode -- as straight-line predicated code with two wave-uniform branches (sigmoid, texture); since r02 only render_small_kernel
// instantiates it, for Alpha blending (Beer-Lambert: TAIL_SCALAR_IDENTITY / TAIL_SCALAR_TEXTURE below).
// TAIL_SCALAR_LOOP: the same frame around Piecewise / Gaussian TFs, which loop over their control points (a separate
// instantiation, so that the loops stay out of the Identity / Texture instruction stream).  TAIL_RGBO: colour networks (rgbo |
// rgbo:direct, no transfer function) in the same frame.
// TAIL_SCALAR_IDENTITY: the Identity TF with Beer-Lambert blending alone -- no texture branch and no blend-mode
// select in the step, every constant folded on the host, r = g = b = density * emission scale kept as ONE accumulator of w * density that is
// scaled once per ray: 12 vector instructions per step instead of 20 (as a run-time branch inside TAIL_SCALAR_TABLE the single accumulator
// alone measured 2 % slower, r02).
// TAIL_SCALAR_TEXTURE: the same for the Texture TF -- what convert_to_texture_tf() of the evaluation scripts produces.
enum { TAIL_GENERIC = 0, TAIL_SCALAR_TABLE = 1, TAIL_SCALAR_LOOP = 2, TAIL_RGBO = 3, TAIL_SCALAR_IDENTITY = 4, TAIL_SCALAR_TEXTURE = 5 };

// x in the lanes whose bit is set in the wave mask m (an SGPR pair), 0 elsewhere: one v_cndmask_b32 with the mask as its selector
__device__ __forceinline__ float select_by_mask(unsigned long long m, float x) {
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}

// NLC > 0 (render_small_kernel): the network stays in registers (ResidentNet, srn_device.hpp); CD = 2, GRID = 0, rotation path
// SCHED: srn_layers' schedule (1: fragment-major for every width, render_stripe_kernel)
// SHADED: 0 = plain renderer; 1 = render_shaded_kernel (finite differences / predicted gradients / BRDF, and the adjoint mode at 96 and
// 128 channels); 2 = render_adjoint_kernel (up to 64 channels, adjoint mode only, see there)
constexpr bool adjoint_in_its_own_kernel(int CD) { return CD >= 2 && CD <= 4; }  // (render_adjoint_kernel is built for 32 / 48 / 64 channels)
// CELLS (render_small_kernel<.., SGRID = 2>, render_cells_kernel, render_shaded_cells_kernel): the decoded latent grid enters through the cell
// table (srn_device.hpp: cell_prepare / cells_accumulate)
template <int CD, int ACT, int GRID, bool HAS_DIR, int FMODE, int SHADED, int TAIL = TAIL_GENERIC, int NLC = 0, int SCHED = 0, bool CELLS = false, bool ADVANCE = true>
__device__ __forceinline__ void render_body(const NetParams& P, const SceneParams& S, const char* lds, const float* tfLds,
                                            float* __restrict__ out, unsigned long long* __restrict__ stats) {
    const int lane = lane_id();
    const int tilesX = (S.width + 7) >> 3;
    const int tilesY = (S.numLocalRows + 7) >> 3;
    const int numTiles = tilesX * tilesY;
    // Persistent waves: the launch holds as many workgroups as fit on the chip; every wave renders launch slot
    // (its own index), then keeps taking the next unrendered slot from a device counter until none is left.  Ray lengths
    // differ by 500x between tiles, so a static assignment leaves a quarter of the wave slots idle on average (r01 PMC:
    // SQ_WAVE_CYCLES).  The counter of the NEXT launch is zeroed here (launches of one scene are stream-ordered).
    const int wavesPerBlock = int(blockDim.x >> 6);
    const int totalWaves = int(gridDim.x) * wavesPerBlock;
    if (S.tileCounterNext && blockIdx.x == 0 && threadIdx.x == 0) *S.tileCounterNext = 0;
    unsigned nValid = 0, nSteps = 0;  // wave-uniform (scalar registers): lane-exact samples / executed wave steps
#ifdef FVSRN_PROF_SECTIONS
    for (int k = 0; k < kProfSections; ++k) P.prof[k] = 0;
#endif
    const bool rgboNet = P.outputMode == FVSRN_OUT_RGBO || P.outputMode == FVSRN_OUT_RGBO_DIRECT;
    // networks that predict the gradient (and, in the curvature modes, two curvature values the DVR path has no use for)
    const bool gradNet = P.outputMode >= FVSRN_OUT_DENSITY_GRADIENT && P.outputMode <= FVSRN_OUT_DENSITY_CURVATURE_DIRECT;
    // Depth segments: when a launch has fewer pixel tiles than a few times the wave slots of the chip (small images, one
    // rank's stripes of a multi-GPU frame) the host cuts every ray into K consecutive step ranges; a work unit is (tile,
    // segment), the partial results are composited front to back afterwards (blending is associative:
    // C = C1 + (1 - A1) C2).  The sample positions t = tmin + i * stepsize are the same as without segments.
#ifdef FVSRN_NO_SEGMENTS
    constexpr int K = 1;
#else
    const int K = S.segments;
#endif
    const int unitsPerFrame = numTiles * K;
    const int numFrames = S.frames > 1 ? S.frames : 1;  // wave-uniform (device_params.hpp: several camera poses in one launch)
    const int numUnits = unitsPerFrame * numFrames;
    ResidentNet<(NLC > 0 ? NLC : 1), (NLC > 0 && GRID == 1 ? (CELLS ? 2 : 1) : 0)> resident;
    if constexpr (NLC > 0) load_resident(P, lds, resident);
    // TAIL_SCALAR_TABLE: loop-invariant scalars
    const float alphaLimit = S.earlyOut ? S.alphaEarlyOut : __builtin_inff();
    // r06: the density bias of the straight-line tails kept in a vector register (their fma has two scalar operands otherwise: one v_mov per step), and the end of
    // a unit's step range folded into the ray's far end (below: two compares per step instead of three).  Same-box A/B of the two, profiles/r06/tail_bias_fold_ab_r06.txt:
    // headline 163.5 -> 165.7 G samples/s (+ 1.3 %), latent-grid and small-frame lines unchanged.  -DFVSRN_NO_TAIL_BIAS_FOLD builds the r05 form.
#ifndef FVSRN_NO_TAIL_BIAS_FOLD
    float densityBiasV = S.densityBias;
    asm volatile("" : "+v"(densityBiasV));
#else
    const float densityBiasV = S.densityBias;
#endif
    const bool sigmoidNet = P.outputMode == FVSRN_OUT_DENSITY;
    const bool textureTf = S.tfKind == FVSRN_TF_TEXTURE;
    const bool beerLambert = S.blendMode == FVSRN_BLEND_BEER_LAMBERT;
    int slot = int(blockIdx.x) * wavesPerBlock + int(threadIdx.x >> 6);
    int quota = S.unitQuota;  // wave-uniform
    if (quota > 0) {  // bounded waves: every unit comes from the counter
        int first = 0;
        if (lane == 0) first = atomicAdd(S.tileCounter, 1);
        slot = __builtin_amdgcn_readfirstlane(first);
    }
    for (; slot < numUnits;) {
    int frame = 0, unit = slot;
    if (numFrames > 1) { frame = slot / unitsPerFrame; unit = slot - frame * unitsPerFrame; }
    const float* cam = S.cams[frame];  // eye, right, up, front (scalar loads; the host fills cams[0] for single frames too)
    const int tileSlot = K > 1 ? unit / K : unit;
    const int seg = unit - tileSlot * K;
    // launch slot -> pixel tile: the host orders tiles by expected ray length (centre of the projected box first), so
    // the long tiles start first and the empty ones fill the tail of the launch
    const int tile = S.tileOrder ? S.tileOrder[tileSlot] : tileSlot;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int x = tx * 8 + (lane & 7);
    const int lrow = ty * 8 + (lane >> 3);
    const int y = S.y0 + ((lrow / S.stripeRows) * S.stripeWorld + S.stripeRank) * S.stripeRows + lrow % S.stripeRows;
    const bool inImage = x < S.width && lrow < S.numLocalRows && y < S.y1;

    // camera ray (renderer_image_evaluator_simple.cuh:84-88, renderer_camera.cuh:33-52), box intersection (renderer_utils.cuh:91-105)
    // and the ray in unit-box coordinates.  Once per ray, so every operation is spelled out (explicit fma's, IEEE division and square
    // root, no contraction by the compiler): the DEVICE arithmetic model of the parity tests (tests/test_fuzz_parity.py) restates this sequence
    // operation by operation, which makes the sample positions of the two bit-identical -- a network behind a 2^9 frequency ladder turns
    // one ulp of a position into a different fp16 rounding and that into percents of a colour.
    float dx, dy, dz, tmin, tmax, pn0x, pn0y, pn0z, dnx, dny, dnz;
    const float ox = cam[0], oy = cam[1], oz = cam[2];
    {
#pragma clang fp contract(off)
        const float ndcx = 2.f * (float(x) + 0.5f) / float(S.width) - 1.f;
        const float ndcy = 2.f * (float(y) + 0.5f) / float(S.height) - 1.f;
        const float ax = ndcx * S.tanFovX, ay = ndcy * S.tanFovY;
        dx = __builtin_fmaf(ay, cam[6], __builtin_fmaf(ax, cam[3], cam[9]));
        dy = __builtin_fmaf(ay, cam[7], __builtin_fmaf(ax, cam[4], cam[10]));
        dz = __builtin_fmaf(ay, cam[8], __builtin_fmaf(ax, cam[5], cam[11]));
        const float invLen = 1.0f / __builtin_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
        dx *= invLen; dy *= invLen; dz *= invLen;
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
        const float t1 = (P.boxMin[0] - ox) * ix, t2 = (P.boxMin[0] + P.boxSize[0] - ox) * ix;
        const float t3 = (P.boxMin[1] - oy) * iy, t4 = (P.boxMin[1] + P.boxSize[1] - oy) * iy;
        const float t5 = (P.boxMin[2] - oz) * iz, t6 = (P.boxMin[2] + P.boxSize[2] - oz) * iz;
        tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
        // position in unit-box coordinates as a function of t: p = pn0 + dn * t  (the reference evaluates ((o + d t) - boxMin) / boxSize,
        // renderer_volume_tensorcores.cuh:746; same value up to fp32 rounding)
        pn0x = (ox - P.boxMin[0]) * P.invBoxSize[0]; dnx = dx * P.invBoxSize[0];
        pn0y = (oy - P.boxMin[1]) * P.invBoxSize[1]; dny = dy * P.invBoxSize[1];
        pn0z = (oz - P.boxMin[2]) * P.invBoxSize[2]; dnz = dz * P.invBoxSize[2];
    }
    tmin = fmaxf(tmin, 0.f);  // stepping_dvr.cuh:66-67 (tmax input of the image evaluator is FLT_MAX)
    if (!inImage) tmax = -1.f;  // padding lanes never become valid
    // this unit's step range [i0, i1) of the ray (the last segment is open: it ends with t <= tmax like the unsegmented loop)
    int i0 = 0, i1 = 0x7fffffff;
    if (K > 1) {
        const float span = tmax - tmin;
        const int n = span >= 0.f ? int(span / S.stepsize) + 1 : 0;
        i0 = (n * seg) / K;
        if (seg + 1 < K) i1 = (n * (seg + 1)) / K;
    }

    float cr = 0, cg = 0, cb = 0, ca = 0, nx = 0, ny = 0, nz = 0, depth = 0;
    float previousDensity = -1.f;  // pre-integrated transfer functions (shaded kernel only), stepping_dvr.cuh:81
    // 32-wide Fourier-only networks keep the input features of the current sample and their per-step rotation in
    // registers (measured r01, 1024^2 x 512: 111.7 -> 118.1 Gsamples/s at 3 waves/SIMD; with a latent grid the 64 extra
    // registers cost more than the saved v_cos: 64.9 -> 62.4)
#ifdef FVSRN_NO_ROTATE
    constexpr bool kRotate = false;
#else
#ifndef FVSRN_ROTATE_GRID
#define FVSRN_ROTATE_GRID 0
#endif
#ifndef FVSRN_CELLS_ROTATE
#define FVSRN_CELLS_ROTATE 1  // (0: the cell-table kernels with direct features, A/B builds)
#endif
    constexpr bool kRotate = CD == 2 && (GRID == 0 || (GRID == 1 && FVSRN_ROTATE_GRID && NLC == 0) || (GRID == 1 && NLC > 0 && CELLS && FVSRN_CELLS_ROTATE)) &&
                             FMODE != FM_FIRST_LAYER && !SHADED;
#endif
    // r03 experiment, measured and NOT shipped (profiles/r03/sgrid_rotation_experiment.md): the register-resident kernel with one latent
    // chunk with rotated features, the per-ray rotation parked in LDS (8 KiB per wave behind the TF table; srn_forward_rotating_resident_grid):
    // 56 v_cos / v_sin and 2 phase MFMAs per wave step less, 32 packed-fp32 instructions and 8 LDS reads more -- 82.7 against 85.7
    // Gsamples/s for the direct features (32x4 + 16^3 grid).  -DFVSRN_ROTATE_SGRID=1 builds it (tools/variant.sh).
#ifndef FVSRN_ROTATE_SGRID
#define FVSRN_ROTATE_SGRID 0
#endif
    constexpr bool kRotateLds = FVSRN_ROTATE_SGRID && NLC > 0 && GRID == 1 && CD == 2 && !SHADED && FMODE == FM_COS;
    float* dfeatLds = nullptr;
    if constexpr (kRotateLds) dfeatLds = const_cast<float*>(tfLds) + S.tfLdsFloats + int(threadIdx.x >> 6) * (64 * 32);
    floatx16 feat[2][mtiles(CD)], dfeat[2][mtiles(CD)];
    // srn_forward's view of the latent grid: 3 = through the cell table (monomial form), 4 = through the corner-weight table of the shaded kernels (srn_device.hpp)
    constexpr int FG = (GRID == 1 && CELLS) ? (SHADED ? 4 : 3) : GRID;
    // the adjoint gradient mode lives in this instantiation (render_shaded_kernel except at 48 / 64 channels, render_adjoint_kernel there)
    constexpr bool kAdjointHere = SHADED == 2 || (SHADED == 1 && !adjoint_in_its_own_kernel(CD));
    [[maybe_unused]] bool normalsAtPreviousStep = false;  // (wave-uniform) per ray tile
    float stepIndex = float(i0);
    const float stepEnd = float(i1);  // INT_MAX -> 2^31: never reached
#ifndef FVSRN_NO_TAIL_BIAS_FOLD
    const float tSegEnd = fmaf(stepEnd - 0.5f, S.stepsize, tmin);
    [[maybe_unused]] const float tmaxEff = tSegEnd < tmax ? tSegEnd : tmax;  // (a NaN far end stays NaN: never valid)
#endif
    // render_small_kernel<.., SGRID = 2>: the wave's cell pair and its table fragment, kept from step to step (srn_device.hpp); none at the start of a work unit
    [[maybe_unused]] CellResident<1> cellPair;
    cellPair.valid = 0; cellPair.cA = cellPair.cB = kNoCell;
#ifdef FVSRN_CELLS_WIDE_RESIDENT_PAIR  // (A/B build: the kept pair in render_cells_kernel too; 4 more registers per M tile across the step loop)
    [[maybe_unused]] CellResident<mtiles(CD)> cellPairWide;
    cellPairWide.valid = 0; cellPairWide.cA = cellPairWide.cB = kNoCell;
#endif
    [[maybe_unused]] const float gdnx = dnx * P.gridXf, gdny = dny * P.gridYf, gdnz = dnz * P.gridZf, gp0x = pn0x * P.gridXf, gp0y = pn0y * P.gridYf, gp0z = pn0z * P.gridZf;

#ifdef FVSRN_PROF_SECTIONS
    P.profLast = __builtin_readcyclecounter();
#endif
    for (int i = 0;; ++i) {
        FVSRN_MARK(P, 0);  // tail of the previous step: output parametrization, TF, blending
        float t;
        bool inRange, valid;
        unsigned long long validMask;
        if constexpr (TAIL != TAIL_GENERIC) {  // straight-line: i0 = 0, i1 = INT_MAX without segments
            // the global step index i0 + i as a float counter (exact below 2^24): one add per step instead of add + convert.
            // The lane predicates live as wave masks in scalar registers only (three v_cmp, two s_and): kept as `bool`s hipcc
            // materialises them in VGPRs across the network code (v_cndmask + v_cmp_ne per step).
            t = fmaf(stepIndex, S.stepsize, tmin);
#ifndef FVSRN_NO_TAIL_BIAS_FOLD
            // the end of this unit's step range is part of the ray's far end (tmaxEff: t_i <= tmin + (i1 - 1/2) stepsize <=> i < i1): two compares
            validMask = __builtin_amdgcn_ballot_w64(t <= tmaxEff) & __builtin_amdgcn_ballot_w64(ca < alphaLimit);
#else
            validMask = __builtin_amdgcn_ballot_w64(t <= tmax) & __builtin_amdgcn_ballot_w64(stepIndex < stepEnd) &
                        __builtin_amdgcn_ballot_w64(ca < alphaLimit);
#endif
            stepIndex += 1.f;
            inRange = valid = false;  // (unused in the straight-line tails)
        } else {
            if (K > 1) {  // wave-uniform; kept as a branch so that the unsegmented loop does no per-lane index arithmetic
                const int gi = i0 + i;
                t = fmaf(float(gi), S.stepsize, tmin);
                inRange = (t <= tmax) & (gi < i1);
            } else {
                t = fmaf(float(i), S.stepsize, tmin);
                inRange = t <= tmax;
            }
            const bool notOpaque = ca < S.alphaEarlyOut;
            valid = bool(int(inRange) & (int(!S.earlyOut) | int(notOpaque)));  // branch-free
            // wave mask straight from the two v_cmp results (scalar ops only)
            validMask = __builtin_amdgcn_ballot_w64(inRange) & (S.earlyOut ? __builtin_amdgcn_ballot_w64(notOpaque) : ~0ull);
        }
        if (validMask == 0) break;  // wave-uniform: no lane of the wave is valid any more
        ++nSteps;
        nValid += unsigned(__builtin_popcountll(validMask));

        float px = fmaf(dnx, t, pn0x), py = fmaf(dny, t, pn0y), pz = fmaf(dnz, t, pn0z);
        if constexpr (!SHADED && NLC == 0) {  // (the register-resident kernels never see such a network: api.cpp)
            if (P.fourierClampPos) {  // wave-uniform; see device_params.hpp
                px = __builtin_amdgcn_fmed3f(px, 0.f, 1.f); py = __builtin_amdgcn_fmed3f(py, 0.f, 1.f); pz = __builtin_amdgcn_fmed3f(pz, 0.f, 1.f);
            }
        }
        float4_t o;
        [[maybe_unused]] bool fusedGradient = false;
        [[maybe_unused]] float fgx = 0.f, fgy = 0.f, fgz = 0.f;
        if constexpr (kRotate) {
            // Fourier features by rotation (fourier_advance): exact features every kFourierResync steps, the per-step
            // rotation once per ray
            if ((i & S.resyncMask) == 0) {  // wave-uniform
                const bool hilo = S.resyncMask != 0;  // features that get advanced: from the fp32 position (srn_device.hpp, fourier_features)
                fourier_features<CD, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz, feat, nullptr, hilo);
                if (i == 0)
                    fourier_features<CD, HAS_DIR, FM_COS, true>(P, lds, dnx * S.stepsize, dny * S.stepsize, dnz * S.stepsize, 0.f, 0.f, 0.f, dfeat, nullptr, hilo);
            }
            // ADVANCE = false: every step re-derives its features (FVSRN_OPT_FOURIER_RESYNC = 1, the reference's arithmetic): nothing to advance.  A
            // compile-time variant (render_small_kernel<.., ADVANCE>, chosen by the host): as a wave-uniform branch inside the step it cost the
            // default path 1.7 % (r04: 162.1 -> 159.4 Gsamples/s), leaving it out costs the exact mode 10 - 17 % (121.5 -> 134.6 .. 145.8)
            constexpr bool advance = ADVANCE;
#if FVSRN_CELLS_SCALED_POS
            // the cell table's coordinates p N straight from the ray parameter (per-ray constants gdn = dn N, gp0 = pn0 N): the position itself is only
            // needed where the features are re-derived
            if constexpr (NLC > 0 && CELLS) o = srn_forward_rotating_resident_cells<ACT, HAS_DIR, NLC, TAIL == TAIL_RGBO>(P, resident, cellPair, feat, dfeat, fmaf(gdnx, t, gp0x), fmaf(gdny, t, gp0y), fmaf(gdnz, t, gp0z), validMask, advance);
#else
            if constexpr (NLC > 0 && CELLS) o = srn_forward_rotating_resident_cells<ACT, HAS_DIR, NLC, TAIL == TAIL_RGBO>(P, resident, cellPair, feat, dfeat, px, py, pz, validMask, advance);
#endif
            else if constexpr (NLC > 0) o = srn_forward_rotating_resident<ACT, HAS_DIR, NLC, TAIL == TAIL_RGBO>(P, resident, feat, dfeat, advance);
            else o = srn_forward_rotating<CD, ACT, GRID, HAS_DIR>(P, lds, feat, dfeat, px, py, pz, advance);
        } else if constexpr (kRotateLds) {
            if ((i & S.resyncMask) == 0) {  // wave-uniform: exact features; the per-step rotation once per ray (and depth segment) -> LDS
                fourier_features<CD, HAS_DIR, FM_COS>(P, lds, px, py, pz, dx, dy, dz, feat);  // (phase fragment: from the LDS image)
                if (i == 0) {
                    fourier_features<CD, HAS_DIR, FM_COS, true>(P, lds, dnx * S.stepsize, dny * S.stepsize, dnz * S.stepsize, 0.f, 0.f, 0.f, dfeat);
                    park_dfeat(dfeatLds, dfeat);
                }
            }
            o = srn_forward_rotating_resident_grid<ACT, HAS_DIR, NLC, TAIL == TAIL_RGBO>(P, resident, feat, dfeatLds, px, py, pz);
        } else if constexpr (NLC > 0 && GRID == 1 && CELLS) {
            o = srn_forward_resident_cells<ACT, HAS_DIR, NLC, TAIL == TAIL_RGBO>(P, resident, px, py, pz, dx, dy, dz, validMask);
        } else if constexpr (NLC > 0 && GRID == 1) {
            o = srn_forward_resident_grid<ACT, HAS_DIR, NLC, TAIL == TAIL_RGBO>(P, resident, px, py, pz, dx, dy, dz);
        } else if constexpr (kAdjointHere) {
            // Adjoint mode: the gradient pass computes the value as well (its first column), so where the wave needed normals at the
            // previous step -- inside the volume that is every step -- the value comes out of that pass and the plain evaluation is
            // skipped: 4 tile evaluations per step instead of 5.  (The reference evaluates, then runs evalNormal where any lane needs it:
            // same numbers up to the rounding of the two evaluations, <= 5e-4 of the value, tests/test_gpu_parity.py.)
            fusedGradient = S.gradientMode == FVSRN_GRADIENT_ADJOINT_METHOD && !rgboNet && normalsAtPreviousStep;
            if (fusedGradient) o = srn_forward_gradient<CD, ACT, GRID, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz, S.gridDiffStep, fgx, fgy, fgz);
            else o = srn_forward<CD, ACT, FG, HAS_DIR, FMODE, SCHED>(P, lds, px, py, pz, dx, dy, dz, validMask);
        } else {
#ifdef FVSRN_CELLS_WIDE_RESIDENT_PAIR
            if constexpr (FG == 3) o = srn_forward<CD, ACT, FG, HAS_DIR, FMODE, SCHED>(P, lds, px, py, pz, dx, dy, dz, validMask, &cellPairWide);
            else
#endif
            o = srn_forward<CD, ACT, FG, HAS_DIR, FMODE, SCHED>(P, lds, px, py, pz, dx, dy, dz, validMask);
        }

        FVSRN_MARK(P, 5);  // last layer (+ the other half of the rotation)
#ifdef FVSRN_ABL_NOTAIL  // ablation build (tools/ablate.sh): no output parametrization / TF / blending
        cr += o[0] * 1e-30f;
        continue;
#endif
        if constexpr (TAIL == TAIL_RGBO) {  // stepping_dvr.cuh:104-109, 138-150: the network's colour, no TF, no normals
            float c0, c1, c2, c3;
            if (P.outputMode == FVSRN_OUT_RGBO) {  // wave-uniform
                asm volatile("");
                c0 = sigmoid_f(o[0]); c1 = sigmoid_f(o[1]); c2 = sigmoid_f(o[2]); c3 = softplus_f(o[3]);
            } else {
                c0 = fminf(fmaxf(o[0], 0.f), 1.f); c1 = fminf(fmaxf(o[1], 0.f), 1.f); c2 = fminf(fmaxf(o[2], 0.f), 1.f); c3 = fmaxf(o[3], 0.f);
            }
            c3 *= S.stepsize;
            const float aBeer = 1.f - __expf(-c3), aAlpha = fminf(1.f, c3);
            const float a = beerLambert ? aBeer : aAlpha;
            const float w = select_by_mask(validMask & __builtin_amdgcn_ballot_w64(c3 > 0.f), (1.f - ca) * a);
            cr += w * c0; cg += w * c1; cb += w * c2;
            depth += w * t;
            ca += w;
            continue;
        }
        if constexpr (TAIL != TAIL_GENERIC) {
            // stepping_dvr.cuh:110-150 for a scalar density network behind a transfer function, no normals: predicated, no
            // exec-mask regions.  Same operations in the same order as the generic tail below.
            float value = o[0];
            if (sigmoidNet) {  // wave-uniform; the empty asm keeps hipcc from turning the branch into a select (2 transcendentals)
                asm volatile("");
                value = sigmoid_f(value);
            }
            if constexpr (TAIL == TAIL_SCALAR_IDENTITY) {
                // Identity TF + Beer-Lambert blending (the host routes Alpha blending to TAIL_SCALAR_TABLE), with every constant folded on the
                // host: density = clamp01(value * s + b), alpha = 1 - 2^(density * k), k = -absorption * stepsize * log2 e.  The sample
                // counts where it is valid and value >= densityMin; "absorption > 0" of the reference is implied (density 0 gives alpha 0).
                const float density = __builtin_amdgcn_fmed3f(fmaf(value, S.divDensityRange, densityBiasV), 0.f, 1.f);
                const float a = 1.f - __builtin_amdgcn_exp2f(density * S.tfAbsorptionStepLog2e);
#ifndef FVSRN_NO_TAIL_FMA_WEIGHT  // r06: (1 - ca) a as a - ca a -- one fma where the product form takes a copy, a packed subtract and a product: headline 167.9 -> 170.4 G samples/s
                                   // same box (profiles/r06/tail_fma_weight_ab_r06.txt; the transmittance form T' = T e, w = T - T' of the same file gained nothing)
                const float w = select_by_mask(validMask & __builtin_amdgcn_ballot_w64(value >= S.densityMin), fmaf(-ca, a, a));
#else
                const float w = select_by_mask(validMask & __builtin_amdgcn_ballot_w64(value >= S.densityMin), (1.f - ca) * a);
#endif
                cr += w * density;  // emission scale behind the loop
                depth += w * t;
                ca += w;
                continue;
            }
            if constexpr (TAIL == TAIL_SCALAR_TEXTURE) {
                // Texture TF (renderer_tf_texture.cuh:46-55) + Beer-Lambert blending, constants folded like in TAIL_SCALAR_IDENTITY:
                // alpha = 1 - 2^(opacity * k), k = -stepsize * log2 e
                const float density = __builtin_amdgcn_fmed3f(fmaf(value, S.divDensityRange, densityBiasV), 0.f, 1.f);
                const int R = S.tfRows;
                const float d = fmaf(density, S.tfRowsF, -0.5f);
                const float fl = floorf(d);
                const int di = int(fl);
                const float df = d - fl;
                const float4_t ta = *reinterpret_cast<const float4_t*>(tfLds + 4 * min(max(di, 0), R - 1));
                const float4_t tb = *reinterpret_cast<const float4_t*>(tfLds + 4 * min(max(di + 1, 0), R - 1));
                const float a = 1.f - __builtin_amdgcn_exp2f((ta[3] + df * (tb[3] - ta[3])) * S.stepLog2e);
#ifndef FVSRN_NO_TAIL_FMA_WEIGHT
                const float w = select_by_mask(validMask & __builtin_amdgcn_ballot_w64(value >= S.densityMin), fmaf(-ca, a, a));
#else
                const float w = select_by_mask(validMask & __builtin_amdgcn_ballot_w64(value >= S.densityMin), (1.f - ca) * a);
#endif
                cr += w * (ta[0] + df * (tb[0] - ta[0]));
                cg += w * (ta[1] + df * (tb[1] - ta[1]));
                cb += w * (ta[2] + df * (tb[2] - ta[2]));
                depth += w * t;
                ca += w;
                continue;
            }
            const float density = fminf(fmaxf((value - S.densityMin) * S.divDensityRange, 0.f), 1.f);  // tf_eval clamps
            float c0, c1, c2, c3;
            if constexpr (TAIL == TAIL_SCALAR_LOOP) {  // Piecewise / Gaussian: per-lane loops over the control points
                const float4_t c = tf_eval(S, tfLds, density);
                c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3];
            } else if (textureTf) {  // wave-uniform; renderer_tf_texture.cuh:46-55
                const int R = S.tfRows;
                const float d = density * S.tfRowsF - 0.5f;
                const float fl = floorf(d);
                const int di = int(fl);
                const float df = d - fl;
                const float4_t a = *reinterpret_cast<const float4_t*>(tfLds + 4 * min(max(di, 0), R - 1));
                const float4_t b = *reinterpret_cast<const float4_t*>(tfLds + 4 * min(max(di + 1, 0), R - 1));
                c0 = a[0] + df * (b[0] - a[0]); c1 = a[1] + df * (b[1] - a[1]); c2 = a[2] + df * (b[2] - a[2]);
                c3 = (a[3] + df * (b[3] - a[3])) * S.stepsize;
            } else {  // renderer_tf_identity.cuh:36-54
                c0 = c1 = c2 = density * S.tfScaleEmission;
                c3 = density * S.tfAbsorptionStep;  // absorption * stepsize, multiplied on the host
            }
            // Blending::eval (renderer_blending.cuh:35-51) where the sample counts: valid, value >= densityMin, absorption > 0
            const float aBeer = 1.f - __expf(-c3), aAlpha = fminf(1.f, c3);
            const float a = beerLambert ? aBeer : aAlpha;
            const float w = select_by_mask(validMask & __builtin_amdgcn_ballot_w64(value >= S.densityMin) & __builtin_amdgcn_ballot_w64(c3 > 0.f),
                                           (1.f - ca) * a);
            cr += w * c0; cg += w * c1; cb += w * c2;
            depth += w * t;
            ca += w;
            continue;
        }
        float4_t color = {0, 0, 0, 0};
        float gx = 0, gy = 0, gz = 0;
        if (rgboNet) {  // stepping_dvr.cuh:104-109
            if (P.outputMode == FVSRN_OUT_RGBO)
                color = float4_t{sigmoid_f(o[0]), sigmoid_f(o[1]), sigmoid_f(o[2]), softplus_f(o[3])};
            else
                color = float4_t{fminf(fmaxf(o[0], 0.f), 1.f), fminf(fmaxf(o[1], 0.f), 1.f), fminf(fmaxf(o[2], 0.f), 1.f), fmaxf(o[3], 0.f)};
            color[3] *= S.stepsize;
        } else {  // stepping_dvr.cuh:110-135
            const bool sigmoidDensity = P.outputMode == FVSRN_OUT_DENSITY || P.outputMode == FVSRN_OUT_DENSITY_GRADIENT ||
                                        P.outputMode == FVSRN_OUT_DENSITY_CURVATURE;
            const float value = sigmoidDensity ? sigmoid_f(o[0]) : o[0];
            const float density2 = (value - S.densityMin) * S.divDensityRange;
            const bool requireNormal = valid && (value >= S.densityMin);
            if (SHADED == 1 && S.gradientMode == FVSRN_GRADIENT_FINITE_DIFFERENCES) {
                // evalNormal, GRADIENT_MODE_FINITE_DIFFERENCES (renderer_volume_tensorcores.cuh:1185-1196): central
                // differences of valueNoClamping with a world-space step, evaluated by the whole wave if ANY lane needs a
                // normal (stepping_dvr.cuh:122-128); takes precedence over gradients the network predicts
                if (__builtin_amdgcn_ballot_w64(requireNormal) != 0) {
                    const float h = S.fdStep;
                    const float hx = h * P.invBoxSize[0], hy = h * P.invBoxSize[1], hz = h * P.invBoxSize[2];
                    float v[6];
#pragma unroll 1
                    for (int k = 0; k < 6; ++k) {
                        const float sgn = (k & 1) ? -1.f : 1.f;
                        const int axis = k >> 1;
                        const float4_t ok = srn_forward<CD, ACT, FG, HAS_DIR, FMODE>(
                            P, lds, px + (axis == 0 ? sgn * hx : 0.f), py + (axis == 1 ? sgn * hy : 0.f), pz + (axis == 2 ? sgn * hz : 0.f), dx, dy, dz, validMask);
                        v[k] = sigmoidDensity ? sigmoid_f(ok[0]) : ok[0];
                    }
                    const float inv2h = 1.0f / (2.0f * h);
                    gx = (v[0] - v[1]) * inv2h; gy = (v[2] - v[3]) * inv2h; gz = (v[4] - v[5]) * inv2h;
                }
            } else if (SHADED && S.gradientMode == FVSRN_GRADIENT_ADJOINT_METHOD) {
                // evalNormal, GRADIENT_MODE_ADJOINT_METHOD (renderer_volume_tensorcores.cuh:1198-1540): the analytic gradient w.r.t.
                // the normalized position, by the whole wave if ANY lane needs a normal; forward mode, see srn_gradient.hpp
                if constexpr (kAdjointHere) {
                    const bool any = __builtin_amdgcn_ballot_w64(requireNormal) != 0;
                    if (fusedGradient) { gx = fgx; gy = fgy; gz = fgz; }
                    else if (any) (void)srn_forward_gradient<CD, ACT, GRID, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz, S.gridDiffStep, gx, gy, gz);
                    normalsAtPreviousStep = any;
                }
            } else if (gradNet) {
                gx = o[1]; gy = o[2]; gz = o[3];
                if (P.outputMode == FVSRN_OUT_DENSITY_GRADIENT_CUBIC) { gx = gx * gx * gx; gy = gy * gy * gy; gz = gz * gz * gz; }
            }
            if (requireNormal) {
                if (SHADED && S.tfPreintegration != FVSRN_PREINTEGRATE_NONE)
                    color = tf_eval_preintegrated(S, tfLds, fminf(fmaxf(density2, 0.f), 1.f), previousDensity);
                else
                    color = tf_eval(S, tfLds, density2, sqrtf(gx * gx + gy * gy + gz * gz), previousDensity);
            }
            if (SHADED) previousDensity = density2;  // stepping_dvr.cuh:135
        }
        if (SHADED && color[3] > 0.f && (S.brdfMagnitudeScaling | S.brdfPhong)) {
            // BRDFLambert::eval (renderer_brdf_lambert.cuh:56-103) on the un-normalised gradient; "gradientNorm" is the
            // reference's rsqrt(|g|^2)
            const float g2 = gx * gx + gy * gy + gz * gz;
            if (S.brdfMagnitudeScaling) color[3] *= 1.f - __expf(-S.brdfMagScale * g2);
            if (S.brdfPhong) {
                const float gradientNorm = rsqrtf(g2);
                float nX = gx, nY = gy, nZ = gz;
                if (g2 >= 1e-8f) { nX *= gradientNorm; nY *= gradientNorm; nZ *= gradientNorm; }  // safeNormalize
                float lx, ly, lz;
                if (S.brdfLightType == FVSRN_LIGHT_DIRECTIONAL) { lx = -S.brdfLight[0]; ly = -S.brdfLight[1]; lz = -S.brdfLight[2]; }
                else {  // point light: towards the light from the world position of the sample
                    lx = S.brdfLight[0] - (ox + dx * t); ly = S.brdfLight[1] - (oy + dy * t); lz = S.brdfLight[2] - (oz + dz * t);
                }
                const float il = rsqrtf(lx * lx + ly * ly + lz * lz);
                lx *= il; ly *= il; lz *= il;
                const float lo = S.brdfMagCenter - S.brdfMagRadius, hi = S.brdfMagCenter + S.brdfMagRadius;
                const float ys = fminf(fmaxf((gradientNorm - lo) / (hi - lo), 0.f), 1.f);
                const float phongStrength = ys * ys * (3.f - 2.f * ys);                       // smoothstep
                const float ambientStrength = 1.f + phongStrength * (S.brdfAmbient - 1.f);    // lerp(1, ambient, s)
                const float ndl = fabsf(nX * lx + nY * ly + nZ * lz);
                // reflect(lightDirection, -normal) = l - 2 (-n) dot(-n, l) = l - 2 n dot(n, l)
                const float nl = nX * lx + nY * ly + nZ * lz;
                const float rx = lx - 2.f * nX * nl, ry = ly - 2.f * nY * nl, rz = lz - 2.f * nZ * nl;
                const float e = float(S.brdfSpecularExponent);
                const float spec = (e + 2.f) * 0.159155f * powf(fmaxf(0.f, dx * rx + dy * ry + dz * rz), e);
                for (int c = 0; c < 3; ++c)
                    color[c] = ambientStrength * color[c] + (1.f - ambientStrength) * (ndl * color[c] + S.brdfSpecular * spec);
            }
        }
        if (color[3] > 0.f && valid) {
            // Blending::eval (renderer_blending.cuh:35-51)
            const float a = S.blendMode == FVSRN_BLEND_BEER_LAMBERT ? 1.f - __expf(-color[3]) : fminf(1.f, color[3]);
            const float w = (1.f - ca) * a;
            cr += w * color[0]; cg += w * color[1]; cb += w * color[2];
            depth += w * t;
            ca += w;
            if (gradNet || (SHADED && S.gradientMode != FVSRN_GRADIENT_OFF_OR_DIRECT)) {  // wave-uniform: there is a normal
                // safeNormalize (helper_math.cuh:2443-2448)
                const float l2 = gx * gx + gy * gy + gz * gz;
                if (l2 >= 1e-8f) { const float il = rsqrtf(l2); gx *= il; gy *= il; gz *= il; }
                nx += w * gx; ny += w * gy; nz += w * gz;
            }
        }
    }

    if constexpr (TAIL == TAIL_SCALAR_IDENTITY) { cr *= S.tfScaleEmission; cg = cr; cb = cr; }
    if (K > 1) {
        if (inImage) {  // raw accumulators of this segment; composite_kernel finishes the pixel
            const size_t plane = size_t(S.width) * (S.compact ? S.numLocalRows : S.height);
            float* p = S.partial + (size_t(frame) * size_t(K) + size_t(seg)) * 8 * plane + size_t(S.compact ? lrow : y) * S.width + x;
            p[0] = cr; p[plane] = cg; p[2 * plane] = cb; p[3 * plane] = ca;
            p[4 * plane] = nx; p[5 * plane] = ny; p[6 * plane] = nz; p[7 * plane] = depth;
        }
    } else if (inImage) {  // renderer_image_evaluator_simple.cuh:100-124 with samples == 1
        const size_t plane = size_t(S.width) * (S.compact ? S.numLocalRows : S.height);
        const size_t o = size_t(frame) * 8 * plane + size_t(S.compact ? lrow : y) * S.width + x;
        out[o] = cr;
        out[plane + o] = cg;
        out[2 * plane + o] = cb;
        out[3 * plane + o] = ca;
        out[4 * plane + o] = nx * ca;  // out.normal += nout.normal * nout.color.w
        out[5 * plane + o] = ny * ca;
        out[6 * plane + o] = nz * ca;
        out[7 * plane + o] = depth * ca / ca;  // (depth*alpha)/alpha, NaN for alpha == 0 like the reference
    }
    if (S.unitQuota > 0) {
        if (--quota == 0) break;
        int next = 0;
        if (lane == 0) next = atomicAdd(S.tileCounter, 1);
        slot = __builtin_amdgcn_readfirstlane(next);
    } else if (S.tileCounter) {
        int next = 0;
        if (lane == 0) next = atomicAdd(S.tileCounter, 1);
        slot = totalWaves + __builtin_amdgcn_readfirstlane(next);
    } else {
        slot += totalWaves;
    }
    }  // slot loop
    if (stats && lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)nValid);
        atomicAdd(&stats[1], (unsigned long long)nSteps * 64ull);
#ifdef FVSRN_PROF_SECTIONS
        for (int k = 0; k < kProfSections; ++k) atomicAdd(&stats[2 + k], P.prof[k]);
#endif
    }
}

// network image + TF table (behind it) -> LDS
__device__ __forceinline__ float* render_prologue(const NetParams& P, const SceneParams& S, char* lds) {
    load_network_to_lds(P, lds);
    float* tfLds = reinterpret_cast<float*>(lds + P.ldsBytes);
    const int cols = S.tfKind == FVSRN_TF_GAUSSIAN ? 6 : (S.tfKind == FVSRN_TF_PIECEWISE ? 5 : (S.tfKind == FVSRN_TF_TEXTURE ? 4 : 0));
    for (int i = threadIdx.x; i < cols * S.tfRows; i += int(blockDim.x)) tfLds[i] = S.tfTable[i];
    __syncthreads();
    return tfLds;
}

template <int CD, int ACT, int GRID, bool HAS_DIR, int SHADED, int SCHED = 0, bool CELLS = false>
__device__ __forceinline__ void render_entry(const NetParams& P, const SceneParams& S, float* __restrict__ out,
                                             unsigned long long* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* tfLds = render_prologue(P, S, lds);
    if constexpr (GRID == 0) {
        if (P.noFourier) return render_body<CD, ACT, GRID, HAS_DIR, FM_FIRST_LAYER, SHADED, TAIL_GENERIC, 0, SCHED, CELLS>(P, S, lds, tfLds, out, stats);
    }
    if constexpr (!SHADED) {
        // The straight-line tails: a scalar network behind an Identity or Texture TF with Beer-Lambert blending, phases inside the v_cos /
        // v_sin domain (with the sine rows on v_sin_f32 that includes the 2^9 NeRF ladder of a 64-wide network, which took the generic tail
        // until r02: 64x6 + 32^3 grid 24.3 -> 25.9 Gsamples/s).  Alpha blending and phases that need v_fract take the generic tail.
        const bool scalarNet = P.outputMode == FVSRN_OUT_DENSITY || P.outputMode == FVSRN_OUT_DENSITY_DIRECT;
        if (scalarNet && !P.fourierNeedsFract && S.blendMode == FVSRN_BLEND_BEER_LAMBERT) {
            if (S.tfKind == FVSRN_TF_IDENTITY && S.tfAbsorptionStepLog2e <= 0.f)
                return render_body<CD, ACT, GRID, HAS_DIR, FM_COS, SHADED, TAIL_SCALAR_IDENTITY, 0, SCHED, CELLS>(P, S, lds, tfLds, out, stats);
            if (S.tfKind == FVSRN_TF_TEXTURE && S.tfOpacityNonNegative)  // (a negative opacity: the generic tail skips the sample, stepping_dvr.cuh:137)
                return render_body<CD, ACT, GRID, HAS_DIR, FM_COS, SHADED, TAIL_SCALAR_TEXTURE, 0, SCHED, CELLS>(P, S, lds, tfLds, out, stats);
        }
    }
    if (P.fourierNeedsFract) return render_body<CD, ACT, GRID, HAS_DIR, FM_FRACT_COS, SHADED, TAIL_GENERIC, 0, SCHED, CELLS>(P, S, lds, tfLds, out, stats);
    render_body<CD, ACT, GRID, HAS_DIR, FM_COS, SHADED, TAIL_GENERIC, 0, SCHED, CELLS>(P, S, lds, tfLds, out, stats);
}

// The layer order of the 48- / 64-wide latent-grid variants (the gather path: BYTE_GAUSSIAN grids, launches outside the cell table's footprint rule) is the
// fragment-major one (SCHED = 1) since r05: the pipelined order of srn_layers spilled 67 - 79 (FLOAT / BYTE_LINEAR gathers) and 172 - 213 registers
// (BYTE_GAUSSIAN) there, the fragment-major order none / one, at 25.8 - 26.1 against 26.3 and 14.0 against 13.8 Gsamples/s (64x6 + 32^3 grid; r03 built it as a
// separate render_stripe_kernel, r05 measured it against one wave per SIMD -- 8.9 -- and made it THE kernel of these variants: profiles/r05/experiments_r05.md).
constexpr int render_layer_schedule(int CD, int GRID) { return GRID != 0 && (CD == 3 || CD == 4) ? 1 : 0; }

template <int CD, int ACT, int GRID, bool HAS_DIR>
__global__ __launch_bounds__(kBlockThreads, min_waves_per_simd(CD, GRID)) void render_kernel(NetParams P, SceneParams S, float* __restrict__ out,
                                                               unsigned long long* __restrict__ stats) {
    render_entry<CD, ACT, GRID, HAS_DIR, 0, render_layer_schedule(CD, GRID)>(P, S, out, stats);
}

// render_kernel<CD, ACT, 1, HAS_DIR> with the decoded latent grid through the cell table (r04: srn_device.hpp cell_prepare / cells_accumulate; the
// unshaded renderer of every network with a FLOAT / BYTE_LINEAR grid that has a table, FVSRN_OPT_CELL_TABLE)
template <int CD, int ACT, bool HAS_DIR>
__global__ __launch_bounds__(kBlockThreads, min_waves_per_simd(CD, 1)) void render_cells_kernel(NetParams P, SceneParams S, float* __restrict__ out,
                                                                     unsigned long long* __restrict__ stats) {
    render_entry<CD, ACT, 1, HAS_DIR, 0, 0, true>(P, S, out, stats);
}

// 32-wide Fourier-only scalar networks with NLC <= 3 C->C layers behind a transfer function, phases inside the v_cos domain
// (the host checks all of that, api.cpp): weights and biases in registers, 2 waves per SIMD (256 registers)
// TAILK: TAIL_SCALAR_IDENTITY / TAIL_SCALAR_TEXTURE (Identity / Texture TF with Beer-Lambert blending), TAIL_SCALAR_TABLE (the same two with Alpha
// blending), TAIL_SCALAR_LOOP (Piecewise / Gaussian TF) or TAIL_RGBO (colour network)
// SGRID = 1: the same with ONE 16-channel latent grid chunk of decoded values (direct Fourier features instead of the rotation:
// the registers of the rotation state hold the grid fetch)
// SGRID = 2 (r04): a latent grid of any channel count through the cell table -- one MFMA K step on the trilinear weights, no gathers,
// rotated features (srn_forward_rotating_resident_cells, srn_device.hpp); taken while a pixel tile spans less than a grid cell (FVSRN_OPT_CELL_TABLE, api.cpp)
// ADVANCE = false: the variant for FVSRN_OPT_FOURIER_RESYNC = 1 (every step re-derives its features like the reference: no rotation to advance; its own
// kernel -- as a second copy of the loop inside one kernel it cost the default path 6 % through the shared register allocation)
template <int ACT, bool HAS_DIR, int NLC, int TAILK, int SGRID = 0, bool ADVANCE = true>
__global__ __launch_bounds__(kBlockThreads, 2) void render_small_kernel(NetParams P, SceneParams S, float* __restrict__ out,
                                                                      unsigned long long* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* tfLds = render_prologue(P, S, lds);
    render_body<2, ACT, (SGRID ? 1 : 0), HAS_DIR, FM_COS, 0, TAILK, NLC, 0, SGRID == 2, ADVANCE>(P, S, lds, tfLds, out, stats);
}

// finite-difference normals / shading BRDF (SceneParams::gradientMode, brdf*): 7 network evaluations per sample
template <int CD, int ACT, int GRID, bool HAS_DIR>
// (80 / 112 / 128 channels: FVSRN_WAVES_PER_EU_SHADED_WIDE above)
__global__ __launch_bounds__(kBlockThreads, ((CD >= 7 || CD == 5) ? FVSRN_WAVES_PER_EU_SHADED_WIDE : 2)) void render_shaded_kernel(NetParams P, SceneParams S, float* __restrict__ out,
                                                               unsigned long long* __restrict__ stats) {
    render_entry<CD, ACT, GRID, HAS_DIR, 1>(P, S, out, stats);
}

// render_shaded_kernel<CD, ACT, 1, HAS_DIR> with the decoded latent grid through the cell table in every plain evaluation -- the sample itself and the
// six of the finite differences, whose positions share cells like the samples of a step do (r04).  The adjoint mode's gradient pass keeps its records.
template <int CD, int ACT, bool HAS_DIR>
__global__ __launch_bounds__(kBlockThreads, ((CD >= 7 || CD == 5) ? FVSRN_WAVES_PER_EU_SHADED_WIDE : 2)) void render_shaded_cells_kernel(NetParams P, SceneParams S, float* __restrict__ out,
                                                                     unsigned long long* __restrict__ stats) {
    render_entry<CD, ACT, 1, HAS_DIR, 1, 0, true>(P, S, out, stats);
}

// GRADIENT_MODE_ADJOINT_METHOD up to 64 channels: the shaded renderer without the finite-difference code (and render_shaded_kernel without
// the gradient pass: with both in one kernel the finite differences of the 32-wide latent-grid network lost 14 %, 38.9 -> 44.4 ms, to the
// spills of the other mode).  At 48 / 64 channels with the 512-register budget of one wave per SIMD: what the four column sets of the
// gradient pass do not fit into 256 registers goes to AGPRs instead of scratch memory (measured r03, 64x6 + 32^3 grid: 130 -> 106 ms per
// frame; finite differences keep two waves per SIMD, which are worth 35 % to them: 113 vs 153 ms).  At 32 channels the same holds with a
// latent grid (12 records in flight: 37.6 -> 36.2 ms, 28 scratch stores per wave step -> none); without one 256 registers are enough and the
// second wave is worth more (20.5 vs 24.0 ms).
template <int CD, int ACT, int GRID, bool HAS_DIR>
__global__ __launch_bounds__(kBlockThreads, ((CD >= 3 || GRID != 0) ? 1 : 2)) void render_adjoint_kernel(NetParams P, SceneParams S, float* __restrict__ out,
                                                               unsigned long long* __restrict__ stats) {
    static_assert(adjoint_in_its_own_kernel(CD), "only instantiated where render_shaded_kernel leaves the adjoint mode out");
    render_entry<CD, ACT, GRID, HAS_DIR, 2>(P, S, out, stats);
}

}  // namespace fvsrn
