// Device side of the SRN evaluator for gfx950 (wave64, v_mfma_f32_32x32x16_f16).
//
// Replaces kernel::VolumeInterpolationTensorcores::eval
// (reference renderer/renderer_volume_tensorcores.cuh:735-1164) with a different dataflow:
//   * one wave evaluates 64 samples (one per lane) as two 32-sample MFMA column tiles;
//     lane (c,h) = (lane&31, lane>>5) owns sample c+32h and, inside tile t, half of the channels
//     of sample c+32t
//   * the Fourier phases are an MFMA too (positions as a K=16 B operand, matrix split hi/lo in f16
//     so phases carry ~22 bits, in revolutions for v_cos_f32), replacing the hmul/hfma chain :797-806
//   * the fp32 accumulator tile of layer l is converted in registers (v_cvt_pk_f16_f32) into the
//     B operand of layer l+1 -- weights were permuted on the host for that (pack.cpp); the
//     reference round-trips activations through shared memory every layer (:1019-1023)
//   * biases enter as the MFMA C operand (fp32), accumulation is fp32 (reference: half, :965)
//   * the last (C -> 1|4) layer is a v_mfma_f32_16x16x32_f16 per K step and tile on the same B fragments (its weight
//     fragment routes every output to the lane that owns the sample, pack.cpp), replacing the per-lane hfma loop :1138-1143
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_params.hpp"
#include "srn_device_enums.hpp"

namespace fvsrn {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));



__device__ __forceinline__ int lane_id() { return int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); }

// Hardware erratum found in r04 (profiles/r04/nondeterminism_r04.md, tools/microbench/r04_pk_opsel_sweep.hip): a packed-fp32 instruction
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose LOW pass selects src0's low and src1's HIGH register -- op_sel:[0,1] -- reads one operand
// as 0 in lanes 48-63 while another wave of the SIMD has MFMAs in flight.  hipcc emits that selection when it vectorises scalar fp32 code
// (the latent-grid tap arithmetic below): the launch-to-launch differences of the latent-grid kernels that r03 spaced out with s_nops.  The
// build rewrites every such instruction with its two (commuting) sources exchanged (tools/fix_pk_opsel.py via hipcc_fixed.sh); hand-written
// packed-fp32 assembly in this file uses the selections measured clean ([0,0], [1,0], [1,1]); a CPU test scans the built objects.
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2_t lane_half_swap(unsigned a, unsigned b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return uint2_t{r[0], r[1]};
}

// own value v (one per lane = per sample) -> value of sample (lane&31) of tile 0 / tile 1
__device__ __forceinline__ void tile_bcast(float v, float& t0, float& t1) {
    const unsigned u = __float_as_uint(v);
    auto r = lane_half_swap(u, u);
    t0 = __uint_as_float(r[0]);
    t1 = __uint_as_float(r[1]);
}

template <int ACT>
__device__ __forceinline__ float act_f32(float x, float a, float b) {
    if constexpr (ACT == ACT_RELU || ACT == ACT_RELU01) {
        return fmaxf(x, 0.f);
    } else if constexpr (ACT == ACT_SINE) {  // sin(p x), a = p/(2 pi)          renderer_activations.cuh: Sine
        return __builtin_amdgcn_sinf(x * a);
    } else if constexpr (ACT == ACT_SNAKE) {  // x + sin^2(p x)/p = x + (1 - cos(2 p x))/(2p), a = p/pi, b = 1/(2p)
        const float c = __builtin_amdgcn_cosf(x * a);
        return fmaf(-c, b, x + b);
    } else if constexpr (ACT == ACT_SIGMOID) {  // 1 / (1 + exp(-x)), a = -log2(e)
        return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * a));
    } else if constexpr (ACT == ACT_SNAKEALT0) {  // x - cos(2 p x): the rest of SnakeAlt sits in the next layer (pack.cpp)
        return x - __builtin_amdgcn_cosf(x * a);
    } else {  // SnakeAlt: (x + 1 - cos(2 p x)) / (2p)
        const float c = __builtin_amdgcn_cosf(x * a);
        return fmaf(x - c, b, b);
    }
}

// Two activations.  Packed fp32 instructions (v_pk_mul/add/fma_f32) are the default.  Measured on MI355X (r02,
// tools/microbench/r02_issue.hip, profiles/r02/microbench_issue_model_r02.md): one of them placed BEHIND an MFMA of the same
// wave costs ~7.5 cycles and a partner wave issues only 0.25 of them per MFMA of this wave (against 2 - 4 scalar fp32
// instructions) -- but in the vector phase of the step, where they sit, the halved instruction count wins: scalar forms
// (FVSRN_PK_F32=0) 153.5 -> 145.5 Gsamples/s (32x4 ReLU), 63.4 -> 55.0 (SnakeAlt).
#ifndef FVSRN_PK_F32
#define FVSRN_PK_F32 1
#endif
// Wave priorities in the LDS kernels like in srn_layers_resident (MFMA chain 0, vector phases 3).  Measured r02 (1024^2 x 512):
// 32-wide + 16^3 grid 73.9 -> 76.0 Gsamples/s, 32x4 Fourier-only 139.1 -> 140.2, but 64x6 + grid 24.65 -> 24.3: 32-wide networks only
#ifndef FVSRN_LDS_PRIO
#define FVSRN_LDS_PRIO 1
#endif
template <int ACT>
__device__ __forceinline__ float2_t act_f32x2(float2_t x, float a, float b) {
#if !FVSRN_PK_F32
    if constexpr (ACT == ACT_RELU || ACT == ACT_RELU01) return float2_t{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    else return float2_t{act_f32<ACT>(x[0], a, b), act_f32<ACT>(x[1], a, b)};
#else
    if constexpr (ACT == ACT_SINE) {
        const float2_t t = x * a;
        return float2_t{__builtin_amdgcn_sinf(t[0]), __builtin_amdgcn_sinf(t[1])};
    } else if constexpr (ACT == ACT_SNAKE) {
        const float2_t t = x * a;
        const float2_t c = {__builtin_amdgcn_cosf(t[0]), __builtin_amdgcn_cosf(t[1])};
        const float2_t bb = {b, b};
        return __builtin_elementwise_fma(-c, bb, x + bb);
    } else if constexpr (ACT == ACT_SNAKEALT) {
        const float2_t t = x * a;
        const float2_t c = {__builtin_amdgcn_cosf(t[0]), __builtin_amdgcn_cosf(t[1])};
        const float2_t bb = {b, b};
        return __builtin_elementwise_fma(x - c, bb, bb);
    } else if constexpr (ACT == ACT_SNAKEALT0) {
        const float2_t t = x * a;
        const float2_t c = {__builtin_amdgcn_cosf(t[0]), __builtin_amdgcn_cosf(t[1])};
        return x - c;
    } else if constexpr (ACT == ACT_SIGMOID) {  // renderer_activations.cuh:152-179: 1 / (1 + exp(-x)); a = -log2(e)
        const float2_t t = x * a;
        return float2_t{__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t[0])), __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t[1]))};
    } else {
        return float2_t{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    }
#endif
}

// 16 fp32 accumulator values of one M tile -> two B fragments (K steps 2m, 2m+1) with activation
template <int ACT>
__device__ __forceinline__ void act_pack(const floatx16& d, float a, float b, half8_t& f0, half8_t& f1) {
    if constexpr (ACT == ACT_RELU01) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float2_t v0 = {d[2 * i], d[2 * i + 1]};
            float2_t v1 = {d[8 + 2 * i], d[8 + 2 * i + 1]};
            half2_t h0 = __builtin_convertvector(v0, half2_t);
            half2_t h1 = __builtin_convertvector(v1, half2_t);
            const half2_t z = {0, 0}, o = {1, 1};
            h0 = __builtin_elementwise_min(__builtin_elementwise_max(h0, z), o);  // folds into the convert's clamp bit
            h1 = __builtin_elementwise_min(__builtin_elementwise_max(h1, z), o);
            f0[2 * i] = h0[0]; f0[2 * i + 1] = h0[1];
            f1[2 * i] = h1[0]; f1[2 * i + 1] = h1[1];
        }
    } else if constexpr (ACT == ACT_RELU) {
        // convert first, then one packed max per register pair
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float2_t v0 = {d[2 * i], d[2 * i + 1]};
            float2_t v1 = {d[8 + 2 * i], d[8 + 2 * i + 1]};
            half2_t h0 = __builtin_convertvector(v0, half2_t);
            half2_t h1 = __builtin_convertvector(v1, half2_t);
            const half2_t z = {0, 0};
            h0 = __builtin_elementwise_max(h0, z);
            h1 = __builtin_elementwise_max(h1, z);
            f0[2 * i] = h0[0]; f0[2 * i + 1] = h0[1];
            f1[2 * i] = h1[0]; f1[2 * i + 1] = h1[1];
        }
    } else {
        // two values per instruction where the ISA has a packed fp32 form (v_pk_mul/add/fma_f32); cos/sin are scalar
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2_t x0 = {d[2 * i], d[2 * i + 1]};
            const float2_t x1 = {d[8 + 2 * i], d[8 + 2 * i + 1]};
            const half2_t h0 = __builtin_convertvector(act_f32x2<ACT>(x0, a, b), half2_t);
            const half2_t h1 = __builtin_convertvector(act_f32x2<ACT>(x1, a, b), half2_t);
            f0[2 * i] = h0[0]; f0[2 * i + 1] = h0[1];
            f1[2 * i] = h1[0]; f1[2 * i + 1] = h1[1];
        }
    }
}

__device__ __forceinline__ half8_t lds_frag(const char* lds, int byteOff, int lane) {
    return *reinterpret_cast<const half8_t*>(lds + byteOff + 16 * lane);
}

// bias rows of M tile m for this lane half as an MFMA C operand
__device__ __forceinline__ floatx16 lds_bias(const char* lds, int byteOff, int h) {
    floatx16 c;
    const float4_t* p = reinterpret_cast<const float4_t*>(lds + byteOff + 16 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4_t v = p[2 * g];  // rows 8g + 4h .. +3
        c[4 * g + 0] = v[0]; c[4 * g + 1] = v[1]; c[4 * g + 2] = v[2]; c[4 * g + 3] = v[3];
    }
    return c;
}

// ---- latent grid ------------------------------------------------------------------------------------------
// Texture semantics of the reference (normalized coords, clamp addressing, linear filter;
// renderer_volume_tensorcores.cuh:581-596, volume_interpolation_network.cpp:495-503): texel coordinate = p*N - 0.5.
// The working grid is stored as x-pair records (pack.cpp), so a trilinear fetch of one channel is 4 x
// v_dot2_f32_f16: record (z,y) holds {v(x0), v(x0+1)} and the packed fp16 weight pair is
// {w_zy*(1-wx), w_zy*wx}.  (fp16 filter weights: 11 bits; CUDA texture units filter with 8 fractional bits.)
struct GridTap {
    unsigned off[4];  // byte offsets of the 4 (z,y) records of this sample (channel 0)
    unsigned w[4];    // packed fp16 weight pairs
    unsigned wlo[4];  // their fp16 rounding residuals (only the BYTE_GAUSSIAN path, whose erfinv amplifies errors, uses them)
};

__device__ __forceinline__ GridTap grid_tap(const NetParams& P, float px, float py, float pz) {
    // Record indices in fp32 (exact below 2^24 records, which pack.cpp guarantees): clamps are v_med3_f32, the row / record arithmetic
    // is 8 v_fma_f32 -- the integer form needed 10 min / max, 3 v_mul_lo_u32 and 4 v_mad_u64_u32 (quarter rate) per sample.
    float fx = fmaf(px, P.gridXf, -0.5f), fy = fmaf(py, P.gridYf, -0.5f), fz = fmaf(pz, P.gridZf, -0.5f);
    float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    float wx = fx - x0f, wy = fy - y0f, wz = fz - z0f;
    float xi = __builtin_amdgcn_fmed3f(x0f + 1.f, 0.f, P.gridXf);  // record index: x-clamping is baked into the records
    const float ym = P.gridYf - 1.f, zm = P.gridZf - 1.f;
    float y0 = __builtin_amdgcn_fmed3f(y0f, 0.f, ym), y1 = __builtin_amdgcn_fmed3f(y0f + 1.f, 0.f, ym);
    float z0 = __builtin_amdgcn_fmed3f(z0f, 0.f, zm), z1 = __builtin_amdgcn_fmed3f(z0f + 1.f, 0.f, zm);
    const float rowLen = P.gridXf + 1.f;                     // records per (z, y) row
    const unsigned rec = unsigned(P.gridC) * 4u;             // bytes per record: G channels x 2 x fp16
    float r00 = fmaf(z0, P.gridYf, y0), r01 = fmaf(z0, P.gridYf, y1), r10 = fmaf(z1, P.gridYf, y0), r11 = fmaf(z1, P.gridYf, y1);
    GridTap t;
    t.off[0] = __umul24(unsigned(fmaf(r00, rowLen, xi)), rec);
    t.off[1] = __umul24(unsigned(fmaf(r01, rowLen, xi)), rec);
    t.off[2] = __umul24(unsigned(fmaf(r10, rowLen, xi)), rec);
    t.off[3] = __umul24(unsigned(fmaf(r11, rowLen, xi)), rec);
    float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
    float w4[4] = {uz * uy, uz * wy, wz * uy, wz * wy};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float2_t v = {w4[k] * ux, w4[k] * wx};
        const half2_t hi = __builtin_convertvector(v, half2_t);
        const float2_t r = v - __builtin_convertvector(hi, float2_t);
        t.w[k] = __builtin_bit_cast(unsigned, hi);
        t.wlo[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2_t));
    }
    return t;
}

// own-sample tap -> taps of sample (lane&31) of tile 0 / tile 1, their offsets INCLUDING this lane half's 32 bytes (kTapHalfInOffset for
// grid_load / grid_features; taps straight from grid_tap() take the lane half there)
constexpr int kTapHalfInOffset = 0;
template <bool WITH_LO>
__device__ __forceinline__ void grid_tap_bcast(const GridTap& own, GridTap& t0, GridTap& t1) {
    // the consumer's lane-half offset (lane half h reads channels [16 g + 8 h, +8): 32 bytes of a record) rides along: the second
    // operand of the swap is a copy of the first anyway, here it is an add (8 v_add_u32 per wave step less in grid_load)
    unsigned o0[4], o1[4], w0[4], w1[4], l0[4], l1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o0[k] = own.off[k]; o1[k] = own.off[k] + 32u;
        w0[k] = own.w[k]; w1[k] = own.w[k];
        l0[k] = own.wlo[k]; l1[k] = own.wlo[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto a = lane_half_swap(o0[k], o1[k]);
        t0.off[k] = a[0]; t1.off[k] = a[1];
        auto b = lane_half_swap(w0[k], w1[k]);
        t0.w[k] = b[0]; t1.w[k] = b[1];
        if constexpr (WITH_LO) {
            auto c = lane_half_swap(l0[k], l1[k]);
            t0.wlo[k] = c[0]; t1.wlo[k] = c[1];
        }
    }
}

// single-precision inverse error function (M. Giles 2010), relative error ~1e-7 like CUDA's erfinvf
__device__ __forceinline__ float erfinv_dev(float x) {
    float w = -__logf((1.0f - x) * (1.0f + x)), p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f; p = fmaf(p, w, 3.43273939e-07f); p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f); p = fmaf(p, w, 0.00021858087f); p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f); p = fmaf(p, w, 0.246640727f); p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f; p = fmaf(p, w, 0.000100950558f); p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f); p = fmaf(p, w, 0.00573950773f); p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f); p = fmaf(p, w, 1.00167406f); p = fmaf(p, w, 2.83297682f);
    }
    return p * x;
}

// The trilinear fetch of 8 channels [16*g + 8*h, +8) of one working grid, in three pieces so that the loads can be
// issued long before the arithmetic: 8 x 16-byte loads -> 32 x v_dot2_f32_f16 (4 records) -> 4 x v_cvt_pk.
struct GridRaw {
    uint4_t v[4][2];
};

__device__ __forceinline__ void grid_load(const void* grid, const GridTap& t, int g, int h, GridRaw& r) {
    // wave-uniform base + 32-bit per-lane offset: the loads take the SGPR-base form (no 64-bit address arithmetic per record)
    const char* base = reinterpret_cast<const char*>(grid) + g * 64;
    const unsigned hoff = unsigned(h) * 32u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#ifdef FVSRN_ABL_NOGRIDLOAD  // ablation build (tools/ablate.sh): no memory access
        r.v[k][0] = uint4_t{t.off[k], t.w[k], 0x3c003c00u, 0x38003800u};
        r.v[k][1] = uint4_t{t.w[k], t.off[k], 0x3c003c00u, 0x38003800u};
#else
        const uint4_t* p = reinterpret_cast<const uint4_t*>(base + (t.off[k] + hoff));
        r.v[k][0] = p[0];
        r.v[k][1] = p[1];
#endif
    }
}

// FIRST: record 0 starts the sums (v_dot2_f32_f16 with the constant 0 as its addend: hipcc otherwise picks the accumulate-in-place
// form v_dot2c and spends a v_mov per accumulator on the zeros -- 16 of the 57 v_mov of the r01 wave step)
// HAZARD: hipcc's hazard recognizer does not look into inline assembly.  On gfx950 an instruction of another opcode (VALU, store, export)
// that reads the result of a DOT instruction needs three wait states (LLVM GCNHazardRecognizer::checkMAIVALUHazards,
// DotWriteDifferentVALURead); the compiler inserts `s_nop 2` for DOT instructions it emitted itself, not for this one (found r03: fp32
// arithmetic right behind these came out wrong in three of eight channels).  Callers keep three instructions between this and the first
// reader -- the sums below continue with seven more records' v_dot2c first -- and tools/check_dot_hazard.py (a CPU test runs it over
// the built objects) disassembles the build and fails on any reader that is closer.  Code outside the hot path takes SAFE = true.
__device__ __forceinline__ float dot2_from_zero(unsigned a, unsigned w) {
#ifdef FVSRN_DOT2_NO_ASM  // experiment: the compiler's own form everywhere (a v_mov per accumulator, its hazard handling)
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a), __builtin_bit_cast(half2_t, w), 0.f, false);
#else
    float r;
    asm("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(w));
    return r;
#endif
}
template <bool WITH_LO, bool FIRST = false, bool SAFE = false>
__device__ __forceinline__ void grid_reduce_record(const GridRaw& r, const GridTap& t, int k, float acc[8]) {
    const half2_t w = __builtin_bit_cast(half2_t, t.w[k]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // NB: __builtin_bit_cast applied directly to a vector ELEMENT (v[j]) is miscompiled by clang 22 /
        // ROCm 7.2 (only element 0 survives): go through a scalar temporary
        const unsigned u0 = r.v[k][0][j], u1 = r.v[k][1][j];
        if constexpr (FIRST && SAFE) {  // (the compiler's own form: a v_mov for the zero, and its hazard handling)
            acc[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u0), w, 0.f, false);
            acc[4 + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u1), w, 0.f, false);
        } else if constexpr (FIRST) {
            acc[j] = dot2_from_zero(u0, t.w[k]);
            acc[4 + j] = dot2_from_zero(u1, t.w[k]);
        } else {
            acc[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u0), w, acc[j], false);
            acc[4 + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u1), w, acc[4 + j], false);
        }
        if constexpr (WITH_LO) {
            const half2_t wl = __builtin_bit_cast(half2_t, t.wlo[k]);
            acc[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u0), wl, acc[j], false);
            acc[4 + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u1), wl, acc[4 + j], false);
        }
    }
}

__device__ __forceinline__ half8_t grid_pack(const float acc[8]) {
    half8_t out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float2_t v = {acc[2 * i], acc[2 * i + 1]};
        const half2_t hh = __builtin_convertvector(v, half2_t);
        out[2 * i] = hh[0]; out[2 * i + 1] = hh[1];
    }
    return out;
}

template <bool WITH_LO, bool SAFE = false>
__device__ __forceinline__ void grid_fetch8(const void* grid, const GridTap& t, int g, int h, float acc[8]) {
    GridRaw r;
    grid_load(grid, t, g, h, r);
    grid_reduce_record<WITH_LO, true, SAFE>(r, t, 0, acc);
#pragma unroll
    for (int k = 1; k < 4; ++k) grid_reduce_record<WITH_LO>(r, t, k, acc);
}

// 8 channels [16*g + 8*h, +8) of the tile sample described by `t`, as the B fragment of latent K step g
// GRID: 1 = `grid` holds decoded, time-blended values (FLOAT, BYTE_LINEAR); 2 = BYTE_GAUSSIAN
// hLoad: the lane half for the record offset (kTapHalfInOffset for taps from grid_tap_bcast)
template <int GRID>
__device__ __forceinline__ half8_t grid_features(const NetParams& P, const GridTap& t, int g, int h, int hLoad) {
    float acc[8];
    grid_fetch8<GRID == 2, GRID == 2>(P.grid, t, g, hLoad, acc);  // (BYTE_GAUSSIAN: SAFE, its decode spills around the sums)
    if constexpr (GRID == 2) {  // EncodeGridValue<BYTE_GAUSSIAN> :370-383
        const int c0 = 16 * g + 8 * h;
        const bool isTime = c0 < P.gridTimeChannels;  // a 16-channel chunk never straddles time / ensemble channels
        const float* mean = isTime ? P.gridMeanTime + c0 : P.gridMeanEns + (c0 - P.gridTimeChannels);
        const float* sd = isTime ? P.gridStdTime + c0 : P.gridStdEns + (c0 - P.gridTimeChannels);
        // r05: the time fraction is wave-uniform (a kernel argument), and at frac == 0 -- every static grid, every frame on a key frame -- the result
        // va + 0 (vb - va) is va, bit for bit: key frame B is neither fetched nor decoded (half the gathers, 8 of the 16 erfinv per chunk and sample)
        if (P.gridFrac == 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xa = acc[j] * (1.0f / 255.0f);  // cudaReadModeNormalizedFloat
                const float ya = 1.4142135623730950488f * erfinv_dev((2.0f - 1e-4f) * (xa - 0.5f));
                acc[j] = mean[j] + ya * sd[j];
            }
        } else {
            // key frame A is decoded before key frame B is fetched (a scheduling barrier in between): the two sets of sums, records and filter
            // weights are not live at once
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xa = acc[j] * (1.0f / 255.0f);
                acc[j] = mean[j] + 1.4142135623730950488f * erfinv_dev((2.0f - 1e-4f) * (xa - 0.5f)) * sd[j];
            }
            __builtin_amdgcn_sched_barrier(0);
            float accB[8];
            grid_fetch8<true, true>(P.gridB, t, g, hLoad, accB);
            const float f = isTime ? P.gridFrac : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xb = accB[j] * (1.0f / 255.0f);
                const float yb = 1.4142135623730950488f * erfinv_dev((2.0f - 1e-4f) * (xb - 0.5f));
                const float va = acc[j], vb = mean[j] + yb * sd[j];
                acc[j] = va + f * (vb - va);
            }
        }
    }
    return grid_pack(acc);
}

// ---- latent grid through the cell table (r04) --------------------------------------------------------------------------------------
// The samples a wave evaluates in one step belong to the rays of an 8 x 8 pixel tile at (almost) the same depth: a patch of a hundredth of
// the box, against grid cells of 1/15 .. 1/31.  Nearly always the 32 samples of an MFMA column tile lie in ONE cell, sometimes in two.  Inside
// a cell the trilinear fetch is linear in its eight weights, and so is the first layer behind it:
//     W_latent . (sum_c w_c G_c) = sum_c w_c (W_latent . G_c) = T_cell . w,        T_cell = [W_latent G_c]_c   (C x 8, fp16, NetParams::cellTable)
// i.e. ONE MFMA K step whose A operand is the cell's table entry and whose B operand holds the sample's eight weights -- K = 16 takes two
// cells (lane half 0: slots of cell A, lane half 1: cell B; a sample's weights are zero in the slots of the cell it is not in).  That
// replaces, per wave step, 16 x 16-byte gathers per lane, 64 v_dot2_f32_f16, the record address arithmetic and 8 converts by ONE coalesced
// fetch of the table entry (512 bytes per M tile), 8 v_cndmask + 4 lane-half swaps and a handful of scalar instructions, with the same number of MFMAs as the latent K step it stands in for
// (and any number of latent channels costs the same).  Samples in a third cell (corners of the cell lattice; a few percent of the steps have any)
// get further cell pairs, accumulated into the same first-layer accumulators (cells_accumulate: wave-uniform).
// Same texture semantics as grid_tap (clamp-to-edge, through the ghost cells of the table: cell_tap); the B operand is in fp16 like the weights there; the
// table entries carry one fp16 rounding of W.(combination of the corner vectors) where the gather path rounds the interpolated feature.
struct CellTap {
    unsigned w[4];  // packed fp16 monomial pairs {1, x}, {y, xy}, {z, xz}, {yz, xyz} of the sample's cell-centred coordinates: the eight K slots of its cell
    unsigned cell;
};

// r06: MONOMIALS instead of corner weights, GHOST cells instead of clamps.  Inside a cell the trilinear interpolant is sum c_abc x^a y^b z^c over a, b, c in
// {0, 1} in coordinates centred on the cell (x, y, z in [-1/2, 1/2]); the table entry holds the eight coefficient vectors (the corner vectors combined on
// the host side of the table build, grid_cell_table_kernel) in K slot 4 c + 2 b + a.  The table runs over the grid extended by one ghost cell per side whose
// outer nodes repeat the boundary nodes (= the texture unit's clamp-to-edge), cell e = round(p N) in [0, N] per axis spans the texel coordinates [e - 1, e].
// Per sample: 3 x (mul, round, sub) + two fma, a convert and a min for the cell index, one product + two packed products, four converts = 20 vector
// instructions where the corner form took 34 (three clamps, floors and clamps again, three 1 - w, twelve products).  A sample exactly on a cell face may
// land in either cell: the interpolant is continuous.  fp16 error against the corner form: 1.4 x on random data (tools/dev/slab_precision.py).
template <bool SCALED = false>  // SCALED: the arguments are p N already (render_small_kernel derives them from the ray parameter, kernels.hpp)
__device__ __forceinline__ CellTap cell_tap(const NetParams& P, float px, float py, float pz) {
    const float gx = SCALED ? px : px * P.gridXf, gy = SCALED ? py : py * P.gridYf, gz = SCALED ? pz : pz * P.gridZf;
    const float ex = __builtin_rintf(gx), ey = __builtin_rintf(gy), ez = __builtin_rintf(gz);
    const float x = gx - ex, y = gy - ey, z = gz - ez;
    CellTap t;
    // (exact in fp32: fewer than 2^24 cells; positions of rays that miss the box may be anything, also NaN: any valid cell will do -- the convert saturates)
    t.cell = min(unsigned(fmaf(fmaf(ez, P.gridYf + 1.f, ey), P.gridXf + 1.f, ex)), P.cellCount - 1u);
    const float2_t a = {1.f, x}, b = {y, x * y};
    const float2_t zz = {z, z};
    const float2_t c = a * zz, d = b * zz;
    t.w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, half2_t));
    t.w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(b, half2_t));
    t.w[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(c, half2_t));
    t.w[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, half2_t));
    return t;
}

// The corner-weight form of r04 / r05 (weights {w_zy (1 - wx), w_zy wx}, zy = 2 dz + dy, of the cell x0 = min(floor(texel), N - 2) with clamped coordinates; table over
// the (X - 1)(Y - 1)(Z - 1) cells of the grid itself, CellTableParams::corners).  The SHADED cell-table kernels keep it: central differences amplify the fp16
// noise of an evaluation by 1 / 2h, and the monomial form carries 1.4 x the noise of this one on rough grids (the 64-wide finite-difference case of
// tests/test_gpu_parity.py::test_shaded_render_matches_oracle read 1.9e-2 against a bar of 1.2e-2 with monomials).
__device__ __forceinline__ CellTap cell_tap_corners(const NetParams& P, float px, float py, float pz) {
    const float xm = P.gridXf - 1.f, ym = P.gridYf - 1.f, zm = P.gridZf - 1.f;
    const float fx = __builtin_amdgcn_fmed3f(fmaf(px, P.gridXf, -0.5f), 0.f, xm);
    const float fy = __builtin_amdgcn_fmed3f(fmaf(py, P.gridYf, -0.5f), 0.f, ym);
    const float fz = __builtin_amdgcn_fmed3f(fmaf(pz, P.gridZf, -0.5f), 0.f, zm);
    // min(floor, N - 2) as a median with -1 (floor >= 0): v_med3_f32 takes its operands as they are, fminf would canonicalise the bound at every step
    const float x0 = __builtin_amdgcn_fmed3f(floorf(fx), -1.f, xm - 1.f), y0 = __builtin_amdgcn_fmed3f(floorf(fy), -1.f, ym - 1.f),
                z0 = __builtin_amdgcn_fmed3f(floorf(fz), -1.f, zm - 1.f);
    const float wx = fx - x0, wy = fy - y0, wz = fz - z0;
    CellTap t;
    // (exact in fp32: fewer than 2^24 cells; positions of rays that miss the box may be anything, also NaN: any valid cell will do)
    t.cell = min(unsigned(fmaf(fmaf(z0, ym, y0), xm, x0)), P.cellCount - 1u);
    const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
    const float w4[4] = {uz * uy, uz * wy, wz * uy, wz * wy};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float2_t v = {w4[k] * ux, w4[k] * wx};
        t.w[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
    }
    return t;
}

__device__ __forceinline__ unsigned select_bits_by_mask(unsigned long long m, unsigned x) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}

// One step of a tile's cell bookkeeping: the next (up to) two cells among the tile samples whose bit is set in `rem` (the wave mask of the
// samples still to cover, 32 bits: both lane halves hold the same sample), the weights of the samples in them as the B fragment, the
// cells' table entries (one per M tile) as the A fragments.  Returns the samples covered.
template <int MT>
__device__ __forceinline__ unsigned cell_pair(const NetParams& P, const unsigned (&w)[4], unsigned cell, unsigned rem, int h, unsigned laneOff,
                                              half8_t& bfrag, half8_t (&afrag)[MT]) {
    const unsigned cA = __builtin_amdgcn_readlane(cell, rem ? __builtin_ctz(rem) : 0);
    const unsigned long long mA = __builtin_amdgcn_ballot_w64(cell == cA);
    const unsigned rem1 = rem & ~unsigned(mA);
    const unsigned cB = __builtin_amdgcn_readlane(cell, rem1 ? __builtin_ctz(rem1) : 0);
    const unsigned long long mB = rem1 ? __builtin_amdgcn_ballot_w64(cell == cB) : 0ull;
    const unsigned long long sel = (mA & 0xffffffffull) | (mB & 0xffffffff00000000ull);
    uint4_t b;
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = select_bits_by_mask(sel, w[k]);
    bfrag = __builtin_bit_cast(half8_t, b);
#ifdef FVSRN_ABL_CELL0
    const unsigned off = __umul24((h ? cB : cA) & 63u, P.cellStride) + laneOff;
#else
    const unsigned off = __umul24(h ? cB : cA, P.cellStride) + laneOff;  // (both below 2^24, the table below 2^31 bytes: api.cpp)
#endif
    const char* base = static_cast<const char*>(P.cellTable) + off;
#pragma unroll
    for (int m = 0; m < MT; ++m) afrag[m] = *reinterpret_cast<const half8_t*>(base + 512 * m);
    return unsigned(mA) | unsigned(mB);
}

// What the first layer needs of a wave step's latent grid.  The first cell pair is chosen among ALL 64 samples of the step (the rays of an
// 8 x 8 pixel tile: one selection, one table fetch for both column tiles); the samples it does not cover -- a few percent of the steps have
// any -- get further pairs tile by tile (cells_accumulate).
template <int MT>
struct CellPre {
    unsigned wOwn[4];        // the weights of this lane's own sample
    unsigned cellOwn;        // ... and its cell
    unsigned long long rem;  // valid samples the first pair does not cover
    half8_t gf[2];           // B fragments of the first pair
    half8_t ga[MT];          // its A fragments
};

template <int MT, bool MONO = true>
__device__ __forceinline__ void cell_prepare(const NetParams& P, float px, float py, float pz, unsigned long long validMask, int h, unsigned laneOff,
                                             CellPre<MT>& C) {
    const CellTap own = MONO ? cell_tap<false>(P, px, py, pz) : cell_tap_corners(P, px, py, pz);
#pragma unroll
    for (int k = 0; k < 4; ++k) C.wOwn[k] = own.w[k];
    C.cellOwn = own.cell;
    const unsigned cA = __builtin_amdgcn_readlane(own.cell, validMask ? __builtin_ctzll(validMask) : 0);
    const unsigned long long mA = __builtin_amdgcn_ballot_w64(own.cell == cA);
    const unsigned long long rem1 = validMask & ~mA;
    const unsigned cB = __builtin_amdgcn_readlane(own.cell, rem1 ? __builtin_ctzll(rem1) : 0);
    const unsigned long long mB = rem1 ? __builtin_amdgcn_ballot_w64(own.cell == cB) : 0ull;
    C.rem = rem1 & ~mB;
    // Own sample's weights masked for the K slots of cell A and of cell B, then ONE lane-half swap per register: it leaves {A slots of sample c,
    // B slots of sample c} in the two halves of the first register -- tile 0's B fragment (lane half 0 carries cell A's K slots, lane half 1
    // cell B's) -- and tile 1's in the second
    uint4_t b0, b1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto r = lane_half_swap(select_bits_by_mask(mA, own.w[k]), select_bits_by_mask(mB, own.w[k]));
        b0[k] = r[0]; b1[k] = r[1];
    }
    C.gf[0] = __builtin_bit_cast(half8_t, b0);
    C.gf[1] = __builtin_bit_cast(half8_t, b1);
#ifdef FVSRN_ABL_CELL0  // ablation (wrong image, same instruction stream): every table fetch reads cell (index & 63) -- a 64-entry working set that stays in L2 --
                        // to price the table's L2 misses of the 64-wide latent-grid frames (291 MB per frame for 33.5 MB of output)
    const char* base = static_cast<const char*>(P.cellTable) + (__umul24((h ? cB : cA) & 63u, P.cellStride) + laneOff);
#else
    const char* base = static_cast<const char*>(P.cellTable) + (__umul24(h ? cB : cA, P.cellStride) + laneOff);
#endif
#pragma unroll
    for (int m = 0; m < MT; ++m) C.ga[m] = *reinterpret_cast<const half8_t*>(base + 512 * m);
}

// The cell pair of a wave KEPT across the steps of a ray tile (r06; render_small_kernel<.., SGRID = 2>): the samples of a step move by 0.03 cells per step, so
// the pair (A, B) the previous step picked still holds every valid sample on 87 % of the steps of the 16^3 headline frame (tools/dev/slab_sim_cells.py).
// While it does, the step needs neither the two v_readlane chains of the pick nor the table fetch (the A fragments stay in four registers): two compares
// against the kept indices give the masks.  A valid sample in neither cell -> the pick of cell_prepare, which replaces the pair.
// (and its coordinates p N come straight from the ray parameter, FVSRN_CELLS_SCALED_POS: kernels.hpp.  Same-box A/B, profiles/r06/cells_resident_pair_ab_r06.txt:
// 128.8 -> 130.8 G samples/s with the kept pair, 132.5 with both, 32 x 4 + 16^3)
#ifndef FVSRN_CELLS_SCALED_POS
#define FVSRN_CELLS_SCALED_POS 1
#endif
constexpr unsigned kNoCell = 0xffffffffu;
template <int MT>
struct CellResident {
    unsigned cA, cB;  // wave-uniform; cB = kNoCell: no second cell
    int valid;
    half8_t ga[MT];   // lane half 0: A's entry row, lane half 1: B's (A's where there is no B)
};

template <int MT, bool SCALED = false>  // SCALED: the position arguments are p N already (cell_tap)
__device__ __forceinline__ void cell_prepare_resident(const NetParams& P, CellResident<MT>& S, float px, float py, float pz, unsigned long long validMask, int h,
                                                      unsigned laneOff, CellPre<MT>& C) {
    const CellTap own = cell_tap<SCALED>(P, px, py, pz);
#pragma unroll
    for (int k = 0; k < 4; ++k) C.wOwn[k] = own.w[k];
    C.cellOwn = own.cell;
    unsigned long long mA = __builtin_amdgcn_ballot_w64(own.cell == S.cA), mB = __builtin_amdgcn_ballot_w64(own.cell == S.cB);
    unsigned long long rem = validMask & ~(mA | mB);
    if (!S.valid || rem != 0ull) {  // wave-uniform
        const unsigned cA = __builtin_amdgcn_readlane(own.cell, validMask ? __builtin_ctzll(validMask) : 0);
        mA = __builtin_amdgcn_ballot_w64(own.cell == cA);
        const unsigned long long rem1 = validMask & ~mA;
        const unsigned cB = rem1 ? unsigned(__builtin_amdgcn_readlane(own.cell, __builtin_ctzll(rem1))) : kNoCell;
        mB = rem1 ? __builtin_amdgcn_ballot_w64(own.cell == cB) : 0ull;
        rem = rem1 & ~mB;
        S.cA = cA; S.cB = cB; S.valid = 1;
        const char* base = static_cast<const char*>(P.cellTable) + (__umul24(h && cB != kNoCell ? cB : cA, P.cellStride) + laneOff);
#pragma unroll
        for (int m = 0; m < MT; ++m) S.ga[m] = *reinterpret_cast<const half8_t*>(base + 512 * m);
    }
    C.rem = rem;
    uint4_t b0, b1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto r = lane_half_swap(select_bits_by_mask(mA, own.w[k]), select_bits_by_mask(mB, own.w[k]));
        b0[k] = r[0]; b1[k] = r[1];
    }
    C.gf[0] = __builtin_bit_cast(half8_t, b0);
    C.gf[1] = __builtin_bit_cast(half8_t, b1);
#pragma unroll
    for (int m = 0; m < MT; ++m) C.ga[m] = S.ga[m];
}

// The latent K step(s) of tile t into the first layer's accumulators: the prepared pair, then -- wave-uniform, rare -- further pairs until
// every valid sample of the tile has met its cell
template <int MT>
__device__ __forceinline__ void cells_accumulate(const NetParams& P, const CellPre<MT>& C, int t, int h, unsigned laneOff, floatx16* acc) {
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(C.ga[m], C.gf[t], acc[m], 0, 0, 0);
    unsigned rem = unsigned(C.rem >> (32 * t));
    if (rem) {
        auto c = lane_half_swap(C.cellOwn, C.cellOwn);  // the tile's cells and weights in both lane halves
        const unsigned cellT = t ? c[1] : c[0];
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            auto a = lane_half_swap(C.wOwn[k], C.wOwn[k]);
            w[k] = t ? a[1] : a[0];
        }
        do {
            half8_t gf, ga[MT];
            rem &= ~cell_pair<MT>(P, w, cellT, rem, h, laneOff, gf, ga);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[m], gf, acc[m], 0, 0, 0);
        } while (rem);
    }
}

// ---- building blocks of the forward pass -------------------------------------------------------------------
constexpr int mtiles(int CD) { return (16 * CD + 31) / 32; }

// quarter q (0..3) of one M tile: accumulator registers {2q,2q+1} / {8+2q,9+2q} -> half pairs q of the two B fragments
// (K steps 2m and 2m+1 of the next layer), with the activation (ACT_NONE: plain convert)
constexpr int ACT_NONE = -1;
template <int ACT>
__device__ __forceinline__ void act_pack_quarter(const floatx16& d, int q, float a, float b, half8_t& f0, half8_t& f1) {
    const float2_t v0 = {d[2 * q], d[2 * q + 1]};
    const float2_t v1 = {d[8 + 2 * q], d[9 + 2 * q]};
    half2_t h0, h1;
    if constexpr (ACT == ACT_NONE) {
        h0 = __builtin_convertvector(v0, half2_t);
        h1 = __builtin_convertvector(v1, half2_t);
    } else if constexpr (ACT == ACT_RELU01) {
        const half2_t z = {0, 0}, o = {1, 1};
        h0 = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_convertvector(v0, half2_t), z), o);  // = the convert's clamp bit
        h1 = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_convertvector(v1, half2_t), z), o);
    } else if constexpr (ACT == ACT_RELU) {
        const half2_t z = {0, 0};
        h0 = __builtin_elementwise_max(__builtin_convertvector(v0, half2_t), z);
        h1 = __builtin_elementwise_max(__builtin_convertvector(v1, half2_t), z);
    } else {
        h0 = __builtin_convertvector(act_f32x2<ACT>(v0, a, b), half2_t);
        h1 = __builtin_convertvector(act_f32x2<ACT>(v1, a, b), half2_t);
    }
    f0[2 * q] = h0[0]; f0[2 * q + 1] = h0[1];
    f1[2 * q] = h1[0]; f1[2 * q + 1] = h1[1];
}

// Instruction-order helper.  Issues fm(0..NMF-1) (one MFMA each) with the NCH pieces of independent VALU work
// fv(0..NCH-1) spread between them, and pins that order (hipcc would otherwise group the MFMAs).  Measured on MI355X
// (profiles/r01/microbench_issue_model.md): while a wave has an MFMA ready, no OTHER wave of the SIMD issues VALU work,
// but up to ~4 VALU instructions of the SAME wave placed behind an MFMA run in the 32 cycles the matrix pipe needs.
template <int NMF, int NCH, class FM, class FV>
__device__ __forceinline__ void interleave(FM&& fm, FV&& fv) {
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
        fm(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = (i * NCH) / NMF; j < ((i + 1) * NCH) / NMF; ++j) fv(j);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// B operand of the phase MFMA for one tile: K slots [x,x,y,y,z,z,1,1 | dx,dx,dy,dy,dz,dz,0,0]
template <bool HAS_DIR>
__device__ __forceinline__ half8_t phase_operand(const unsigned (&tp)[3], const float (&tdir)[3], int h) {
    if constexpr (HAS_DIR) {
        const float2_t vx = {tdir[0], tdir[0]}, vy = {tdir[1], tdir[1]}, vz = {tdir[2], tdir[2]};
        const unsigned dxh = __builtin_bit_cast(unsigned, __builtin_convertvector(vx, half2_t));
        const unsigned dyh = __builtin_bit_cast(unsigned, __builtin_convertvector(vy, half2_t));
        const unsigned dzh = __builtin_bit_cast(unsigned, __builtin_convertvector(vz, half2_t));
        const uint4_t u = {h ? dxh : tp[0], h ? dyh : tp[1], h ? dzh : tp[2], h ? 0u : 0x3c003c00u};
        return __builtin_bit_cast(half8_t, u);
    } else {
        // lane half 1 supplies K slots 8..15, whose phase-matrix entries are all zero: its B values are
        // multiplied by 0 and only need to be finite, so no select is needed
        const uint4_t u = {tp[0], tp[1], tp[2], 0x3c003c00u /* (1.0h, 1.0h) */};
        return __builtin_bit_cast(half8_t, u);
    }
}

// Input features of the first Linear layer for the two sample tiles of the wave, in fp32 and in accumulator layout:
// register r of f[t][m] = row 32m + (r&3) + 8(r>>2) + 4h = [pass-through x,y | z,time (, direction)] on the first 2 (4)
// registers of M tile 0, (cos, sin) of one Fourier feature on every further register pair (pack.cpp, rowToChannel).
//   (px,py,pz): this lane's sample position, already normalized to the unit box
//   (dx,dy,dz): this lane's view direction (only read when the network uses it)
// The phases are an MFMA (positions as a K=16 B operand, matrix in revolutions, split hi/lo in fp16), replacing the
// half hmul/hfma chain + hcos/hsin of the reference (renderer_volume_tensorcores.cuh:797-806).
// DELTA: (px,py,pz) is the per-step position increment of a ray instead: the pairs become (cos, sin) of the per-step
// phase increment and the pass-through registers the position increment (0 for time / direction) -- see fourier_advance.
// What follows the "phase" MFMA (FMODE, resolved once per kernel launch):
//   FM_COS / FM_FRACT_COS: Fourier features, cos of the phase (after v_fract if it can leave the +-256 revolution
//                          domain of v_cos_f32)
//   FM_FIRST_LAYER:        network without Fourier features: the fragments are the scalar first layer 3|6 -> C
//                          (renderer_volume_tensorcores.cuh:810-823) with the bias in the constant slot, followed by the
//                          hidden activation
enum { FM_COS = 0, FM_FRACT_COS = 1, FM_FIRST_LAYER = 2 };
template <int FMODE>
__device__ __forceinline__ void phase_cos(floatx16& d, int npass) {
    static_assert(FMODE != FM_FIRST_LAYER, "no cos stage without Fourier features");
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (r >= npass) {  // (cos_j, sin_j) on registers (2i, 2i + 1): pack.cpp, rowToChannel
            float x = d[r];
            if constexpr (FMODE == FM_FRACT_COS) x = __builtin_amdgcn_fractf(x);
            d[r] = (r & 1) ? __builtin_amdgcn_sinf(x) : __builtin_amdgcn_cosf(x);
        }
}

// B operands of the phase MFMA for the two tiles from this lane's own position / direction
template <bool HAS_DIR, bool WITH_DIR_VALUES>
__device__ __forceinline__ void phase_operands(float px, float py, float pz, float dx, float dy, float dz, int h, half8_t (&b0)[2]) {
    // Positions of the two sample tiles: fp16 images (both halves of a register = the hi/lo slot pair of the phase
    // matrix), then exchanged between the lane halves.
    unsigned tph[2][3];
    const float pp[3] = {px, py, pz};
    unsigned pu[3], pv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float2_t v = {pp[i], pp[i]};
        pu[i] = pv[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        auto r = lane_half_swap(pu[i], pv[i]);
        tph[0][i] = r[0];
        tph[1][i] = r[1];
    }

    float td[2][3] = {{0, 0, 0}, {0, 0, 0}};
    if constexpr (HAS_DIR && WITH_DIR_VALUES) {
        tile_bcast(dx, td[0][0], td[1][0]);
        tile_bcast(dy, td[0][1], td[1][1]);
        tile_bcast(dz, td[0][2], td[1][2]);
    }
    b0[0] = phase_operand<HAS_DIR>(tph[0], td[0], h);
    b0[1] = phase_operand<HAS_DIR>(tph[1], td[1], h);
}

// The same operands for the fp16 rounding residuals of the position (x - half(x): a second fp16 value, so that x = hi + lo carries
// ~22 bits): constant and direction slots zero.  A second phase MFMA on these, accumulating into the first one's result, gives the
// phases of the fp32 position (fourier_features, hilo).
template <bool HAS_DIR>
__device__ __forceinline__ void phase_operands_residual(float px, float py, float pz, int h, half8_t (&b1)[2]) {
    const float pp[3] = {px, py, pz};
    unsigned pu[3], pv[3], tph[2][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float r = pp[i] - float(_Float16(pp[i]));
        const float2_t v = {r, r};
        pu[i] = pv[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        auto r = lane_half_swap(pu[i], pv[i]);
        tph[0][i] = r[0];
        tph[1][i] = r[1];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const bool zero = HAS_DIR && h != 0;  // (without a direction the matrix entries of K slots 8..15 are zero: any finite value)
        const uint4_t u = {zero ? 0u : tph[t][0], zero ? 0u : tph[t][1], zero ? 0u : tph[t][2], 0u};
        b1[t] = __builtin_bit_cast(half8_t, u);
    }
}

// wp != nullptr: the phase fragments are given (resident in registers) instead of read from the LDS image
// hilo (wave-uniform; r04): the phases of the fp32 position instead of its fp16 rounding (a second MFMA on the rounding residuals).  The
// rotating renderers re-derive with it: a feature set that is then ADVANCED for 64 steps should not carry the rounding of one position
// (up to 0.025 rad on a 2^4 octave) and of the step vector coherently along the ray -- the reference rounds every sample's position anew,
// which averages out in the integral (the CPU model of the parity tests restates both forms; image error of the rotation 3.1e-3 -> 1.7e-3
// on the scene that exposed it).  With FVSRN_OPT_FOURIER_RESYNC = 1 (the reference's per-sample arithmetic) it stays off.
template <int CD, bool HAS_DIR, int FMODE, bool DELTA = false>
__device__ __forceinline__ void fourier_features(const NetParams& P, const char* lds, float px, float py, float pz, float dx,
                                                 float dy, float dz, floatx16 (&f)[2][mtiles(CD)], const half8_t* wp = nullptr, bool hilo = false) {
    constexpr int MT = mtiles(CD);
    constexpr int NPASS = HAS_DIR ? 4 : 2;
    const int lane = lane_id();
    const int h = lane >> 5;
    half8_t aph[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) aph[m] = wp ? wp[m] : lds_frag(lds, P.offPhase + m * kFragBytes, lane);
    half8_t b0[2];
    phase_operands<HAS_DIR, !DELTA>(px, py, pz, dx, dy, dz, h, b0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const floatx16 z = {0};
            f[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aph[m], b0[t], z, 0, 0, 0);
        }
    if (hilo) {
        half8_t b1[2];
        phase_operands_residual<HAS_DIR>(px, py, pz, h, b1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) f[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aph[m], b1[t], f[t][m], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) phase_cos<FMODE>(f[t][m], m == 0 ? NPASS : 0);
    if constexpr (DELTA) {
        // the constant K slot also carries the time pass-through (row 5: lane half 1, register 1; with direction row 3:
        // lane half 0, register 3): its increment is 0
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (HAS_DIR) f[t][0][3] = h ? f[t][0][3] : 0.f;
            else f[t][0][1] = h ? 0.f : f[t][0][1];
        }
    }
}

// the same features as fp16 B fragments of the first layer, tile by tile (few live registers)
// one tile: b0 = its phase operand (phase_operands)
template <int CD, int ACT, bool HAS_DIR, int FMODE>
__device__ __forceinline__ void fourier_fragments_tile(const NetParams& P, const char* lds, const half8_t& b0, half8_t (&xbt)[2 * mtiles(CD)]) {
    constexpr int MT = mtiles(CD);
    constexpr int NPASS = HAS_DIR ? 4 : 2;
    const int lane = lane_id();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        floatx16 d = {0};
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(lds, P.offPhase + m * kFragBytes, lane), b0, d, 0, 0, 0);
        if constexpr (FMODE == FM_FIRST_LAYER) {
            constexpr int A = (ACT == ACT_RELU01) ? ACT_RELU : (ACT == ACT_SNAKEALT0 ? ACT_SNAKEALT : ACT);  // no re-scaled image without Fourier features
#pragma unroll
            for (int q = 0; q < 4; ++q) act_pack_quarter<A>(d, q, P.actA, P.actB, xbt[2 * m], xbt[2 * m + 1]);
        } else {
            phase_cos<FMODE>(d, m == 0 ? NPASS : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(d, q, 0.f, 0.f, xbt[2 * m], xbt[2 * m + 1]);
        }
    }
}

template <int CD, int ACT, bool HAS_DIR, int FMODE>
__device__ __forceinline__ void fourier_fragments(const NetParams& P, const char* lds, float px, float py, float pz, float dx,
                                                  float dy, float dz, half8_t (&xb)[2][2 * mtiles(CD)]) {
    const int h = lane_id() >> 5;
    half8_t b0[2];
    phase_operands<HAS_DIR, true>(px, py, pz, dx, dy, dz, h, b0);
    fourier_fragments_tile<CD, ACT, HAS_DIR, FMODE>(P, lds, b0[0], xb[0]);
    fourier_fragments_tile<CD, ACT, HAS_DIR, FMODE>(P, lds, b0[1], xb[1]);
}

// fourier_fragments for CD = 2 (one M tile) with BOTH phase MFMAs ahead of the cosines: the second one runs while the first tile's
// transcendentals issue -- one exposed MFMA latency per batch instead of two (evaluate_small_kernel, r05; 16 more live registers in the
// Fourier stage, which is not where that kernel's register peak is)
template <bool HAS_DIR, int FMODE>
__device__ __forceinline__ void fourier_fragments_ahead(const NetParams& P, const char* lds, float px, float py, float pz, float dx,
                                                        float dy, float dz, half8_t (&xb)[2][2]) {
    static_assert(FMODE != FM_FIRST_LAYER, "Fourier networks only");
    constexpr int NPASS = HAS_DIR ? 4 : 2;
    const int lane = lane_id();
    half8_t b0[2];
    phase_operands<HAS_DIR, true>(px, py, pz, dx, dy, dz, lane >> 5, b0);
    const half8_t a = lds_frag(lds, P.offPhase, lane);
    floatx16 d[2];
    {
        const floatx16 z = {0};
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0[0], z, 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0[1], z, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        phase_cos<FMODE>(d[t], NPASS);
#pragma unroll
        for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(d[t], q, 0.f, 0.f, xb[t][0], xb[t][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CD>
__device__ __forceinline__ void feature_fragments(const floatx16 (&f)[2][mtiles(CD)], half8_t (&xb)[2][2 * mtiles(CD)]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < mtiles(CD); ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(f[t][m], q, 0.f, 0.f, xb[t][2 * m], xb[t][2 * m + 1]);
}

// Features of the next sample along the rays: every ray advances by the same step, so each phase grows by a constant
// and (cos, sin) rotate by a constant angle:  c' = c cd - s sd,  s' = s cd + c sd  -- two packed-fp32 instructions per
// feature instead of two v_cos_f32 (8.3 cycles each, profiles/r01/microbench_issue_model.md) plus the phase MFMA, the
// position converts and lane exchanges.  Rounding errors grow linearly with the number of rotations (~6e-8 each); the
// renderer re-derives the features from the positions every kFourierResync steps.
// piece c of the rotation, c in [0, 16 * mtiles(CD)): one register pair (<= 2 VALU instructions)
template <int CD, bool HAS_DIR>
__device__ __forceinline__ void fourier_advance_piece(floatx16 (&f)[2][mtiles(CD)], const floatx16 (&d)[2][mtiles(CD)], int c) {
    constexpr int MT = mtiles(CD);
    constexpr int NPASS = HAS_DIR ? 4 : 2;
    const int t = c / (8 * MT), m = (c / 8) % MT, k = c % 8;
#if !FVSRN_PK_F32
    // scalar fp32 (experiment, see act_f32x2): c' = c cd - s sd, s' = s cd + c sd as 2 v_mul + 2 v_fma; pinned with asm so
    // that hipcc's SLP vectorizer does not fuse them back into v_pk_*_f32
    float cs_ = f[t][m][2 * k], sn = f[t][m][2 * k + 1];
    const float cd = d[t][m][2 * k], sd = d[t][m][2 * k + 1];
    if (m == 0 && 2 * k < NPASS) {
        asm("v_add_f32 %0, %1, %2" : "=v"(cs_) : "v"(cs_), "v"(cd));
        asm("v_add_f32 %0, %1, %2" : "=v"(sn) : "v"(sn), "v"(sd));
    } else {
        float t0, t1;
        asm("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(cs_), "v"(cd));
        asm("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(sn), "v"(cd));
        asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(t0) : "v"(sn), "v"(sd), "v"(t0));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(cs_), "v"(sd), "v"(t1));
        cs_ = t0; sn = t1;
    }
    f[t][m][2 * k] = cs_;
    f[t][m][2 * k + 1] = sn;
#else
    float2_t cs = {f[t][m][2 * k], f[t][m][2 * k + 1]};
    const float2_t dd = {d[t][m][2 * k], d[t][m][2 * k + 1]};
    if (m == 0 && 2 * k < NPASS) {
        cs += dd;
    } else {
        float2_t tmp;
        // tmp = (c cd, s cd);  cs = (s * -sd + tmp.x, c * sd + tmp.y)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(tmp) : "v"(cs), "v"(dd));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(cs) : "v"(cs), "v"(dd), "v"(tmp));
    }
    f[t][m][2 * k] = cs[0];
    f[t][m][2 * k + 1] = cs[1];
#endif
}

// The network behind the input features, for the 64 samples of this wave: returns the raw last-layer outputs (before
// the output parametrization) of this lane's sample.  EXEC must be all ones.
//   * the fp32 accumulator tile of layer l is converted in registers (v_cvt_pk_f16_f32) into the B operand of layer
//     l+1 -- weights were permuted on the host for that (pack.cpp); the reference round-trips activations through shared
//     memory every layer (renderer_volume_tensorcores.cuh:1019-1023)
//   * biases enter as the MFMA C operand (fp32), accumulation is fp32 (reference: half, :965)
//   * the last (C -> 1|4) layer is a v_mfma_f32_16x16x32_f16 per K step and tile on the same B fragments (its weight
//     fragment routes every output to the lane that owns the sample, pack.cpp), replacing the per-lane hfma loop :1138-1143
//
// Schedule (Fourier-only networks up to 64 wide).  The two sample tiles t0,t1 of the wave run half a layer apart:
//     A_l: MFMAs of layer l for t0   ||  activation+convert of layer l-1 for t1, bias(l)   -> accumulators of t1
//     B_l: MFMAs of layer l for t1   ||  activation+convert of layer l   for t0, bias(l+1) -> accumulators of t0,
//                                        weight fragments of layer l+1 -> registers (each right after its last use)
// so inside ONE wave the matrix pipe always has the other tile's chain to work on while the VALU converts, and every
// LDS read (weights, biases) is issued at least half a layer before its use.  Biases are read straight into the
// accumulator registers (the MFMA C operand), weight fragments are read once per layer and shared by both tiles.
//   pre():    produces xb (the first layer's B fragments); called after the first layer's LDS reads are issued
//   fill(j):  NFILL pieces of independent VALU work for the MFMAs that have none of their own (first layer of tile 0,
//             last layer of tile 1)
// Latent-grid work a caller has done ahead of the layers (srn_forward): the taps of both sample tiles and -- valid != 0 -- the
// fetched first 16-channel chunk of both tiles as B fragments.  Measured r02: fetching inside the layer loop leaves the L1 / L2
// latency of every tile's 8 gathers exposed once per tile and wave step (the real 32x4 + grid kernel ran 32 % below its own
// instruction skeleton, tools/microbench/r02_issue.hip part 6); srn_forward issues them in front of the tile's Fourier work.
struct GridPre {
    GridTap gt[2];
    half8_t gf[2];
    int valid;
};

template <int CD, int ACT, int GRID, bool HAS_DIR, int NFILL, class Pre, class Fill>
__device__ __forceinline__ float4_t srn_layers_pipelined(const NetParams& P, const char* lds, half8_t (&xb)[2][2 * mtiles(CD)],
                                                         float px, float py, float pz, Pre&& pre, Fill&& fill, const GridPre* gpre = nullptr,
                                                         const CellPre<mtiles(CD)>* cells = nullptr) {
    // GRID = 3: the decoded latent grid through the cell table (`cells`: cell_prepare / cells_accumulate) instead of gathers
    constexpr int MT = mtiles(CD), KS = CD, NM = MT * KS, NV = 4 * MT;
    const int lane = lane_id();
    const int h = lane >> 5;
    const float actA = P.actA, actB = P.actB;
    const char* ldsA = lds + 16 * lane;  // A fragments are lane-linear
    const char* ldsB = lds + 16 * h;     // bias blocks: this lane half's rows 8g + 4h .. +3 at 32g bytes
#ifdef FVSRN_ABL_NOLDS  // ablation build (tools/ablate.sh): the layer loop without its LDS reads (registers left undefined)
    auto frag = [&](int) { half8_t v; asm volatile("" : "=v"(v)); return v; };
#else
    auto frag = [&](int byteOff) { return *reinterpret_cast<const half8_t*>(ldsA + byteOff); };
#endif
    auto bias = [&](int byteOff) {
        floatx16 c;
        const float4_t* p = reinterpret_cast<const float4_t*>(ldsB + byteOff);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#ifdef FVSRN_ABL_NOLDS
            float4_t v; asm volatile("" : "=v"(v));
#else
            const float4_t v = p[2 * g];
#endif
            c[4 * g + 0] = v[0]; c[4 * g + 1] = v[1]; c[4 * g + 2] = v[2]; c[4 * g + 3] = v[3];
        }
        return c;
    };

    // weight fragments of the first layer (its Fourier part has the layout of a hidden layer), its bias, latent taps
    half8_t a[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) a[i] = frag(P.offLayer0 + i * kFragBytes);
    floatx16 acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = bias(P.offBias + m * 128);
    const int offGridW = P.offLayer0 + NM * kFragBytes;  // [g][m] fragments of the latent K steps
    // (the latent-step fragments are only prefetched where registers allow: at 64 wide it would spill)
    constexpr bool PREFETCH_AG = MT == 1 && GRID != 3;
    // first latent chunk fetched ahead and reduced behind the first layer's MFMAs: only where its 32 registers fit
    constexpr bool GRID_AHEAD = GRID == 1 && MT == 1;
    half8_t ag[MT];
    GridTap gt[2];
    GridRaw raw;
    if constexpr (GRID != 0 && GRID != 3) {
        if constexpr (PREFETCH_AG) {
#pragma unroll
            for (int m = 0; m < MT; ++m) ag[m] = frag(offGridW + m * kFragBytes);
        }
        if (gpre) { gt[0] = gpre->gt[0]; gt[1] = gpre->gt[1]; }
        else grid_tap_bcast<GRID == 2>(grid_tap(P, px, py, pz), gt[0], gt[1]);
        if constexpr (GRID_AHEAD) grid_load(P.grid, gt[0], 0, kTapHalfInOffset, raw);
    }
    const bool havePre = GRID == 1 && gpre && gpre->valid;
    __builtin_amdgcn_sched_barrier(0);
    FVSRN_MARK(P, 1);  // loop head, LDS reads of the first layer issued
    pre();
    __builtin_amdgcn_sched_barrier(0);
    FVSRN_MARK(P, 2);  // pre(): B fragments of the first layer (+ half of the rotation)
    if constexpr (FVSRN_LDS_PRIO && CD == 2) __builtin_amdgcn_s_setprio(0);  // the MFMA chain yields to the vector phases of the SIMD's other waves

    const int NL = P.numLayers;
    constexpr int kBiasLayer = 32 * MT * 4;
    constexpr int NF0 = NFILL / 2;
    // ---- first layer, tile 0 -------------------------------------------------------------------------------------
    half8_t gf;
    if constexpr (GRID_AHEAD) {
        float gacc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gacc[j] = 0.f;
        interleave<NM, 5>([&](int i) { acc[0][i / KS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], xb[0][i % KS], acc[0][i / KS], 0, 0, 0); },
                          [&](int j) {
                              if (j < 4) grid_reduce_record<false>(raw, gt[0], j, gacc);
                              else gf = grid_pack(gacc);
                          });
        grid_load(P.grid, gt[1], 0, kTapHalfInOffset, raw);  // tile 1's chunk: in flight behind tile 1's first-layer MFMAs
#pragma unroll
        for (int j = 0; j < NF0; ++j) fill(j);
    } else {
        interleave<NM, NF0>([&](int i) { acc[0][i / KS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], xb[0][i % KS], acc[0][i / KS], 0, 0, 0); },
                            [&](int j) { fill(j); });
    }
    if constexpr (GRID == 3) {
        cells_accumulate<MT>(P, *cells, 0, h, unsigned(lane & 31) * 16u, acc[0]);
    } else if constexpr (GRID != 0) {
        for (int g = 0; g < P.gridK; ++g) {
            if (havePre && g == 0) gf = gpre->gf[0];
            else if (!GRID_AHEAD || g > 0) gf = grid_features<GRID>(P, gt[0], g, h, kTapHalfInOffset);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const half8_t w = (PREFETCH_AG && g == 0) ? ag[m] : frag(offGridW + (g * MT + m) * kFragBytes);
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, gf, acc[0][m], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // One half-layer slot: MFMAs of tile T over the fragments in a[] || activation+convert of the other tile, whose
    // accumulators then take the bias block at `biasOff`; RELOAD: a[i] <- fragment i at `nextW` right after its use.
    // The bias block of the converted tile lands in its accumulator registers as soon as they are free: 16-byte block g
    // (registers 4g..4g+3) after the quarters {0,1} (g = 0, 2) resp. {2,3} (g = 1, 3) are converted.  The conversion items
    // are front-loaded (kSlotItems per MFMA) so that the slot's remaining MFMAs cover the LDS latency of those reads: the
    // next slot starts with an MFMA that takes them as its C operand.
#ifndef FVSRN_SLOT_ITEMS
#define FVSRN_SLOT_ITEMS 2
#endif
    constexpr int kSlotItems = FVSRN_SLOT_ITEMS;
    auto slot = [&](auto tileTag, auto reloadTag, int biasOff, int nextW) {
        constexpr int T = decltype(tileTag)::value, O = 1 - T;
        constexpr bool RELOAD = decltype(reloadTag)::value;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            acc[T][i / KS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], xb[T][i % KS], acc[T][i / KS], 0, 0, 0);
            if constexpr (RELOAD) a[i] = frag(nextW + i * kFragBytes);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = i * kSlotItems; j < (i + 1) * kSlotItems && j < NV; ++j) {
                const int m = j / 4, q = j % 4;
                act_pack_quarter<ACT>(acc[O][m], q, actA, actB, xb[O][2 * m], xb[O][2 * m + 1]);
                if (q & 1) {
                    const float4_t* p = reinterpret_cast<const float4_t*>(ldsB + biasOff + m * 128);
#pragma unroll
                    for (int g = q >> 1; g < 4; g += 2) {
#ifdef FVSRN_ABL_NOLDS
                        float4_t v; asm volatile("" : "=v"(v)); (void)p;
#else
                        const float4_t v = p[2 * g];
#endif
                        acc[O][m][4 * g + 0] = v[0]; acc[O][m][4 * g + 1] = v[1]; acc[O][m][4 * g + 2] = v[2]; acc[O][m][4 * g + 3] = v[3];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using Yes = std::true_type;
    using No = std::false_type;

    // ---- first layer, tile 1 || convert tile 0 -----------------------------------------------------------------------
    slot(T1{}, Yes{}, P.offBias + kBiasLayer, NL > 1 ? P.offHidden : P.offLast);
    if constexpr (GRID == 3) {
        cells_accumulate<MT>(P, *cells, 1, h, unsigned(lane & 31) * 16u, acc[1]);
        __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (GRID != 0) {
        for (int g = 0; g < P.gridK; ++g) {
            if (GRID_AHEAD && g == 0) {
                float gacc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) gacc[j] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) grid_reduce_record<false>(raw, gt[1], k, gacc);
                gf = grid_pack(gacc);
            } else if (havePre && g == 0) {
                gf = gpre->gf[1];
            } else {
                gf = grid_features<GRID>(P, gt[1], g, h, kTapHalfInOffset);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const half8_t w = (PREFETCH_AG && g == 0) ? ag[m] : frag(offGridW + (g * MT + m) * kFragBytes);
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, gf, acc[1][m], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    FVSRN_MARK(P, 3);  // first layer
    // ---- layers 1 .. NL-1 -----------------------------------------------------------------------------------------------
#ifdef FVSRN_ABL_NOHIDDEN
    for (int l = 1; l < NL; l += 1000) {
#else
    for (int l = 1; l < NL; ++l) {
#endif
        slot(T0{}, No{}, P.offBias + l * kBiasLayer, 0);
        slot(T1{}, Yes{}, P.offBias + (l + 1) * kBiasLayer, l + 1 < NL ? P.offHidden + l * NM * kFragBytes : P.offLast);
    }

    FVSRN_MARK(P, 4);  // hidden layers
    // ---- last layer: 16x16x32 MFMAs (pack.cpp) on the same B fragments; its weight fragments are a[0..KS-1]; output r of
    // this lane's tile-(lane>>5) sample lands in register r ----------------------------------------------------------------
#ifdef FVSRN_ABL_NOLDS
    float4_t biasLast; asm volatile("" : "=v"(biasLast));
#else
    const float4_t biasLast = *reinterpret_cast<const float4_t*>(lds + P.offBias + NL * kBiasLayer);
#endif
    float4_t o0 = biasLast, o1 = biasLast;
    interleave<KS, NV>([&](int s) { o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], xb[0][s], o0, 0, 0, 0); },
                       [&](int j) {
                           const int m = j / 4, q = j % 4;
                           act_pack_quarter<ACT>(acc[1][m], q, actA, actB, xb[1][2 * m], xb[1][2 * m + 1]);
                       });
    interleave<KS, NFILL - NF0>([&](int s) { o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], xb[1][s], o1, 0, 0, 0); },
                                [&](int j) { fill(NF0 + j); });
    if constexpr (FVSRN_LDS_PRIO && CD == 2) __builtin_amdgcn_s_setprio(3);
    float4_t out = {0, 0, 0, 0};
    out[0] = h ? o1[0] : o0[0];
    if (P.outputMode >= FVSRN_OUT_RGBO) {  // wave-uniform: only colour / gradient networks have outputs 1..3
        out[1] = h ? o1[1] : o0[1];
        out[2] = h ? o1[2] : o0[2];
        out[3] = h ? o1[3] : o0[3];
    }
    return out;
}

// Weight-fragment-major order: each fragment is read from LDS right before its two MFMAs (one per tile).  Used for wide
// networks (C >= 96: a layer's fragments do not fit into registers next to the accumulators) and for latent-grid
// networks (measured r01: there the extra registers of the pipelined order cost more occupancy than the schedule gains).
template <int CD, int ACT, int GRID, bool HAS_DIR, int NFILL, class Pre, class Fill>
__device__ __forceinline__ float4_t srn_layers_kmajor(const NetParams& P, const char* lds, half8_t (&xb)[2][2 * mtiles(CD)],
                                                      float px, float py, float pz, Pre&& pre, Fill&& fill, const GridPre* gpre = nullptr,
                                                      const CellPre<mtiles(CD)>* cells = nullptr) {
    constexpr int MT = mtiles(CD), KS = CD;
    const int lane = lane_id();
    const int h = lane >> 5;
    const float actA = P.actA, actB = P.actB;
    GridTap gt[2];
    if constexpr (GRID != 0 && GRID != 3) {
        if (gpre) { gt[0] = gpre->gt[0]; gt[1] = gpre->gt[1]; }
        else grid_tap_bcast<GRID == 2>(grid_tap(P, px, py, pz), gt[0], gt[1]);
    }
    const bool havePre = GRID == 1 && gpre && gpre->valid;
    pre();
#pragma unroll
    for (int j = 0; j < NFILL; ++j) fill(j);
    if constexpr (FVSRN_LDS_PRIO && CD == 2) __builtin_amdgcn_s_setprio(0);

    // ---- C -> C layers -----------------------------------------------------------------------------------
    const int NL = P.numLayers;
    for (int l = 0; l < NL; ++l) {
        const int wOff = l == 0 ? P.offLayer0 : P.offHidden + (l - 1) * MT * KS * kFragBytes;
        const int bOff = P.offBias + l * 32 * MT * 4;
        floatx16 acc[2][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const floatx16 bias = lds_bias(lds, bOff + m * 128, h);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const half8_t a = lds_frag(lds, wOff + (m * KS + s) * kFragBytes, lane);
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[0][s], s == 0 ? bias : acc[0][m], 0, 0, 0);
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[1][s], s == 0 ? bias : acc[1][m], 0, 0, 0);
            }
        }
        if constexpr (GRID == 3) {
            if (l == 0) {
                cells_accumulate<MT>(P, *cells, 0, h, unsigned(lane & 31) * 16u, acc[0]);
                cells_accumulate<MT>(P, *cells, 1, h, unsigned(lane & 31) * 16u, acc[1]);
            }
        } else if constexpr (GRID != 0) {
            if (l == 0) {
                for (int g = 0; g < P.gridK; ++g) {
                    const half8_t g0 = (havePre && g == 0) ? gpre->gf[0] : grid_features<GRID>(P, gt[0], g, h, kTapHalfInOffset);
                    const half8_t g1 = (havePre && g == 0) ? gpre->gf[1] : grid_features<GRID>(P, gt[1], g, h, kTapHalfInOffset);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const half8_t a = lds_frag(lds, wOff + (MT * KS + g * MT + m) * kFragBytes, lane);
                        acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, g0, acc[0][m], 0, 0, 0);
                        acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, g1, acc[1][m], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) act_pack<ACT>(acc[t][m], actA, actB, xb[t][2 * m], xb[t][2 * m + 1]);
    }

    // ---- last layer: 16x16x32 MFMAs (pack.cpp); output r of this lane's tile-(lane>>5) sample lands in register r ------
    const float4_t biasLast = *reinterpret_cast<const float4_t*>(lds + P.offBias + NL * 32 * MT * 4);
    float4_t o0 = biasLast, o1 = biasLast;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const half8_t a = lds_frag(lds, P.offLast + s * kFragBytes, lane);
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[0][s], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[1][s], o1, 0, 0, 0);
    }
    if constexpr (FVSRN_LDS_PRIO && CD == 2) __builtin_amdgcn_s_setprio(3);
    float4_t out = {0, 0, 0, 0};
    out[0] = h ? o1[0] : o0[0];
    if (P.outputMode >= FVSRN_OUT_RGBO) {  // wave-uniform: only colour / gradient networks have outputs 1..3
        out[1] = h ? o1[1] : o0[1];
        out[2] = h ? o1[2] : o0[2];
        out[3] = h ? o1[3] : o0[3];
    }
    return out;
}

// MAYBE_NO_LAYERS: the network may consist of first + last layer only (possible without Fourier features)
// SCHED = 1: the fragment-major order for every width (render_kernel<3|4, *, 1|2, *>: no register spills, kernels.hpp render_layer_schedule)
template <int CD, int ACT, int GRID, bool HAS_DIR, int NFILL, bool MAYBE_NO_LAYERS, int SCHED = 0, class Pre, class Fill>
__device__ __forceinline__ float4_t srn_layers(const NetParams& P, const char* lds, half8_t (&xb)[2][2 * mtiles(CD)], float px,
                                               float py, float pz, Pre&& pre, Fill&& fill, const GridPre* gpre = nullptr,
                                               const CellPre<mtiles(CD)>* cells = nullptr) {
#ifndef FVSRN_NO_PIPELINE
    // measured (r01, 1024^2 x 512): the pipelined order wins for Fourier-only networks (32x4: 106.7 -> 110.3 Gsamples/s) and
    // for 64-wide latent-grid networks (23.1 -> 23.9); for 32-wide ones with a grid its extra registers cost more
    // occupancy than the schedule gains (64.2 -> 62.2)
    if constexpr (SCHED == 0 && CD >= 2 && CD <= 4 && (GRID == 0 || CD >= 3)) {  // (16 channels: one K step, the fragment-major order)
        if constexpr (MAYBE_NO_LAYERS) {
            if (P.numLayers == 0)  // wave-uniform
                return srn_layers_kmajor<CD, ACT, GRID, HAS_DIR, NFILL>(P, lds, xb, px, py, pz, pre, fill, gpre, cells);
        }
        return srn_layers_pipelined<CD, ACT, GRID, HAS_DIR, NFILL>(P, lds, xb, px, py, pz, pre, fill, gpre, cells);
    } else
#endif
        return srn_layers_kmajor<CD, ACT, GRID, HAS_DIR, NFILL>(P, lds, xb, px, py, pz, pre, fill, gpre, cells);
}

// positions -> raw network outputs (evaluate_points, and render steps of networks without the rotation shortcut)
#ifndef FVSRN_GRID_PRE
#define FVSRN_GRID_PRE 1
#endif
// keep (GRID = 3, render_cells_kernel): the wave's cell pair kept from step to step (cell_prepare_resident); null: a pick and a fetch at every call
template <int CD, int ACT, int GRID, bool HAS_DIR, int FMODE, int SCHED = 0>
__device__ __forceinline__ float4_t srn_forward(const NetParams& P, const char* lds, float px, float py, float pz,
                                                float dx, float dy, float dz, unsigned long long validMask = ~0ull, CellResident<mtiles(CD)>* keep = nullptr) {
    half8_t xb[2][2 * mtiles(CD)];
    if constexpr (GRID == 3 || GRID == 4) {  // (4: the cell table in its corner-weight form -- the shaded kernels; the layer code sees 3 either way)
        // latent grid through the cell table: weights, cells and the first cell pair's fragments ahead of the Fourier work (validMask: the
        // samples that count -- the others need no cell of their own)
        const int lane = lane_id();
        CellPre<mtiles(CD)> C;
        if (GRID == 3 && keep) cell_prepare_resident<mtiles(CD), false>(P, *keep, px, py, pz, validMask, lane >> 5, unsigned(lane & 31) * 16u, C);
        else cell_prepare<mtiles(CD), GRID == 3>(P, px, py, pz, validMask, lane >> 5, unsigned(lane & 31) * 16u, C);
        __builtin_amdgcn_sched_barrier(0);
        return srn_layers<CD, ACT, 3, HAS_DIR, 0, false, SCHED>(
            P, lds, xb, px, py, pz, [&]() { fourier_fragments<CD, ACT, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz, xb); }, [](int) {}, nullptr, &C);
    } else if constexpr (GRID == 1 && FVSRN_GRID_PRE) {
        // Latent grid with decoded working values: the 8 gathers of a tile's first 16-channel chunk are issued in FRONT of that
        // tile's Fourier work (phase MFMA, v_cos, converts), which covers their L1 / L2 latency; one 32-register buffer serves
        // both tiles in turn (see GridPre).
        const int h = lane_id() >> 5;
        GridPre G;
        grid_tap_bcast<false>(grid_tap(P, px, py, pz), G.gt[0], G.gt[1]);
        G.valid = 1;
        GridRaw raw;
        grid_load(P.grid, G.gt[0], 0, kTapHalfInOffset, raw);
        half8_t b0[2];
        phase_operands<HAS_DIR, true>(px, py, pz, dx, dy, dz, h, b0);
        __builtin_amdgcn_sched_barrier(0);
        fourier_fragments_tile<CD, ACT, HAS_DIR, FMODE>(P, lds, b0[0], xb[0]);
        __builtin_amdgcn_sched_barrier(0);
        {
            float acc[8];
            grid_reduce_record<false, true>(raw, G.gt[0], 0, acc);
#pragma unroll
            for (int k = 1; k < 4; ++k) grid_reduce_record<false>(raw, G.gt[0], k, acc);
            G.gf[0] = grid_pack(acc);
        }
        grid_load(P.grid, G.gt[1], 0, kTapHalfInOffset, raw);
        __builtin_amdgcn_sched_barrier(0);
        fourier_fragments_tile<CD, ACT, HAS_DIR, FMODE>(P, lds, b0[1], xb[1]);
        __builtin_amdgcn_sched_barrier(0);
        {
            float acc[8];
            grid_reduce_record<false, true>(raw, G.gt[1], 0, acc);
#pragma unroll
            for (int k = 1; k < 4; ++k) grid_reduce_record<false>(raw, G.gt[1], k, acc);
            G.gf[1] = grid_pack(acc);
        }
        return srn_layers<CD, ACT, GRID, HAS_DIR, 0, false, SCHED>(P, lds, xb, px, py, pz, []() {}, [](int) {}, &G);
    } else {
        return srn_layers<CD, ACT, GRID, HAS_DIR, 0, FMODE == FM_FIRST_LAYER, SCHED>(
            P, lds, xb, px, py, pz, [&]() { fourier_fragments<CD, ACT, HAS_DIR, FMODE>(P, lds, px, py, pz, dx, dy, dz, xb); }, [](int) {});
    }
}

// the same from the input features `feat` of the current sample, which are advanced to the next sample of the rays
template <int CD, int ACT, int GRID, bool HAS_DIR>
__device__ __forceinline__ float4_t srn_forward_rotating(const NetParams& P, const char* lds, floatx16 (&feat)[2][mtiles(CD)],
                                                         const floatx16 (&dfeat)[2][mtiles(CD)], float px, float py, float pz, bool advance = true) {
    // rotation pieces behind MFMAs without conversion work of their own (NFILL) vs in pre(), where they also cover the LDS
    // latency of the first layer's reads: all of them in pre() is fastest (r01: NFILL = NP/2 -> 0: 132 -> 137 Gsamples/s)
    constexpr int NP = 16 * mtiles(CD);
    constexpr int NFILL = 0;
    half8_t xb[2][2 * mtiles(CD)];
    if constexpr (GRID == 1) {
        // latent grid + rotated features: the gathers of a tile are in flight behind that tile's converts and rotations (GridPre)
        constexpr int MT = mtiles(CD);
        GridPre G;
        grid_tap_bcast<false>(grid_tap(P, px, py, pz), G.gt[0], G.gt[1]);
        G.valid = 1;
        GridRaw raw;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            grid_load(P.grid, G.gt[t], 0, kTapHalfInOffset, raw);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(feat[t][m], q, 0.f, 0.f, xb[t][2 * m], xb[t][2 * m + 1]);
#pragma unroll
            for (int c = t * 8 * MT; c < (t + 1) * 8 * MT; ++c) fourier_advance_piece<CD, HAS_DIR>(feat, dfeat, c);
            __builtin_amdgcn_sched_barrier(0);
            float acc[8];
            grid_reduce_record<false, true>(raw, G.gt[t], 0, acc);
#pragma unroll
            for (int k = 1; k < 4; ++k) grid_reduce_record<false>(raw, G.gt[t], k, acc);
            G.gf[t] = grid_pack(acc);
        }
        return srn_layers<CD, ACT, GRID, HAS_DIR, 0, false>(P, lds, xb, px, py, pz, []() {}, [](int) {}, &G);
    } else {
        return srn_layers<CD, ACT, GRID, HAS_DIR, NFILL, false>(
            P, lds, xb, px, py, pz,
            [&]() {
                feature_fragments<CD>(feat, xb);
                if (advance) {  // (wave-uniform; off when every step re-derives its features: FVSRN_OPT_FOURIER_RESYNC = 1)
#pragma unroll
                    for (int c = 0; c < NP - NFILL; ++c) fourier_advance_piece<CD, HAS_DIR>(feat, dfeat, c);
                }
            },
            [&](int j) { fourier_advance_piece<CD, HAS_DIR>(feat, dfeat, NP - NFILL + j); });
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Small networks entirely in registers (render_small_kernel): a 32-wide Fourier-only network with NLC <= 3 C->C layers has
// 2 (NLC + 1) weight fragments (4 registers each) and NLC bias blocks (16 registers): 80 registers at NLC = 3.  With the 256
// registers of a 2-waves-per-SIMD kernel they stay resident for the whole launch, and the sample loop has no LDS access at
// all: the same schedule as srn_layers_pipelined without its 33 ds_read_b128 per wave step (8 weight fragments, 6 x 4 bias
// blocks), their address arithmetic and their waits, and the bias block enters as the C operand of each tile's first MFMA
// instead of passing through the accumulator registers.  Measured r01 on the headline frame with the LDS reads of the
// 3-wave kernel ablated: 126.7 -> 145.6 Gsamples/s at 3 waves, 140.6 at 2 waves (what fits), tools/ablate.sh.
#ifndef FVSRN_SLOT_ORDER
#define FVSRN_SLOT_ORDER 0
#endif
// RGRID = 1: the network has one 16-channel latent grid chunk: the phase fragment and the latent K-step fragment of layer 0 stay
// resident too (render_small_kernel with a latent grid: direct Fourier features, no LDS access in the sample loop either)
template <int NLC, int RGRID = 0>
struct ResidentNet {
    half8_t w[2 * (NLC + 1)];  // [layer 0 | hidden 1..NLC-1 | last][K step]
    floatx16 b[NLC];           // bias rows of this lane half, layers 0..NLC-1 (RGRID: b[0] is never read -- the first layer's bias is folded
                               // into its weights, NetParams::bias0Folded, and the compiler drops the 16 registers)
    float4_t bLast;
    half8_t wg[RGRID == 1 ? 1 : 0];  // latent K step of layer 0 (RGRID = 2: the cell table takes its place, srn_forward_resident_cells)
    half8_t wp[RGRID ? 1 : 0];       // phase fragment
};

template <int NLC, int RGRID>
__device__ __forceinline__ void load_resident(const NetParams& P, const char* lds, ResidentNet<NLC, RGRID>& R) {
    const int lane = lane_id();
    const char* ldsA = lds + 16 * lane;
    const char* ldsB = lds + 16 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i) R.w[i] = *reinterpret_cast<const half8_t*>(ldsA + P.offLayer0 + i * kFragBytes);
#pragma unroll
    for (int i = 0; i < 2 * (NLC - 1); ++i) R.w[2 + i] = *reinterpret_cast<const half8_t*>(ldsA + P.offHidden + i * kFragBytes);
#pragma unroll
    for (int i = 0; i < 2; ++i) R.w[2 * NLC + i] = *reinterpret_cast<const half8_t*>(ldsA + P.offLast + i * kFragBytes);
#pragma unroll
    for (int l = 0; l < NLC; ++l) {
        const float4_t* p = reinterpret_cast<const float4_t*>(ldsB + P.offBias + l * 128);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4_t v = p[2 * g];
            R.b[l][4 * g + 0] = v[0]; R.b[l][4 * g + 1] = v[1]; R.b[l][4 * g + 2] = v[2]; R.b[l][4 * g + 3] = v[3];
        }
    }
    R.bLast = *reinterpret_cast<const float4_t*>(lds + P.offBias + NLC * 128);
    if constexpr (RGRID == 1) R.wg[0] = *reinterpret_cast<const half8_t*>(ldsA + P.offLayer0 + 2 * kFragBytes);  // [g = 0][m = 0] behind the MT * KS Fourier fragments
    if constexpr (RGRID != 0) R.wp[0] = *reinterpret_cast<const half8_t*>(ldsA + P.offPhase);
}

// srn_layers_pipelined for CD = 2, GRID = 0 on a ResidentNet; returns the raw outputs of this lane's sample (ALL4: all four,
// colour networks; otherwise only output 0)
template <int ACT, int NLC, int NFILL, bool ALL4, int RGRID = 0, class Pre, class Fill>
__device__ __forceinline__ float4_t srn_layers_resident(const NetParams& P, const ResidentNet<NLC, RGRID>& R, half8_t (&xb)[2][2], Pre&& pre, Fill&& fill,
                                                        const half8_t* gf = nullptr, const CellPre<1>* cells = nullptr) {
    // the latent K step of layer 0: RGRID = 1: gf = B fragments of the two tiles' latent features, A = R.wg[0] (the first layer's latent
    // columns); RGRID = 2: through the cell table (cells_accumulate)
    [[maybe_unused]] const int lane_ = lane_id();
    auto latent = [&](int T, floatx16& a) {
        if constexpr (RGRID == 2) cells_accumulate<1>(P, *cells, T, lane_ >> 5, unsigned(lane_ & 31) * 16u, &a);
        else a = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.wg[0], gf[T], a, 0, 0, 0);
    };
    const float actA = P.actA, actB = P.actB;
    constexpr int NF0 = NFILL / 2;
    floatx16 acc[2];
    __builtin_amdgcn_sched_barrier(0);
    pre();
    // Wave priority: the vector-only stretches of the step (ray bookkeeping, feature converts, transfer function, blending)
    // run at priority 3, the MFMA chain at 0, so that the SIMD's other wave gets its VALU work issued while this one
    // only waits for the matrix pipe (r01: 136.8 -> 139.8 Gsamples/s; the reverse assignment: 138.0).
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // first layer, tile 0 (its MFMAs carry rotation pieces)
    const floatx16 zero16 = {0};
    interleave<2, NF0>([&](int i) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.w[i], xb[0][i], i == 0 ? (RGRID ? zero16 : R.b[0]) : acc[0], 0, 0, 0); },
                       [&](int j) { fill(j); });
    if constexpr (RGRID != 0) {  // latent K step of layer 0 (gf: B fragments of the two tiles' latent features)
        latent(0, acc[0]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // half-layer slot: layer l for tile T || activation + convert of the other tile
    auto slot = [&](auto tileTag, auto layerTag) {
        constexpr int T = decltype(tileTag)::value, O = 1 - T, L = decltype(layerTag)::value;
#if FVSRN_SLOT_ORDER == 1
        // both MFMAs of the slot first: the other tile's accumulators (its last MFMA was issued just before this slot) are
        // complete by the time the converts start, and the converts run in the shadow of this slot's second MFMA
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.w[2 * L], xb[T][0], R.b[L], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.w[2 * L + 1], xb[T][1], acc[T], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) act_pack_quarter<ACT>(acc[O], q, actA, actB, xb[O][0], xb[O][1]);
        __builtin_amdgcn_sched_barrier(0);
#else
        interleave<2, 4>([&](int i) { acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.w[2 * L + i], xb[T][i], i == 0 ? ((RGRID && L == 0) ? zero16 : R.b[L]) : acc[T], 0, 0, 0); },
                         [&](int q) { act_pack_quarter<ACT>(acc[O], q, actA, actB, xb[O][0], xb[O][1]); });
#endif
        if constexpr (RGRID != 0 && L == 0) {
            latent(T, acc[T]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    slot(T1{}, std::integral_constant<int, 0>{});
    if constexpr (NLC > 1) { slot(T0{}, std::integral_constant<int, 1>{}); slot(T1{}, std::integral_constant<int, 1>{}); }
    if constexpr (NLC > 2) { slot(T0{}, std::integral_constant<int, 2>{}); slot(T1{}, std::integral_constant<int, 2>{}); }
    static_assert(NLC >= 1 && NLC <= 3, "resident networks: 1..3 C->C layers");
    // last layer (16x16x32, see srn_layers_pipelined)
    float4_t o0 = R.bLast, o1 = R.bLast;
    interleave<2, 4>([&](int s) { o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(R.w[2 * NLC + s], xb[0][s], o0, 0, 0, 0); },
                     [&](int q) { act_pack_quarter<ACT>(acc[1], q, actA, actB, xb[1][0], xb[1][1]); });
    interleave<2, NFILL - NF0>([&](int s) { o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(R.w[2 * NLC + s], xb[1][s], o1, 0, 0, 0); },
                               [&](int j) { fill(NF0 + j); });
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_sched_barrier(0);
    const bool h = (lane_id() >> 5) != 0;
    float4_t out = {h ? o1[0] : o0[0], 0.f, 0.f, 0.f};
    if constexpr (ALL4) { out[1] = h ? o1[1] : o0[1]; out[2] = h ? o1[2] : o0[2]; out[3] = h ? o1[3] : o0[3]; }
    return out;
}

// 32-wide network with one 16-channel latent grid chunk, resident in registers: direct Fourier features (one phase MFMA + v_cos per
// tile), the tile's 8 gathers in flight behind them (GridPre order), then srn_layers_resident with the latent K step
template <int ACT, bool HAS_DIR, int NLC, bool ALL4, int FMODE = FM_COS>
__device__ __forceinline__ float4_t srn_forward_resident_grid(const NetParams& P, const ResidentNet<NLC, 1>& R, float px, float py, float pz,
                                                             float dx, float dy, float dz) {
    constexpr int NPASS = HAS_DIR ? 4 : 2;
    const int h = lane_id() >> 5;
    half8_t xb[2][2], gf[2];
    GridTap gt[2];
    grid_tap_bcast<false>(grid_tap(P, px, py, pz), gt[0], gt[1]);
    half8_t b0[2];
    phase_operands<HAS_DIR, true>(px, py, pz, dx, dy, dz, h, b0);
    GridRaw raw;
    grid_load(P.grid, gt[0], 0, kTapHalfInOffset, raw);
    // both phase MFMAs first: the second runs while the first tile's cosines issue (one exposed MFMA latency instead of two)
    floatx16 d[2];
    {
        const floatx16 z = {0};
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.wp[0], b0[0], z, 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.wp[0], b0[1], z, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1) grid_load(P.grid, gt[1], 0, kTapHalfInOffset, raw);
        __builtin_amdgcn_sched_barrier(0);
        phase_cos<FMODE>(d[t], NPASS);
#pragma unroll
        for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(d[t], q, 0.f, 0.f, xb[t][0], xb[t][1]);
        __builtin_amdgcn_sched_barrier(0);
        float acc[8];
        grid_reduce_record<false, true>(raw, gt[t], 0, acc);
#pragma unroll
        for (int k = 1; k < 4; ++k) grid_reduce_record<false>(raw, gt[t], k, acc);
        gf[t] = grid_pack(acc);
    }
    return srn_layers_resident<ACT, NLC, 0, ALL4, 1>(P, R, xb, []() {}, [](int) {}, gf);
}

template <int ACT, bool HAS_DIR, int NLC, bool ALL4>
__device__ __forceinline__ float4_t srn_forward_resident_cells(const NetParams& P, const ResidentNet<NLC, 2>& R, float px, float py, float pz,
                                                              float dx, float dy, float dz, unsigned long long validMask) {
    constexpr int NPASS = HAS_DIR ? 4 : 2;
    const int lane = lane_id();
    const int h = lane >> 5;
    CellPre<1> C;
    cell_prepare<1>(P, px, py, pz, validMask, h, unsigned(lane & 31) * 16u, C);
    half8_t b0[2], xb[2][2];
    phase_operands<HAS_DIR, true>(px, py, pz, dx, dy, dz, h, b0);
    floatx16 d[2];
    {
        const floatx16 z = {0};
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.wp[0], b0[0], z, 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.wp[0], b0[1], z, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        phase_cos<FM_COS>(d[t], NPASS);
#pragma unroll
        for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(d[t], q, 0.f, 0.f, xb[t][0], xb[t][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    return srn_layers_resident<ACT, NLC, 0, ALL4, 2>(P, R, xb, []() {}, [](int) {}, nullptr, &C);
}

// The same with rotated Fourier features (r03): the current features stay in 32 registers (fp32, both tiles), their per-step rotation
// -- 32 more values per lane that are constant along a ray -- is parked in LDS (`dfeatLds`: this wave's 8 KiB, [tile][quad][lane] float4,
// conflict-free ds_read_b128 / ds_write_b128) and read back 16 registers at a time while the tile's gathers are in flight.  Per wave step
// this replaces 2 phase MFMAs, 56 v_cos_f32 / v_sin_f32 (quarter rate), the position converts and their lane swaps by 32 packed-fp32
// instructions and 8 LDS reads; the LDS is idle in this kernel otherwise (SQ_INSTS_LDS = 0).
__device__ __forceinline__ void park_dfeat(float* dfeatLds, const floatx16 (&dfeat)[2][1]) {
    const int lane = lane_id();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4_t*>(dfeatLds + ((t * 4 + q) * 64 + lane) * 4) =
                float4_t{dfeat[t][0][4 * q], dfeat[t][0][4 * q + 1], dfeat[t][0][4 * q + 2], dfeat[t][0][4 * q + 3]};
}

template <int ACT, bool HAS_DIR, int NLC, bool ALL4>
__device__ __forceinline__ float4_t srn_forward_rotating_resident_grid(const NetParams& P, const ResidentNet<NLC, 1>& R, floatx16 (&feat)[2][1],
                                                                      const float* dfeatLds, float px, float py, float pz) {
    const int lane = lane_id();
    half8_t xb[2][2], gf[2];
    GridTap gt[2];
    grid_tap_bcast<false>(grid_tap(P, px, py, pz), gt[0], gt[1]);
    GridRaw raw;
    grid_load(P.grid, gt[0], 0, kTapHalfInOffset, raw);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1) grid_load(P.grid, gt[1], 0, kTapHalfInOffset, raw);
        float4_t rot[4];  // this tile's rotation: four LDS reads in flight behind the converts
#pragma unroll
        for (int q = 0; q < 4; ++q) rot[q] = *reinterpret_cast<const float4_t*>(dfeatLds + ((t * 4 + q) * 64 + lane) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(feat[t][0], q, 0.f, 0.f, xb[t][0], xb[t][1]);
        __builtin_amdgcn_sched_barrier(0);  // (converts first: the rotation then updates the feature registers in place, no copies)
        // -> the next sample of the rays (fourier_advance_piece, in place)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {  // piece k = 2q + j of this tile: registers (2k, 2k + 1)
                constexpr int NPASS = HAS_DIR ? 4 : 2;
                const int k = 2 * q + j;
                float2_t cs = {feat[t][0][2 * k], feat[t][0][2 * k + 1]};
                const float2_t dd = {rot[q][2 * j], rot[q][2 * j + 1]};
                if (2 * k < NPASS) {
                    cs += dd;
                } else {
                    float2_t tmp;
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(tmp) : "v"(cs), "v"(dd));
                    asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(cs) : "v"(dd), "v"(tmp));
                }
                feat[t][0][2 * k] = cs[0];
                feat[t][0][2 * k + 1] = cs[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float acc[8];
        grid_reduce_record<false, true>(raw, gt[t], 0, acc);
#pragma unroll
        for (int k = 1; k < 4; ++k) grid_reduce_record<false>(raw, gt[t], k, acc);
        gf[t] = grid_pack(acc);
    }
    return srn_layers_resident<ACT, NLC, 0, ALL4, 1>(P, R, xb, []() {}, [](int) {}, gf);
}

// srn_forward_rotating on a ResidentNet
template <int ACT, bool HAS_DIR, int NLC, bool ALL4>
__device__ __forceinline__ float4_t srn_forward_rotating_resident(const NetParams& P, const ResidentNet<NLC>& R, floatx16 (&feat)[2][1],
                                                               const floatx16 (&dfeat)[2][1], bool advance = true) {
#ifndef FVSRN_SMALL_NFILL
#define FVSRN_SMALL_NFILL 0
#endif
    // rotation pieces behind MFMAs; the rest runs in pre().  r01: 0 / 4 / 8 / 12 / 16 pieces behind MFMAs -> 157.4 / 155.1 / 154.1 /
    // 154.4 / 155.1 Gsamples/s: the whole rotation belongs to the high-priority vector phase of the step
    constexpr int NP = 16, NFILL = FVSRN_SMALL_NFILL;
    half8_t xb[2][2];
    return srn_layers_resident<ACT, NLC, NFILL, ALL4>(
        P, R, xb,
        [&]() {
            feature_fragments<2>(feat, xb);
            if (advance) {  // (wave-uniform; off when every step re-derives its features: FVSRN_OPT_FOURIER_RESYNC = 1, the reference's arithmetic)
#pragma unroll
                for (int c = 0; c < NP - NFILL; ++c) fourier_advance_piece<2, HAS_DIR>(feat, dfeat, c);
            }
        },
        [&](int j) { fourier_advance_piece<2, HAS_DIR>(feat, dfeat, NP - NFILL + j); });
}

// srn_forward_resident_cells with rotated Fourier features (the registers the gathers held take the rotation state)
template <int ACT, bool HAS_DIR, int NLC, bool ALL4>
__device__ __forceinline__ float4_t srn_forward_rotating_resident_cells(const NetParams& P, const ResidentNet<NLC, 2>& R, CellResident<1>& cells, floatx16 (&feat)[2][1],
                                                                       const floatx16 (&dfeat)[2][1], float px, float py, float pz,
                                                                       unsigned long long validMask, bool advance = true) {
    const int lane = lane_id();
    CellPre<1> C;
#ifdef FVSRN_CELLS_NO_RESIDENT_PAIR  // (A/B build: the pick and the table fetch at every step, r04 / r05)
    cell_prepare<1>(P, px, py, pz, validMask, lane >> 5, unsigned(lane & 31) * 16u, C);
#else
    cell_prepare_resident<1, FVSRN_CELLS_SCALED_POS != 0>(P, cells, px, py, pz, validMask, lane >> 5, unsigned(lane & 31) * 16u, C);
#endif
    half8_t xb[2][2];
    return srn_layers_resident<ACT, NLC, 0, ALL4, 2>(
        P, R, xb,
        [&]() {
            feature_fragments<2>(feat, xb);
#ifndef FVSRN_CELLS_NO_CVT_BARRIER
            // r06: both tiles' converts BEFORE the rotation.  Without the barrier hipcc sinks tile 1's eight converts behind tile 0's first-layer MFMAs, which
            // keeps tile 1's un-rotated features alive across its rotation: the rotation then writes fresh registers and the loop ends with eight v_mov_b64
            // back into place (seen in the r05 listing of render_small_kernel<4,false,3,4,2>; the kernel without a grid has none).
            // (an empty asm that ties tile 1's fragments and its features together: a sched_barrier alone does not stop the IR-level sinking)
            asm volatile("" : "+v"(feat[1][0]), "+v"(xb[1][0]), "+v"(xb[1][1]));
#endif
            if (advance) {
#pragma unroll
                for (int c = 0; c < 16; ++c) fourier_advance_piece<2, HAS_DIR>(feat, dfeat, c);
            }
        },
        [](int) {}, nullptr, &C);
}

// output parametrization, renderer_volume_tensorcores.cuh:1054-1158. The reference rounds the
// last-layer result to half before the fp32 output activation; we keep fp32.
// v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division: the result feeds a transfer function / an fp32 blend
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : __logf(1.f + __expf(x)); }

}  // namespace fvsrn
