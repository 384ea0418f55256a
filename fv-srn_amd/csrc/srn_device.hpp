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
//   * the last (C -> 1|4) layer is one more MFMA whose rows are replicated for both lane halves,
//     replacing the per-lane hfma loop :1138-1143
#pragma once
#include <hip/hip_runtime.h>

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

// own value v (one per lane = per sample) -> value of sample (lane&31) of tile 0 / tile 1
__device__ __forceinline__ void tile_bcast(float v, float& t0, float& t1) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    t0 = __uint_as_float(r[0]);
    t1 = __uint_as_float(r[1]);
}

template <int ACT>
__device__ __forceinline__ float act_f32(float x, float a, float b) {
    if constexpr (ACT == ACT_RELU) {
        return fmaxf(x, 0.f);
    } else if constexpr (ACT == ACT_SINE) {  // sin(p x), a = p/(2 pi)          renderer_activations.cuh: Sine
        return __builtin_amdgcn_sinf(x * a);
    } else if constexpr (ACT == ACT_SNAKE) {  // x + sin^2(p x)/p = x + (1 - cos(2 p x))/(2p), a = p/pi, b = 1/(2p)
        const float c = __builtin_amdgcn_cosf(x * a);
        return fmaf(-c, b, x + b);
    } else {  // SnakeAlt: (x + 1 - cos(2 p x)) / (2p)
        const float c = __builtin_amdgcn_cosf(x * a);
        return fmaf(x - c, b, b);
    }
}

template <int ACT>
__device__ __forceinline__ float2_t act_f32x2(float2_t x, float a, float b) {
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
    } else {
        return float2_t{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    }
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
    const float fx = px * float(P.gridX) - 0.5f, fy = py * float(P.gridY) - 0.5f, fz = pz * float(P.gridZ) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    const float wx = fx - x0f, wy = fy - y0f, wz = fz - z0f;
    const int xi = min(max(int(x0f) + 1, 0), P.gridX);  // record index: x-clamping is baked into the records
    const int y0 = min(max(int(y0f), 0), P.gridY - 1), y1 = min(max(int(y0f) + 1, 0), P.gridY - 1);
    const int z0 = min(max(int(z0f), 0), P.gridZ - 1), z1 = min(max(int(z0f) + 1, 0), P.gridZ - 1);
    const unsigned rec = unsigned(P.gridC) * 4u;             // bytes per record: G channels x 2 x fp16
    const unsigned row = unsigned(P.gridX + 1) * rec;
    const unsigned xo = unsigned(xi) * rec;
    GridTap t;
    t.off[0] = unsigned(z0 * P.gridY + y0) * row + xo;
    t.off[1] = unsigned(z0 * P.gridY + y1) * row + xo;
    t.off[2] = unsigned(z1 * P.gridY + y0) * row + xo;
    t.off[3] = unsigned(z1 * P.gridY + y1) * row + xo;
    const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
    const float w4[4] = {uz * uy, uz * wy, wz * uy, wz * wy};
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

// own-sample tap -> taps of sample (lane&31) of tile 0 / tile 1
template <bool WITH_LO>
__device__ __forceinline__ void grid_tap_bcast(const GridTap& own, GridTap& t0, GridTap& t1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto a = __builtin_amdgcn_permlane32_swap(own.off[k], own.off[k], false, false);
        t0.off[k] = a[0]; t1.off[k] = a[1];
        auto b = __builtin_amdgcn_permlane32_swap(own.w[k], own.w[k], false, false);
        t0.w[k] = b[0]; t1.w[k] = b[1];
        if constexpr (WITH_LO) {
            auto c = __builtin_amdgcn_permlane32_swap(own.wlo[k], own.wlo[k], false, false);
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

// trilinear fetch of 8 channels [16*g + 8*h, +8) from one working grid: fp32 results
template <bool WITH_LO>
__device__ __forceinline__ void grid_fetch8(const void* grid, const GridTap& t, int g, int h, float acc[8]) {
    const char* base = reinterpret_cast<const char*>(grid) + (g * 64 + h * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4_t* p = reinterpret_cast<const uint4_t*>(base + t.off[k]);
        const uint4_t v0 = p[0], v1 = p[1];
        const half2_t w = __builtin_bit_cast(half2_t, t.w[k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // NB: __builtin_bit_cast applied directly to a vector ELEMENT (v0[j]) is miscompiled by clang 22 /
            // ROCm 7.2 (only element 0 survives): go through a scalar temporary
            const unsigned u0 = v0[j], u1 = v1[j];
            acc[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u0), w, acc[j], false);
            acc[4 + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u1), w, acc[4 + j], false);
            if constexpr (WITH_LO) {
                const half2_t wl = __builtin_bit_cast(half2_t, t.wlo[k]);
                acc[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u0), wl, acc[j], false);
                acc[4 + j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, u1), wl, acc[4 + j], false);
            }
        }
    }
}

// 8 channels [16*g + 8*h, +8) of the tile sample described by `t`, as the B fragment of latent K step g
// GRID: 1 = `grid` holds decoded, time-blended values (FLOAT, BYTE_LINEAR); 2 = BYTE_GAUSSIAN
template <int GRID>
__device__ __forceinline__ half8_t grid_features(const NetParams& P, const GridTap& t, int g, int h) {
    float acc[8];
    grid_fetch8<GRID == 2>(P.grid, t, g, h, acc);
    if constexpr (GRID == 2) {  // EncodeGridValue<BYTE_GAUSSIAN> :370-383
        float accB[8];
        grid_fetch8<true>(P.gridB, t, g, h, accB);
        const int c0 = 16 * g + 8 * h;
        const bool isTime = c0 < P.gridTimeChannels;  // a 16-channel chunk never straddles time / ensemble channels
        const float* mean = isTime ? P.gridMeanTime + c0 : P.gridMeanEns + (c0 - P.gridTimeChannels);
        const float* sd = isTime ? P.gridStdTime + c0 : P.gridStdEns + (c0 - P.gridTimeChannels);
        const float f = isTime ? P.gridFrac : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xa = acc[j] * (1.0f / 255.0f), xb = accB[j] * (1.0f / 255.0f);  // cudaReadModeNormalizedFloat
            const float ya = 1.4142135623730950488f * erfinv_dev((2.0f - 1e-4f) * (xa - 0.5f));
            const float yb = 1.4142135623730950488f * erfinv_dev((2.0f - 1e-4f) * (xb - 0.5f));
            const float va = mean[j] + ya * sd[j], vb = mean[j] + yb * sd[j];
            acc[j] = va + f * (vb - va);
        }
    }
    half8_t out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float2_t v = {acc[2 * i], acc[2 * i + 1]};
        const half2_t hh = __builtin_convertvector(v, half2_t);
        out[2 * i] = hh[0]; out[2 * i + 1] = hh[1];
    }
    return out;
}

// Evaluates the network for the 64 samples of this wave.
//   (px,py,pz): this lane's sample position, already normalized to the unit box
//   (dx,dy,dz): this lane's view direction (only read when the network uses it)
// Returns the raw last-layer outputs (before the output parametrization) of this lane's sample.
// EXEC must be all ones.
template <int CD, int ACT, int GRID, bool HAS_DIR>
__device__ __forceinline__ float4_t srn_forward(const NetParams& P, const char* lds, float px, float py, float pz,
                                                float dx, float dy, float dz) {
    constexpr int C = 16 * CD;
    constexpr int MT = (C + 31) / 32;
    constexpr int KS = CD;
    const int lane = lane_id();
    const int h = lane >> 5;
    const float actA = P.actA, actB = P.actB;

    // Positions of the two sample tiles.  Without a latent grid only their fp16 images are needed: convert first
    // (one v_cvt_pk per coordinate, both halves = the hi/lo slot pair of the phase matrix), then exchange.
    unsigned tph[2][3];
    {
        const float pp[3] = {px, py, pz};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float2_t v = {pp[i], pp[i]};
            const half2_t hh = __builtin_convertvector(v, half2_t);
            const unsigned u = __builtin_bit_cast(unsigned, hh);
            auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            tph[0][i] = r[0];
            tph[1][i] = r[1];
        }
    }
    GridTap gt[2];
    if constexpr (GRID != 0) grid_tap_bcast<GRID == 2>(grid_tap(P, px, py, pz), gt[0], gt[1]);
    float td[2][3] = {{0, 0, 0}, {0, 0, 0}};
    if constexpr (HAS_DIR) {
        tile_bcast(dx, td[0][0], td[1][0]);
        tile_bcast(dy, td[0][1], td[1][1]);
        tile_bcast(dz, td[0][2], td[1][2]);
    }

    half8_t xb[2][KS + (KS & 1)];  // B fragments of the current layer input, per tile

    // ---- Fourier layer: phases by MFMA, then cos ------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        half8_t b0;
        if constexpr (HAS_DIR) {
            const float2_t vx = {td[t][0], td[t][0]}, vy = {td[t][1], td[t][1]}, vz = {td[t][2], td[t][2]};
            const unsigned dxh = __builtin_bit_cast(unsigned, __builtin_convertvector(vx, half2_t));
            const unsigned dyh = __builtin_bit_cast(unsigned, __builtin_convertvector(vy, half2_t));
            const unsigned dzh = __builtin_bit_cast(unsigned, __builtin_convertvector(vz, half2_t));
            const uint4_t u = {h ? dxh : tph[t][0], h ? dyh : tph[t][1], h ? dzh : tph[t][2], h ? 0u : 0x00003c00u};
            b0 = __builtin_bit_cast(half8_t, u);
        } else {
            // lane half 1 supplies K slots 8..15, whose phase-matrix entries are all zero: its B values are
            // multiplied by 0 and only need to be finite, so no select is needed
            const uint4_t u = {tph[t][0], tph[t][1], tph[t][2], 0x00003c00u /* (1.0h, 0) */};
            b0 = __builtin_bit_cast(half8_t, u);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            floatx16 d = {0};
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(lds, P.offPhase + m * kFragBytes, lane), b0, d, 0, 0, 0);
            // registers 0,1 (0..3 with direction) of M tile 0 are pass-through channels
            floatx16 x;
            if (P.fourierNeedsFract) {  // keep v_cos_f32 inside its +-256 revolution domain
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(d[r]));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_cosf(d[r]);
            }
            if (m == 0) {
                constexpr int NP = HAS_DIR ? 4 : 2;
#pragma unroll
                for (int r = 0; r < NP; ++r) x[r] = d[r];
            }
            half8_t f0, f1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float2_t v0 = {x[2 * i], x[2 * i + 1]};
                float2_t v1 = {x[8 + 2 * i], x[8 + 2 * i + 1]};
                half2_t h0 = __builtin_convertvector(v0, half2_t);
                half2_t h1 = __builtin_convertvector(v1, half2_t);
                f0[2 * i] = h0[0]; f0[2 * i + 1] = h0[1];
                f1[2 * i] = h1[0]; f1[2 * i + 1] = h1[1];
            }
            xb[t][2 * m] = f0;
            if (2 * m + 1 < KS + (KS & 1)) xb[t][2 * m + 1] = f1;
        }
    }

    // ---- C -> C layers -----------------------------------------------------------------------------------
    const int NL = P.numLayers;
    for (int l = 0; l < NL; ++l) {
        const int ks = (GRID != 0 && l == 0) ? KS + P.gridK : KS;
        const int wOff = l == 0 ? P.offLayer0 : P.offHidden + (l - 1) * MT * KS * kFragBytes;
        const int bOff = P.offBias + l * 32 * MT * 4;
        floatx16 acc[2][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const floatx16 bias = lds_bias(lds, bOff + m * 128, h);
#ifndef FVSRN_K_MAJOR
            if constexpr (KS <= 4) {
                // tile-major issue order: tile 0's chain completes while tile 1's MFMAs still run, so the VALU work
                // on tile 0's accumulators overlaps the matrix pipe inside one wave
                half8_t a[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) a[s] = lds_frag(lds, wOff + (m * ks + s) * kFragBytes, lane);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], xb[t][s], s == 0 ? bias : acc[t][m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);  // keep the two chains apart (hipcc would re-interleave them)
                }
            } else
#endif
            {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const half8_t a = lds_frag(lds, wOff + (m * ks + s) * kFragBytes, lane);
                    acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[0][s], s == 0 ? bias : acc[0][m], 0, 0, 0);
                    acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[1][s], s == 0 ? bias : acc[1][m], 0, 0, 0);
                }
            }
        }
        if constexpr (GRID != 0) {
            if (l == 0) {
                for (int g = 0; g < P.gridK; ++g) {
                    const half8_t g0 = grid_features<GRID>(P, gt[0], g, h);
                    const half8_t g1 = grid_features<GRID>(P, gt[1], g, h);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const half8_t a = lds_frag(lds, wOff + (m * ks + KS + g) * kFragBytes, lane);
                        acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, g0, acc[0][m], 0, 0, 0);
                        acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, g1, acc[1][m], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                half8_t f0, f1;
                act_pack<ACT>(acc[t][m], actA, actB, f0, f1);
                xb[t][2 * m] = f0;
                if (2 * m + 1 < KS + (KS & 1)) xb[t][2 * m + 1] = f1;
            }
    }

    // ---- last layer ----------------------------------------------------------------------------------------
    const floatx16 biasL = lds_bias(lds, P.offBias + NL * 32 * MT * 4, 0);  // rows 0..7 hold the (replicated) bias
    floatx16 o0 = biasL, o1 = biasL;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const half8_t a = lds_frag(lds, P.offLast + s * kFragBytes, lane);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[0][s], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[1][s], o1, 0, 0, 0);
    }
    // rows 0..3 (lane half 0) and rows 4..7 (lane half 1) both carry outputs 0..3 of sample c of the tile
    float4_t out = {0, 0, 0, 0};
    out[0] = h ? o1[0] : o0[0];
    if (P.outputMode >= FVSRN_OUT_RGBO) {  // wave-uniform: only colour / gradient networks have outputs 1..3
        out[1] = h ? o1[1] : o0[1];
        out[2] = h ? o1[2] : o0[2];
        out[3] = h ? o1[3] : o0[3];
    }
    return out;
}

// output parametrization, renderer_volume_tensorcores.cuh:1054-1158. The reference rounds the
// last-layer result to half before the fp32 output activation; we keep fp32.
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : __logf(1.f + __expf(x)); }

}  // namespace fvsrn
