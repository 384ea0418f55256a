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

namespace fvsrn {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

enum { ACT_RELU = 0, ACT_SINE = 1, ACT_SNAKE = 2, ACT_SNAKEALT = 3 };

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

// 16 fp32 accumulator values of one M tile -> two B fragments (K steps 2m, 2m+1) with activation
template <int ACT>
__device__ __forceinline__ void act_pack(const floatx16& d, float a, float b, half8_t& f0, half8_t& f1) {
    if constexpr (ACT == ACT_RELU) {
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
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float2_t v0 = {act_f32<ACT>(d[2 * i], a, b), act_f32<ACT>(d[2 * i + 1], a, b)};
            float2_t v1 = {act_f32<ACT>(d[8 + 2 * i], a, b), act_f32<ACT>(d[8 + 2 * i + 1], a, b)};
            half2_t h0 = __builtin_convertvector(v0, half2_t);
            half2_t h1 = __builtin_convertvector(v1, half2_t);
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

// trilinear latent features: 8 channels [16*g + 8*h, +8) of the sample at normalized position p,
// texture semantics of the reference (normalized coords, clamp, linear; :581-596):
// texel coordinate = p*N - 0.5.
__device__ __forceinline__ half8_t grid_features(const NetParams& P, float px, float py, float pz, int g, int h) {
    const float fx = px * float(P.gridX) - 0.5f, fy = py * float(P.gridY) - 0.5f, fz = pz * float(P.gridZ) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    const float wx = fx - x0f, wy = fy - y0f, wz = fz - z0f;
    const int x0 = min(max(int(x0f), 0), P.gridX - 1), x1 = min(max(int(x0f) + 1, 0), P.gridX - 1);
    const int y0 = min(max(int(y0f), 0), P.gridY - 1), y1 = min(max(int(y0f) + 1, 0), P.gridY - 1);
    const int z0 = min(max(int(z0f), 0), P.gridZ - 1), z1 = min(max(int(z0f) + 1, 0), P.gridZ - 1);
    const half8_t* base = reinterpret_cast<const half8_t*>(P.grid) + (2 * g + h);
    const int cs = P.gridC >> 3;  // half8 per voxel
    auto at = [&](int z, int y, int x) { return base[size_t((z * P.gridY + y) * P.gridX + x) * cs]; };
    const half8_t v000 = at(z0, y0, x0), v001 = at(z0, y0, x1), v010 = at(z0, y1, x0), v011 = at(z0, y1, x1);
    const half8_t v100 = at(z1, y0, x0), v101 = at(z1, y0, x1), v110 = at(z1, y1, x0), v111 = at(z1, y1, x1);
    const float ux = 1.f - wx, uy = 1.f - wy, uz = 1.f - wz;
    const float w000 = uz * uy * ux, w001 = uz * uy * wx, w010 = uz * wy * ux, w011 = uz * wy * wx;
    const float w100 = wz * uy * ux, w101 = wz * uy * wx, w110 = wz * wy * ux, w111 = wz * wy * wx;
    half8_t out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = float(v000[j]) * w000;
        a = fmaf(float(v001[j]), w001, a);
        a = fmaf(float(v010[j]), w010, a);
        a = fmaf(float(v011[j]), w011, a);
        a = fmaf(float(v100[j]), w100, a);
        a = fmaf(float(v101[j]), w101, a);
        a = fmaf(float(v110[j]), w110, a);
        a = fmaf(float(v111[j]), w111, a);
        out[j] = _Float16(a);
    }
    return out;
}

// Evaluates the network for the 64 samples of this wave.
//   (px,py,pz): this lane's sample position, already normalized to the unit box
//   (dx,dy,dz): this lane's view direction (only read when the network uses it)
// Returns the raw last-layer outputs (before the output parametrization) of this lane's sample.
// EXEC must be all ones.
template <int CD, int ACT, bool HAS_GRID, bool HAS_DIR>
__device__ __forceinline__ float4_t srn_forward(const NetParams& P, const char* lds, float px, float py, float pz,
                                                float dx, float dy, float dz) {
    constexpr int C = 16 * CD;
    constexpr int MT = (C + 31) / 32;
    constexpr int KS = CD;
    const int lane = lane_id();
    const int h = lane >> 5;
    const float actA = P.actA, actB = P.actB;

    // tile positions
    float tp[2][3];
    tile_bcast(px, tp[0][0], tp[1][0]);
    tile_bcast(py, tp[0][1], tp[1][1]);
    tile_bcast(pz, tp[0][2], tp[1][2]);
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
            const float sx = h ? td[t][0] : tp[t][0], sy = h ? td[t][1] : tp[t][1], sz = h ? td[t][2] : tp[t][2];
            const _Float16 hx = _Float16(sx), hy = _Float16(sy), hz = _Float16(sz);
            b0 = half8_t{hx, hx, hy, hy, hz, hz, h ? _Float16(0) : _Float16(1), _Float16(0)};
        } else {
            const _Float16 hx = _Float16(tp[t][0]), hy = _Float16(tp[t][1]), hz = _Float16(tp[t][2]);
            const half8_t v = {hx, hx, hy, hy, hz, hz, _Float16(1), _Float16(0)};
            const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            b0 = h ? z : v;
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
        const int ks = (HAS_GRID && l == 0) ? KS + P.gridK : KS;
        const int wOff = l == 0 ? P.offLayer0 : P.offHidden + (l - 1) * MT * KS * kFragBytes;
        const int bOff = P.offBias + l * 32 * MT * 4;
        floatx16 acc[2][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const floatx16 bias = lds_bias(lds, bOff + m * 128, h);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const half8_t a = lds_frag(lds, wOff + (m * ks + s) * kFragBytes, lane);
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[0][s], s == 0 ? bias : acc[0][m], 0, 0, 0);
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[1][s], s == 0 ? bias : acc[1][m], 0, 0, 0);
            }
        }
        if constexpr (HAS_GRID) {
            if (l == 0) {
                for (int g = 0; g < P.gridK; ++g) {
                    const half8_t g0 = grid_features(P, tp[0][0], tp[0][1], tp[0][2], g, h);
                    const half8_t g1 = grid_features(P, tp[1][0], tp[1][1], tp[1][2], g, h);
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
    float4_t out;
    out[0] = h ? o1[0] : o0[0];
    out[1] = h ? o1[1] : o0[1];
    out[2] = h ? o1[2] : o0[2];
    out[3] = h ? o1[3] : o0[3];
    return out;
}

// output parametrization, renderer_volume_tensorcores.cuh:1054-1158. The reference rounds the
// last-layer result to half before the fp32 output activation; we keep fp32.
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : __logf(1.f + __expf(x)); }

}  // namespace fvsrn
