// GRADIENT_MODE_ADJOINT_METHOD of the reference (renderer_volume_tensorcores.cuh:1198-1540): the analytic gradient of the network's
// density with respect to the (normalized) sample position.
//
// The reference walks the network backwards: it keeps the pre-activation of every hidden unit of every sample in shared memory
// during the forward pass (32 x C x L halfs per warp, which is what limits its warps per block, computeMaxWarps(.., adjoint)),
// then multiplies the adjoint by the TRANSPOSED weight matrices layer by layer (:1285-1335) and finishes with the derivative of
// the Fourier features (:1466-1500) and central differences of the latent grid (:609-735).
//
// On CDNA4 the same derivative is computed in FORWARD mode, in the same MFMA pass as the value: the three tangents
// d/dx, d/dy, d/dz of every layer's output are three more sample tiles behind the SAME weight fragments,
//     dx_l = W_l dy_{l-1},   dy_l = act'(x_l) * dx_l,
// with act'(x_l) taken from the accumulator registers that hold x_l at that moment.  Nothing is stored, no transposed weight
// image is needed, and each weight fragment read from LDS feeds 4 MFMAs instead of 1 (a backward pass would need 2 x the
// forward work plus the activation store; this needs 4 x the forward work and no memory) -- the 6 extra network evaluations
// of the finite-difference mode cost 7 x.  Mathematically the result is the reference's: value-gradient = J^T e_density = the
// density row of J; the rounding points differ (tangents are rounded to fp16 between layers like activations are, the
// reference rounds its adjoints to half), tolerance in tests/test_gpu_parity.py::test_adjoint_*.
//   * inputs: tangent of a pass-through position row = 1 on its own axis; of a (cos, sin) pair = 2 pi c (-sin, cos) with c =
//     the row's coefficient for that axis, obtained from the phase MFMA applied to the unit vector of the axis; time / direction
//     inputs have zero tangents (the reference differentiates w.r.t. the position only)
//   * latent grid: central differences with step latentGridDifferencesStepSize = 1 / (resolution * 4) in normalized
//     coordinates, in fp32 with hi+lo filter weights (the quotient amplifies rounding by 2 * resolution * 4)
//   * output: density / densitygrad / densitycurvature: times the derivative of the sigmoid at the stored pre-sigmoid value
//     (:1228-1231); the :direct and :cubic modes: un-clamped (:1233-1237).  Colour networks: not supported (:1262), like there.
#pragma once
#include "srn_device.hpp"

namespace fvsrn {

// act'(x) (fp32); a, b as in act_f32
template <int ACT>
__device__ __forceinline__ float act_derivative(float x, float a, float b) {
    if constexpr (ACT == ACT_RELU || ACT == ACT_RELU01) {
        return x > 0.f ? 1.f : 0.f;  // activations::ReLU::adjoint (float / pre-sm_80 half form): v > 0 ? zAdj : 0
    } else if constexpr (ACT == ACT_SINE) {  // p cos(p x), a = p / (2 pi)
        return 6.28318530717958647692f * a * __builtin_amdgcn_cosf(x * a);
    } else if constexpr (ACT == ACT_SNAKE) {  // 1 + sin(2 p x), a = p / pi
        return 1.f + __builtin_amdgcn_sinf(x * a);
    } else if constexpr (ACT == ACT_SIGMOID) {  // e^x / (1 + e^x)^2 = s (1 - s)
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * a));
        return s * (1.f - s);
    } else {  // SnakeAlt: sin(2 p x) + 1 / (2p)
        return __builtin_amdgcn_sinf(x * a) + b;
    }
}

// 8 fp32 channel values [16 g + 8 h, +8) of the sample described by tap `t` (grid_features before the fp16 packing); hi+lo weights
template <int GRID>
__device__ __forceinline__ void grid_values8(const NetParams& P, const GridTap& t, int g, int h, float (&acc)[8]) {
    grid_fetch8<true>(P.grid, t, g, h, acc);
    if constexpr (GRID == 2) {  // EncodeGridValue<BYTE_GAUSSIAN> :370-383, as in grid_features
        float accB[8];
        grid_fetch8<true>(P.gridB, t, g, h, accB);
        const int c0 = 16 * g + 8 * h;
        const bool isTime = c0 < P.gridTimeChannels;
        const float* mean = isTime ? P.gridMeanTime + c0 : P.gridMeanEns + (c0 - P.gridTimeChannels);
        const float* sd = isTime ? P.gridStdTime + c0 : P.gridStdEns + (c0 - P.gridTimeChannels);
        const float f = isTime ? P.gridFrac : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xa = acc[j] * (1.0f / 255.0f), xb = accB[j] * (1.0f / 255.0f);
            const float ya = 1.4142135623730950488f * erfinv_dev((2.0f - 1e-4f) * (xa - 0.5f));
            const float yb = 1.4142135623730950488f * erfinv_dev((2.0f - 1e-4f) * (xb - 0.5f));
            const float va = mean[j] + ya * sd[j], vb = mean[j] + yb * sd[j];
            acc[j] = va + f * (vb - va);
        }
    }
}

// Value and position gradient of the network for the 64 samples of the wave (EXEC all ones).  Returns the raw last-layer outputs of
// this lane's sample in `out` (like srn_forward) and d(out[0]) / d(normalized position) in (gx, gy, gz).
//   (px,py,pz): this lane's sample position in unit-box coordinates; (dx,dy,dz): view direction (networks that use it)
//   gridStep: central-difference step of the latent grid in unit-box coordinates
// NT = tangents carried per pass: 3 (one pass) while the accumulators of 4 sample tiles fit the register budget, else 1 (three
// passes, the value is recomputed in each).
template <int CD, int ACT, int GRID, bool HAS_DIR, int FMODE>
__device__ __forceinline__ float4_t srn_forward_gradient(const NetParams& P, const char* lds, float px, float py, float pz, float dx, float dy,
                                                         float dz, float gridStep, float& gx, float& gy, float& gz) {
    constexpr int MT = mtiles(CD), KS = CD;
    constexpr int NT = MT <= 2 ? 3 : 1;  // tangents per pass
    constexpr int NPASS_ROWS = HAS_DIR ? 4 : 2;  // pass-through registers of M tile 0 (fourier_features)
    const int lane = lane_id();
    const int h = lane >> 5;
    const float actA = P.actA, actB = P.actB;
    const int NL = P.numLayers;
    float4_t result = {0, 0, 0, 0};
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;

    // positions / directions of the two sample tiles (fp32) and their fp16 images for the phase MFMA
    float tp[2][3], td[2][3] = {{0, 0, 0}, {0, 0, 0}};
    tile_bcast(px, tp[0][0], tp[1][0]);
    tile_bcast(py, tp[0][1], tp[1][1]);
    tile_bcast(pz, tp[0][2], tp[1][2]);
    if constexpr (HAS_DIR) {
        tile_bcast(dx, td[0][0], td[1][0]);
        tile_bcast(dy, td[0][1], td[1][1]);
        tile_bcast(dz, td[0][2], td[1][2]);
    }

#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        // (selects instead of tp[t][i]: a run-time index would put the arrays into scratch memory)
        const float tq[3] = {t ? tp[1][0] : tp[0][0], t ? tp[1][1] : tp[0][1], t ? tp[1][2] : tp[0][2]};
        const float tdir[3] = {t ? td[1][0] : td[0][0], t ? td[1][1] : td[0][1], t ? td[1][2] : td[0][2]};
        unsigned tph[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float2_t v = {tq[i], tq[i]};
            tph[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
        }
        const half8_t bpos = phase_operand<HAS_DIR>(tph, tdir, h);
        GridTap tap{};
        if constexpr (GRID != 0) tap = grid_tap(P, tq[0], tq[1], tq[2]);

#pragma unroll 1
        for (int pass = 0; pass < 3 / NT; ++pass) {
            // column sets: 0 = value, 1 + i = tangent of axis (pass * NT + i)
            floatx16 acc[1 + NT][MT];
            half8_t xb[1 + NT][2 * MT];
            // ---- input features and their tangents -----------------------------------------------------------------------
            {
                floatx16 f[MT], c[NT][MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const half8_t aph = lds_frag(lds, P.offPhase + m * kFragBytes, lane);
                    const floatx16 z = {0};
                    f[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aph, bpos, z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        // unit vector of the axis in both (hi, lo) slots of its input, nothing in the constant slot
                        const int axis = pass * NT + i;
                        const uint4_t u = {axis == 0 ? 0x3c003c00u : 0u, axis == 1 ? 0x3c003c00u : 0u, axis == 2 ? 0x3c003c00u : 0u, 0u};
                        const uint4_t zero = {0u, 0u, 0u, 0u};
                        const half8_t e = __builtin_bit_cast(half8_t, (HAS_DIR && h) ? zero : u);
                        c[i][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aph, e, z, 0, 0, 0);
                    }
                }
                if constexpr (FMODE == FM_FIRST_LAYER) {
                    // scalar first layer: f = W p + b, c = W[:, axis]; activation like a hidden layer
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float d = act_derivative<ACT>(f[m][r], actA, actB);
#pragma unroll
                            for (int i = 0; i < NT; ++i) c[i][m][r] *= d;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            act_pack_quarter<ACT>(f[m], q, actA, actB, xb[0][2 * m], xb[0][2 * m + 1]);
#pragma unroll
                            for (int i = 0; i < NT; ++i) act_pack_quarter<ACT_NONE>(c[i][m], q, 0.f, 0.f, xb[1 + i][2 * m], xb[1 + i][2 * m + 1]);
                        }
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        phase_cos<FMODE>(f[m], m == 0 ? NPASS_ROWS : 0);
                        // (cos u, sin u) with u in revolutions: d/dp = 2 pi c (-sin u, cos u); pass-through rows: c itself (1 on the own axis)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            if (m == 0 && 2 * k < NPASS_ROWS) continue;
#pragma unroll
                            for (int i = 0; i < NT; ++i) {
                                const float c0 = 6.28318530717958647692f * c[i][m][2 * k], c1 = 6.28318530717958647692f * c[i][m][2 * k + 1];
                                c[i][m][2 * k] = -c0 * f[m][2 * k + 1];
                                c[i][m][2 * k + 1] = c1 * f[m][2 * k];
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            act_pack_quarter<ACT_NONE>(f[m], q, 0.f, 0.f, xb[0][2 * m], xb[0][2 * m + 1]);
#pragma unroll
                            for (int i = 0; i < NT; ++i) act_pack_quarter<ACT_NONE>(c[i][m], q, 0.f, 0.f, xb[1 + i][2 * m], xb[1 + i][2 * m + 1]);
                        }
                    }
                }
            }
            // ---- C -> C layers --------------------------------------------------------------------------------------------
            for (int l = 0; l < NL; ++l) {
                const int wOff = l == 0 ? P.offLayer0 : P.offHidden + (l - 1) * MT * KS * kFragBytes;
                const int bOff = P.offBias + l * 32 * MT * 4;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const floatx16 bias = lds_bias(lds, bOff + m * 128, h);
                    const floatx16 z = {0};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const half8_t a = lds_frag(lds, wOff + (m * KS + s) * kFragBytes, lane);
#pragma unroll
                        for (int v = 0; v <= NT; ++v)
                            acc[v][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[v][s], s == 0 ? (v == 0 ? bias : z) : acc[v][m], 0, 0, 0);
                    }
                }
                if constexpr (GRID != 0) {
                    if (l == 0) {
                        for (int g = 0; g < P.gridK; ++g) {
                            float val[8];
                            grid_values8<GRID>(P, tap, g, h, val);
                            half8_t gb[1 + NT];
                            gb[0] = grid_pack(val);
#pragma unroll
                            for (int i = 0; i < NT; ++i) {
                                const int axis = pass * NT + i;
                                float hi[8], lo[8];
                                grid_values8<GRID>(P, grid_tap(P, tq[0] + (axis == 0 ? gridStep : 0.f), tq[1] + (axis == 1 ? gridStep : 0.f),
                                                               tq[2] + (axis == 2 ? gridStep : 0.f)), g, h, hi);
                                grid_values8<GRID>(P, grid_tap(P, tq[0] - (axis == 0 ? gridStep : 0.f), tq[1] - (axis == 1 ? gridStep : 0.f),
                                                               tq[2] - (axis == 2 ? gridStep : 0.f)), g, h, lo);
                                const float s2 = 0.5f / gridStep;
#pragma unroll
                                for (int j = 0; j < 8; ++j) hi[j] = s2 * (hi[j] - lo[j]);
                                gb[1 + i] = grid_pack(hi);
                            }
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                const half8_t a = lds_frag(lds, wOff + (MT * KS + g * MT + m) * kFragBytes, lane);
#pragma unroll
                                for (int v = 0; v <= NT; ++v) acc[v][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, gb[v], acc[v][m], 0, 0, 0);
                            }
                        }
                    }
                }
                // y = act(x), dy = act'(x) dx
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = act_derivative<ACT>(acc[0][m][r], actA, actB);
#pragma unroll
                        for (int i = 0; i < NT; ++i) acc[1 + i][m][r] *= d;
                    }
                    act_pack<ACT>(acc[0][m], actA, actB, xb[0][2 * m], xb[0][2 * m + 1]);
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(acc[1 + i][m], q, 0.f, 0.f, xb[1 + i][2 * m], xb[1 + i][2 * m + 1]);
                }
            }
            // ---- last layer (16x16x32, srn_layers_kmajor): outputs of the tile-(lane>>5) sample land in this lane when t == h ----
            const float4_t biasLast = *reinterpret_cast<const float4_t*>(lds + P.offBias + NL * 32 * MT * 4);
            float4_t o[1 + NT];
            o[0] = biasLast;
#pragma unroll
            for (int i = 0; i < NT; ++i) o[1 + i] = float4_t{0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const half8_t a = lds_frag(lds, P.offLast + s * kFragBytes, lane);
#pragma unroll
                for (int v = 0; v <= NT; ++v) o[v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[v][s], o[v], 0, 0, 0);
            }
            if (t == h) {
                result = o[0];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int axis = pass * NT + i;
                    const float v = o[1 + i][0];
                    if (axis == 0) g0 = v; else if (axis == 1) g1 = v; else g2 = v;  // (no run-time array index: scratch)
                }
            }
        }
    }
    // adjoint of the output parametrization :1225-1238
    const int om = P.outputMode;
    if (om == FVSRN_OUT_DENSITY || om == FVSRN_OUT_DENSITY_GRADIENT || om == FVSRN_OUT_DENSITY_CURVATURE) {
        const float ev = __expf(result[0]);
        const float ds = ev / ((1.f + ev) * (1.f + ev));  // activations::Sigmoid::adjoint(out, 1)
        g0 *= ds; g1 *= ds; g2 *= ds;
    }
    gx = g0; gy = g1; gz = g2;
    return result;
}

}  // namespace fvsrn
