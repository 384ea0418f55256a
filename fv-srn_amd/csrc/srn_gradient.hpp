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
//     coordinates.  Decoded grids with h * resolution <= 1/2 (the default is 1/4): the piecewise-exact form below (own cell's slope +
//     one neighbour cell's, 12 records); otherwise (larger steps, BYTE_GAUSSIAN grids whose decode is not linear) six more fetches in
//     fp32 with hi+lo filter weights (the quotient amplifies rounding by 2 * resolution * 4)
//   * output: density / densitygrad / densitycurvature: times the derivative of the sigmoid at the stored pre-sigmoid value
//     (:1228-1231); the :direct and :cubic modes: un-clamped (:1233-1237).  Colour networks: not supported (:1262), like there.
#pragma once
#include <type_traits>

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
    grid_fetch8<true, true>(P.grid, t, g, h, acc);  // (SAFE: not the hot path, see dot2_from_zero)
    if constexpr (GRID == 2) {  // EncodeGridValue<BYTE_GAUSSIAN> :370-383, as in grid_features
        float accB[8];
        grid_fetch8<true, true>(P.gridB, t, g, h, accB);
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

// ---- central differences of the latent grid, piecewise exact (r03) --------------------------------------------------------------
// The reference differentiates the latent grid by central differences of the trilinear fetch with step h = 1 / (4 resolution)
// (renderer_volume_tensorcores.cuh:609-735): six more fetches per sample.  Along one axis the trilinear interpolant is piecewise
// linear with knots at the texel centres, so with d = h * resolution <= 1/2 texels the points x - d, x + d lie in the sample's own cell
// or in ONE neighbour cell, and
//     g(x + d) - g(x - d) = a * slope(own cell) + b * slope(neighbour cell),   a + b = 2 d,
// with a = 1 - wx + d (x + d crosses the upper face), wx + d (x - d crosses the lower face) or 2 d (no crossing, b = 0).  That is the same
// number the six fetches produce (clamp addressing included: the slope of a cell outside the grid is 0), from the four records of the
// value fetch plus eight more (four x neighbours, two rows y, two rows z) instead of 24, with one tap computation instead of seven, and
// without the hi + lo split of the filter weights: every difference is formed from values interpolated with the SAME rounded weights.
struct GridDiffTap {
    unsigned off[12];  // records: 0-3 own (z0y0, z0y1, z1y0, z1y1), 4-7 their x neighbours, 8-9 row ynew at z0 / z1, 10-11 row znew at y0 / y1
    unsigned wx2;      // packed fp16 {1 - wx, wx}
    float w4[4];       // uz uy, uz wy, wz uy, wz wy
    float xOwn, xNb;   // a_x / 2h, b_x / 2h: weights of the own / the neighbour cell's slope along x
    float uy, wy, uz, wz;
    float cy[3], cz[3];  // coefficients of (row 0, row 1, new row) along y / z, already divided by 2h
};

// `step` = h in unit-box coordinates; the caller guarantees step * resolution <= 1/2 on every axis (grid_differences_are_local)
__device__ __forceinline__ bool grid_differences_are_local(const NetParams& P, float step) {
    return step * P.gridXf <= 0.5f && step * P.gridYf <= 0.5f && step * P.gridZf <= 0.5f;
}

__device__ __forceinline__ GridDiffTap grid_diff_tap(const NetParams& P, float px, float py, float pz, float step) {
    const float fx = fmaf(px, P.gridXf, -0.5f), fy = fmaf(py, P.gridYf, -0.5f), fz = fmaf(pz, P.gridZf, -0.5f);
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    const float wx = fx - x0f, wy = fy - y0f, wz = fz - z0f;
    const float inv2h = 0.5f / step;
    // per axis: own-cell weight a, neighbour weight b, neighbour side
    const float dx = step * P.gridXf, dy = step * P.gridYf, dz = step * P.gridZf;
    const bool xp = wx + dx >= 1.f, xm = wx - dx < 0.f;
    const bool yp = wy + dy >= 1.f, ym = wy - dy < 0.f;
    const bool zp = wz + dz >= 1.f, zm = wz - dz < 0.f;
    const float ax = xp ? 1.f - wx + dx : (xm ? wx + dx : 2.f * dx), bx = 2.f * dx - ax;
    const float ay = yp ? 1.f - wy + dy : (ym ? wy + dy : 2.f * dy), by = 2.f * dy - ay;
    const float az = zp ? 1.f - wz + dz : (zm ? wz + dz : 2.f * dz), bz = 2.f * dz - az;
    // record / row indices (fp32, exact: see grid_tap)
    const float xi = __builtin_amdgcn_fmed3f(x0f + 1.f, 0.f, P.gridXf);
    const float xn = __builtin_amdgcn_fmed3f(x0f + (xp ? 2.f : 0.f), 0.f, P.gridXf);
    const float yMax = P.gridYf - 1.f, zMax = P.gridZf - 1.f;
    const float y0 = __builtin_amdgcn_fmed3f(y0f, 0.f, yMax), y1 = __builtin_amdgcn_fmed3f(y0f + 1.f, 0.f, yMax);
    const float yn = __builtin_amdgcn_fmed3f(y0f + (yp ? 2.f : -1.f), 0.f, yMax);
    const float z0 = __builtin_amdgcn_fmed3f(z0f, 0.f, zMax), z1 = __builtin_amdgcn_fmed3f(z0f + 1.f, 0.f, zMax);
    const float zn = __builtin_amdgcn_fmed3f(z0f + (zp ? 2.f : -1.f), 0.f, zMax);
    const float rowLen = P.gridXf + 1.f;
    const unsigned rec = unsigned(P.gridC) * 4u;
    const float r00 = fmaf(z0, P.gridYf, y0), r01 = fmaf(z0, P.gridYf, y1), r10 = fmaf(z1, P.gridYf, y0), r11 = fmaf(z1, P.gridYf, y1);
    const float ry0 = fmaf(z0, P.gridYf, yn), ry1 = fmaf(z1, P.gridYf, yn), rz0 = fmaf(zn, P.gridYf, y0), rz1 = fmaf(zn, P.gridYf, y1);
    GridDiffTap t;
    t.off[0] = __umul24(unsigned(fmaf(r00, rowLen, xi)), rec);
    t.off[1] = __umul24(unsigned(fmaf(r01, rowLen, xi)), rec);
    t.off[2] = __umul24(unsigned(fmaf(r10, rowLen, xi)), rec);
    t.off[3] = __umul24(unsigned(fmaf(r11, rowLen, xi)), rec);
    t.off[4] = __umul24(unsigned(fmaf(r00, rowLen, xn)), rec);
    t.off[5] = __umul24(unsigned(fmaf(r01, rowLen, xn)), rec);
    t.off[6] = __umul24(unsigned(fmaf(r10, rowLen, xn)), rec);
    t.off[7] = __umul24(unsigned(fmaf(r11, rowLen, xn)), rec);
    t.off[8] = __umul24(unsigned(fmaf(ry0, rowLen, xi)), rec);
    t.off[9] = __umul24(unsigned(fmaf(ry1, rowLen, xi)), rec);
    t.off[10] = __umul24(unsigned(fmaf(rz0, rowLen, xi)), rec);
    t.off[11] = __umul24(unsigned(fmaf(rz1, rowLen, xi)), rec);
    const float ux = 1.f - wx;
    t.uy = 1.f - wy; t.wy = wy; t.uz = 1.f - wz; t.wz = wz;
    t.w4[0] = t.uz * t.uy; t.w4[1] = t.uz * wy; t.w4[2] = wz * t.uy; t.w4[3] = wz * wy;
    {
        const float2_t v = {ux, wx};
        t.wx2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
    }
    t.cy[0] = (-ay + (ym ? by : 0.f)) * inv2h; t.cy[1] = (ay - (yp ? by : 0.f)) * inv2h; t.cy[2] = (yp ? by : -by) * inv2h;
    t.cz[0] = (-az + (zm ? bz : 0.f)) * inv2h; t.cz[1] = (az - (zp ? bz : 0.f)) * inv2h; t.cz[2] = (zp ? bz : -bz) * inv2h;
    t.xOwn = ax * inv2h;  // (the neighbour slope enters with the same sign on either side)
    t.xNb = bx * inv2h;
    return t;
}

// value and d/dx, d/dy, d/dz (central differences, unit-box coordinates) of the 8 channels [16 g + 8 h, +8) of the working grid
__device__ __forceinline__ void grid_value_and_differences8(const NetParams& P, const GridDiffTap& t, int g, int h, float (&val)[8],
                                                            float (&dX)[8], float (&dY)[8], float (&dZ)[8]) {
    const char* base = reinterpret_cast<const char*>(P.grid) + g * 64;
    const unsigned hoff = unsigned(h) * 32u;
    uint4_t r[12][2];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint4_t* p = reinterpret_cast<const uint4_t*>(base + (t.off[k] + hoff));
        r[k][0] = p[0];
        r[k][1] = p[1];
    }
    // NB the order of this function.  dot2_from_zero is inline assembly (it saves the v_mov that zeroes the accumulator of the
    // compiler's v_dot2c form), and hipcc's hazard recognizer does not look into inline assembly: on gfx950 a VALU instruction that
    // reads the result of a DOT instruction of another opcode needs three wait states, which the compiler only inserts (s_nop 2) for
    // DOT instructions it emitted itself.  (Measured: with the fp32 combinations right behind their dot products three of eight
    // channels came out wrong.)  So all the dot products from zero are issued first, a scheduling barrier pins that, and the first
    // consumer behind the barrier reads a result that is 60 instructions old.
    const float A = t.xOwn, B = t.xNb;
    float R[4][8], Ry0[8], Ry1[8], Rz0[8], Rz1[8];
    unsigned pa[4], pb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // d/dx weights {-w, +w}: one rounding for both signs
        const float wa = A * t.w4[k], wb = B * t.w4[k];
        const float2_t va = {-wa, wa}, vb = {-wb, wb};
        pa[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(va, half2_t));
        pb[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(vb, half2_t));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int w = j >> 2, e = j & 3;
        const unsigned o0 = r[0][w][e];
        dX[j] = dot2_from_zero(o0, pa[0]);
    }
    // rows interpolated along x with the one weight pair {1 - wx, wx}
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int w = j >> 2, e = j & 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const unsigned u = r[k][w][e]; R[k][j] = dot2_from_zero(u, t.wx2); }
        const unsigned uy0 = r[8][w][e], uy1 = r[9][w][e], uz0 = r[10][w][e], uz1 = r[11][w][e];
        Ry0[j] = dot2_from_zero(uy0, t.wx2); Ry1[j] = dot2_from_zero(uy1, t.wx2);
        Rz0[j] = dot2_from_zero(uz0, t.wx2); Rz1[j] = dot2_from_zero(uz1, t.wx2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // d/dx: slopes of the own and the neighbour cell
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const half2_t ha = __builtin_bit_cast(half2_t, pa[k]), hb = __builtin_bit_cast(half2_t, pb[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = j >> 2, e = j & 3;
            const unsigned o0 = r[k][w][e], n0 = r[4 + k][w][e];
            if (k > 0) dX[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, o0), ha, dX[j], false);
            dX[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, n0), hb, dX[j], false);
        }
    }
    // value and d/dy, d/dz: fp32 combinations of the rows
    const float cv[4] = {t.w4[0], t.w4[1], t.w4[2], t.w4[3]};
    const float cyk[4] = {t.uz * t.cy[0], t.uz * t.cy[1], t.wz * t.cy[0], t.wz * t.cy[1]};
    const float czk[4] = {t.uy * t.cz[0], t.wy * t.cz[0], t.uy * t.cz[1], t.wy * t.cz[1]};
    const float cyn[2] = {t.uz * t.cy[2], t.wz * t.cy[2]};
    const float czn[2] = {t.uy * t.cz[2], t.wy * t.cz[2]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        val[j] = fmaf(cv[3], R[3][j], fmaf(cv[2], R[2][j], fmaf(cv[1], R[1][j], cv[0] * R[0][j])));
        dY[j] = fmaf(cyn[1], Ry1[j], fmaf(cyn[0], Ry0[j], fmaf(cyk[3], R[3][j], fmaf(cyk[2], R[2][j], fmaf(cyk[1], R[1][j], cyk[0] * R[0][j])))));
        dZ[j] = fmaf(czn[1], Rz1[j], fmaf(czn[0], Rz0[j], fmaf(czk[3], R[3][j], fmaf(czk[2], R[2][j], fmaf(czk[1], R[1][j], czk[0] * R[0][j])))));
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
    // (wave-uniform) the +-h points of the grid's central differences stay within one cell of the sample: piecewise-exact form
    const bool localDiff = GRID == 1 && grid_differences_are_local(P, gridStep);

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

    // ---- latent-grid chunk g: value and central differences of 16 channels as B fragments of the layer-0 grid steps --------
    auto gridChunkAt = [&](const float (&tq)[3], int pass, int g, half8_t (&gb)[1 + NT]) {
        float val[8], dd[3][8];
        if (GRID == 1 && localDiff) {
            const GridDiffTap dtap = grid_diff_tap(P, tq[0], tq[1], tq[2], gridStep);
            grid_value_and_differences8(P, dtap, g, h, val, dd[0], dd[1], dd[2]);
        } else {
            // six more fetches; a rolled loop (one copy of the fetch, few registers): this path is the rare one and must not
            // set the register budget of the kernel
            grid_values8<GRID>(P, grid_tap(P, tq[0], tq[1], tq[2]), g, h, val);
#pragma unroll
            for (int j = 0; j < 8; ++j) dd[0][j] = dd[1][j] = dd[2][j] = 0.f;
            const float s2 = 0.5f / gridStep;
#pragma unroll 1
            for (int k = 0; k < 2 * NT; ++k) {
                const int axis = pass * NT + (k >> 1);
                const float sg = (k & 1) ? -gridStep : gridStep, w = (k & 1) ? -s2 : s2;
                float v[8];
                grid_values8<GRID>(P, grid_tap(P, tq[0] + (axis == 0 ? sg : 0.f), tq[1] + (axis == 1 ? sg : 0.f), tq[2] + (axis == 2 ? sg : 0.f)), g, h, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // (no run-time array index: scratch)
                    dd[0][j] = fmaf(axis == 0 ? w : 0.f, v[j], dd[0][j]);
                    dd[1][j] = fmaf(axis == 1 ? w : 0.f, v[j], dd[1][j]);
                    dd[2][j] = fmaf(axis == 2 ? w : 0.f, v[j], dd[2][j]);
                }
            }
        }
        gb[0] = grid_pack(val);
        if constexpr (NT == 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) gb[1 + i] = grid_pack(dd[i]);
        } else {  // one tangent per pass
            float d1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d1[j] = pass == 0 ? dd[0][j] : (pass == 1 ? dd[1][j] : dd[2][j]);
            gb[1] = grid_pack(d1);
        }
    };
    // With one pass of three tangents the latent-grid fragments of BOTH tiles (chunk 0) are fetched before the tile loop: the gradient pass runs
    // with one wave per SIMD where a latent grid is present (render_adjoint_kernel), nothing else hides the 24 gathers of a tile, and two
    // tiles' gathers in flight at once cost one latency instead of two (2 x 16 registers across the first tile).
    constexpr bool kChunksAhead = GRID != 0 && NT == 3;
    half8_t gbTile0[1 + NT], gbTile1[1 + NT];
    if constexpr (kChunksAhead) {
        const float q0[3] = {tp[0][0], tp[0][1], tp[0][2]}, q1[3] = {tp[1][0], tp[1][1], tp[1][2]};
        gridChunkAt(q0, 0, 0, gbTile0);
        gridChunkAt(q1, 0, 0, gbTile1);
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

#pragma unroll 1
        for (int pass = 0; pass < 3 / NT; ++pass) {
            // column sets: 0 = value, 1 + i = tangent of axis (pass * NT + i)
            half8_t xb[1 + NT][2 * MT];
            // chunk 0 is fetched before anything else of the pass is live (12 records = 96 registers in the piecewise-exact form); its four
            // fragments (16 registers) wait for layer 0
            half8_t gb0[1 + NT];
            if constexpr (kChunksAhead) {
#pragma unroll
                for (int v = 0; v <= NT; ++v) {  // (selects, no run-time array index)
                    const uint4_t a = __builtin_bit_cast(uint4_t, gbTile0[v]), b = __builtin_bit_cast(uint4_t, gbTile1[v]);
                    const uint4_t c = {t ? b[0] : a[0], t ? b[1] : a[1], t ? b[2] : a[2], t ? b[3] : a[3]};
                    gb0[v] = __builtin_bit_cast(half8_t, c);
                }
            } else if constexpr (GRID != 0) {
                gridChunkAt(tq, pass, 0, gb0);
            }
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
            // The (1 + NT) column sets go through a layer in PAIRS -- (value, tangent 0), then (tangent 1, tangent 2) -- so that a weight
            // fragment read from LDS feeds two MFMAs while only 2 x MT accumulator tiles are live (all four at once: 128 accumulator
            // registers at 64 channels, and the shaded renderer spilled 130 registers per wave step around them, r03 PMC:
            // profiles/r03/shaded_adjoint_pmc.md).  A tangent only needs its own inputs and act'(x) of the value column, which the
            // first pair leaves behind in fp16 (dh: 8 x MT registers); outputs replace their inputs in xb.
            constexpr int NP = (1 + NT) / 2;
            half8_t dh[MT][2];
            auto layer = [&](int l, auto withGrid) {
                constexpr bool GRID0 = decltype(withGrid)::value;
                const int wOff = l == 0 ? P.offLayer0 : P.offHidden + (l - 1) * MT * KS * kFragBytes;
                const int bOff = P.offBias + l * 32 * MT * 4;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    floatx16 acc[2][MT];
                    const floatx16 z = {0};
                    if constexpr (GRID0) {  // bias + latent-grid steps first
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const half8_t a = lds_frag(lds, wOff + (MT * KS + m) * kFragBytes, lane);
                            acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, gb0[2 * p], p == 0 ? lds_bias(lds, bOff + m * 128, h) : z, 0, 0, 0);
                            acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, gb0[2 * p + 1], z, 0, 0, 0);
                        }
                        for (int g = 1; g < P.gridK; ++g) {  // (more than 16 latent channels: fetched again for the second pair)
                            half8_t gb[1 + NT];
                            gridChunkAt(tq, pass, g, gb);
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                const half8_t a = lds_frag(lds, wOff + (MT * KS + g * MT + m) * kFragBytes, lane);
                                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, gb[2 * p], acc[0][m], 0, 0, 0);
                                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, gb[2 * p + 1], acc[1][m], 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            const half8_t a = lds_frag(lds, wOff + (m * KS + s) * kFragBytes, lane);
                            if (s == 0 && !GRID0) {
                                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[2 * p][s], p == 0 ? lds_bias(lds, bOff + m * 128, h) : z, 0, 0, 0);
                                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[2 * p + 1][s], z, 0, 0, 0);
                            } else {
                                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[2 * p][s], acc[0][m], 0, 0, 0);
                                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, xb[2 * p + 1][s], acc[1][m], 0, 0, 0);
                            }
                        }
                    }
                    // y = act(x), dy = act'(x) dx.  The tangents are converted to fp16 first and scaled there (v_pk_mul_f16 with act'(x) as
                    // packed fp16: one instruction per two values instead of two multiplies and a convert back): the product is rounded to
                    // fp16 either way, and act'(x) of a ReLU is exactly 0 or 1.
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (p == 0) {
                            act_pack<ACT>(acc[0][m], actA, actB, xb[0][2 * m], xb[0][2 * m + 1]);
                            if constexpr (ACT == ACT_RELU) {
                                // act'(x) = [x > 0] = [y != 0] from the packed outputs: 1.0h where the half is non-zero (two packed integer
                                // instructions per pair instead of two compares, two selects and a convert)
                                // (inline assembly: hipcc turns the vector form into 16 compares, 16 selects and 8 packs.  Plain VALU
                                // instructions on plain VALU results: none of the hazard classes of dot2_from_zero)
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const uint4_t y = __builtin_bit_cast(uint4_t, xb[0][2 * m + e]);
                                    uint4_t d;
#pragma unroll
                                    for (int w = 0; w < 4; ++w) {
                                        const unsigned yw = y[w];
                                        unsigned t, dw;
                                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(yw), "s"(0x00010001u));
                                        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(dw) : "v"(t), "s"(0x3c003c00u));
                                        d[w] = dw;
                                    }
                                    dh[m][e] = __builtin_bit_cast(half8_t, d);
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; r += 2) {
                                    const float2_t dv = {act_derivative<ACT>(acc[0][m][r], actA, actB), act_derivative<ACT>(acc[0][m][r + 1], actA, actB)};
                                    const half2_t dp = __builtin_convertvector(dv, half2_t);
                                    dh[m][r >> 3][r & 7] = dp[0];
                                    dh[m][r >> 3][(r & 7) + 1] = dp[1];
                                }
                            }
                        } else {
                            half8_t t0, t1;
#pragma unroll
                            for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(acc[0][m], q, 0.f, 0.f, t0, t1);
                            xb[2 * p][2 * m] = t0 * dh[m][0];
                            xb[2 * p][2 * m + 1] = t1 * dh[m][1];
                        }
                        half8_t t0, t1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) act_pack_quarter<ACT_NONE>(acc[1][m], q, 0.f, 0.f, t0, t1);
                        xb[2 * p + 1][2 * m] = t0 * dh[m][0];
                        xb[2 * p + 1][2 * m + 1] = t1 * dh[m][1];
                    }
                }
            };
            if constexpr (GRID != 0) layer(0, std::true_type{});
            else if (NL > 0) layer(0, std::false_type{});  // (networks without Fourier features may have no C -> C layer at all)
            for (int l = 1; l < NL; ++l) layer(l, std::false_type{});
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
