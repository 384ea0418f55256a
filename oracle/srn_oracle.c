/*
 * srn_oracle.c -- CPU restatement of the fV-SRN hot path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this file's
 * library.  Nothing in the product path (fv-srn_amd/) links, imports or calls it.
 *
 * It restates, in scalar C, the algorithm of the reference CUDA path (paths relative to the
 * reference checkout, shamanDevel/fV-SRN):
 *   network     renderer/renderer_volume_tensorcores.cuh:735-1164   (eval)
 *   activations renderer/renderer_activations.cuh
 *   latent grid renderer/renderer_volume_tensorcores.cuh:572-605 + tex3D semantics set up in
 *               renderer/volume_interpolation_network.cpp:470-504 (normalized, clamp, linear)
 *   camera      renderer/renderer_camera.cuh:33-52
 *   box test    renderer/renderer_utils.cuh:91-105
 *   DVR loop    renderer/renderer_ray_evaluation_stepping_dvr.cuh:48-157
 *   TFs         renderer/renderer_tf_{identity,gaussian,piecewise,texture}.cuh
 *   blending    renderer/renderer_blending.cuh:35-51
 *   image       renderer/renderer_image_evaluator_simple.cuh:53-125
 *
 * Inputs are the arrays of the reference's __constant__ block
 * (kernel::VolumeInterpolationTensorcoresParameters, renderer_volume_tensorcores.cuh:195-249),
 * i.e. the half matrices exactly as SceneNetwork::fillConstantMemory lays them out.
 *
 * Two arithmetic models (ORACLE_ACC_*):
 *   HALF  : the reference's: half storage AND half accumulation (WMMA accumulator fragments of
 *           type half, bias pre-loaded, :849-858/:965-972), half activations, half Fourier chain,
 *           sequential hfma last layer.  A WMMA k=16 step is modelled as one exact dot product
 *           added to the accumulator and rounded to half once.
 *   FLOAT : the reference's network with the accumulation the MI355X kernels use: same half-quantised
 *           inputs/weights/activations, but fp32 accumulation over the whole K, fp32 activations, exact
 *           Fourier phases (from the fp16 position of every sample, like the reference).
 *   DEVICE: FLOAT plus a statement of what the HIP kernels do in the Fourier stage (fv-srn_amd/csrc/pack.cpp,
 *           srn_device.hpp; no reference counterpart, this is the kernels' own arithmetic): phases in
 *           revolutions as an fp32 sum of exact fp16 x fp16 products of the position with a hi + lo split
 *           of the matrix (2^-22 relative) and a whole-revolution centring constant; and, in the renderer
 *           for 32-wide Fourier-only networks (OracleScene::rotationResync), features that are derived
 *           from the fp16 position only every rotationResync steps and advanced in between by rotating
 *           (cos, sin) with the per-step phase increment -- i.e. they follow the UN-rounded ray there.
 *           The parity tests hold the HIP path against this model with one absolute tolerance, and this
 *           model against FLOAT / HALF as the reference-side bars.
 *   EXACT : the network itself: the stored fp16 weights, everything else in fp32 / double -- positions, Fourier
 *           features, latent features and activations are NOT rounded to half (forward path only).  This is what
 *           the reference's PyTorch model computes in fp32; it ranks the other three models.
 * The pinning of this file against the reference's own Python implementation lives in
 * tests/test_oracle_golden.py (golden vectors made by tests/golden/make_golden.py).
 */
#include "srn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ half */
static uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (x - 0x38000000u) >> 13;
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}

static float h2f(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int k = 0;
            while (!(m & 0x400u)) { m <<= 1; ++k; }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(113 - k) << 23) | (m << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* round a double to the nearest half (ties to even) in ONE rounding, result as float */
static float rh_d(double v) {
    if (v != v || v == 0.0) return (float)v;
    const double a = fabs(v);
    if (a >= 65520.0) return v < 0 ? -INFINITY : INFINITY;
    int e;
    (void)frexp(a, &e); /* a = m * 2^e, m in [0.5,1) */
    const int ulp_exp = (e - 1 < -14) ? -24 : (e - 1 - 10);
    const double r = nearbyint(ldexp(a, -ulp_exp)); /* default rounding mode: nearest even */
    const double q = ldexp(r, ulp_exp);
    return (float)(v < 0 ? -q : q);
}
static float rh(float v) { return h2f(f2h(v)); }
/* half intrinsics of the reference (cuda_fp16): one rounding per operation */
static float hmul(float a, float b) { return rh(a * b); /* product of two halves is exact in fp32 */ }
static float hadd(float a, float b) { return rh_d((double)a + (double)b); }
static float hsub(float a, float b) { return rh_d((double)a - (double)b); }
static float hfma(float a, float b, float c) { return rh_d((double)a * (double)b + (double)c); }
static float hdivf(float a, float b) { return rh_d((double)a / (double)b); }
static float hcosf(float a) { return rh_d(cos((double)a)); }
static float hsinf(float a) { return rh_d(sin((double)a)); }
/* the fp16 roundings of inputs, features and activations that FLOAT / DEVICE share with the reference; EXACT leaves them out */
static float rq(const OracleNet* n, float v) { return n->accMode == ORACLE_ACC_EXACT ? v : rh(v); }
static float rq_d(const OracleNet* n, double v) { return n->accMode == ORACLE_ACC_EXACT ? (float)v : rh_d(v); }

/* ------------------------------------------------------------------------------------ activations */
/* half versions: renderer_activations.cuh (ReLU :40-60, Sine :97-120, Sigmoid :152-179, Snake :263-285, SnakeAlt :329-358) */
static float act_half(int act, float v, float param) {
    switch (act) {
        case ORACLE_ACT_RELU: return v > 0.f ? v : 0.f;
        case ORACLE_ACT_SINE: return hsinf(hmul(v, rh(param)));
        case ORACLE_ACT_SNAKE: {
            const float f = rh(param), divf = rh(1.0f / param);
            const float v2 = hsinf(hmul(f, v));
            return hadd(v, hmul(divf, hmul(v2, v2)));
        }
        case ORACLE_ACT_SNAKEALT: {
            const float f2 = rh(2 * param);
            const float x0 = hcosf(hmul(f2, v));
            const float x1 = hsub(hadd(v, 1.0f), x0);
            return hdivf(x1, f2);
        }
        case ORACLE_ACT_SIGMOID: /* :152-160: __hdiv(ONE, __hadd(ONE, hexp(__hneg(v)))) */
            return hdivf(1.0f, hadd(1.0f, rh_d(exp(-(double)v))));
        default: return v;
    }
}
/* fp32 versions: the double-precision overloads of the same file evaluated in fp32 */
static float act_float(int act, float v, float param) {
    switch (act) {
        case ORACLE_ACT_RELU: return v > 0.f ? v : 0.f;
        case ORACLE_ACT_SINE: return (float)sin((double)v * param);
        case ORACLE_ACT_SNAKE: { const double s = sin((double)param * v); return (float)(v + s * s / param); }
        case ORACLE_ACT_SNAKEALT: return (float)(((double)v + 1.0 - cos(2.0 * param * v)) / (2.0 * param));
        case ORACLE_ACT_SIGMOID: return (float)(1.0 / (1.0 + exp(-(double)v)));
        default: return v;
    }
}

/* ------------------------------------------------------------------------------------ latent grid */
/* Accurate single-precision erfinv (Giles 2010); the reference calls CUDA's erfinvf (:374). */
static float erfinv_f(float x) {
    float w = -logf((1.0f - x) * (1.0f + x)), p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w; p = 0.00021858087f + p * w; p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w; p = 0.246640727f + p * w; p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w; p = 0.00573950773f + p * w; p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w; p = 1.00167406f + p * w; p = 2.83297682f + p * w;
    }
    return p * x;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* tex3D<float4> with normalized coordinates, clamp addressing, linear filtering; byte textures are
 * read as normalized floats.  One texture = 4 channels, layout [Z][Y][X][4]. */
static void tex3d(const OracleNet* n, const void* tex, float px, float py, float pz, float out[4]) {
    const int X = n->gridX, Y = n->gridY, Z = n->gridZ;
    const float fx = px * (float)X - 0.5f, fy = py * (float)Y - 0.5f, fz = pz * (float)Z - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    const float wx = fx - x0f, wy = fy - y0f, wz = fz - z0f;
    const int xs[2] = {clampi((int)x0f, 0, X - 1), clampi((int)x0f + 1, 0, X - 1)};
    const int ys[2] = {clampi((int)y0f, 0, Y - 1), clampi((int)y0f + 1, 0, Y - 1)};
    const int zs[2] = {clampi((int)z0f, 0, Z - 1), clampi((int)z0f + 1, 0, Z - 1)};
    for (int c = 0; c < 4; ++c) out[c] = 0.f;
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const float w = (dz ? wz : 1.f - wz) * (dy ? wy : 1.f - wy) * (dx ? wx : 1.f - wx);
                const size_t o = (((size_t)zs[dz] * Y + ys[dy]) * X + xs[dx]) * 4;
                for (int c = 0; c < 4; ++c) {
                    const float v = n->gridEncoding == ORACLE_GRID_FLOAT ? ((const float*)tex)[o + c]
                                                                       : ((const uint8_t*)tex)[o + c] / 255.0f;
                    out[c] += w * v;
                }
            }
}

/* EncodeGridValue<enc, false> -- note the reference decodes grid B with A's coefficients (:586-587) */
static float decode_grid(const OracleNet* n, float v, int channel) {
    if (n->gridEncoding == ORACLE_GRID_FLOAT) return v;
    if (n->gridEncoding == ORACLE_GRID_BYTE_GAUSSIAN)
        v = 1.4142135623730950488f * erfinv_f((2.0f - 1e-4f) * (v - 0.5f));
    return n->gridOffsetA[channel] + v * n->gridScaleA[channel];
}

/* LoadVolumetricFeatures :572-605: 16 channels starting at 16*chunk, as half values */
static void grid_features(const OracleNet* n, const float p[3], int chunk, float out16[16]) {
    for (int i = 0; i < 4; ++i) {
        const int t = 4 * chunk + i;
        float a[4], b[4];
        tex3d(n, n->gridTexA[t], p[0], p[1], p[2], a);
        tex3d(n, n->gridTexB[t], p[0], p[1], p[2], b);
        const float time = n->gridInterpolation[t];
        const float f = time - floorf(time);
        for (int c = 0; c < 4; ++c) {
            const float va = decode_grid(n, a[c], 4 * t + c), vb = decode_grid(n, b[c], 4 * t + c);
            out16[4 * i + c] = rq(n, va + f * (vb - va));
        }
    }
}

/* ---------------------------------------------------------------------------------------- network */
#define ORACLE_MAX_C 256

/* one WMMA-style layer: y = act(W x + b), W row-major [cout][cin]; pre (optional): the pre-activations, as the adjoint pass
 * finds them (HALF: the half accumulator, sStorageForAdjointHidden :1000-1010; FLOAT: the fp32 sum) */
static void dense_layer_tape(const OracleNet* n, const uint16_t* W, const uint16_t* b, int cin, int cout, const float* x,
                             float* y, float* pre) {
    for (int o = 0; o < cout; ++o) {
        const uint16_t* w = W + (size_t)o * cin;
        float r;
        if (n->accMode == ORACLE_ACC_HALF) {
            float acc = h2f(b[o]);
            for (int k0 = 0; k0 < cin; k0 += 16) {
                double s = 0;
                for (int k = k0; k < k0 + 16 && k < cin; ++k) s += (double)h2f(w[k]) * (double)x[k];
                acc = rh_d((double)acc + s);
            }
            if (pre) pre[o] = acc;
            r = act_half(n->activation, acc, n->actParam);
        } else {
            double s = h2f(b[o]);
            for (int k = 0; k < cin; ++k) s += (double)h2f(w[k]) * (double)x[k];
            if (pre) pre[o] = (float)s;
            r = rq(n, act_float(n->activation, (float)s, n->actParam));
        }
        y[o] = r;
    }
}
static void dense_layer(const OracleNet* n, const uint16_t* W, const uint16_t* b, int cin, int cout, const float* x, float* y) {
    dense_layer_tape(n, W, b, cin, cout, x, y, NULL);
}

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static float softplusf_(float x) { return x > 20.f ? x : logf(1.f + expf(x)); }

/* ORACLE_ACC_DEVICE: phase of Fourier feature i in REVOLUTIONS the way the kernels' phase MFMA computes it (pack.cpp: matrix entry
 * / 2 pi split into two halves hi + lo, the constant slot carries minus the integer nearest to the centre of the feature's phase
 * range over the unit box / directions in [-1,1]; srn_device.hpp: B operand = the fp16 inputs; exact products, fp32 sum).
 * in6: fp16 values (as floats) of x, y, z, dx, dy, dz -- a position or, for the per-step rotation, a position increment */
static float device_phase_rev(const OracleNet* n, int i, const float in6[6]) {
    const int F = n->F, cols = n->useDirection == 2 ? 6 : 3;
    double lo = 0, hi = 0, sum = 0;
    for (int c = 0; c < cols; ++c) {
        const double v = (double)h2f(n->fourier[i + (size_t)F * c]) / (2.0 * 3.14159265358979323846);
        const float vh = rh((float)v);
        const float vl = rh((float)(v - (double)vh));
        sum += (double)vh * in6[c] + (double)vl * in6[c];
        if (c < 3) { lo += v < 0 ? v : 0; hi += v > 0 ? v : 0; }
        else { lo -= fabs(v); hi += fabs(v); }
    }
    const double konst = -nearbyint(0.5 * (lo + hi));
    const float kh = rh((float)konst);
    const float kl = rh((float)(konst - (double)kh));
    return (float)(sum + (double)kh + (double)kl);
}

/* the linear part of device_phase_rev alone (no constant slot): the contribution of a second operand, e.g. the fp16 residual of a position */
static double device_phase_lin(const OracleNet* n, int i, const float in6[6]) {
    const int F = n->F, cols = n->useDirection == 2 ? 6 : 3;
    double sum = 0;
    for (int c = 0; c < cols; ++c) {
        const double v = (double)h2f(n->fourier[i + (size_t)F * c]) / (2.0 * 3.14159265358979323846);
        const float vh = rh((float)v);
        const float vl = rh((float)(v - (double)vh));
        sum += (double)vh * in6[c] + (double)vl * in6[c];
    }
    return sum;
}

/* The padded input vector x[0 .. C) of the first layer: [x, y, z, (time | 0), (dx, dy, dz, 0), cos(0..F-1), sin(0..F-1)] as fp16
 * values, from the normalized position p (renderer_volume_tensorcores.cuh:768-808) */
static void fourier_inputs(const OracleNet* n, const float p[3], const float dir[3], float* x) {
    const int F = n->F;
    const int base = n->useDirection >= 1 ? 8 : 4; /* fourierOffset :794 */
    const float vx = rq(n, p[0]), vy = rq(n, p[1]), vz = rq(n, p[2]);
    x[0] = vx; x[1] = vy; x[2] = vz;
    x[3] = n->passTime ? rq(n, n->gridInterpolation[0]) : 0.f;
    float dxh = 0, dyh = 0, dzh = 0;
    if (n->useDirection >= 1) { /* :784-792 */
        dxh = rq(n, dir[0]); dyh = rq(n, dir[1]); dzh = rq(n, dir[2]);
        x[4] = dxh; x[5] = dyh; x[6] = dzh; x[7] = 0.f;
    }
    const float in6[6] = {vx, vy, vz, dxh, dyh, dzh};
    for (int i = 0; i < F; ++i) {
        const float f0 = h2f(n->fourier[i]), f1 = h2f(n->fourier[i + F]), f2 = h2f(n->fourier[i + 2 * F]);
        const int d6 = n->useDirection == 2; /* USE_DIRECTION==2: direction inside the Fourier matrix :800-804 */
        const float f3 = d6 ? h2f(n->fourier[i + 3 * F]) : 0.f, f4 = d6 ? h2f(n->fourier[i + 4 * F]) : 0.f,
                    f5 = d6 ? h2f(n->fourier[i + 5 * F]) : 0.f;
        if (n->accMode == ORACLE_ACC_HALF) {
            float c = hmul(vx, f0);
            c = hfma(vy, f1, c);
            c = hfma(vz, f2, c);
            if (d6) { c = hfma(dxh, f3, c); c = hfma(dyh, f4, c); c = hfma(dzh, f5, c); }
            x[base + i] = hcosf(c);
            x[base + F + i] = hsinf(c);
        } else if (n->accMode == ORACLE_ACC_DEVICE) {
            const double c = 2.0 * 3.14159265358979323846 * (double)device_phase_rev(n, i, in6);
            x[base + i] = rh((float)cos(c));      /* v_cos_f32 / v_sin_f32 of the fp32 phase, then v_cvt_pk_f16_f32 */
            x[base + F + i] = rh((float)sin(c));
        } else {
            double c = (double)vx * f0 + (double)vy * f1 + (double)vz * f2;
            if (d6) c += (double)dxh * f3 + (double)dyh * f4 + (double)dzh * f5;
            x[base + i] = rq_d(n, cos(c));
            x[base + F + i] = rq_d(n, sin(c));
        }
    }
}

/* eval<>: world position (+ view direction) -> out[0..3] = value (1 or 4 channels used), nrm[3] = predicted normal
 * xFeat != NULL: the first layer's padded input vector (fourier_inputs) is given; pNorm != NULL: the normalized position is given */
static void srn_eval_x(const OracleNet* n, const float wpos[3], const float dir[3], const float* xFeat, const float* pNorm, float out[4], float nrm[3],
                       float curv[2]) {
    const int C = n->C, F = n->F, G = n->G;
    float p[3];
    for (int i = 0; i < 3; ++i) p[i] = pNorm ? pNorm[i] : (wpos[i] - n->boxMin[i]) / n->boxSize[i]; /* :746; pNorm: the renderer's DEVICE model */
    float x[ORACLE_MAX_C + 64], y[ORACLE_MAX_C];
    /* Fourier layer :768-808 */
    const float vx = rq(n, p[0]), vy = rq(n, p[1]), vz = rq(n, p[2]);
    float dxh = 0, dyh = 0, dzh = 0;
    if (n->useDirection >= 1) { dxh = rq(n, dir[0]); dyh = rq(n, dir[1]); dzh = rq(n, dir[2]); } /* :784-792 */
    if (xFeat) memcpy(x, xFeat, sizeof(float) * (size_t)C); /* ORACLE_ACC_DEVICE: rotated features of the renderer (render_pixel) */
    else fourier_inputs(n, p, dir, x);
    if (F == 0) { /* no Fourier features: scalar first layer :810-823, weights [cin][cout], half fma chain */
        for (int co = 0; co < C; ++co) {
            float r;
            if (n->accMode == ORACLE_ACC_HALF) {
                float c = h2f(n->bFirst[co]);
                c = hfma(vx, h2f(n->wFirst[co + C * 0]), c);
                c = hfma(vy, h2f(n->wFirst[co + C * 1]), c);
                c = hfma(vz, h2f(n->wFirst[co + C * 2]), c);
                if (n->useDirection >= 1) {
                    c = hfma(dxh, h2f(n->wFirst[co + C * 3]), c);
                    c = hfma(dyh, h2f(n->wFirst[co + C * 4]), c);
                    c = hfma(dzh, h2f(n->wFirst[co + C * 5]), c);
                }
                r = act_half(n->activation, c, n->actParam);
            } else {
                double c = h2f(n->bFirst[co]);
                c += (double)vx * h2f(n->wFirst[co + C * 0]) + (double)vy * h2f(n->wFirst[co + C * 1]) + (double)vz * h2f(n->wFirst[co + C * 2]);
                if (n->useDirection >= 1)
                    c += (double)dxh * h2f(n->wFirst[co + C * 3]) + (double)dyh * h2f(n->wFirst[co + C * 4]) + (double)dzh * h2f(n->wFirst[co + C * 5]);
                r = rq(n, act_float(n->activation, (float)c, n->actParam));
            }
            y[co] = r;
        }
        memcpy(x, y, sizeof(float) * (size_t)C);
    }
    /* latent grid layer :839-948 */
    if (G > 0) {
        for (int g = 0; g < G / 16; ++g) grid_features(n, p, g, x + C + 16 * g);
        dense_layer(n, n->wFirst, n->bFirst, C + G, C, x, y);
        memcpy(x, y, sizeof(float) * (size_t)C);
    }
    /* hidden layers :956-1033 */
    for (int l = 0; l < n->NH; ++l) {
        dense_layer(n, n->wHidden + (size_t)l * C * C, n->bHidden + (size_t)l * C, C, C, x, y);
        memcpy(x, y, sizeof(float) * (size_t)C);
    }
    /* last layer :1045-1053, :1138-1143; weights stored [cin][cout] */
    const int Co = n->Cout;
    float o[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < Co; ++c) {
        if (n->accMode == ORACLE_ACC_HALF) {
            float acc = h2f(n->bLast[c]);
            for (int k = 0; k < C; ++k) acc = hfma(x[k], h2f(n->wLast[(size_t)k * Co + c]), acc);
            o[c] = acc;
        } else {
            double s = h2f(n->bLast[c]);
            for (int k = 0; k < C; ++k) s += (double)x[k] * (double)h2f(n->wLast[(size_t)k * Co + c]);
            o[c] = (float)s;
        }
    }
    nrm[0] = nrm[1] = nrm[2] = 0.f;
    curv[0] = curv[1] = 0.f;
    out[0] = out[1] = out[2] = out[3] = 0.f;
    switch (n->outputMode) { /* :1054-1158 */
        case ORACLE_OUT_DENSITY: out[0] = sigmoidf_(o[0]); break;
        case ORACLE_OUT_DENSITY_DIRECT: out[0] = o[0]; break;
        case ORACLE_OUT_RGBO:
            out[0] = sigmoidf_(o[0]); out[1] = sigmoidf_(o[1]); out[2] = sigmoidf_(o[2]); out[3] = softplusf_(o[3]);
            break;
        case ORACLE_OUT_RGBO_DIRECT:
            for (int c = 0; c < 3; ++c) out[c] = fminf(fmaxf(o[c], 0.f), 1.f);
            out[3] = fmaxf(o[3], 0.f);
            break;
        case ORACLE_OUT_DENSITY_GRADIENT: out[0] = sigmoidf_(o[0]); nrm[0] = o[1]; nrm[1] = o[2]; nrm[2] = o[3]; break;
        case ORACLE_OUT_DENSITY_GRADIENT_DIRECT: out[0] = o[0]; nrm[0] = o[1]; nrm[1] = o[2]; nrm[2] = o[3]; break;
        case ORACLE_OUT_DENSITY_GRADIENT_CUBIC:
            out[0] = o[0]; nrm[0] = o[1] * o[1] * o[1]; nrm[1] = o[2] * o[2] * o[2]; nrm[2] = o[3] * o[3] * o[3];
            break;
        case ORACLE_OUT_DENSITY_CURVATURE: /* :1117-1123 */
            out[0] = sigmoidf_(o[0]); nrm[0] = o[1]; nrm[1] = o[2]; nrm[2] = o[3]; curv[0] = o[4]; curv[1] = o[5];
            break;
        case ORACLE_OUT_DENSITY_CURVATURE_DIRECT: /* :1124-1132 */
            out[0] = o[0]; nrm[0] = o[1]; nrm[1] = o[2]; nrm[2] = o[3]; curv[0] = o[4]; curv[1] = o[5];
            break;
        default: break;
    }
}

static void srn_eval(const OracleNet* n, const float wpos[3], const float dir[3], float out[4], float nrm[3], float curv[2]) {
    srn_eval_x(n, wpos, dir, NULL, NULL, out, nrm, curv);
}

/* ------------------------------------------------------------------------------------ adjoint method */
/* activations::*::adjoint (renderer_activations.cuh): vAdj = act'(v) * zAdj.  HALF: the half overloads (ReLU: the intended
 * v > 0 ? zAdj : 0 of the float / pre-sm_80 forms -- the sm_80 half branch :70 returns v instead of zAdj); FLOAT: fp32 */
static float act_adjoint(const OracleNet* n, float v, float zAdj) {
    const float p = n->actParam;
    if (n->accMode == ORACLE_ACC_HALF) {
        switch (n->activation) {
            case ORACLE_ACT_RELU: return v > 0.f ? zAdj : 0.f;
            case ORACLE_ACT_SINE: return hmul(zAdj, hmul(rh(p), hcosf(hmul(v, rh(p)))));
            case ORACLE_ACT_SNAKE: return hmul(zAdj, hadd(1.0f, hsinf(hmul(rh(2 * p), v))));
            case ORACLE_ACT_SNAKEALT: return hmul(zAdj, hadd(hsinf(hmul(rh(2 * p), v)), rh(1 / (2 * p))));
            case ORACLE_ACT_SIGMOID: { const float ev = rh_d(exp((double)v)), ev1 = hadd(1.0f, ev); return hmul(zAdj, hdivf(ev, hmul(ev1, ev1))); }
            default: return zAdj;
        }
    }
    switch (n->activation) {
        case ORACLE_ACT_RELU: return v > 0.f ? zAdj : 0.f;
        case ORACLE_ACT_SINE: return (float)((double)zAdj * p * cos((double)v * p));
        case ORACLE_ACT_SNAKE: return (float)((double)zAdj * (1.0 + sin(2.0 * p * v)));
        case ORACLE_ACT_SNAKEALT: return (float)((double)zAdj * (sin(2.0 * p * v) + 1.0 / (2.0 * p)));
        case ORACLE_ACT_SIGMOID: { const double ev = exp((double)v); return (float)((double)zAdj * ev / ((ev + 1) * (ev + 1))); }
        default: return zAdj;
    }
}

/* adjIn[cin] = sum_cout W[cout][c0 + cin] * z[cout] for cin in [0, ncin): the transposed product :1285-1335 (HALF: half
 * accumulators of the 16-wide WMMA tiles; FLOAT: fp32) */
static void dense_transposed(const OracleNet* n, const uint16_t* W, int stride, int c0, int ncin, int cout, const float* z, float* adjIn) {
    for (int i = 0; i < ncin; ++i) {
        if (n->accMode == ORACLE_ACC_HALF) {
            float acc = 0.f;
            for (int k0 = 0; k0 < cout; k0 += 16) {
                double s = 0;
                for (int k = k0; k < k0 + 16 && k < cout; ++k) s += (double)h2f(W[(size_t)k * stride + c0 + i]) * (double)z[k];
                acc = rh_d((double)acc + s);
            }
            adjIn[i] = acc;
        } else {
            double s = 0;
            for (int k = 0; k < cout; ++k) s += (double)h2f(W[(size_t)k * stride + c0 + i]) * (double)z[k];
            adjIn[i] = (float)s;
        }
    }
}

/* one decoded, time-blended texel group of the latent grid at p (4 channels of texture t), fp32: LoadVolumetricFeaturesAdjoint :662-676 */
static void grid_value4(const OracleNet* n, int t, float px, float py, float pz, float out[4]) {
    float a[4], b[4];
    tex3d(n, n->gridTexA[t], px, py, pz, a);
    tex3d(n, n->gridTexB[t], px, py, pz, b);
    const float time = n->gridInterpolation[t];
    const float f = time - floorf(time);
    for (int c = 0; c < 4; ++c) {
        const float va = decode_grid(n, a[c], 4 * t + c), vb = decode_grid(n, b[c], 4 * t + c);
        out[c] = va + f * (vb - va);
    }
}

/* _evalNormalAdjoint (renderer_volume_tensorcores.cuh:1198-1540): d(density before clamping) / d(NORMALIZED position).
 * gridStep = latentGridDifferencesStepSize.  Colour networks: zero (the reference asserts). */
static void srn_adjoint(const OracleNet* n, const float wpos[3], const float dir[3], float gridStep, float grad[3]) {
    const int C = n->C, F = n->F, G = n->G, NH = n->NH;
    const int base = n->useDirection >= 1 ? 8 : 4;
    const int half = n->accMode == ORACLE_ACC_HALF;
    grad[0] = grad[1] = grad[2] = 0.f;
    if (n->outputMode == ORACLE_OUT_RGBO || n->outputMode == ORACLE_OUT_RGBO_DIRECT) return;
    float p[3];
    for (int i = 0; i < 3; ++i) p[i] = (wpos[i] - n->boxMin[i]) / n->boxSize[i];
    /* ---- forward pass with the pre-activations kept (the reference stores them in shared memory as halves) ---- */
    static _Thread_local float x[ORACLE_MAX_C + 64], y[ORACLE_MAX_C], pre[16][ORACLE_MAX_C], adj[ORACLE_MAX_C + 64], z[ORACLE_MAX_C];
    const float vx = rh(p[0]), vy = rh(p[1]), vz = rh(p[2]);
    float dxh = 0, dyh = 0, dzh = 0;
    if (n->useDirection >= 1) { dxh = rh(dir[0]); dyh = rh(dir[1]); dzh = rh(dir[2]); }
    float phase[ORACLE_MAX_C];
    x[0] = vx; x[1] = vy; x[2] = vz; x[3] = n->passTime ? rh(n->gridInterpolation[0]) : 0.f;
    if (n->useDirection >= 1) { x[4] = dxh; x[5] = dyh; x[6] = dzh; x[7] = 0.f; }
    const int d6 = n->useDirection == 2;
    for (int i = 0; i < F; ++i) {
        const float f0 = h2f(n->fourier[i]), f1 = h2f(n->fourier[i + F]), f2 = h2f(n->fourier[i + 2 * F]);
        const float f3 = d6 ? h2f(n->fourier[i + 3 * F]) : 0.f, f4 = d6 ? h2f(n->fourier[i + 4 * F]) : 0.f, f5 = d6 ? h2f(n->fourier[i + 5 * F]) : 0.f;
        if (half) {
            float c = hmul(vx, f0);
            c = hfma(vy, f1, c); c = hfma(vz, f2, c);
            if (d6) { c = hfma(dxh, f3, c); c = hfma(dyh, f4, c); c = hfma(dzh, f5, c); }
            phase[i] = c;
            x[base + i] = hcosf(c); x[base + F + i] = hsinf(c);
        } else {
            double c = (double)vx * f0 + (double)vy * f1 + (double)vz * f2;
            if (d6) c += (double)dxh * f3 + (double)dyh * f4 + (double)dzh * f5;
            phase[i] = (float)c; /* (only its cos / sin are used below, from the double) */
            x[base + i] = rh_d(cos(c)); x[base + F + i] = rh_d(sin(c));
        }
    }
    float preFirst[ORACLE_MAX_C];
    if (F == 0) { /* scalar first layer :810-823 */
        for (int co = 0; co < C; ++co) {
            float c;
            if (half) {
                c = h2f(n->bFirst[co]);
                c = hfma(vx, h2f(n->wFirst[co + C * 0]), c); c = hfma(vy, h2f(n->wFirst[co + C * 1]), c); c = hfma(vz, h2f(n->wFirst[co + C * 2]), c);
                if (n->useDirection >= 1) { c = hfma(dxh, h2f(n->wFirst[co + C * 3]), c); c = hfma(dyh, h2f(n->wFirst[co + C * 4]), c); c = hfma(dzh, h2f(n->wFirst[co + C * 5]), c); }
                y[co] = act_half(n->activation, c, n->actParam);
            } else {
                double cd = h2f(n->bFirst[co]);
                cd += (double)vx * h2f(n->wFirst[co + C * 0]) + (double)vy * h2f(n->wFirst[co + C * 1]) + (double)vz * h2f(n->wFirst[co + C * 2]);
                if (n->useDirection >= 1) cd += (double)dxh * h2f(n->wFirst[co + C * 3]) + (double)dyh * h2f(n->wFirst[co + C * 4]) + (double)dzh * h2f(n->wFirst[co + C * 5]);
                c = (float)cd;
                y[co] = rh(act_float(n->activation, c, n->actParam));
            }
            preFirst[co] = c;
        }
        memcpy(x, y, sizeof(float) * (size_t)C);
    }
    float preGrid[ORACLE_MAX_C];
    if (G > 0) {
        for (int g = 0; g < G / 16; ++g) grid_features(n, p, g, x + C + 16 * g);
        dense_layer_tape(n, n->wFirst, n->bFirst, C + G, C, x, y, preGrid);
        memcpy(x, y, sizeof(float) * (size_t)C);
    }
    for (int l = 0; l < NH; ++l) {
        dense_layer_tape(n, n->wHidden + (size_t)l * C * C, n->bHidden + (size_t)l * C, C, C, x, y, pre[l]);
        memcpy(x, y, sizeof(float) * (size_t)C);
    }
    const int Co = n->Cout;
    float out0;
    if (half) {
        float acc = h2f(n->bLast[0]);
        for (int k = 0; k < C; ++k) acc = hfma(x[k], h2f(n->wLast[(size_t)k * Co]), acc);
        out0 = acc;
    } else {
        double s = h2f(n->bLast[0]);
        for (int k = 0; k < C; ++k) s += (double)x[k] * (double)h2f(n->wLast[(size_t)k * Co]);
        out0 = (float)s;
    }
    /* ---- adjoint of the output parametrization :1225-1262 ---- */
    float adjOut = 1.f;
    if (n->outputMode == ORACLE_OUT_DENSITY || n->outputMode == ORACLE_OUT_DENSITY_GRADIENT || n->outputMode == ORACLE_OUT_DENSITY_CURVATURE) {
        const double ev = half ? (double)expf(out0) : exp((double)out0); /* Sigmoid::adjoint(float) :204-209 */
        adjOut = (float)(ev / ((ev + 1) * (ev + 1)));
    }
    if (half) adjOut = rh(adjOut);
    for (int cin = 0; cin < C; ++cin) adj[cin] = half ? hmul(h2f(n->wLast[(size_t)cin * Co]), adjOut) : rh(h2f(n->wLast[(size_t)cin * Co]) * adjOut);
    /* ---- hidden layers, last to first :1269-1336 ---- */
    for (int l = NH - 1; l >= 0; --l) {
        for (int co = 0; co < C; ++co) { const float a = act_adjoint(n, pre[l][co], adj[co]); z[co] = half ? a : rh(a); }
        dense_transposed(n, n->wHidden + (size_t)l * C * C, C, 0, C, C, z, adj);
        if (half) for (int i = 0; i < C; ++i) adj[i] = rh(adj[i]); /* store_matrix_sync of half accumulators: already half */
    }
    /* ---- latent grid layer :1342-1460 ---- */
    if (G > 0) {
        for (int co = 0; co < C; ++co) { const float a = act_adjoint(n, preGrid[co], adj[co]); z[co] = half ? a : rh(a); }
        const float s2 = 1.0f / (2.0f * gridStep);
        for (int g = 0; g < G / 16; ++g) {
            float adjG[16];
            dense_transposed(n, n->wFirst, C + G, C + 16 * g, 16, C, z, adjG);
            for (int i = 0; i < 4; ++i) { /* LoadVolumetricFeaturesAdjoint :609-735: central differences, fp32 */
                const int t = 4 * g + i;
                for (int axis = 0; axis < 3; ++axis) {
                    float hi[4], lo[4];
                    float q[3] = {p[0], p[1], p[2]};
                    q[axis] = p[axis] + gridStep; grid_value4(n, t, q[0], q[1], q[2], hi);
                    q[axis] = p[axis] - gridStep; grid_value4(n, t, q[0], q[1], q[2], lo);
                    for (int c = 0; c < 4; ++c) grad[axis] += adjG[4 * i + c] * (s2 * (hi[c] - lo[c]));
                }
            }
        }
        dense_transposed(n, n->wFirst, C + G, 0, C, C, z, adj);
    }
    /* ---- first layer :1466-1531 ---- */
    if (F > 0) {
        grad[0] += adj[0]; grad[1] += adj[1]; grad[2] += adj[2];
        for (int i = 0; i < F; ++i) {
            const float f0 = h2f(n->fourier[i]), f1 = h2f(n->fourier[i + F]), f2 = h2f(n->fourier[i + 2 * F]);
            if (half) {
                const float c = phase[i];
                const float adjC = hsub(hmul(rh(adj[base + F + i]), hcosf(c)), hmul(rh(adj[base + i]), hsinf(c)));
                grad[0] += hmul(f0, adjC); grad[1] += hmul(f1, adjC); grad[2] += hmul(f2, adjC);
            } else {
                double c = (double)vx * f0 + (double)vy * f1 + (double)vz * f2;
                if (d6) c += (double)dxh * h2f(n->fourier[i + 3 * F]) + (double)dyh * h2f(n->fourier[i + 4 * F]) + (double)dzh * h2f(n->fourier[i + 5 * F]);
                const double adjC = (double)adj[base + F + i] * cos(c) - (double)adj[base + i] * sin(c);
                grad[0] += (float)(f0 * adjC); grad[1] += (float)(f1 * adjC); grad[2] += (float)(f2 * adjC);
            }
        }
    } else {
        for (int co = 0; co < C; ++co) {
            const float a1 = act_adjoint(n, preFirst[co], adj[co]);
            for (int axis = 0; axis < 3; ++axis) {
                const float w = h2f(n->wFirst[co + C * axis]);
                grad[axis] += half ? hmul(w, a1) : w * a1;
            }
        }
    }
}

int oracle_eval_adjoint(const OracleNet* n, const float* pos, const float* dirs, size_t count, float gridStep, float* out3) {
    if (!n || !pos || !out3 || n->C > ORACLE_MAX_C || n->NH > 16) return -1;
    static const float zero[3] = {0, 0, 0};
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)count; ++i) srn_adjoint(n, pos + 3 * i, dirs ? dirs + 3 * i : zero, gridStep, out3 + 3 * i);
    return 0;
}

int oracle_eval_points(const OracleNet* n, const float* pos, const float* dirs, size_t count, float* out) {
    if (!n || n->C > ORACLE_MAX_C || n->G > 64) return -1;
    const int oc = (n->outputMode == ORACLE_OUT_RGBO || n->outputMode == ORACLE_OUT_RGBO_DIRECT) ? 4 : 1;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)count; ++i) {
        float o[4], nr[3], cv[2];
        const float zero[3] = {0, 0, 0};
        srn_eval(n, pos + 3 * i, dirs ? dirs + 3 * i : zero, o, nr, cv);
        for (int c = 0; c < oc; ++c) out[(size_t)i * oc + c] = o[c];
    }
    return 0;
}

int oracle_eval_points_full(const OracleNet* n, const float* pos, const float* dirs, size_t count, float* out9) {
    if (!n || n->C > ORACLE_MAX_C || n->G > 64) return -1;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)count; ++i) {
        float o[4], nr[3], cv[2];
        const float zero[3] = {0, 0, 0};
        srn_eval(n, pos + 3 * i, dirs ? dirs + 3 * i : zero, o, nr, cv);
        float* r = out9 + (size_t)i * 9;
        r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = o[3]; r[4] = nr[0]; r[5] = nr[1]; r[6] = nr[2]; r[7] = cv[0]; r[8] = cv[1];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- TFs */
static float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

/* tex1D<float4> with linear filtering, normalized coordinates, clamp addressing (exact fp32 weights; the hardware filter of
 * the reference has 8 fractional bits) */
static void tex1d(const float* tex, int R, float x, float o[4]) {
    const float d = x * R - 0.5f;
    const int di = (int)floorf(d);
    const float f = d - di;
    const int i0 = di < 0 ? 0 : (di > R - 1 ? R - 1 : di), i1 = di + 1 < 0 ? 0 : (di + 1 > R - 1 ? R - 1 : di + 1);
    for (int k = 0; k < 4; ++k) o[k] = tex[4 * i0 + k] + f * (tex[4 * i1 + k] - tex[4 * i0 + k]);
}
static void tex2d(const float* tex, int R, float x, float y, float o[4]) {
    const float dx = x * R - 0.5f, dy = y * R - 0.5f;
    const int ix = (int)floorf(dx), iy = (int)floorf(dy);
    const float fx = dx - ix, fy = dy - iy;
#define CL(v) ((v) < 0 ? 0 : ((v) > R - 1 ? R - 1 : (v)))
    const int x0 = CL(ix), x1 = CL(ix + 1), y0 = CL(iy), y1 = CL(iy + 1);
#undef CL
    for (int k = 0; k < 4; ++k) {
        const float a = tex[4 * ((size_t)y0 * R + x0) + k], b = tex[4 * ((size_t)y0 * R + x1) + k];
        const float c = tex[4 * ((size_t)y1 * R + x0) + k], d = tex[4 * ((size_t)y1 * R + x1) + k];
        const float lo = a + fx * (b - a), hi = c + fx * (d - c);
        o[k] = lo + fy * (hi - lo);
    }
}

void oracle_tf_preintegrate(const float* tex, int R, int mode, float stepsize, int N, float* out) {
    if (mode == 1) { /* Compute1DPreintegrationTableKernel :9-36 */
        float integral[4] = {0, 0, 0, 0}, last[4], v[4];
        float lastDensity = 0.f;
        tex1d(tex, R, lastDensity, last);
        for (int i = 0; i < R; ++i) {
            const float cur = ((float)i + 0.5f) / (float)R;
            tex1d(tex, R, cur, v);
            const float w = cur - lastDensity;
            for (int k = 0; k < 3; ++k) integral[k] += w * 0.5f * (last[k] * last[3] + v[k] * v[3]);
            integral[3] += w * 0.5f * (last[3] + v[3]);
            for (int k = 0; k < 4; ++k) { out[4 * i + k] = integral[k]; last[k] = v[k]; }
            lastDensity = cur;
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int iend = 0; iend < R; ++iend) /* Compute2DPreintegrationTableKernel :50-79 */
        for (int istart = 0; istart < R; ++istart) {
            const float dstart = ((float)istart + 0.5f) / (float)R, dend = ((float)iend + 0.5f) / (float)R;
            float rgb[3] = {0, 0, 0}, alphaSum = 0, v[4];
            const float h = 1.0f / (float)N;
            for (int i = 1; i <= N; ++i) {
                const float omega = i * h;
                tex1d(tex, R, (1 - omega) * dstart + omega * dend, v);
                alphaSum += v[3] * h * stepsize;
                const float k = h * v[3] * stepsize * expf(-alphaSum);
                rgb[0] += k * v[0]; rgb[1] += k * v[1]; rgb[2] += k * v[2];
            }
            float* o = out + 4 * ((size_t)iend * R + istart);
            o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = 1 - expf(-alphaSum);
        }
}

/* TransferFunctionTexture::eval with pre-integration, renderer_tf_texture.cuh:55-93; density is clamped */
static void tf_eval_preintegrated(const OracleScene* s, float density, float previousDensity, float c[4]) {
    density = density < 0 ? 0 : (density > 1 ? 1 : density);
    if (previousDensity < 0) previousDensity = density;
    if (s->tfPreintegration == 1) {
        if (fabsf(previousDensity - density) < 1e-3f) {
            tex1d(s->tfTable, s->tfRows, density, c);
            c[3] *= s->stepsize;
        } else {
            float f[4], b[4];
            tex1d(s->tfPreintegrated, s->tfRows, previousDensity, f);
            tex1d(s->tfPreintegrated, s->tfRows, density, b);
            const float inv = 1.0f / (density - previousDensity);
            for (int k = 0; k < 3; ++k) c[k] = s->stepsize * (b[k] - f[k]) * inv;
            c[3] = 1 - expf(-s->stepsize * (b[3] - f[3]) * inv);
            if (c[3] > 1e-5f) { c[0] /= c[3]; c[1] /= c[3]; c[2] /= c[3]; }
        }
    } else {
        tex2d(s->tfPreintegrated, s->tfRows, previousDensity, density, c);
        if (c[3] > 1e-5f) { c[0] /= c[3]; c[1] /= c[3]; c[2] /= c[3]; }
    }
}

/* normalLength: |gradient| the volume handed to the TF (0 where it provides none); previousDensity: the un-clamped mapped density of
 * the previous sample of the ray, -1 for the first one (stepping_dvr.cuh:81,135) -- only the Gaussian variants read them */
static void tf_eval(const OracleScene* s, float density, float normalLength, float previousDensity, float c[4]) {
    density = clamp01(density);
    c[0] = c[1] = c[2] = c[3] = 0.f;
    const float* T = s->tfTable;
    switch (s->tfKind) {
        case ORACLE_TF_IDENTITY: /* renderer_tf_identity.cuh:36-54 */
            c[0] = c[1] = c[2] = density * s->tfScaleEmission;
            c[3] = density * s->tfScaleAbsorption * s->stepsize;
            break;
        case ORACLE_TF_GAUSSIAN: /* renderer_tf_gaussian.cuh:43-86 */
            for (int i = 0; i < s->tfRows; ++i) {
                const float* r = T + 6 * i;
                const float mu = r[4];
                float sigma = r[5], ni;
                if (s->tfGaussianMode == 1) sigma *= fmaxf(1e-5f, normalLength * 0.1f); /* :55-57 "some arbitrary scaling factor" */
                if (s->tfGaussianMode == 2 && !(previousDensity < 0 || previousDensity == density)) {
                    /* :61-72 piecewise analytic integration, constant colour per segment */
                    const float SQRT_PI_2 = 0.8862269254527580136f;
                    ni = SQRT_PI_2 / (previousDensity - density) * sigma * (erff((previousDensity - mu) / sigma) + erff((mu - density) / sigma));
                } else {
                    ni = expf(-(density - mu) * (density - mu) / (sigma * sigma));
                }
                for (int k = 0; k < 4; ++k) c[k] += r[k] * ni;
            }
            c[3] *= s->stepsize;
            break;
        case ORACLE_TF_PIECEWISE: { /* renderer_tf_piecewise.cuh:29-62 */
            int i;
            for (i = 0; i < s->tfRows - 2; ++i)
                if (T[5 * (i + 1) + 4] > density) break;
            const float* a = T + 5 * i;
            const float* b = T + 5 * (i + 1);
            const float d = fminf(fmaxf(density, a[4]), b[4]);
            const float f = (d - a[4]) / (b[4] - a[4]);
            for (int k = 0; k < 4; ++k) c[k] = a[k] + f * (b[k] - a[k]);
            c[3] *= s->stepsize;
        } break;
        case ORACLE_TF_TEXTURE: { /* renderer_tf_texture.cuh:46-55 (tensor mode) */
            const int R = s->tfRows;
            const float d = density * R - 0.5f;
            const int di = (int)floorf(d);
            const float df = d - di;
            const float* a = T + 4 * clampi(di, 0, R - 1);
            const float* b = T + 4 * clampi(di + 1, 0, R - 1);
            for (int k = 0; k < 4; ++k) c[k] = a[k] + df * (b[k] - a[k]);
            c[3] *= s->stepsize;
        } break;
        default: break;
    }
}

void oracle_tf_evaluate(const OracleScene* s, const float* density, const float* previous, size_t n, float* out4) {
    const float divRange = 1.0f / (s->densityMax - s->densityMin);
    for (size_t i = 0; i < n; ++i) {
        float c[4] = {0, 0, 0, 0};
        const float d = density[i];
        if (d >= s->densityMin) {
            const float d2 = (d - s->densityMin) * divRange;
            if (s->tfPreintegration) {
                const float p = previous ? previous[i] : -1.f;
                tf_eval_preintegrated(s, d2, p >= 0 ? (p - s->densityMin) * divRange : -1.f, c);
            } else { /* renderer_tf_kernels.cuh:30,61: zero normal; previous density mapped when it is >= 0 */
                const float p = previous ? previous[i] : -1.f;
                tf_eval(s, d2, 0.f, p >= 0 ? (p - s->densityMin) * divRange : -1.f, c);
            }
        }
        for (int k = 0; k < 4; ++k) out4[4 * i + k] = c[k];
    }
}

/* ------------------------------------------------------------------------------------------- DVR */
static void render_pixel(const OracleNet* n, const OracleScene* s, int W, int H, int x, int y, float px8[8],
                         unsigned long long* samples) {
    /* Camera ray (renderer_image_evaluator_simple.cuh:84-88, renderer_camera.cuh:41-52) and box test (renderer_utils.cuh:91-105).
     * The reference leaves the contraction of these expressions to nvcc; here they are spelled out operation by operation (explicit
     * fma's, correctly rounded division / square root; this file is compiled with -ffp-contract=off) in exactly the sequence of the HIP
     * kernels (fv-srn_amd/csrc/kernels.hpp, render_body), so that the sample positions of the two are bit-identical. */
    const float ndcx = 2 * ((float)x + 0.5f) / (float)W - 1, ndcy = 2 * ((float)y + 0.5f) / (float)H - 1;
    const float tanFovY = tanf(s->fovY / 2), tanFovX = tanFovY * ((float)W / (float)H);
    const float* eye = s->eye; const float* right = s->right; const float* up = s->up;
    const float front[3] = {up[1] * right[2] - up[2] * right[1], up[2] * right[0] - up[0] * right[2],
                            up[0] * right[1] - up[1] * right[0]};
    const float ax = ndcx * tanFovX, ay = ndcy * tanFovY;
    float dir[3];
    for (int i = 0; i < 3; ++i) dir[i] = fmaf(ay, up[i], fmaf(ax, right[i], front[i]));
    const float il = 1.0f / sqrtf(fmaf(dir[2], dir[2], fmaf(dir[1], dir[1], dir[0] * dir[0])));
    for (int i = 0; i < 3; ++i) dir[i] *= il;
    float tlo[3], thi[3];
    for (int i = 0; i < 3; ++i) {
        const float inv = 1.0f / dir[i];
        const float ta = (n->boxMin[i] - eye[i]) * inv, tb = (n->boxMin[i] + n->boxSize[i] - eye[i]) * inv;
        tlo[i] = fminf(ta, tb);
        thi[i] = fmaxf(ta, tb);
    }
    float tmin = fmaxf(fmaxf(tlo[0], tlo[1]), tlo[2]);
    const float tmax = fminf(fminf(thi[0], thi[1]), thi[2]);
    tmin = fmaxf(tmin, 0.f); /* stepping_dvr.cuh:66 */
    const float alphaEarlyOut = 1.0f - 1e-5f;
    const float divRange = 1.0f / (s->densityMax - s->densityMin);
    const int rgbo = n->outputMode == ORACLE_OUT_RGBO || n->outputMode == ORACLE_OUT_RGBO_DIRECT;
    float col[4] = {0, 0, 0, 0}, nacc[3] = {0, 0, 0}, depth = 0;
    float previousDensity = -1.f; /* :81 */
    unsigned long long cnt = 0;
    /* ORACLE_ACC_DEVICE with OracleScene::rotationResync: the kernels' feature rotation (fv-srn_amd/csrc/kernels.hpp render_body,
     * srn_device.hpp fourier_advance_piece), stated step by step.  feat / dfeat: the first layer's padded input vector in fp32 and
     * its per-step increment (positions: additive; (cos, sin) pairs: the rotation by the phase increment of one step). */
    const int rotate = n->accMode == ORACLE_ACC_DEVICE && s->rotationResync > 0 && n->F > 0; /* (r04: latent-grid networks of the cell-table kernels too) */
    const int hilo = s->rotationHiLo && s->rotationResync > 1;
    const int K = s->segments > 1 ? s->segments : 1;
    const int nsteps = (tmax - tmin) >= 0.f ? (int)((tmax - tmin) / s->stepsize) + 1 : 0; /* the host's cut of a ray into K step ranges */
    float feat[ORACLE_MAX_C], dfeat[ORACLE_MAX_C], xrot[ORACLE_MAX_C];
    float pn0[3], dn[3];
    for (int k = 0; k < 3; ++k) { /* the kernels' p(t) = pn0 + dn t in unit-box coordinates */
        const float inv = 1.0f / n->boxSize[k];
        pn0[k] = (eye[k] - n->boxMin[k]) * inv;
        dn[k] = dir[k] * inv;
    }
    for (int i = 0;; ++i) { /* :84-154, per-lane view of the warp-synchronous loop */
        const float t = fmaf((float)i, s->stepsize, tmin); /* tmin + i * stepsize, one rounding (kernels.hpp) */
        const int valid = (t <= tmax) && (!s->earlyOut || col[3] < alphaEarlyOut);
        if (!valid) break; /* later iterations of an invalid lane never blend (:151) and never turn valid again */
        ++cnt;
        const float pos[3] = {eye[0] + dir[0] * t, eye[1] + dir[1] * t, eye[2] + dir[2] * t};
        float v[4], g[3], cv[2], c[4] = {0, 0, 0, 0};
        /* DEVICE: the kernels evaluate p(t) = pn0 + dn t with one fma per axis (kernels.hpp) instead of ((o + d t) - boxMin) / boxSize */
        const float pd[3] = {fmaf(dn[0], t, pn0[0]), fmaf(dn[1], t, pn0[1]), fmaf(dn[2], t, pn0[2])};
        if (rotate) {
            const int F = n->F, base = n->useDirection >= 1 ? 8 : 4, C = n->C;
            int seg = 0; /* the step range [nsteps seg / K, nsteps (seg + 1) / K) that holds step i; the last one is open */
            while (seg + 1 < K && (nsteps * (seg + 1)) / K <= i) ++seg;
            const int local = i - (nsteps * seg) / K;
            if ((local & (s->rotationResync - 1)) == 0) {
                for (int pass = (local == 0 ? 1 : 0); pass >= 0; --pass) { /* pass 1: the per-step increment, once per ray and segment */
                    float* f = pass ? dfeat : feat;
                    float in6[6] = {0, 0, 0, 0, 0, 0};
                    float lo6[6] = {0, 0, 0, 0, 0, 0};
                    for (int k = 0; k < 3; ++k) {
                        const float v = pass ? dn[k] * s->stepsize : pd[k];
                        in6[k] = rh(v);
                        if (hilo) lo6[k] = rh(v - in6[k]); /* second phase MFMA on the residual (constant and direction slots zero) */
                    }
                    if (!pass && n->useDirection >= 1) for (int k = 0; k < 3; ++k) in6[3 + k] = rh(dir[k]);
                    for (int k = 0; k < C; ++k) f[k] = 0.f;
                    f[0] = in6[0] + lo6[0]; f[1] = in6[1] + lo6[1]; f[2] = in6[2] + lo6[2];
                    f[3] = (!pass && n->passTime) ? rh(n->gridInterpolation[0]) : 0.f;
                    if (n->useDirection >= 1) { f[4] = in6[3]; f[5] = in6[4]; f[6] = in6[5]; }
                    for (int q = 0; q < F; ++q) {
                        double rev = (double)device_phase_rev(n, q, in6);
                        if (hilo) rev = (double)(float)(rev + device_phase_lin(n, q, lo6)); /* accumulated by the second MFMA: fp32 */
                        const double ph = 2.0 * 3.14159265358979323846 * rev;
                        f[base + q] = (float)cos(ph);
                        f[base + F + q] = (float)sin(ph);
                    }
                }
            }
            for (int k = 0; k < C; ++k) xrot[k] = rh(feat[k]); /* v_cvt_pk_f16_f32 of the current features */
            for (int k = 0; k < base; ++k) feat[k] += dfeat[k];  /* advance to the next sample: positions ... */
            for (int q = 0; q < F; ++q) {                        /* ... and c' = c cd - s sd, s' = s cd + c sd (v_pk_mul_f32 + v_pk_fma_f32) */
                const float cc = feat[base + q], ss = feat[base + F + q], cd = dfeat[base + q], sd = dfeat[base + F + q];
                const float t0 = cc * cd, t1 = ss * cd;
                feat[base + q] = fmaf(ss, -sd, t0);
                feat[base + F + q] = fmaf(cc, sd, t1);
            }
            srn_eval_x(n, pos, dir, xrot, pd, v, g, cv);
        } else if (n->accMode == ORACLE_ACC_DEVICE) {
            srn_eval_x(n, pos, dir, NULL, pd, v, g, cv);
        } else {
            srn_eval(n, pos, dir, v, g, cv);
        }
        if (rgbo) {
            c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3] * s->stepsize; /* :104-108 */
        } else {
            const float density2 = (v[0] - s->densityMin) * divRange;
            if (v[0] >= s->densityMin) {
                if (s->gradientMode == 1) { /* evalNormal, finite differences of valueNoClamping :1185-1196 */
                    const float h = s->fdStep;
                    float vv[6];
                    for (int k = 0; k < 6; ++k) {
                        float q[3] = {pos[0], pos[1], pos[2]}, vk[4], gk[3], ck[2];
                        q[k >> 1] += (k & 1) ? -h : h;
                        srn_eval(n, q, dir, vk, gk, ck);
                        vv[k] = vk[0];
                    }
                    g[0] = (vv[0] - vv[1]) / (2 * h); g[1] = (vv[2] - vv[3]) / (2 * h); g[2] = (vv[4] - vv[5]) / (2 * h);
                } else if (s->gradientMode == 2) { /* GRADIENT_MODE_ADJOINT_METHOD :1198-1540 */
                    srn_adjoint(n, pos, dir, s->gridDiffStep, g);
                }
                if (s->tfPreintegration) tf_eval_preintegrated(s, density2, previousDensity, c);
                else tf_eval(s, density2, sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), previousDensity, c); /* :113-133 */
            }
            previousDensity = density2; /* :135 */
        }
        if (c[3] > 0 && (s->brdfMagnitudeScaling || s->brdfPhong)) { /* BRDFLambert::eval, renderer_brdf_lambert.cuh:56-103 */
            const float g2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            if (s->brdfMagnitudeScaling) c[3] *= 1.0f - expf(-s->brdfMagScale * g2);
            if (s->brdfPhong) {
                const float gradientNorm = 1.0f / sqrtf(g2); /* the reference's rsqrt(gradientNormSqr) */
                float nn[3] = {g[0], g[1], g[2]};
                if (g2 >= 1e-8f) { nn[0] *= gradientNorm; nn[1] *= gradientNorm; nn[2] *= gradientNorm; }
                float l[3];
                for (int k = 0; k < 3; ++k) l[k] = s->brdfLightType == 1 ? -s->brdfLight[k] : s->brdfLight[k] - pos[k];
                const float il = 1.0f / sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
                l[0] *= il; l[1] *= il; l[2] *= il;
                const float lo = s->brdfMagCenter - s->brdfMagRadius, hi = s->brdfMagCenter + s->brdfMagRadius;
                const float y = clamp01((gradientNorm - lo) / (hi - lo));
                const float phongStrength = y * y * (3.0f - 2.0f * y);
                const float ambientStrength = 1.0f + phongStrength * (s->brdfAmbient - 1.0f);
                const float nl = nn[0] * l[0] + nn[1] * l[1] + nn[2] * l[2];
                const float r[3] = {l[0] - 2 * nn[0] * nl, l[1] - 2 * nn[1] * nl, l[2] - 2 * nn[2] * nl}; /* reflect(l, -n) */
                const float e = (float)s->brdfSpecularExponent;
                const float spec = (e + 2.0f) * 0.159155f * powf(fmaxf(0.f, dir[0] * r[0] + dir[1] * r[1] + dir[2] * r[2]), e);
                for (int k = 0; k < 3; ++k)
                    c[k] = ambientStrength * c[k] + (1 - ambientStrength) * (fabsf(nl) * c[k] + s->brdfSpecular * spec);
            }
        }
        if (c[3] > 0) { /* :138-153 */
            const float l2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            if (l2 >= 1e-8f) { const float k = 1.0f / sqrtf(l2); g[0] *= k; g[1] *= k; g[2] *= k; }
            const float a = s->blendMode == ORACLE_BLEND_BEER_LAMBERT ? 1.0f - expf(-c[3]) : fminf(1.0f, c[3]);
            const float w = (1 - col[3]) * a;
            for (int k = 0; k < 3; ++k) { col[k] += w * c[k]; nacc[k] += w * g[k]; }
            depth += w * t;
            col[3] += w;
        }
    }
    /* renderer_image_evaluator_simple.cuh:100-124, samples == 1 */
    px8[0] = col[0]; px8[1] = col[1]; px8[2] = col[2]; px8[3] = col[3];
    for (int k = 0; k < 3; ++k) px8[4 + k] = nacc[k] * col[3];
    px8[7] = depth * col[3] / col[3];
    *samples = cnt;
}

int oracle_render(const OracleNet* n, const OracleScene* s, int W, int H, int y0, int y1, float* out8,
                  unsigned long long* evaluatedSamples) {
    if (!n || !s || n->C > ORACLE_MAX_C || n->G > 64 || W <= 0 || H <= 0 || y0 < 0 || y1 > H) return -1;
    unsigned long long total = 0;
    const size_t plane = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < W; ++x) {
            float p[8];
            unsigned long long c;
            render_pixel(n, s, W, H, x, y, p, &c);
            total += c;
            for (int k = 0; k < 8; ++k) out8[k * plane + (size_t)y * W + x] = p[k];
        }
    if (evaluatedSamples) *evaluatedSamples = total;
    return 0;
}

/* lane-exact evaluated sample count without evaluating the network (early-out off), SURVEY 8(d) */
unsigned long long oracle_count_samples(const OracleNet* n, const OracleScene* s, int W, int H, int y0, int y1) {
    unsigned long long total = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < W; ++x) {
            const float ndcx = 2 * ((float)x + 0.5f) / (float)W - 1, ndcy = 2 * ((float)y + 0.5f) / (float)H - 1;
            const float tanFovY = tanf(s->fovY / 2), tanFovX = tanFovY * ((float)W / (float)H);
            const float* right = s->right; const float* up = s->up;
            const float front[3] = {up[1] * right[2] - up[2] * right[1], up[2] * right[0] - up[0] * right[2],
                                    up[0] * right[1] - up[1] * right[0]};
            const float ax = ndcx * tanFovX, ay = ndcy * tanFovY; /* the sequence of render_pixel */
            float dir[3];
            for (int i = 0; i < 3; ++i) dir[i] = fmaf(ay, up[i], fmaf(ax, right[i], front[i]));
            const float il = 1.0f / sqrtf(fmaf(dir[2], dir[2], fmaf(dir[1], dir[1], dir[0] * dir[0])));
            float tmin = -INFINITY, tmax = INFINITY;
            for (int i = 0; i < 3; ++i) {
                const float inv = 1.0f / (dir[i] * il);
                const float ta = (n->boxMin[i] - s->eye[i]) * inv, tb = (n->boxMin[i] + n->boxSize[i] - s->eye[i]) * inv;
                tmin = fmaxf(tmin, fminf(ta, tb));
                tmax = fminf(tmax, fmaxf(ta, tb));
            }
            tmin = fmaxf(tmin, 0.f);
            for (int i = 0;; ++i) {
                if (!(fmaf((float)i, s->stepsize, tmin) <= tmax)) break;
                ++total;
            }
        }
    return total;
}

/* --------------------------------------------------------------------------- dense grid volumes
 * kernel::VolumeInterpolationGrid, renderer/renderer_volume_grid.cuh */
static float vol_fetch(const OracleVolume* v, int x, int y, int z) {
    x = clampi(x, 0, v->res[0] - 1); y = clampi(y, 0, v->res[1] - 1); z = clampi(z, 0, v->res[2] - 1);
    return v->data[(size_t)x + (size_t)v->res[0] * ((size_t)y + (size_t)v->res[1] * (size_t)z)];
}
static float lerpf_(float a, float b, float t) { return a + t * (b - a); } /* helper_math.cuh lerp */

/* sampleLinear, :102-139 */
static float vol_linear(const OracleVolume* v, float x, float y, float z) {
    if (v->source == 1) { /* tensor branch: ipos = make_int3(posObject) truncates, nodes at integer coordinates */
        const int ix = (int)x, iy = (int)y, iz = (int)z;
        const float fx = x - (float)ix, fy = y - (float)iy, fz = z - (float)iz;
        const float d000 = vol_fetch(v, ix, iy, iz), d001 = vol_fetch(v, ix, iy, iz + 1);
        const float d010 = vol_fetch(v, ix, iy + 1, iz), d011 = vol_fetch(v, ix, iy + 1, iz + 1);
        const float d100 = vol_fetch(v, ix + 1, iy, iz), d101 = vol_fetch(v, ix + 1, iy, iz + 1);
        const float d110 = vol_fetch(v, ix + 1, iy + 1, iz), d111 = vol_fetch(v, ix + 1, iy + 1, iz + 1);
        return lerpf_(lerpf_(lerpf_(d000, d100, fx), lerpf_(d010, d110, fx), fy),
                      lerpf_(lerpf_(d001, d101, fx), lerpf_(d011, d111, fx), fy), fz);
    }
    /* tex3D with cudaFilterModeLinear, un-normalised coordinates, clamp addressing (CUDA programming guide, linear
     * filtering): xB = x - 0.5, i = floor(xB), alpha = frac(xB) stored in 1.8 fixed point */
    const float xb = x - 0.5f, yb = y - 0.5f, zb = z - 0.5f;
    const float fxi = floorf(xb), fyi = floorf(yb), fzi = floorf(zb);
    const float a = rintf((xb - fxi) * 256.f) * (1.f / 256.f), b = rintf((yb - fyi) * 256.f) * (1.f / 256.f),
                c = rintf((zb - fzi) * 256.f) * (1.f / 256.f);
    const int i = (int)fxi, j = (int)fyi, k = (int)fzi;
    return (1 - a) * (1 - b) * (1 - c) * vol_fetch(v, i, j, k) + a * (1 - b) * (1 - c) * vol_fetch(v, i + 1, j, k) +
           (1 - a) * b * (1 - c) * vol_fetch(v, i, j + 1, k) + a * b * (1 - c) * vol_fetch(v, i + 1, j + 1, k) +
           (1 - a) * (1 - b) * c * vol_fetch(v, i, j, k + 1) + a * (1 - b) * c * vol_fetch(v, i + 1, j, k + 1) +
           (1 - a) * b * c * vol_fetch(v, i, j + 1, k + 1) + a * b * c * vol_fetch(v, i + 1, j + 1, k + 1);
}

/* bspline_weights + sampleCubic, :141-186 */
static float vol_cubic(const OracleVolume* v, float x, float y, float z) {
    const float c[3] = {x - 0.5f, y - 0.5f, z - 0.5f};
    float g0[3], g1[3], h0[3], h1[3];
    for (int d = 0; d < 3; ++d) {
        const float index = floorf(c[d]), f = c[d] - index;
        const float one_frac = 1.0f - f, squared = f * f, one_sqd = one_frac * one_frac;
        const float w0 = 1.0f / 6.0f * one_sqd * one_frac;
        const float w1 = 2.0f / 3.0f - 0.5f * squared * (2.0f - f);
        const float w2 = 2.0f / 3.0f - 0.5f * one_sqd * (2.0f - one_frac);
        const float w3 = 1.0f / 6.0f * squared * f;
        g0[d] = w0 + w1;
        g1[d] = w2 + w3;
        h0[d] = (w1 / g0[d]) - 0.5f + index;
        h1[d] = (w3 / g1[d]) + 1.5f + index;
    }
    // eight linear fetches, weighted along x, then y, then z (the order of the reference's sums)
    float alongZ[2];
    for (int kz = 0; kz < 2; ++kz) {
        const float z = kz ? h1[2] : h0[2];
        float alongY[2];
        for (int ky = 0; ky < 2; ++ky) {
            const float y = ky ? h1[1] : h0[1];
            alongY[ky] = g0[0] * vol_linear(v, h0[0], y, z) + g1[0] * vol_linear(v, h1[0], y, z);
        }
        alongZ[kz] = g0[1] * alongY[0] + g1[1] * alongY[1];
    }
    return g0[2] * alongZ[0] + g1[2] * alongZ[1];
}

/* eval, :193-232 */
static void vol_to_object(const OracleVolume* v, const float w[3], float p[3]) {
    for (int d = 0; d < 3; ++d) {
        const float scale = (float)(v->newBehavior ? v->res[d] : v->res[d] - 1);
        p[d] = (w[d] - v->boxMin[d]) / v->boxSize[d] * scale;
    }
}
static float vol_sample(const OracleVolume* v, float x, float y, float z) {
    if (v->interpolation == 0) return vol_fetch(v, (int)roundf(x), (int)roundf(y), (int)roundf(z)); /* :88-101,189 */
    if (v->interpolation == 1) return vol_linear(v, x, y, z);
    return vol_cubic(v, x, y, z);
}
static float vol_eval(const OracleVolume* v, const float w[3]) {
    float p[3];
    vol_to_object(v, w, p);
    return vol_sample(v, p[0], p[1], p[2]);
}
/* evalNormalImpl, :234-283: normalStep = 1 voxel, normalScale = 0.5 / voxelSize (volume_interpolation_grid.cpp:1097-1104) */
static void vol_normal(const OracleVolume* v, const float w[3], float n[3]) {
    float p[3];
    vol_to_object(v, w, p);
    for (int d = 0; d < 3; ++d) {
        const float voxel = v->boxSize[d] / (float)(v->newBehavior ? v->res[d] : v->res[d] - 1);
        const float scale = 0.5f / voxel;
        float a[3] = {p[0], p[1], p[2]}, b[3] = {p[0], p[1], p[2]};
        a[d] += 1.f;
        b[d] -= 1.f;
        n[d] = scale * (vol_sample(v, a[0], a[1], a[2]) - vol_sample(v, b[0], b[1], b[2]));
    }
}

void oracle_volume_eval_points(const OracleVolume* v, const float* pos, size_t count, float* out) {
    for (size_t i = 0; i < count; ++i) out[i] = vol_eval(v, pos + 3 * i);
}

/* render_pixel with the grid as the volume: no normals, no BRDF (the DVR loop ignores isInside like for networks) */
static void render_pixel_volume(const OracleVolume* v, const OracleScene* s, int W, int H, int x, int y, float px8[8],
                                unsigned long long* samples) {
    const float ndcx = 2 * ((float)x + 0.5f) / (float)W - 1, ndcy = 2 * ((float)y + 0.5f) / (float)H - 1;
    const float tanFovY = tanf(s->fovY / 2), tanFovX = tanFovY * ((float)W / (float)H);
    const float* eye = s->eye; const float* right = s->right; const float* up = s->up;
    const float front[3] = {up[1] * right[2] - up[2] * right[1], up[2] * right[0] - up[0] * right[2],
                            up[0] * right[1] - up[1] * right[0]};
    float dir[3];
    for (int i = 0; i < 3; ++i) dir[i] = front[i] + ndcx * tanFovX * right[i] + ndcy * tanFovY * up[i];
    const float il = 1.0f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    for (int i = 0; i < 3; ++i) dir[i] *= il;
    float tlo[3], thi[3];
    for (int i = 0; i < 3; ++i) {
        const float inv = 1.0f / dir[i];
        const float ta = (v->boxMin[i] - eye[i]) * inv, tb = (v->boxMin[i] + v->boxSize[i] - eye[i]) * inv;
        tlo[i] = fminf(ta, tb);
        thi[i] = fmaxf(ta, tb);
    }
    float tmin = fmaxf(fmaxf(tlo[0], tlo[1]), tlo[2]);
    const float tmax = fminf(fminf(thi[0], thi[1]), thi[2]);
    tmin = fmaxf(tmin, 0.f);
    const float alphaEarlyOut = 1.0f - 1e-5f;
    const float divRange = 1.0f / (s->densityMax - s->densityMin);
    float col[4] = {0, 0, 0, 0}, nacc[3] = {0, 0, 0}, depth = 0;
    float previousDensity = -1.f;
    const int normals = v->provideNormals || s->brdfPhong || s->brdfMagnitudeScaling || s->tfGaussianMode == 1; /* transfer_function_gaussian.cpp:271-272 */
    unsigned long long cnt = 0;
    for (int i = 0;; ++i) {
        const float t = tmin + (float)i * s->stepsize;
        const int valid = (t <= tmax) && (!s->earlyOut || col[3] < alphaEarlyOut);
        if (!valid) break;
        ++cnt;
        const float pos[3] = {eye[0] + dir[0] * t, eye[1] + dir[1] * t, eye[2] + dir[2] * t};
        const float value = vol_eval(v, pos);
        const float density2 = (value - s->densityMin) * divRange;
        float c[4] = {0, 0, 0, 0}, g[3] = {0, 0, 0};
        if (value >= s->densityMin) {
            if (normals) vol_normal(v, pos, g);
            if (s->tfPreintegration) tf_eval_preintegrated(s, density2, previousDensity, c);
            else tf_eval(s, density2, sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), previousDensity, c);
        }
        previousDensity = density2;
        if (c[3] > 0 && (s->brdfMagnitudeScaling || s->brdfPhong)) { /* BRDFLambert::eval, as in render_pixel */
            const float g2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            if (s->brdfMagnitudeScaling) c[3] *= 1.0f - expf(-s->brdfMagScale * g2);
            if (s->brdfPhong) {
                const float gradientNorm = 1.0f / sqrtf(g2);
                float nn[3] = {g[0], g[1], g[2]};
                if (g2 >= 1e-8f) { nn[0] *= gradientNorm; nn[1] *= gradientNorm; nn[2] *= gradientNorm; }
                float l[3];
                for (int k = 0; k < 3; ++k) l[k] = s->brdfLightType == 1 ? -s->brdfLight[k] : s->brdfLight[k] - pos[k];
                const float ill = 1.0f / sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
                l[0] *= ill; l[1] *= ill; l[2] *= ill;
                const float lo = s->brdfMagCenter - s->brdfMagRadius, hi = s->brdfMagCenter + s->brdfMagRadius;
                const float yy = clamp01((gradientNorm - lo) / (hi - lo));
                const float phongStrength = yy * yy * (3.0f - 2.0f * yy);
                const float ambientStrength = 1.0f + phongStrength * (s->brdfAmbient - 1.0f);
                const float nl = nn[0] * l[0] + nn[1] * l[1] + nn[2] * l[2];
                const float r[3] = {l[0] - 2 * nn[0] * nl, l[1] - 2 * nn[1] * nl, l[2] - 2 * nn[2] * nl};
                const float e = (float)s->brdfSpecularExponent;
                const float spec = (e + 2.0f) * 0.159155f * powf(fmaxf(0.f, dir[0] * r[0] + dir[1] * r[1] + dir[2] * r[2]), e);
                for (int k = 0; k < 3; ++k)
                    c[k] = ambientStrength * c[k] + (1 - ambientStrength) * (fabsf(nl) * c[k] + s->brdfSpecular * spec);
            }
        }
        if (c[3] > 0) {
            const float l2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            if (l2 >= 1e-8f) { const float k = 1.0f / sqrtf(l2); g[0] *= k; g[1] *= k; g[2] *= k; }
            const float a = s->blendMode == ORACLE_BLEND_BEER_LAMBERT ? 1.0f - expf(-c[3]) : fminf(1.0f, c[3]);
            const float w = (1 - col[3]) * a;
            for (int k = 0; k < 3; ++k) { col[k] += w * c[k]; nacc[k] += w * g[k]; }
            depth += w * t;
            col[3] += w;
        }
    }
    px8[0] = col[0]; px8[1] = col[1]; px8[2] = col[2]; px8[3] = col[3];
    for (int k = 0; k < 3; ++k) px8[4 + k] = nacc[k] * col[3];
    px8[7] = depth * col[3] / col[3];
    *samples = cnt;
}

int oracle_render_volume(const OracleVolume* v, const OracleScene* s, int W, int H, float* out8, unsigned long long* evaluatedSamples) {
    if (!v || !s || !v->data || W <= 0 || H <= 0) return -1;
    unsigned long long total = 0;
    const size_t plane = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float p[8];
            unsigned long long c;
            render_pixel_volume(v, s, W, H, x, y, p, &c);
            total += c;
            for (int k = 0; k < 8; ++k) out8[k * plane + (size_t)y * W + x] = p[k];
        }
    if (evaluatedSamples) *evaluatedSamples = total;
    return 0;
}

uint16_t oracle_float_to_half(float f) { return f2h(f); }
float oracle_half_to_float(uint16_t h) { return h2f(h); }
