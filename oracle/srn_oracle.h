/* srn_oracle.h -- CPU restatement of the fV-SRN hot path.  TEST INFRASTRUCTURE ONLY (see srn_oracle.c). */
#ifndef SRN_ORACLE_H_
#define SRN_ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* arithmetic models, see srn_oracle.c.  DEVICE = FLOAT + the MI355X kernels' statement of the Fourier stage (phases in revolutions
 * from a hi + lo fp16 split of the matrix, fp32 sum; with OracleScene::rotationResync the features advance along a ray by rotation) */
enum { ORACLE_ACC_HALF = 0, ORACLE_ACC_FLOAT = 1, ORACLE_ACC_DEVICE = 2, ORACLE_ACC_EXACT = 3 /* no fp16 rounding of inputs / activations */ };
enum { ORACLE_ACT_RELU = 0, ORACLE_ACT_SINE = 1, ORACLE_ACT_SNAKE = 2, ORACLE_ACT_SNAKEALT = 3, ORACLE_ACT_SIGMOID = 4 };
enum {
    ORACLE_OUT_DENSITY = 0, ORACLE_OUT_DENSITY_DIRECT = 1, ORACLE_OUT_RGBO = 2, ORACLE_OUT_RGBO_DIRECT = 3,
    ORACLE_OUT_DENSITY_GRADIENT = 4, ORACLE_OUT_DENSITY_GRADIENT_DIRECT = 5, ORACLE_OUT_DENSITY_GRADIENT_CUBIC = 6,
    ORACLE_OUT_DENSITY_CURVATURE = 7, ORACLE_OUT_DENSITY_CURVATURE_DIRECT = 8
};
enum { ORACLE_GRID_FLOAT = 0, ORACLE_GRID_BYTE_LINEAR = 1, ORACLE_GRID_BYTE_GAUSSIAN = 2 };
enum { ORACLE_TF_NONE = 0, ORACLE_TF_IDENTITY = 1, ORACLE_TF_GAUSSIAN = 2, ORACLE_TF_PIECEWISE = 3, ORACLE_TF_TEXTURE = 4 };
enum { ORACLE_BLEND_ALPHA = 0, ORACLE_BLEND_BEER_LAMBERT = 1 };

/* The reference's constant block, kernel::VolumeInterpolationTensorcoresParameters
 * (renderer/renderer_volume_tensorcores.cuh:195-249); all matrices are half bits. */
typedef struct {
    int C;           /* HIDDEN_CHANNELS = 4 + 2F (8 + 2F with direction) */
    int F;           /* NUM_FOURIER_FEATURES                             */
    int G;           /* latent grid channels (0 = none)                  */
    int NH;          /* NUM_HIDDEN_LAYERS (C x C)                        */
    int Cout;        /* last layer outputs: 1, 4 or 6 (curvature modes)  */
    int outputMode, activation, gridEncoding, passTime, accMode;
    int useDirection; /* USE_DIRECTION: 0 none, 1 extra inputs, 2 extra inputs + inside the Fourier matrix (6 columns) */
    float actParam;
    float boxMin[3], boxSize[3];
    const uint16_t* fourier; /* cWeightsFourier [3*F], feature-fastest   */
    const uint16_t* wFirst;  /* cWeightsLatentGrid [C][(C+G)] row-major; F == 0: cWeightsFirst [3|6][C] (output fastest) */
    const uint16_t* bFirst;  /* cBiasLatentGrid [C]; F == 0: cBiasFirst [C] */
    const uint16_t* wHidden; /* cWeightsHidden [NH][C][C] row-major      */
    const uint16_t* bHidden; /* cBiasHidden [NH][C]                      */
    const uint16_t* wLast;   /* cWeightsLast [C][Cout]                   */
    const uint16_t* bLast;   /* cBiasLast [Cout]                         */
    int gridX, gridY, gridZ;
    const void* const* gridTexA; /* G/4 textures, each [Z][Y][X][4] float | uint8 (cLatentGridA)  */
    const void* const* gridTexB; /* cLatentGridB                                                     */
    const float* gridOffsetA;    /* per channel [G] (cLatentGridOffsetA)                             */
    const float* gridScaleA;     /* per channel [G]                                                  */
    const float* gridInterpolation; /* per texture [G/4] (cLatentGridInterpolation.x)                */
} OracleNet;

typedef struct {
    float eye[3], right[3], up[3], fovY;
    float stepsize, densityMin, densityMax;
    int earlyOut, blendMode, tfKind, tfRows;
    float tfScaleAbsorption, tfScaleEmission;
    const float* tfTable;
    /* GRADIENT_MODE (renderer_volume_tensorcores.cuh:1166-1201): 0 off / direct, 1 finite differences, 2 adjoint method */
    int gradientMode;
    float fdStep;
    float gridDiffStep; /* adjoint: latentGridDifferencesStepSize, normalized coordinates */
    /* BRDFLambert (renderer_brdf_lambert.cuh:19-103) */
    int brdfMagnitudeScaling, brdfPhong, brdfLightType /* 0 point, 1 directional */, brdfSpecularExponent;
    float brdfMagScale, brdfAmbient, brdfSpecular, brdfMagCenter, brdfMagRadius;
    float brdfLight[3];
    /* TransferFunctionTexture pre-integration (renderer_tf_texture.cuh:55-93): 0 none, 1: table [R][4], 2: [R][R][4];
     * built by oracle_tf_preintegrate (transfer_function_texture_cuda.cu:9-90) */
    int tfPreintegration;
    const float* tfPreintegrated;
    /* TRANSFER_FUNCTION_GAUSSIAN__SCALE_WITH_GRADIENT (1) / __ANALYTIC (2), renderer_tf_gaussian.cuh:55-73; 0: neither */
    int tfGaussianMode;
    /* ORACLE_ACC_DEVICE only -- what fv-srn_amd/csrc/kernels.hpp (render_body) does to 32-wide Fourier-only networks: the input
     * features are derived from the fp16-rounded position every rotationResync steps of a depth segment and advanced by the
     * rotation of the per-step phase increment in between (0: derived at every step, like the reference); `segments` = the K
     * consecutive step ranges a ray was cut into (the step count restarts in each; <= 1: none).  fvsrn_scene_last_render_info
     * reports both for a render. */
    int rotationResync, segments;
    /* ... and (r04) rotationHiLo != 0: at a re-derivation the phase MFMA runs on the position AND on its fp16 rounding residual (x = hi + lo,
     * two fp16 values: ~22 bits), so the rotated features follow the fp32 sample positions instead of carrying the rounding of one
     * position (and of the step vector) coherently over rotationResync steps.  Not used with rotationResync = 1 (the reference's arithmetic). */
    int rotationHiLo;
} OracleScene;
/* tex [R][4] -> out [R][4] (mode 1) or [R][R][4] (mode 2, N quadrature steps, world step size) */
/* EvaluateTF / EvaluateTFWithPrevious (renderer_tf_kernels.cuh:11-70) with the scene's TF, density range and step size;
 * previous == NULL: no previous density */
void oracle_tf_evaluate(const OracleScene* s, const float* density, const float* previous, size_t n, float* out4);
void oracle_tf_preintegrate(const float* tex, int R, int mode, float stepsize, int N, float* out);

int oracle_eval_points(const OracleNet* n, const float* worldPos, const float* directions /* or NULL */, size_t count, float* out);
/* all raw outputs of eval<>: out[count][9] = value[4], normal[3], curvature[2] */
int oracle_eval_points_full(const OracleNet* n, const float* worldPos, const float* directions /* or NULL */, size_t count, float* out9);
/* _evalNormalAdjoint (:1198-1540): out3[count][3] = d(un-clamped density) / d(normalized position) */
int oracle_eval_adjoint(const OracleNet* n, const float* worldPos, const float* directions /* or NULL */, size_t count, float gridStep, float* out3);
int oracle_render(const OracleNet* n, const OracleScene* s, int W, int H, int y0, int y1, float* out8,
                  unsigned long long* evaluatedSamples);
unsigned long long oracle_count_samples(const OracleNet* n, const OracleScene* s, int W, int H, int y0, int y1);
/* Dense grid volumes: kernel::VolumeInterpolationGrid (renderer/renderer_volume_grid.cuh:89-232) behind the same DVR loop.
 * data: fp32, index x + X (y + Y z) (Volume::MipmapLevel::idx, volume.h:126-132). */
typedef struct {
    const float* data;
    int res[3];
    float boxMin[3], boxSize[3];
    int interpolation; /* 0 nearest, 1 trilinear, 2 tricubic (VOLUME_INTERPOLATION_GRID__INTERPOLATION) */
    int source;        /* 0: CUDA texture addressing (VolumeSource::VOLUME), 1: tensor accessor (VolumeSource::TORCH_TENSOR) */
    int newBehavior;   /* !VOLUME_INTERPOLATION_GRID__GRID_RESOLUTION_OLD_BEHAVIOR */
    int provideNormals; /* VOLUME_INTERPOLATION_GRID__REQUIRES_NORMAL (shading BRDF, normal channel) */
} OracleVolume;
void oracle_volume_eval_points(const OracleVolume* v, const float* worldPos, size_t count, float* out);
int oracle_render_volume(const OracleVolume* v, const OracleScene* s, int W, int H, float* out8, unsigned long long* evaluatedSamples);

uint16_t oracle_float_to_half(float f);
float oracle_half_to_float(uint16_t h);

#ifdef __cplusplus
}
#endif
#endif
