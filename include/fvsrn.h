/*
 * fvsrn.h -- C ABI of the MI355X-native fV-SRN inference renderer (libfvsrn.so).
 *
 * This is the drop-in boundary for ONE hot path of shamanDevel/fV-SRN: the fused
 * SRN-MLP + DVR ray-stepping renderer.  Every entry point names
 * the reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C: pointers, sizes, PODs; no torch / pybind / C++ types cross this line
 *   - every function returns FVSRN_OK (0) or a negative error code; the text of the
 *     last error of the calling thread is available from fvsrn_last_error()
 *     (reference: C++ exceptions -> Python RuntimeError, renderer/kernel_loader.cpp:19-31)
 *   - pointers named d_* are DEVICE pointers (HIP), h_* / unprefixed are host pointers
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all device work is
 *     enqueued asynchronously on it, like the reference uses the current torch stream
 *     (renderer/iimage_evaluator.cpp:167-170)
 *   - handles are re-entrant per object; distinct objects may be used from distinct threads
 *   - a handle's device state (weight image, latent key frames, TF tables) is created at its first use on the HIP device
 *     that is current then; every later call must run with that device current (FVSRN_ERR_WRONG_DEVICE otherwise) --
 *     one handle per device, the reference's "one renderer per CUDA context".  A network may be used from several
 *     streams: its first use (upload of the weight images and key frames) and every time change (blend into a working grid)
 *     are enqueued on the stream of the call that triggers them, and calls on other streams wait for them by events; a blend
 *     waits for every kernel that still reads the working grid it overwrites (r03; before, callers had to synchronise)
 *   - nothing on a per-frame path reads the process environment; the FVSRN_* variables only seed the option defaults once
 */
#ifndef FVSRN_H_
#define FVSRN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVSRN_OK 0
#define FVSRN_ERR_INVALID_ARGUMENT (-1) /* bad pointer / size / enum                       */
#define FVSRN_ERR_FORMAT (-2)           /* malformed or unsupported .volnet bytes          */
#define FVSRN_ERR_INVALID_NETWORK (-3)  /* SceneNetwork::valid() failed                    */
#define FVSRN_ERR_UNSUPPORTED (-4)      /* outside the ahead-of-time compiled variant set  */
#define FVSRN_ERR_DEVICE (-5)           /* HIP runtime error                               */
#define FVSRN_ERR_NO_DEVICE (-6)        /* no GPU / kernels not usable                     */
#define FVSRN_ERR_IO (-7)               /* file cannot be opened / written                   */
#define FVSRN_ERR_WRONG_DEVICE (-8)     /* the handle's device state lives on another HIP device than the current one (exercised on one-GPU boxes
                                          through FVSRN_DEBUG_DEVICE_SKEW: tests/test_gpu_parity.py test_wrong_device_check_fires_on_one_gpu) */

/* activation of the hidden layers: renderer/volume_interpolation_network.cpp:223-230 */
typedef enum {
    FVSRN_ACT_RELU = 0, FVSRN_ACT_SINE = 1, FVSRN_ACT_SNAKE = 2,
    FVSRN_ACT_SNAKEALT = 3, FVSRN_ACT_SIGMOID = 4, FVSRN_ACT_NONE = 5
} fvsrn_activation;

/* output parametrization: renderer/volume_interpolation_network.cpp:165-181 */
typedef enum {
    FVSRN_OUT_DENSITY = 0, FVSRN_OUT_DENSITY_DIRECT = 1, FVSRN_OUT_RGBO = 2, FVSRN_OUT_RGBO_DIRECT = 3,
    FVSRN_OUT_DENSITY_GRADIENT = 4, FVSRN_OUT_DENSITY_GRADIENT_DIRECT = 5, FVSRN_OUT_DENSITY_GRADIENT_CUBIC = 6,
    FVSRN_OUT_DENSITY_CURVATURE = 7, FVSRN_OUT_DENSITY_CURVATURE_DIRECT = 8
} fvsrn_output_mode;

/* latent grid encodings: renderer/volume_interpolation_network.h (LatentGrid::Encoding) */
typedef enum { FVSRN_GRID_FLOAT = 0, FVSRN_GRID_BYTE_LINEAR = 1, FVSRN_GRID_BYTE_GAUSSIAN = 2 } fvsrn_grid_encoding;

/* transfer functions: renderer/renderer_tf_{identity,gaussian,piecewise,texture}.cuh */
typedef enum {
    FVSRN_TF_NONE = 0,      /* network emits rgbo directly (RAY_EVALUATION_STEPPING__SKIP_TRANSFER_FUNCTION) */
    FVSRN_TF_IDENTITY = 1,  /* table unused; tf_scale_absorption / tf_scale_emission                          */
    FVSRN_TF_GAUSSIAN = 2,  /* table (R,6): r,g,b,opacity*absorptionScaling,mean,sigma                        */
    FVSRN_TF_PIECEWISE = 3, /* table (R,5): r,g,b,absorption,position (sorted)                                */
    FVSRN_TF_TEXTURE = 4    /* table (R,4): rgba texels, sampled as d*R-0.5 lerp (tensor mode)                */
} fvsrn_tf_kind;

/* blending: renderer/renderer_blending.cuh:16-52 */
typedef enum { FVSRN_BLEND_ALPHA = 0, FVSRN_BLEND_BEER_LAMBERT = 1 } fvsrn_blend_mode;

typedef struct fvsrn_network fvsrn_network; /* SceneNetwork + its device image          */
typedef struct fvsrn_scene fvsrn_scene;     /* camera + DVR + TF + blending parameters   */

/* ----------------------------------------------------------------------------------------
 * library
 * -------------------------------------------------------------------------------------- */
/* text of the last error raised on the calling thread ("" if none) */
const char* fvsrn_last_error(void);
/* "fvsrn <version> gfx950" */
const char* fvsrn_version(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int fvsrn_device_count(void);
/* How many of these streams run side by side on the current device?  Puts one wave that spins for `microseconds` on each of the
 * `streams` (2 .. 16) streams -- the caller's own (stream_handles, HIP stream handles) or, with stream_handles == NULL, freshly created ones --
 * and returns streams x microseconds / elapsed time in *concurrent (1.0: one after the other; = streams: all at once).  ROCm maps all
 * streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4; read when the HIP runtime starts) and two streams that share a
 * queue run in submission order: the frame pipeline of a rank checks here whether ITS render and collective streams got queues of their
 * own instead of trusting the environment (measured r04: n fresh streams on q queues give n / ceil(n / q); inside a process that has
 * initialised RCCL six fresh streams ran 2.8-wide even with GPU_MAX_HW_QUEUES=8).  Use >= 2000 microseconds (300: launch overhead shows).
 * Synchronises the device.  No reference counterpart. */
int fvsrn_probe_stream_concurrency(void* const* stream_handles, int streams, int microseconds, float* concurrent);

/* ----------------------------------------------------------------------------------------
 * SceneNetwork  (replaces renderer::SceneNetwork, renderer/volume_interpolation_network.{h,cpp})
 * -------------------------------------------------------------------------------------- */
/* SceneNetwork::load(std::istream&)   volume_interpolation_network.cpp:1059-1086 */
int fvsrn_network_create_from_volnet(const void* bytes, size_t len, fvsrn_network** out);
/* SceneNetwork()                      volume_interpolation_network.cpp:798-804 */
int fvsrn_network_create(fvsrn_network** out);
void fvsrn_network_destroy(fvsrn_network* net);

/* InputParametrization fields + setFourierMatrixFromTensor   :129-156
 * matrix is row-major (num_fourier, 3|6) fp32; multiplied by 2*pi unless premultiplied. */
int fvsrn_network_set_input(fvsrn_network* net, int has_time, int has_direction,
                            const float* fourier_matrix, int num_fourier, int fourier_cols,
                            int premultiplied);
/* OutputParametrization::outputMode   :183-200 */
int fvsrn_network_set_output_mode(fvsrn_network* net, fvsrn_output_mode mode);
/* SceneNetwork::addLayerFromTorch     :896-921 (weights (out,in) row-major fp32, bias (out)) */
int fvsrn_network_add_layer(fvsrn_network* net, const float* weights, const float* bias,
                            int channels_out, int channels_in, fvsrn_activation act, float act_param);
/* SceneNetwork::setBoxMin/setBoxSize */
int fvsrn_network_set_box(fvsrn_network* net, const float box_min[3], const float box_size[3]);
/* LatentGridTimeAndEnsemble(time_min,time_num,time_step,ensemble_min,ensemble_num)   .h:328-338 */
int fvsrn_network_set_latent_grid_layout(fvsrn_network* net, int time_min, int time_num, int time_step,
                                         int ensemble_min, int ensemble_num);
/* setTimeGridFromTorch / setEnsembleGridFromTorch  :616-630; grid is (C,Z,Y,X) fp32.
 * *encoding_error receives the average absolute encoding error (LastEncodingError). */
int fvsrn_network_set_latent_grid(fvsrn_network* net, int is_ensemble, int index, const float* grid,
                                  int C, int Z, int Y, int X, fvsrn_grid_encoding enc,
                                  double* encoding_error);
/* SceneNetwork::valid()               :940-985   (1 valid, 0 invalid + message) */
int fvsrn_network_valid(const fvsrn_network* net);
/* SceneNetwork::save(std::ostream&)   :1088-1104. Call with buf==NULL to query *len. */
int fvsrn_network_save_volnet(const fvsrn_network* net, void* buf, size_t cap, size_t* len);
/* SceneNetwork::setTimeAndEnsemble    :923-938 (clamps silently) */
int fvsrn_network_set_time_and_ensemble(fvsrn_network* net, float time, int ensemble);
/* Brings the device state of a network up to date on `stream` without launching anything else: uploads at first use, and -- after
 * fvsrn_network_set_time_and_ensemble -- the blend of the selected key frames into a working grid (what the next evaluate / render
 * call would do first).  A frame pipeline calls it on a side stream as soon as the NEXT frame's time is known, so that the blend
 * runs beside the current frame's render instead of between two renders (with two working grids, FVSRN_OPT_WORKING_GRIDS; the
 * render on another stream waits for it by an event).  The reference uploads / re-fills at the start of render()
 * (volume_interpolation_network.cpp:923-938,1308-1328). */
int fvsrn_network_prepare(fvsrn_network* net, void* stream);

/* SceneNetwork::clearGPUResources     :1106-1112 */
int fvsrn_network_clear_gpu_resources(fvsrn_network* net);

typedef struct {
    int num_layers;          /* hidden_.size() (all Linear layers incl. the last)          */
    int hidden_channels;     /* HIDDEN_CHANNELS of getDefines()                             */
    int num_fourier;         /* NUM_FOURIER_FEATURES                                        */
    int has_direction, has_time, use_direction_in_fourier;
    int output_mode;         /* fvsrn_output_mode                                           */
    int output_channels;     /* OutputModeNumChannelsOut: 1 or 4                            */
    int activation;          /* fvsrn_activation of the hidden layers                       */
    float activation_param;
    int grid_channels;       /* total latent channels (0 = none)                            */
    int grid_encoding;
    int grid_res[3];         /* X,Y,Z of the first grid                                     */
    int time_num, ensemble_num;
    int num_parameters;      /* SceneNetwork::numParameters()  :1043-1055                   */
    int max_warps_shared, max_warps_mixed; /* computeMaxWarps(true/false,false) :987-1041   */
    double flops_per_sample; /* algorithmic FLOPs, SURVEY.md 8(d)                           */
    double mfma_flops_per_sample; /* padded FLOPs actually issued to the matrix cores       */
    float box_min[3], box_size[3];
} fvsrn_network_info;
int fvsrn_network_get_info(const fvsrn_network* net, fvsrn_network_info* info);
/* Layer i as stored after addLayer (weights half bits, rows=out, cols=in): for round-trip tests
 * (unittests/testSRN.cpp:413-430). Pass NULL buffers to query sizes. */
int fvsrn_network_get_layer(const fvsrn_network* net, int index, int* channels_out, int* channels_in,
                            int* activation, float* act_param, uint16_t* weights, uint16_t* bias);
/* Fourier matrix as stored (half bits, feature-fastest [cin*F + f]); returns F*cols entries */
int fvsrn_network_get_fourier(const fvsrn_network* net, uint16_t* matrix, int cap, int* count);

/* ----------------------------------------------------------------------------------------
 * Tuning / developer options of a handle.  No reference counterpart (the reference takes its kernel choices from NVRTC
 * #defines); they replace what used to be FVSRN_* environment variables read on every call.  A new handle starts from the
 * process defaults, which are read from the environment ONCE (variable named per option below).
 * -------------------------------------------------------------------------------------- */
typedef enum {
    FVSRN_OPT_SMALL_KERNEL = 0,      /* network + scene: 0 = never take the register-resident kernels, -1 auto  [FVSRN_SMALL_KERNEL]  */
    FVSRN_OPT_PERSISTENT = 1,        /* scene: persistent render waves 0 / 1; -1 auto = 1 for whole frames, 0 (bounded waves) for the stripes of
                                        a multi-GPU frame: persistent stripes (with FVSRN_OPT_PERSISTENT_RESERVE slots left free) pay only when
                                        the streams of the process run side by side -- fvsrn_probe_stream_concurrency measures that, the
                                        frame pipeline (tiles.StripeRenderer) opts in on its result              [FVSRN_PERSISTENT]    */
    FVSRN_OPT_DEPTH_SEGMENTS = 2,    /* scene: cut rays into k step ranges composited afterwards, 0 auto       [FVSRN_SEGMENTS]      */
    FVSRN_OPT_FOURIER_RESYNC = 3,    /* scene: exact Fourier features every k steps (power of two), 0 = 64; 1 = the reference's
                                        per-step arithmetic (fp16 position at every sample), no feature rotation; k > 1: derived from the
                                        fp32 position (two fp16 operands) and rotated in between               [FVSRN_FOURIER_RESYNC] */
    FVSRN_OPT_UNIT_QUOTA = 4,        /* scene: work units per bounded wave of a stripe launch, -1 auto          [FVSRN_UNIT_QUOTA]    */
    FVSRN_OPT_TILE_ORDER = 5,        /* scene: 1 = centre-first tile order, 0 = raster, -1 auto                 [FVSRN_TILE_ORDER]    */
    FVSRN_OPT_WAVES_PER_BLOCK = 6,   /* network (evaluate) + scene (render): 1 / 2 / 4, 0 auto                  [FVSRN_WAVES_PER_BLOCK] */
    FVSRN_OPT_MAX_BLOCKS_PER_CU = 7, /* scene: occupancy cap for experiments, 0 = none                          [FVSRN_MAX_BLOCKS_PER_CU] */
    FVSRN_OPT_RELU_CLAMP = 8,        /* network: 0 = do not use the [0,1]-scaled ReLU weight image              [FVSRN_DISABLE_RELU_CLAMP] */
    FVSRN_OPT_KEYFRAME_SLOTS = 9,    /* network: time key frames of the latent grid resident in HBM; 0 = all, k >= 2 = at most k, the
                                        others are streamed from pinned host memory on a copy stream, overlapped with rendering
                                        (k >= 3: prefetched one key-frame interval ahead)                      [FVSRN_KEYFRAME_SLOTS] */
    FVSRN_OPT_WORKING_GRIDS = 10,    /* network: blended fp16 working grids (what the kernels read); 0 = auto: 2 for a network with several
                                        key frames -- a time change then blends into the grid the frame in flight does not read, so
                                        two frames at different times may be in flight on two streams --, 1 otherwise  [FVSRN_WORKING_GRIDS] */
    FVSRN_OPT_OVERLAP_KERNEL = 11,   /* scene: 48 / 64-wide latent-grid networks: 1 = render with the gather kernel (fragment-major layer order, no
                                        register spills) also where the cell table would be taken; 0 / -1 = by the footprint rule
                                        (r03 - r04: a separate spill-free kernel variant on request; r05: THE gather kernel)   [FVSRN_OVERLAP_KERNEL] */
    FVSRN_OPT_PERSISTENT_RESERVE = 12, /* scene: persistent launches leave this many workgroup slots of the chip unused, so that a kernel on
                                        another stream (the all-gather of the previous frame) finds room while the frame renders; -1 auto
                                        (0 for whole frames, 1/16 of the slots for the stripes of a multi-GPU frame)  [FVSRN_PERSISTENT_RESERVE] */
    FVSRN_OPT_CELL_TABLE = 13,       /* network + scene: latent grids of FLOAT / BYTE_LINEAR encoding enter the unshaded renderer through the cell
                                        table (one MFMA K step on the trilinear weights instead of gathers + dot products; DESIGN.md section 4
                                        item 16); 0 = off (the gather path; network: no table is built), 1 = on where a kernel variant exists,
                                        -1 (scene default) = on while an 8 x 8 pixel tile spans less than ~0.6 - 0.8 grid cells at the box
                                        centre, where it pays (measured: tools/dev/cell_footprint_sweep.py)            [FVSRN_CELL_TABLE]    */
    FVSRN_OPT_COUNT_ = 14
} fvsrn_option;
int fvsrn_network_set_option(fvsrn_network* net, int option, int value);
int fvsrn_network_get_option(const fvsrn_network* net, int option, int* value);
int fvsrn_scene_set_option(fvsrn_scene* scene, int option, int value);
int fvsrn_scene_get_option(fvsrn_scene* scene, int option, int* value);
/* time key frames of a network's latent grid: out = { key frames, device slots, uploads, of which a blend had to wait for
 * (on demand), prefetched ahead of need, bytes uploaded } since the device state was created */
int fvsrn_network_keyframe_stats(const fvsrn_network* net, unsigned long long out[6]);
/* cell tables of a network's latent grid (FVSRN_OPT_CELL_TABLE): out = { bytes of one table (0: this network has none -- no grid, BYTE_GAUSSIAN, a
 * resolution below 2, above the size cap, option 0), builds of the table of the image the unshaded renderer runs, builds of the shaded renderer's table
 * (its own since r06: corner vectors of the plain image, where the unshaded renderer's holds monomial coefficients), bytes of table memory resident now } since the device state was created.  Tables are
 * built by the first launch that goes through them and then with every key-frame blend for as long as the launches do; a network whose launches take
 * the gathers (footprint rule, adjoint mode, evaluate_points) never allocates one. */
int fvsrn_network_cell_table_stats(const fvsrn_network* net, unsigned long long out[4]);

/* ----------------------------------------------------------------------------------------
 * IVolumeInterpolation::evaluate   (renderer/volume_interpolation.cpp:26-127, kernel
 * EvaluateNoBatches renderer/renderer_volume_kernels1.cuh:15).
 * d_positions (n,3) fp32, d_directions (n,3) or NULL, d_out (n, output_channels) fp32.
 * Like the reference -- which resets the box to [0,1]^3 around this call (:46-49) -- positions are
 * in UNIT-BOX coordinates unless FVSRN_EVAL_WORLD_POSITIONS is set, in which case they are world
 * positions and mapped through the network's box_min / box_size as the renderer does.
 * -------------------------------------------------------------------------------------- */
#define FVSRN_EVAL_WORLD_POSITIONS 1
/* gradient-predicting networks (densitygrad, densitygrad:direct, densitygrad:cubic; also the curvature modes): d_out is (n,4) =
 * value + the predicted gradient, i.e. eval() + evalNormal() in GRADIENT_MODE_OFF_OR_DIRECT
 * (renderer_volume_tensorcores.cuh:1166-1183; IVolumeInterpolation::evaluateWithGradient, volume_interpolation.cpp:128-243) */
#define FVSRN_EVAL_WITH_PREDICTED_GRADIENT 2
/* networks that also predict the curvature (densitycurvature, densitycurvature:direct): d_out is (n,6) = value, predicted gradient,
 * the two predicted curvature values, i.e. eval() + evalNormal() + evalCurvature() (renderer_volume_tensorcores.cuh:1104-1133,1541-1556;
 * IVolumeInterpolation::evaluateWithGradientAndCurvature, volume_interpolation.cpp:245-360).  Other networks: FVSRN_ERR_INVALID_ARGUMENT
 * (the reference traps) */
#define FVSRN_EVAL_WITH_PREDICTED_CURVATURE 4
int fvsrn_evaluate_points(fvsrn_network* net, const float* d_positions, const float* d_directions,
                          size_t n, float* d_out, int flags, void* stream);
/* The same call with fp16 tensors: d_positions_f16 (n,3), d_directions_f16 (n,3) or NULL, d_out_f16 (n, output_channels | 4) -- 8 instead of
 * 16 bytes per point of a scalar network.  The reference's kernel rounds the normalized position and the direction to half before its first
 * layer (renderer_volume_tensorcores.cuh:770-772, 785-787, 813-817) and its network output is a half; with the unit box of this call a
 * caller that holds its positions in fp16 gets bit for bit the network inputs of the fp32 call, and gives up only the bits of the fp32
 * output parametrization that do not fit a half.  The reference's entry point dispatches on the tensor's scalar type, float or double
 * (volume_interpolation.cpp:40-42, KERNEL_DOUBLE_PRECISION :57-59); this is the same dispatch for a third type.  flags: FVSRN_EVAL_WORLD_POSITIONS, FVSRN_EVAL_WITH_PREDICTED_GRADIENT. */
int fvsrn_evaluate_points_half(fvsrn_network* net, const void* d_positions_f16, const void* d_directions_f16,
                               size_t n, void* d_out_f16, int flags, void* stream);
/* IVolumeInterpolation::evaluateWithGradient (volume_interpolation.cpp:128-243) of a VolumeInterpolationNetwork in
 * GRADIENT_MODE_ADJOINT_METHOD: d_out4 is (n,4) = value (as fvsrn_evaluate_points) + the analytic gradient of output 0 w.r.t. the
 * normalized (unit-box) position (evalNormal, renderer_volume_tensorcores.cuh:1198-1540; computed in forward mode,
 * fv-srn_amd/csrc/srn_gradient.hpp).  Latent grids are differentiated by central differences with step adjoint_grid_stepsize in
 * unit-box coordinates (0: 1 / (4 * grid resolution), volume_interpolation_network.cpp:1808-1812).  Scalar networks only.
 * flags: FVSRN_EVAL_WORLD_POSITIONS. */
int fvsrn_evaluate_points_adjoint(fvsrn_network* net, const float* d_positions, const float* d_directions, size_t n, float* d_out4,
                                  float adjoint_grid_stepsize, int flags, void* stream);

/* ----------------------------------------------------------------------------------------
 * scene = ImageEvaluatorSimple + CameraOnASphere(->reference frame) + RayEvaluationSteppingDvr
 *         + TF + BRDFLambert + Blending, as ONE POD
 *   renderer/image_evaluator_simple.cpp:198-361, renderer/camera.cpp:458-551,
 *   renderer/ray_evaluation_stepping.cpp:535-623, renderer/transfer_function*.cpp, blending.cpp
 * -------------------------------------------------------------------------------------- */
typedef struct {
    /* kernel::CameraReferenceFrameParameters (renderer/renderer_camera.cuh:12-24):
     * matrix rows = eye, right, up */
    float cam_eye[3], cam_right[3], cam_up[3];
    float fov_y_radians;
    /* kernel::RayEvaluationSteppingDvrParameters (renderer_ray_evaluation_stepping_dvr.cuh:22-30) */
    float stepsize;       /* world units */
    float density_min, density_max;
    int early_out;        /* RAY_EVALUATION_STEPPING__ENABLE_EARLY_OUT, alphaEarlyOut = 1-1e-5 */
    int blend_mode;       /* fvsrn_blend_mode */
    /* transfer function */
    int tf_kind;          /* fvsrn_tf_kind */
    float tf_scale_absorption, tf_scale_emission; /* Identity */
    const float* tf_table; /* host pointer, (tf_rows, cols(kind)) row-major; copied */
    int tf_rows;
    /* VolumeInterpolationNetwork gradient mode (renderer_volume_tensorcores.cuh:1166-1201, GRADIENT_MODE):
     * FVSRN_GRADIENT_OFF_OR_DIRECT: normals only from networks that predict them; FVSRN_GRADIENT_FINITE_DIFFERENCES:
     * central differences of the (un-clamped) network value, 6 extra evaluations per sample, world step
     * finite_differences_stepsize.  FVSRN_GRADIENT_ADJOINT_METHOD (:1198-1540): the analytic gradient w.r.t. the NORMALIZED
     * position of density networks (colour networks: none, like the reference); computed in forward mode in the same MFMA pass
     * (fv-srn_amd/csrc/srn_gradient.hpp), the latent grid by central differences with step adjoint_grid_stepsize in unit-box
     * coordinates (0 = the reference's default 1 / (grid resolution * 4), volume_interpolation_network.cpp:1809-1812). */
    int gradient_mode;
    float finite_differences_stepsize;
    /* BRDFLambert (renderer/renderer_brdf_lambert.cuh:56-103, host renderer/brdf.cpp:413-508); all zero = pass-through */
    int brdf_enable_magnitude_scaling, brdf_enable_phong;
    float brdf_magnitude_scaling, brdf_ambient, brdf_specular, brdf_magnitude_center, brdf_magnitude_radius;
    int brdf_specular_exponent;
    int brdf_light_type;        /* fvsrn_light_type */
    float brdf_light[3];        /* light position (POINT) or direction (DIRECTIONAL); "light follows camera" is the
                                   caller's job: camera origin / front (brdf.cpp:490-508) */
    /* TransferFunctionTexture::PreintegrationMode (transfer_function.h, device renderer_tf_texture.cuh:55-93): only with
     * FVSRN_TF_TEXTURE and a 256-texel table; the tables are built on the device (transfer_function_texture_cuda.cu:9-90,
     * 256 entries / 256 x 256 entries with 256 quadrature steps) whenever the table or the step size changes */
    int tf_preintegration;      /* fvsrn_tf_preintegration */
    float adjoint_grid_stepsize; /* FVSRN_GRADIENT_ADJOINT_METHOD: see gradient_mode */
    /* TransferFunctionGaussian's two variants (renderer_tf_gaussian.cuh:55-73; host flags scaleWithGradient_ /
     * usePiecewiseAnalyticIntegration_, transfer_function_gaussian.cpp:238-239,293-303 -- mutually exclusive there too):
     * SCALE_WITH_GRADIENT multiplies every sigma by max(1e-5, 0.1 |gradient|) with the gradient the volume provides (gradient_mode,
     * or a gradient-predicting network; none: |gradient| = 0); ANALYTIC integrates every Gaussian between the previous and the
     * current sample's density in closed form (erf), falling back to the point value for the first sample of a ray and equal densities */
    int tf_gaussian_mode;       /* fvsrn_tf_gaussian_mode, FVSRN_TF_GAUSSIAN only */
} fvsrn_scene_desc;
typedef enum { FVSRN_TF_GAUSSIAN_PLAIN = 0, FVSRN_TF_GAUSSIAN_SCALE_WITH_GRADIENT = 1, FVSRN_TF_GAUSSIAN_ANALYTIC = 2 } fvsrn_tf_gaussian_mode;
typedef enum { FVSRN_PREINTEGRATE_NONE = 0, FVSRN_PREINTEGRATE_1D = 1, FVSRN_PREINTEGRATE_2D = 2 } fvsrn_tf_preintegration;
typedef enum { FVSRN_GRADIENT_OFF_OR_DIRECT = 0, FVSRN_GRADIENT_FINITE_DIFFERENCES = 1, FVSRN_GRADIENT_ADJOINT_METHOD = 2 } fvsrn_gradient_mode;
typedef enum { FVSRN_LIGHT_POINT = 0, FVSRN_LIGHT_DIRECTIONAL = 1 } fvsrn_light_type;

/* sizeof(fvsrn_scene_desc) / sizeof(fvsrn_network_info) of the library: lets an FFI binding verify its struct mirrors */
size_t fvsrn_scene_desc_size(void);
size_t fvsrn_network_info_size(void);
int fvsrn_scene_create(const fvsrn_scene_desc* desc, fvsrn_scene** out);
int fvsrn_scene_update(fvsrn_scene* scene, const fvsrn_scene_desc* desc);
void fvsrn_scene_destroy(fvsrn_scene* scene);

/* CameraOnASphere -> reference frame (renderer/camera.cpp:458-490,553-581); orientation 0..5 =
 * Xp,Xm,Yp,Ym,Zp,Zm. Pure host math in fp64 like the reference, results cast to fp32. */
int fvsrn_camera_on_a_sphere(int orientation, const double center[3], double pitch, double yaw,
                             double distance, float eye[3], float right[3], float up[3]);

/* How the last fvsrn_render / fvsrn_render_stripes of this scene treated the samples of a ray -- for callers that restate the
 * arithmetic (the parity oracle): out = { depth segments K a ray was cut into (1: none; the step count of the feature rotation
 * restarts in each), period in steps of the exact re-derivation of rotated Fourier features (0: the kernel derives the features
 * from the fp16 position at every step, like the reference; FVSRN_OPT_FOURIER_RESYNC), kernel family (0 render_kernel / render_shaded_kernel, 1 register-resident, 2 spill-free
 * stripe variant, 3 render_adjoint_kernel: the adjoint gradient mode up to 64 channels, 4 register-resident with the latent grid through
 * the cell table, 5 render_cells_kernel: render_kernel with the latent grid through the cell table, FVSRN_OPT_CELL_TABLE),
 * waves per workgroup }.  No reference counterpart. */
int fvsrn_scene_last_render_info(fvsrn_scene* scene, int out[4]);
/* The kernel the last fvsrn_render / fvsrn_render_stripes of this scene launched, spelled like rocprofv3 lists it (family + template
 * arguments; "" before the first render).  fvsrn_network_kernel_name only knows the network: whether a latent grid goes through the cell
 * table or the gathers is decided per launch (image size, camera distance: FVSRN_OPT_CELL_TABLE), so profiles and bench lines quote THIS
 * name.  No reference counterpart (the reference's kernel name is fixed: "ImageEvaluatorSimpleKernel", image_evaluator_simple.cpp:300). */
int fvsrn_scene_last_kernel_name(fvsrn_scene* scene, char* buf, size_t cap);
/* Text report for a watchdog (tests/conftest.py asks when a test runs longer than five minutes): every live scene handle with its last launch --
 * kernel, grid, work units, launch shape, whether its stream is still busy and the values of its two device work counters (persistent waves take their
 * tiles from them; a kernel that never finishes shows as a busy stream with a counter below `units`).  Never blocks: handles whose mutex is held are
 * reported as locked, device memory is read by an asynchronous copy polled for at most a second.  May be called from any thread.  No reference counterpart. */
int fvsrn_debug_state(char* buf, size_t cap);

/* ImageEvaluatorSimple::render  (renderer/image_evaluator_simple.cpp:198-361, kernel
 * ImageEvaluatorSimpleKernel renderer/renderer_image_evaluator_simple.cuh:36-127).
 * Renders image rows [y0,y1) of a width x height frame (aspect = width/height) into d_out8,
 * a planar fp32 image [8][height][width] (rgb, alpha, normal xyz, depth); rows outside
 * [y0,y1) are not touched (multi-GPU row stripes).
 * d_stats (optional, may be NULL): 2 x uint64 on the device, ATOMICALLY incremented by
 *   [0] lane-exact evaluated samples, [1] wave-granular executed samples (64 per wave step). */
int fvsrn_render(fvsrn_scene* scene, fvsrn_network* net, int width, int height, int y0, int y1,
                 float* d_out8, unsigned long long* d_stats, void* stream);

/* Multi-GPU row stripes (no reference counterpart): rank r of `world` renders the image rows
 * y with (y / stripe_rows) % world == r -- round-robin stripes balance empty and dense image regions -- into
 * a COMPACT planar image d_out_local [8][fvsrn_stripe_rows(...)][width] whose rows are the owned rows in
 * increasing y.  Equal-sized compact images of all ranks are exchanged by one RCCL all-gather.
 * stripe_rows must be a multiple of 8 (the pixel tile of one wave). */
int fvsrn_stripe_rows(int height, int stripe_rows, int rank, int world);
int fvsrn_render_stripes(fvsrn_scene* scene, fvsrn_network* net, int width, int height, int stripe_rows, int rank,
                         int world, float* d_out_local, unsigned long long* d_stats, void* stream);
/* A sequence of frames in one call (no reference counterpart; the reference's sequence renders -- applications/volnet/eval_NetworkConfigsGrid.py:100-140:
 * one ImageEvaluatorSimple::render per camera of a rotation -- pay one Python call, one scene update and one launch set-up per frame, which is a
 * third of a rank's 0.26 ms share of the headline frame at world 8).  Frame f takes camera cameras9[f] = { eye, right, up } (host memory, like
 * fvsrn_scene_desc), the time times[f] if `times` is non-NULL (fvsrn_network_set_time_and_ensemble with the network's current ensemble), and is
 * rendered like fvsrn_render_stripes into d_out_local + f * 8 * rows * width (rows = fvsrn_stripe_rows(...); world = 1: whole frames).  Frames that
 * share their time (times == NULL) are rendered SEVERAL PER LAUNCH -- up to eight poses, a work unit is (frame, pixel tile), handed out by the same
 * device counter as the tiles of one frame: a rank's 1/8 share of a frame is a small launch whose ramp-up, longest tile and gap to the next launch cost
 * a fifth of its time, eight shares are a whole frame's worth of work -- in groups of ceil(frames / lanes) that the lanes (scenes[lane] on
 * streams[lane]) take in turn; with per-frame times every frame is its own launch on lane f % lanes.  Two lanes let the tail of one launch overlap
 * the head of the next (what tiles.StripeRenderer does from Python with two scenes on two streams).  Pixel for pixel the images fvsrn_render_stripes
 * gives frame by frame, except that a multi-frame launch does not cut rays into depth segments (re-associated sums, <= 1e-4).  The scenes must agree in everything but the camera -- the
 * call overwrites cam_eye / cam_right / cam_up of each scene's description, nothing else (a light that follows the camera, brdf_light, does not) -- and the caller orders the streams against its buffers (events) as for
 * single frames.  d_rgba8 (optional): frame f's rows also as packed RGBA8 words (fvsrn_extract_color_rgba8, FVSRN_CHANNEL_COLOR, use_tonemapping /
 * max_exposure) at d_rgba8 + f * rows * width, enqueued behind its render on the same stream.  d_stats as in fvsrn_render, summed over the frames.
 * Errors: arguments are checked before anything is enqueued (world == 1 renders whole frames: stripe_rows is not used there).  A failure in the middle of
 * a batch returns its code at once: the frames of the groups before it are enqueued, each lane's scene carries the camera of its last group and the
 * network the time of the failing frame -- state every later batch call overwrites. */
int fvsrn_render_stripes_batch(fvsrn_scene* const* scenes, void* const* streams, int lanes, fvsrn_network* net, int width, int height, int stripe_rows,
                               int rank, int world, int frames, const float* cameras9, const float* times, float* d_out_local, unsigned int* d_rgba8,
                               int use_tonemapping, float max_exposure, unsigned long long* d_stats);

/* IImageEvaluator::ExtractColor (renderer/iimage_evaluator.cpp:26-135): the (1,8,H,W) raw image of fvsrn_render ->
 * a displayable RGBA image.  channel_mode: ChannelMode of iimage_evaluator.h:19-26.
 *   COLOR:  rgb, alpha; with use_tonemapping: rgb / max_exposure -> ACES filmic curve -> clamp -> gamma 1/2.4
 *           (tonemappingFunction, iimage_evaluator_cuda.cu:144-165)
 *   DEPTH:  (depth - min) / (max - min) of channel 7 over the whole image (min / max found on the device; like the
 *           reference's tensor min()/max() a NaN depth -- alpha 0 -- makes both NaN), alpha 1
 *   MASK:   alpha in rgb, alpha 1          NORMAL: 0.5 n + 0.5 of channels 4..6, alpha of channel 3
 * fvsrn_extract_color writes planar fp32 (4,H,W) (SelectOutputChannelKernel2 / TonemappingKernel2, :82-101, :232-262),
 * fvsrn_extract_color_rgba8 one packed 0xAABBGGRR word per pixel (the OpenGL-texture overloads, rgbaToInt of
 * renderer_utils.cuh:48-57).  Device pointers; the image must not alias the output. */
typedef enum { FVSRN_CHANNEL_MASK = 0, FVSRN_CHANNEL_NORMAL = 1, FVSRN_CHANNEL_DEPTH = 2, FVSRN_CHANNEL_COLOR = 3 } fvsrn_channel_mode;
int fvsrn_extract_color(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping,
                        float max_exposure, float* d_out4, void* stream);
int fvsrn_extract_color_rgba8(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping,
                              float max_exposure, unsigned int* d_out, void* stream);
/* ExtractColor of an image that is spread over the ranks of a multi-GPU frame (no reference counterpart: its frame lives on one device):
 * every rank converts ITS rows before the collective -- 4 bytes per pixel instead of 32.  Only FVSRN_CHANNEL_DEPTH needs the other ranks: the depth range
 * of the WHOLE image (iimage_evaluator.cpp:60-77, tensor min() / max()).
 *   fvsrn_depth_range           d_range3 = { -min, max, nan flag } of channel 7 of this part, three floats on the device -- a form that merges over
 *                               ranks by an element-wise maximum, i.e. ONE all-reduce (MAX) of three floats;
 *   fvsrn_extract_color_ranged  fvsrn_extract_color / _rgba8 (exactly one of d_out4 / d_out8 is non-NULL) with the depth range taken from d_range3
 *                               instead of this part's own (NULL: this part's own, i.e. the plain functions above; ignored by the other modes).
 * A merged nan flag gives NaN colours like the reference's min() / max() over a depth plane with a NaN (an alpha-0 pixel). */
int fvsrn_depth_range(const float* d_raw8, int width, int height, float* d_range3, void* stream);
int fvsrn_extract_color_ranged(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping, float max_exposure,
                               const float* d_range3, float* d_out4, unsigned int* d_out8, void* stream);

/* ICamera::generateRays (renderer/camera.cpp:37-98, kernel CameraGenerateRayKernel renderer_camera_kernels.cuh:12-43):
 * ray start / direction of every pixel centre of a width x height image for a camera reference frame (eye, right, up as
 * in fvsrn_scene_desc), written as [height][width][3] fp32 each (the reference's (1,H,W,3) tensors). */
int fvsrn_generate_rays(const float eye[3], const float right[3], const float up[3], float fov_y_radians, int width,
                        int height, float* d_ray_start, float* d_ray_dir, void* stream);

/* ITransferFunction::evaluate / evaluate_with_previous (renderer/transfer_function.cpp:132-145, kernels EvaluateTF /
 * EvaluateTFWithPrevious renderer_tf_kernels.cuh:11-70): the scene's transfer function on n densities ->
 * colours [n][4] (rgb, absorption).  d_previous_density == NULL: evaluate() (no previous density, step size 1);
 * otherwise evaluate_with_previous() with the given step size (a negative previous density means "none"). */
int fvsrn_scene_evaluate_tf(fvsrn_scene* scene, const float* d_density, const float* d_previous_density, size_t n,
                            float density_min, float density_max, float stepsize, float* d_colors, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Dense grid volumes: VolumeInterpolationGrid (renderer/volume_interpolation_grid.cpp, device code
 * renderer/renderer_volume_grid.cuh:89-232) behind the same DVR ray loop -- BASELINE.json configs[0], the ground-truth
 * renders of the reference's evaluation scripts.  A volume is one scalar feature of a `Volume` (renderer/volume.h) or a
 * (X,Y,Z) tensor; it is stored as fp32 in HBM (u8 / u16 data are read as normalised floats like the reference's textures,
 * volume.cpp:109-167). */
typedef struct fvsrn_volume fvsrn_volume;
typedef enum { FVSRN_VOLUME_U8 = 0, FVSRN_VOLUME_U16 = 1, FVSRN_VOLUME_F32 = 2 } fvsrn_volume_dtype;          /* volume.h:43-47 */
typedef enum { FVSRN_VOLUME_NEAREST = 0, FVSRN_VOLUME_TRILINEAR = 1, FVSRN_VOLUME_TRICUBIC = 2 } fvsrn_volume_interpolation;
/* VolumeSource (volume_interpolation_grid.h): VOLUME samples like a CUDA texture (un-normalised coordinates, texel centres at
 * +0.5, 8-bit filter weights), TORCH_TENSOR like the tensor branch of sampleLinear (nodes at integer coordinates) */
typedef enum { FVSRN_VOLUME_SOURCE_TEXTURE = 0, FVSRN_VOLUME_SOURCE_TENSOR = 1 } fvsrn_volume_source;

/* host_data: sx*sy*sz values; x_fastest != 0: index x + sx*(y + sy*z) (Volume::idx, volume.h:128-134), else the layout of a
 * contiguous (X,Y,Z) tensor.  The box is the world-space extent the volume is rendered into. */
int fvsrn_volume_create(const void* host_data, int dtype, int sx, int sy, int sz, int x_fastest, const float box_min[3],
                        const float box_size[3], fvsrn_volume** out);
int fvsrn_volume_destroy(fvsrn_volume* volume);
/* .cvol files: version 1 ("CVOL", volume.cpp:278-332,685-740) and the old density-only format ("cvol", volume.cpp:741-793), uncompressed
 * or LZ4-compressed (Flag_Compressed / useCompression: int32 size + LZ4 block per message of <= 64 KiB, one dependent-block stream per
 * file -- the framing of the reference's lz4cpp wrapper, an empty submodule in the snapshot, recovered from the volume the snapshot does
 * hold, applications/volumes/RichtmyerMeshkov/ppm-t0020.cvol).  load: feature `feature_index` at mipmap level 0, box = [-world/2,
 * world/2] like VolumeInterpolationGrid::setSource.  save writes version 1, uncompressed. */
int fvsrn_volume_load_cvol(const char* path, int feature_index, fvsrn_volume** out);
/* The container itself (Volume::Volume(filename) + Feature::load, volume.cpp:278-332,685-793): every feature of the file at level 0 as raw
 * typed voxels -- channel fastest, then x, y, z (MipmapLevel::idx, volume.h:126-132) -- handed to `on_feature` in file order (return
 * non-zero to stop early; `data` is only valid during the call).  world_size (may be NULL) is filled before the first callback.  The
 * pyrenderer `Volume` class is built on this. */
typedef struct {
    char name[256];
    int index, num_features;
    int dtype;    /* fvsrn_volume_dtype */
    int channels;
    int resolution[3];
} fvsrn_cvol_feature;
typedef int (*fvsrn_cvol_feature_callback)(void* user, const fvsrn_cvol_feature* info, const void* data, size_t bytes);
int fvsrn_cvol_read(const char* path, float world_size[3], fvsrn_cvol_feature_callback on_feature, void* user);
int fvsrn_volume_save_cvol(const char* path, const char* feature_name, const void* host_data, int dtype, int sx, int sy, int sz,
                           float world_x, float world_y, float world_z);
/* Volume::save(filename, compression) (renderer/volume.cpp:623-682): compression 0 .. 9 like there -- 0 writes the file of fvsrn_volume_save_cvol,
 * > 0 sets Flag_Compressed and writes the body as LZ4 messages (int32 size + one LZ4 block per 64 KiB; every level is one greedy matcher here:
 * same format, larger files than the reference's LZ4-HC levels).  fvsrn_cvol_read / fvsrn_volume_load_cvol read either. */
int fvsrn_volume_save_cvol_compressed(const char* path, const char* feature_name, const void* host_data, int dtype, int sx, int sy, int sz,
                                      float world_x, float world_y, float world_z, int compression);
/* The writer behind both (and behind the pyrenderer `Volume.save`): any number of features, described like fvsrn_cvol_read reports them (`index` /
 * `num_features` are ignored), data[i] = feature i's typed voxels, channel fastest, then x, y, z. */
int fvsrn_cvol_write(const char* path, const float world_size[3], int num_features, const fvsrn_cvol_feature* features, const void* const* data,
                     int compression);
int fvsrn_volume_info(fvsrn_volume* volume, int resolution[3], float box_min[3], float box_size[3]);
/* the voxels as the kernels see them (u8 / u16 normalised to [0,1]), x fastest: index x + sx*(y + sy*z); count = sx*sy*sz host floats
 * (Volume::MipmapLevel::toTensor / dataCpu of level 0, volume.cpp:169-214) */
int fvsrn_volume_get_data(fvsrn_volume* volume, float* out, size_t count);
/* IVolumeInterpolation::evaluate for a grid volume: world positions [n][3] -> values [n] (device pointers) */
int fvsrn_volume_evaluate_points(fvsrn_volume* volume, int source, int interpolation, int grid_resolution_new_behavior,
                                 const float* d_positions, size_t n, float* d_out, void* stream);
/* fvsrn_render with a grid volume instead of a network: same scene, same (8,H,W) output, same counters.  provide_normals:
 * central-difference gradients one voxel to either side (evalNormalImpl, renderer_volume_grid.cuh:234-283) for the normal
 * channels; implied by a shading BRDF (brdf_enable_phong / _magnitude_scaling).  fvsrn_scene_desc::gradient_mode is ignored. */
int fvsrn_render_volume(fvsrn_scene* scene, fvsrn_volume* volume, int source, int interpolation, int grid_resolution_new_behavior,
                        int provide_normals, int width, int height, float* d_out8, unsigned long long* d_stats, void* stream);

/* Kernel name of the variant fvsrn_render / fvsrn_evaluate_points would launch for this network: writes a 0-terminated string.
 * For renders this is a forecast from the network alone -- a latent grid is reported with the cell-table kernel whenever a table exists,
 * "(cells or gathers by footprint)" marks the automatic mode; the launch's own choice is fvsrn_scene_last_kernel_name. */
int fvsrn_network_kernel_name(fvsrn_network* net, int render, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* FVSRN_H_ */
