// Rendering entry points of the C ABI: which kernel family a launch takes (register-resident, LDS, cell table or gathers by footprint, adjoint, shaded),
// its shape (waves per workgroup, persistent or bounded waves, depth segments, several frames per launch) and the launch itself; the watchdog report.
#include "api_internal.hpp"

void fillSceneParams(fvsrn_scene* scene, const fvsrn_scene_desc& d, int width, int height, SceneParams& S) {
    for (int i = 0; i < 3; ++i) { S.eye[i] = d.cam_eye[i]; S.right[i] = d.cam_right[i]; S.up[i] = d.cam_up[i]; }
    // front = cross(up, right), renderer_camera.cuh:47
    S.front[0] = S.up[1] * S.right[2] - S.up[2] * S.right[1];
    S.front[1] = S.up[2] * S.right[0] - S.up[0] * S.right[2];
    S.front[2] = S.up[0] * S.right[1] - S.up[1] * S.right[0];
    S.frames = 1;  // (render_body reads the camera from cams[]: device_params.hpp)
    for (int i = 0; i < 3; ++i) { S.cams[0][i] = S.eye[i]; S.cams[0][3 + i] = S.right[i]; S.cams[0][6 + i] = S.up[i]; S.cams[0][9 + i] = S.front[i]; }
    S.tanFovY = std::tan(d.fov_y_radians / 2);
    S.tanFovX = S.tanFovY * (float(width) / float(height));  // setAspectRatio, image_evaluator_simple.cpp:204
    S.stepsize = d.stepsize;
    S.alphaEarlyOut = 1.0f - 1e-5f;  // ray_evaluation_stepping.cpp:536
    S.densityMin = d.density_min;
    S.divDensityRange = 1.0f / (d.density_max - d.density_min);
    S.earlyOut = d.early_out;
    S.blendMode = d.blend_mode;
    S.gradientMode = d.gradient_mode;
    S.fdStep = d.finite_differences_stepsize;
    S.gridDiffStep = d.adjoint_grid_stepsize;
    S.brdfMagnitudeScaling = d.brdf_enable_magnitude_scaling;
    S.brdfPhong = d.brdf_enable_phong;
    S.brdfLightType = d.brdf_light_type;
    S.brdfSpecularExponent = d.brdf_specular_exponent;
    S.brdfMagScale = d.brdf_magnitude_scaling; S.brdfAmbient = d.brdf_ambient; S.brdfSpecular = d.brdf_specular;
    S.brdfMagCenter = d.brdf_magnitude_center; S.brdfMagRadius = d.brdf_magnitude_radius;
    for (int i = 0; i < 3; ++i) S.brdfLight[i] = d.brdf_light[i];
    S.tfKind = d.tf_kind;
    S.tfRows = d.tf_rows; S.tfRowsF = float(d.tf_rows);
    S.tfScaleAbsorption = d.tf_scale_absorption;
    S.tfScaleEmission = d.tf_scale_emission;
    S.tfAbsorptionStep = d.tf_scale_absorption * d.stepsize;
    S.tfAbsorptionStepLog2e = float(-double(d.tf_scale_absorption) * double(d.stepsize) * 1.4426950408889634);
    S.densityBias = -d.density_min * S.divDensityRange;
    S.stepLog2e = float(-double(d.stepsize) * 1.4426950408889634);
    S.tfTable = static_cast<const float*>(scene->dTf.ptr);
    S.tfOpacityNonNegative = scene->tfOpacityNonNegative ? 1 : 0;
    S.tfGaussianMode = d.tf_gaussian_mode;
    S.tfPreintegration = d.tf_preintegration;
    S.tfPreintegrated = static_cast<const float*>(scene->dPreint.ptr);
}

extern "C" {

// frames / cameras9: > 1 camera poses { eye, right, up } of the same scene rendered by ONE launch into d_out8 + f * 8 * plane (at most
// kMaxFramesPerLaunch; fvsrn_render_stripes_batch); 1 / nullptr: the scene's own camera
static int renderImpl(fvsrn_scene* scene, fvsrn_network* net, int width, int height, int y0, int y1, int numLocalRows,
                      int stripeRows, int stripeRank, int stripeWorld, int compact, float* d_out8,
                      unsigned long long* d_stats, void* stream, int frames = 1, const float* cameras9 = nullptr) {
    return guarded([&] {
        if (!scene || !net || !d_out8) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        if (width <= 0 || height <= 0 || y0 < 0 || y1 > height || y0 > y1)
            return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad image size or row range");
        std::lock_guard<std::mutex> lockN(net->mu);
        std::lock_guard<std::mutex> lockS(scene->mu);
        try {
            hipStream_t s = static_cast<hipStream_t>(stream);
            net->ensureDevice(s);
            net->syncTime(s);
            const fvsrn_scene_desc& d = scene->desc;
            const NetworkConfig& c = net->packed.cfg;
            const bool rgbo = c.outputMode == FVSRN_OUT_RGBO || c.outputMode == FVSRN_OUT_RGBO_DIRECT;
            // ray_evaluation_stepping.cpp:560-601: the TF is skipped iff the volume emits colour
            if (rgbo && d.tf_kind != FVSRN_TF_NONE)
                return fail(FVSRN_ERR_INVALID_ARGUMENT, "the network outputs colour; the scene must use FVSRN_TF_NONE");
            if (!rgbo && d.tf_kind == FVSRN_TF_NONE)
                return fail(FVSRN_ERR_INVALID_ARGUMENT, "the network outputs densities; the scene needs a transfer function");
            if (numLocalRows == 0) return FVSRN_OK;
            const size_t tfFloats = scene->tfTable.size();
            if (const int rc = scene->uploadTf(d.stepsize, s)) return rc;
            RenderArgs a{};
            a.P = net->packed.params;
            a.shaded = d.gradient_mode != FVSRN_GRADIENT_OFF_OR_DIRECT || d.brdf_enable_phong || d.brdf_enable_magnitude_scaling ||
                       d.tf_preintegration != FVSRN_PREINTEGRATE_NONE || d.tf_gaussian_mode != FVSRN_TF_GAUSSIAN_PLAIN;  // (the Gaussian variants
                       // read the gradient / the previous sample's density: the shaded kernel tracks both)
            // finite differences also sample up to a step outside the box, where the [0,1] bound of the scaled image does
            // not hold: the shaded renderer takes the plain image
            if (!a.shaded) a.P.fourierNeedsFract = a.P.fourierNeedsFractPlain;  // positions inside the box only (pack.cpp)
            if (net->scaledImage && !a.shaded) {  // ReLU network: image with activations scaled into [0,1] (pack.cpp)
                a.P.ldsImage = net->scaledImage;
                a.P.reluClamp = net->keyScaled.act == ACT_RELU01 ? 1 : 0;
                if (!net->packed.scaledBias0Exact) a.P.bias0Folded = 0;  // (a residue of the folded bias sits in the fp32 block: pack.cpp)
            }
            SceneParams& S = a.S;
            fillSceneParams(scene, d, width, height, S);
            // colour networks have no gradient mode (SceneNetwork::getDefines, volume_interpolation_network.cpp:1148)
            if (rgbo) S.gradientMode = FVSRN_GRADIENT_OFF_OR_DIRECT;
            // latentGridDifferencesStepSize of the adjoint mode (VolumeInterpolationNetwork::fillConstantMemory :1808-1812)
            if (S.gridDiffStep <= 0.f) S.gridDiffStep = 1.0f / (float(std::max(1, a.P.gridX)) * 4.0f);
            if (frames > 1) {
                if (frames > kMaxFramesPerLaunch || !cameras9) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad frame count of a multi-frame launch");
                S.frames = frames;
                for (int f = 0; f < frames; ++f) {
                    const float* c = cameras9 + size_t(f) * 9;  // eye, right, up; front = cross(up, right) like fillSceneParams
                    float* o = S.cams[f];
                    for (int i = 0; i < 9; ++i) o[i] = c[i];
                    o[9] = c[7] * c[5] - c[8] * c[4];
                    o[10] = c[8] * c[3] - c[6] * c[5];
                    o[11] = c[6] * c[4] - c[7] * c[3];
                }
                for (int i = 0; i < 3; ++i) { S.eye[i] = S.cams[0][i]; S.right[i] = S.cams[0][3 + i]; S.up[i] = S.cams[0][6 + i]; S.front[i] = S.cams[0][9 + i]; }
            }
            S.width = width; S.height = height; S.y0 = y0; S.y1 = y1;
            S.numLocalRows = numLocalRows; S.stripeRows = stripeRows; S.stripeRank = stripeRank;
            S.stripeWorld = stripeWorld; S.compact = compact;
            a.out = d_out8;
            a.stats = d_stats;
            const int tilesX = (width + 7) / 8, tilesY = (numLocalRows + 7) / 8;
            const int tiles = tilesX * tilesY;
            size_t lds = size_t(a.P.ldsBytes) + ((tfFloats + 3) & ~size_t(3)) * 4;
            S.tfLdsFloats = int((tfFloats + 3) & ~size_t(3));
            const Options& O = scene->opts;
            // automatic: whole frames persistent; the stripes of a multi-GPU frame in bounded waves (below) unless the caller opts in --
            // tiles.StripeRenderer does after it has MEASURED that the process's streams really run side by side (fvsrn_probe_stream_concurrency)
            const bool persistent = O[FVSRN_OPT_PERSISTENT] >= 0 ? O[FVSRN_OPT_PERSISTENT] != 0 : stripeWorld <= 1;
            int wpb = wavesPerBlockFor(lds, O);
            // Small networks in registers (render_small_kernel, kernels.hpp): 32-wide Fourier-only scalar network with at most
            // three C->C layers, phases inside the v_cos domain, a transfer function, no shading.  FVSRN_SMALL_KERNEL=0: off.
            // The cell table pays while the samples of a wave step (the rays of an 8 x 8 pixel tile at one depth) share one or two grid cells:
            // footprint of the tile in cells = 8 pixels x pixel size at the box centre x cells per unit length.  Measured r04 (tools/dev/
            // cell_footprint_sweep.py, 16^3 .. 64^3 grids, 512^2 .. 2048^2 images): 0.72 - 0.82 of the gather path's time up to 0.5 cells, equal at
            // ~0.9 (32 wide) / ~0.7 (64 wide), 1.2 x at 1.3 -- further cell pairs cost a dependent fetch each (profiles/r04/cell_footprint_sweep_r04.txt).
            // -1 = this rule, 1 = always, 0 = never.
            bool useCells = O[FVSRN_OPT_CELL_TABLE] == 1;
            if (O[FVSRN_OPT_CELL_TABLE] == -1 && net->cellTableBytes) {
                const BoxCenter bc = P_boxCenter(a.P);
                const double ex = S.eye[0] - bc.c[0], ey = S.eye[1] - bc.c[1], ez = S.eye[2] - bc.c[2];
                const float* bsz = a.P.boxSize;  // (a camera close to or inside the box: its samples are up to half a box diagonal away)
                const double dist = std::max(0.5 * std::sqrt(double(bsz[0]) * bsz[0] + double(bsz[1]) * bsz[1] + double(bsz[2]) * bsz[2]), std::sqrt(ex * ex + ey * ey + ez * ez));
                const double pixel = std::max(2.0 * S.tanFovX / std::max(1, width), 2.0 * S.tanFovY / std::max(1, height));
                const double cellsPerUnit = std::max({(a.P.gridX - 1) / double(a.P.boxSize[0]), (a.P.gridY - 1) / double(a.P.boxSize[1]), (a.P.gridZ - 1) / double(a.P.boxSize[2])});
                useCells = 8.0 * pixel * dist * cellsPerUnit <= (net->key.CD <= 2 ? 0.8 : 0.7);
            }
            const void* smallFn = nullptr;
            int smallGrid = 0;
            const int smallTail = rgbo ? 3 : (d.tf_kind == FVSRN_TF_PIECEWISE || d.tf_kind == FVSRN_TF_GAUSSIAN ? 2 : (d.blend_mode != FVSRN_BLEND_BEER_LAMBERT ? 1 : (d.tf_kind == FVSRN_TF_IDENTITY && d.tf_scale_absorption >= 0.f ? FVSRN_IDENTITY_TAIL : (d.tf_kind == FVSRN_TF_TEXTURE && scene->tfOpacityNonNegative ? 5 : 1))));  // kernels.hpp TAIL_*
            {
                const VariantKey& k = net->keyScaled;
                const bool scalarNet = a.P.outputMode == FVSRN_OUT_DENSITY || a.P.outputMode == FVSRN_OUT_DENSITY_DIRECT;
                // latent grid: 2 = through the cell table (any number of latent channels), 1 = one decoded 16-channel chunk by gathers;
                // both need the first layer's bias in its weights (bias0Folded: no time input), the resident kernels drop that bias block
                smallGrid = k.grid == 0 ? 0 : (k.grid == 1 && a.P.bias0Folded ? (net->cellTableBytes && useCells ? 2 : (a.P.gridK == 1 ? 1 : 3)) : 3);
                if (O[FVSRN_OPT_SMALL_KERNEL] != 0 && net->opts[FVSRN_OPT_SMALL_KERNEL] != 0 && !a.shaded && k.CD == 2 && smallGrid <= 2 && !a.P.noFourier && !a.P.fourierNeedsFract &&
                    !a.P.fourierClampPos &&  // (the resident kernels compile the position clamp out)
                    a.P.numLayers >= 1 && a.P.numLayers <= 3 && (rgbo || (scalarNet && d.tf_kind != FVSRN_TF_NONE)))
                {
                    smallFn = render_small_fn(k.act, k.dir, a.P.numLayers, smallTail, smallGrid);
                    if (!smallFn && smallGrid == 2 && a.P.gridK == 1) smallFn = render_small_fn(k.act, k.dir, a.P.numLayers, smallTail, smallGrid = 1);
                }
            }
#ifndef FVSRN_ROTATE_SGRID
#define FVSRN_ROTATE_SGRID 0  // kernels.hpp: the rotating variant of the resident latent-grid kernel is an A/B build, not the shipped one
#endif
            if (FVSRN_ROTATE_SGRID && smallFn && smallGrid == 1) {
                // the resident kernel with a latent chunk parks the per-ray feature rotation of every wave in LDS (8 KiB per wave behind the
                // TF table, srn_forward_rotating_resident_grid); it runs 2 waves per SIMD = 8 per CU
                constexpr size_t kRotationBytes = 64 * 32 * 4;
                if (!O[FVSRN_OPT_WAVES_PER_BLOCK]) {
                    wpb = 4;
                    for (int w : {1, 2})
                        if (size_t(8 / w) * (lds + size_t(w) * kRotationBytes) <= 160 * 1024) { wpb = w; break; }
                }
                lds += size_t(wpb) * kRotationBytes;
            }
            // FVSRN_OPT_FOURIER_RESYNC = 1 (every step derives its features like the reference): the variants of the rotating resident kernels that
            // have no rotation to advance
            bool smallExact = false;
            if (smallFn && smallGrid != 1 && (O[FVSRN_OPT_FOURIER_RESYNC] ? O[FVSRN_OPT_FOURIER_RESYNC] : kFourierResync) == 1) {
                if (const void* fn = render_small_exact_fn(net->keyScaled.act, net->keyScaled.dir, a.P.numLayers, smallTail, smallGrid)) { smallFn = fn; smallExact = true; }
            }
            // 48- / 64-wide latent-grid networks on the GATHER path: render_kernel<3|4, *, 1|2, *> in the fragment-major layer order (kernels.hpp,
            // render_layer_schedule; r03 - r04: a separate render_stripe_kernel on request).  FVSRN_OPT_OVERLAP_KERNEL = 1 takes it in place of the cell-table
            // kernel too (A/B of the two latent-grid paths at these widths); 0 / -1: by the footprint rule.
            const void* stripeFn = nullptr;
            const bool wantCellsKernel = net->keyScaled.grid == 1 && net->cellTableBytes && useCells && O[FVSRN_OPT_OVERLAP_KERNEL] != 1;
            if (!smallFn && !a.shaded && !wantCellsKernel)
                stripeFn = render_stripe_fn(net->keyScaled);
            // every other unshaded render of a network whose decoded latent grid has a cell table: render_kernel with the grid through that table
            const void* cellsFn = nullptr;
            if (!smallFn && !stripeFn && !a.shaded && net->keyScaled.grid == 1 && net->cellTableBytes && useCells)
                cellsFn = render_cells_fn(net->keyScaled);
            // the adjoint gradient mode up to 64 channels: its own kernel (render_adjoint_kernel, kernels.hpp)
            const void* adjointFn = (a.shaded && d.gradient_mode == FVSRN_GRADIENT_ADJOINT_METHOD) ? render_adjoint_fn(net->key) : nullptr;
            // the shaded renderer with the grid through the cell table of the plain image: every mode but the adjoint one (whose gradient pass keeps its records)
            if (a.shaded && d.gradient_mode != FVSRN_GRADIENT_ADJOINT_METHOD && net->key.grid == 1 && net->cellTableBytes && useCells)
                cellsFn = render_shaded_cells_fn(net->key);
            // the table itself: built by the first launch that goes through it (and from then on with every blend, until a launch does not)
            if ((smallFn && smallGrid == 2) || cellsFn) {
                a.P.cellTable = net->ensureCellTable(a.shaded, s);
                // (the shaded kernels read the corner-form table over the grid's own cells: srn_device.hpp cell_tap_corners)
                if (a.shaded) a.P.cellCount = unsigned(net->cellTableBytesCorners / size_t(a.P.cellStride ? a.P.cellStride : 512));
            }
            else (a.shaded ? net->cellsPlainWanted : net->cellsWanted) = false;
            net->beginUse(s);
            struct Done { fvsrn_network* n; hipStream_t s; ~Done() { try { n->endUse(s); } catch (...) {} } } done{net, s};
            const void* altFn = smallFn ? smallFn : (stripeFn ? stripeFn : (cellsFn ? cellsFn : adjointFn));
            const int perCU = net->renderBlocksPerCU(unsigned(64 * wpb), lds, a.shaded, altFn, O[FVSRN_OPT_MAX_BLOCKS_PER_CU]);
            const unsigned resident = unsigned(net->numCUs) * unsigned(std::max(perCU, 1));  // workgroups the chip holds at once
            // Depth segments (kernels.hpp): with fewer tiles than ~4x the resident waves (small images, the stripes of one
            // rank of a multi-GPU frame) the longest tile dictates the launch time; cut the rays into K step ranges so that
            // there are enough work units to balance, as long as a segment keeps >= ~48 steps (box diagonal / step size).
            // Measured r01: 512^2 x 256: 79.8 -> see BASELINE.md.  FVSRN_SEGMENTS=k forces K (1 = off).
            int K = 1;
            {
                const double waves = double(resident) * wpb;
                const float* bs = a.P.boxSize;
                const double maxSteps = std::sqrt(double(bs[0]) * bs[0] + double(bs[1]) * bs[1] + double(bs[2]) * bs[2]) / d.stepsize;
                // (a pre-integrated TF looks at the previous sample of the ray: no cuts)
                const bool looksBack = d.tf_preintegration != FVSRN_PREINTEGRATE_NONE || d.tf_gaussian_mode == FVSRN_TF_GAUSSIAN_ANALYTIC;
                while (!looksBack && K < 8 && double(tiles) * frames * K < 4.0 * waves && maxSteps / (2 * K) >= 48.0) K *= 2;
                if (O[FVSRN_OPT_DEPTH_SEGMENTS] >= 1 && !looksBack) K = O[FVSRN_OPT_DEPTH_SEGMENTS];
                if (frames > 1) K = 1;  // (a multi-frame launch: the frames are the extra work units; the composite pass handles one image)
            }
            // FVSRN_OPT_FOURIER_RESYNC: 1 = exact Fourier features at every step (the reference's arithmetic), default every 64 steps
            S.resyncMask = (O[FVSRN_OPT_FOURIER_RESYNC] ? O[FVSRN_OPT_FOURIER_RESYNC] : kFourierResync) - 1;
            S.segments = K;
            S.partial = nullptr;
            const size_t plane = size_t(width) * size_t(compact ? numLocalRows : height);
            if (K > 1) {
                scene->dPartial.ensure(size_t(K) * 8 * plane * sizeof(float));
                S.partial = static_cast<float*>(scene->dPartial.ptr);
            }
            const long long units = (long long)tiles * K * frames;
            unsigned grid = unsigned((units + wpb - 1) / wpb);
            // persistent waves: no more workgroups than the chip holds at once; the rest of the units is handed out by
            // a device counter (kernels.hpp).  FVSRN_PERSISTENT=0: one unit per wave, hardware dispatch order.
            // A rank of a multi-GPU frame (stripeWorld > 1) has its previous frame gathered by a collective's kernel on another stream
            // while this one renders.  Until r03 its launches were therefore not persistent (bounded waves, below), on the assumption that
            // persistent waves hold every wave slot until their launch ends.  Measured r03 (tools/dev/coschedule.py, a 24-workgroup
            // stand-in kernel submitted into a persistent launch): it starts at once and ends on time -- the wide kernels leave ~60
            // registers per lane and SIMD unallocated, enough for a small kernel's waves.  What did serialise the two was ROCm's default of
            // FOUR hardware queues for all streams of a process (GPU_MAX_HW_QUEUES: the comm stream shared a queue with a render stream);
            // with eight, a rank's share at world 8 runs at 97 - 98 % of frame / world persistent against 86 % with bounded waves
            // (profiles/r03/stripe_pipeline_r03.md).  Since a real collective may need more registers than a launch leaves, stripe
            // launches keep 1/16 of the workgroup slots free (FVSRN_OPT_PERSISTENT_RESERVE).  r04: that gain is a one-GPU
            // emulation and depends on a process setting the library cannot make (GPU_MAX_HW_QUEUES is read when HIP starts; with four
            // queues persistent stripes measured 75 - 87 % against 82 - 90 % bounded), so the AUTOMATIC choice for stripes is bounded waves
            // again and persistent stripes are an opt-in (FVSRN_OPT_PERSISTENT = 1).
            S.unitQuota = 0;
            if (persistent) {
                if (perCU > 0 && grid > resident) {
                    // FVSRN_OPT_PERSISTENT_RESERVE: slots left to kernels of other streams while this launch holds the chip
                    const unsigned reserve = O[FVSRN_OPT_PERSISTENT_RESERVE] >= 0 ? unsigned(O[FVSRN_OPT_PERSISTENT_RESERVE]) : (stripeWorld > 1 ? resident / 16 : 0u);
                    grid = resident > reserve + unsigned(net->numCUs) / 2 ? resident - reserve : resident;
                    if (!scene->tileCounters(s, &S.tileCounter, &S.tileCounterNext))
                        return fail(FVSRN_ERR_DEVICE, "could not set up the tile counters");
                }
            } else {
                // Bounded waves for the stripes of a latent-grid network: every wave takes two units from the counter in raster
                // order (neighbouring tiles share grid lines in L2) and retires, so the launch still turns its workgroups over
                // for the gather kernel but copies the network into LDS half as often.  Measured r01 on one rank's share
                // of 1024^2 x 512, 64x6 + grid (tools/stripe_efficiency.py): 75 / 76 / 84 % of frame_time / world at
                // world 2 / 4 / 8 without, 85 / 87 / 85 % with; no gain for Fourier-only networks (small LDS image).
                const int quota = O[FVSRN_OPT_UNIT_QUOTA] >= 0 ? O[FVSRN_OPT_UNIT_QUOTA] : (stripeWorld > 1 && net->key.grid != 0 ? 2 : 0);
                if (quota > 1 && perCU > 0 && grid > resident) {
                    grid = unsigned((units + (long long)wpb * quota - 1) / ((long long)wpb * quota));
                    S.unitQuota = quota;
                    if (!scene->tileCounters(s, &S.tileCounter, &S.tileCounterNext))
                        return fail(FVSRN_ERR_DEVICE, "could not set up the tile counters");
                }
            }
            // Launch order of the pixel tiles.  Persistent waves balance the load themselves and are fastest in raster
            // order (neighbouring tiles share latent-grid lines in L1/L2: 64x6+grid 22.1 centre-first vs 23.2 raster
            // Gsamples/s, r01).  Without them (FVSRN_PERSISTENT=0, or a launch that fits on the chip at once) centre-first
            // starts the long rays first, which pays whenever a workgroup holds several waves or waits on memory.
            const bool useOrder = O[FVSRN_OPT_TILE_ORDER] >= 0 ? O[FVSRN_OPT_TILE_ORDER] == 1 : (S.tileCounter == nullptr && (wpb > 1 || net->key.grid != 0));
            S.tileOrder = useOrder && frames == 1 ? scene->tileOrder(S, P_boxCenter(a.P), tilesX, tilesY, s) : nullptr;  // (the order is one camera's)
            // what this launch does to the samples of a ray, for callers that restate it (fvsrn_scene_last_render_info)
            const bool rotates = net->keyScaled.CD == 2 && (net->keyScaled.grid == 0 || (FVSRN_ROTATE_SGRID && smallFn && smallGrid == 1) || (smallFn && smallGrid == 2)) && !a.P.noFourier &&
                                 !a.shaded;  // kRotate / kRotateLds, kernels.hpp
            scene->lastLaunch.grid = grid; scene->lastLaunch.block = unsigned(64 * wpb); scene->lastLaunch.units = units; scene->lastLaunch.width = width;
            scene->lastLaunch.height = height; scene->lastLaunch.rows = numLocalRows; scene->lastLaunch.stripeWorld = stripeWorld;
            scene->lastLaunch.frames = frames;
            scene->lastLaunch.persistent = S.tileCounter ? (S.unitQuota > 0 ? 2 : 1) : 0; scene->lastLaunch.stream = stream; ++scene->lastLaunch.count;
            scene->lastInfo[0] = K;
            scene->lastInfo[1] = rotates ? S.resyncMask + 1 : 0;
            scene->lastInfo[2] = smallFn ? (smallGrid == 2 ? 4 : 1) : (stripeFn ? 2 : (cellsFn ? 5 : (adjointFn ? 3 : 0)));
            scene->lastInfo[3] = wpb;
            {   // the kernel this launch runs, as rocprofv3 will name it (fvsrn_scene_last_kernel_name)
                const VariantKey& ks = net->keyScaled;
                const std::string v = std::to_string(ks.CD) + ",act " + std::to_string(ks.act) + "," + (ks.dir ? "true" : "false");
                const std::string vp = std::to_string(net->key.CD) + ",act " + std::to_string(net->key.act) + ",grid " + std::to_string(net->key.grid) + "," + (net->key.dir ? "true" : "false");
                if (smallFn)
                    scene->lastKernel = "render_small_kernel<act " + std::to_string(ks.act) + "," + (ks.dir ? "true" : "false") + "," + std::to_string(a.P.numLayers) + ",TAIL=" +
                                        std::to_string(smallTail) + ",SGRID=" + std::to_string(smallGrid) + (smallExact ? ",ADVANCE=false>" : ">");
                else if (stripeFn) scene->lastKernel = net->kinfoScaled.renderName;  // (render_kernel<3|4, ...>: fragment-major order)
                else if (cellsFn) scene->lastKernel = std::string(a.shaded ? "render_shaded_cells_kernel<" + vp : "render_cells_kernel<" + v) + ">";
                else if (adjointFn) scene->lastKernel = "render_adjoint_kernel<" + vp + ">";
                else if (a.shaded) scene->lastKernel = "render_shaded_kernel<" + vp + ">";
                else scene->lastKernel = net->kinfoScaled.renderName;
            }
            hipError_t e = smallFn ? (smallExact ? launch_render_small_exact(net->keyScaled.act, net->keyScaled.dir, a.P.numLayers, smallTail, smallGrid, a, grid, unsigned(64 * wpb), lds, s)
                                                 : launch_render_small(net->keyScaled.act, net->keyScaled.dir, a.P.numLayers, smallTail, smallGrid, a, grid, unsigned(64 * wpb), lds, s))
                                   : (stripeFn ? launch_render_stripe(net->keyScaled, a, grid, unsigned(64 * wpb), lds, s)
                                      : cellsFn ? (a.shaded ? launch_render_shaded_cells(net->key, a, grid, unsigned(64 * wpb), lds, s)
                                                             : launch_render_cells(net->keyScaled, a, grid, unsigned(64 * wpb), lds, s))
                                      : (adjointFn ? launch_render_adjoint(net->key, a, grid, unsigned(64 * wpb), lds, s)
                                                   : launch_render(a.shaded ? net->key : net->keyScaled, a, grid, unsigned(64 * wpb), lds, s)));
            if (e == hipSuccess && K > 1) e = launch_composite(S.partial, d_out8, K, plane, S, s);
            if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("Error during rendering! ") + hipGetErrorString(e));
            return FVSRN_OK;
        } catch (const DeviceError& e) {
            return fail(fvsrn_device_count() == 0 ? FVSRN_ERR_NO_DEVICE : FVSRN_ERR_DEVICE, e.what());
        }
    });
}

int fvsrn_render(fvsrn_scene* scene, fvsrn_network* net, int width, int height, int y0, int y1, float* d_out8,
                 unsigned long long* d_stats, void* stream) {
    return renderImpl(scene, net, width, height, y0, y1, y1 - y0, 8, 0, 1, 0, d_out8, d_stats, stream);
}

int fvsrn_debug_state(char* buf, size_t cap) {
    // Never blocks: a scene whose mutex is held (a call of this library is in flight on another host thread) is reported as such, device memory
    // is read by an asynchronous copy on a stream of its own that is polled for at most a second (a kernel that spins forever keeps the copy engines free).
    if (!buf || cap == 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    std::string out;
    SceneRegistry& r = sceneRegistry();
    std::unique_lock<std::mutex> lr(r.mu, std::try_to_lock);
    if (!lr.owns_lock()) out += "scene registry: locked\n";
    else {
        out += "live scenes: " + std::to_string(r.live.size()) + "\n";
        int idx = 0;
        for (fvsrn_scene* sc : r.live) {
            out += "scene " + std::to_string(idx++) + ": ";
            std::unique_lock<std::mutex> ls(sc->mu, std::try_to_lock);
            if (!ls.owns_lock()) { out += "LOCKED (a library call on this scene is in flight on a host thread)\n"; continue; }
            const fvsrn_scene::LastLaunch& L = sc->lastLaunch;
            out += "launches " + std::to_string(L.count) + ", last kernel '" + sc->lastKernel + "' grid " + std::to_string(L.grid) + " x " + std::to_string(L.block) +
                   ", units " + std::to_string(L.units) + ", image " + std::to_string(L.width) + " x " + std::to_string(L.height) + " (" + std::to_string(L.rows) +
                   " rows, world " + std::to_string(L.stripeWorld) + "), frames " + std::to_string(L.frames) + ", " + (L.persistent == 1 ? "persistent" : (L.persistent == 2 ? "bounded waves" : "one unit per wave")) +
                   ", segments " + std::to_string(sc->lastInfo[0]) + ", waves/workgroup " + std::to_string(sc->lastInfo[3]);
            // r06: this runs on a watchdog thread whose current device is 0 by default -- the copy stream below belongs on the SCENE's device
            // (a rank process bound to GPU r would otherwise open a context on GPU 0 and copy across devices); restored on the way out.  The launch's stream
            // is the caller's: it is only queried while it is the legacy / per-thread default stream (handles that cannot have been destroyed).
            int prevDev = -1;
            (void)hipGetDevice(&prevDev);
            struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{sc->device >= 0 && sc->device != prevDev ? prevDev : -1};
            if (sc->device >= 0 && sc->device != prevDev) (void)hipSetDevice(sc->device);
            const hipStream_t ls_ = static_cast<hipStream_t>(L.stream);
            if (L.count && (ls_ == nullptr || ls_ == hipStreamPerThread || ls_ == hipStreamLegacy))
                out += std::string(", stream ") + (hipStreamQuery(ls_) == hipSuccess ? "idle" : "BUSY");
            else if (L.count) out += ", stream: the caller's (not queried: it may be gone)";
            (void)hipGetLastError();
            if (sc->dCounters.ptr) {
                int host[2] = {-1, -1};
                hipStream_t cs = nullptr;
                if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess) {
                    bool ok = hipMemcpyAsync(host, sc->dCounters.ptr, sizeof(host), hipMemcpyDeviceToHost, cs) == hipSuccess;
                    for (int i = 0; ok && i < 1000 && hipStreamQuery(cs) == hipErrorNotReady; ++i) { struct timespec ts{0, 1000000}; nanosleep(&ts, nullptr); }
                    const bool done = ok && hipStreamQuery(cs) == hipSuccess;
                    out += done ? ", work counters {" + std::to_string(host[0]) + ", " + std::to_string(host[1]) + "} (launch parity " + std::to_string(sc->launches & 1u) + ")"
                                : std::string(", work counters: copy did not complete in 1 s");
                    if (done) (void)hipStreamDestroy(cs);  // (a stream with a stuck copy is leaked: destroying it would block)
                }
                (void)hipGetLastError();
            }
            out += "\n";
        }
    }
    std::strncpy(buf, out.c_str(), cap - 1);
    buf[cap - 1] = 0;
    return FVSRN_OK;
}

int fvsrn_scene_last_kernel_name(fvsrn_scene* scene, char* buf, size_t cap) {
    return guarded([&] {
        if (!scene || !buf || cap == 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> lock(scene->mu);
        std::strncpy(buf, scene->lastKernel.c_str(), cap - 1);
        buf[cap - 1] = 0;
        return FVSRN_OK;
    });
}

int fvsrn_scene_last_render_info(fvsrn_scene* scene, int out[4]) {
    if (!scene || !out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(scene->mu);
    for (int i = 0; i < 4; ++i) out[i] = scene->lastInfo[i];
    return FVSRN_OK;
}

int fvsrn_stripe_rows(int height, int stripe_rows, int rank, int world) {
    if (height <= 0 || stripe_rows <= 0 || world <= 0 || rank < 0 || rank >= world) return -1;
    int rows = 0;
    for (int y = rank * stripe_rows; y < height; y += stripe_rows * world) rows += std::min(stripe_rows, height - y);
    return rows;
}

int fvsrn_render_stripes(fvsrn_scene* scene, fvsrn_network* net, int width, int height, int stripe_rows, int rank,
                         int world, float* d_out_local, unsigned long long* d_stats, void* stream) {
    if (stripe_rows <= 0 || stripe_rows % 8 != 0 || world <= 0 || rank < 0 || rank >= world)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "stripe_rows must be a positive multiple of 8 and 0 <= rank < world");
    // local rows are laid out stripe after stripe; a short last stripe only ever is the LAST local stripe
    const int rows = fvsrn_stripe_rows(height, stripe_rows, rank, world);
    return renderImpl(scene, net, width, height, 0, height, rows, stripe_rows, rank, world, 1, d_out_local, d_stats, stream);
}

int fvsrn_render_stripes_batch(fvsrn_scene* const* scenes, void* const* streams, int lanes, fvsrn_network* net, int width, int height, int stripe_rows,
                               int rank, int world, int frames, const float* cameras9, const float* times, float* d_out_local, unsigned int* d_rgba8,
                               int use_tonemapping, float max_exposure, unsigned long long* d_stats) {
    if (!scenes || !streams || lanes < 1 || lanes > 8 || !net || !cameras9 || !d_out_local || frames < 0)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument, or lanes outside 1 .. 8");
    // (world == 1: a whole frame, the stripe height is not used)
    if ((world != 1 && (stripe_rows <= 0 || stripe_rows % 8 != 0)) || world <= 0 || rank < 0 || rank >= world)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "stripe_rows must be a positive multiple of 8 and 0 <= rank < world");
    // r06: like every entry point, nothing thrown below crosses the C boundary.  What a failure in the MIDDLE of a batch leaves behind is stated in
    // include/fvsrn.h: the frames before the failing group are enqueued, the scenes of the lanes carry the camera of their last group, the network the time of
    // the failing frame -- the caller re-sends both with its next call (a batch always does).
    return guarded([&]() -> int {
    for (int l = 0; l < lanes; ++l) {
        if (!scenes[l]) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null scene");
        for (int m = 0; m < l; ++m)
            if (scenes[m] == scenes[l] && streams[m] != streams[l])
                return fail(FVSRN_ERR_INVALID_ARGUMENT, "one scene on two streams: the launches of a scene are ordered on one stream (fvsrn.h)");
    }
    if (use_tonemapping && !(max_exposure > 0)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "max_exposure must be positive");
    const int rows = world == 1 ? height : fvsrn_stripe_rows(height, stripe_rows, rank, world);
    const size_t planes = size_t(8) * size_t(std::max(rows, 0)) * size_t(std::max(width, 0));
    // Frames that share their time go into ONE launch per lane, up to kMaxFramesPerLaunch poses each (device_params.hpp: a work unit is (frame, tile));
    // lanes take consecutive groups in turn, so with two lanes the tail of one group's launch overlaps the head of the next.  Per-frame times: one
    // launch per frame (every frame blends its own working grid), lane f % lanes.
    const int group = times ? 1 : std::min(kMaxFramesPerLaunch, std::max(1, (frames + lanes - 1) / lanes));
    int lane = 0;
    for (int f0 = 0; f0 < frames; f0 += group, lane = (lane + 1) % lanes) {
        const int n = std::min(group, frames - f0);
        fvsrn_scene* sc = scenes[lane];
        void* st = streams[lane];
        {
            std::lock_guard<std::mutex> lock(sc->mu);
            std::memcpy(sc->desc.cam_eye, cameras9 + size_t(f0) * 9, 3 * sizeof(float));
            std::memcpy(sc->desc.cam_right, cameras9 + size_t(f0) * 9 + 3, 3 * sizeof(float));
            std::memcpy(sc->desc.cam_up, cameras9 + size_t(f0) * 9 + 6, 3 * sizeof(float));
        }
        if (times) {
            std::lock_guard<std::mutex> lock(net->mu);
            net->net->setTimeAndEnsemble(times[f0], net->net->currentEnsemble);
            net->timeDirty = true;
        }
        float* out = d_out_local + size_t(f0) * planes;
        const int rc = world == 1 ? renderImpl(sc, net, width, height, 0, height, height, 8, 0, 1, 0, out, d_stats, st, n, cameras9 + size_t(f0) * 9)
                                  : renderImpl(sc, net, width, height, 0, height, rows, stripe_rows, rank, world, 1, out, d_stats, st, n, cameras9 + size_t(f0) * 9);
        if (rc != FVSRN_OK) return rc;
        if (d_rgba8 && rows > 0)
            for (int f = f0; f < f0 + n; ++f) {
                const int rc2 = extractImpl(d_out_local + size_t(f) * planes, width, rows, FVSRN_CHANNEL_COLOR, use_tonemapping, max_exposure, nullptr,
                                            d_rgba8 + size_t(f) * size_t(rows) * size_t(width), st);
                if (rc2 != FVSRN_OK) return rc2;
            }
    }
    return FVSRN_OK;
    });
}

}  // extern "C"
