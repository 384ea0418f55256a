// C ABI of libfvsrn.so (see include/fvsrn.h for the reference interfaces each entry replaces): handles, options, point evaluation, ExtractColor and the
// ray / TF tensor APIs.  Rendering: launch_plan.cpp; device state of a network: keyframes.cpp; grid volumes and .cvol files: cvol_io.cpp.
#include "api_internal.hpp"

extern "C" {

const char* fvsrn_last_error(void) { return g_lastError.c_str(); }
size_t fvsrn_scene_desc_size(void) { return sizeof(fvsrn_scene_desc); }
size_t fvsrn_network_info_size(void) { return sizeof(fvsrn_network_info); }
const char* fvsrn_version(void) { return "fvsrn 0.1.0 gfx950"; }

int fvsrn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fvsrn_probe_stream_concurrency(void* const* stream_handles, int streams, int microseconds, float* concurrent) {
    if (!concurrent || streams < 2 || streams > 16 || microseconds < 10 || microseconds > 100000)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "2 .. 16 streams, 10 .. 100000 microseconds, a result pointer");
    if (fvsrn_device_count() == 0) return fail(FVSRN_ERR_NO_DEVICE, "no HIP device");
    std::vector<hipStream_t> st(size_t(streams), nullptr);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    for (int i = 0; i < streams; ++i) {
        if (stream_handles) st[size_t(i)] = static_cast<hipStream_t>(stream_handles[i]);  // the caller's own streams (nullptr = the null stream)
        else ok = ok && hipStreamCreateWithFlags(&st[size_t(i)], hipStreamNonBlocking) == hipSuccess;
    }
    float ms = 0.f;
    if (ok) {
        // warm-up (module load, queue creation), then the timed round: the events sit on the first stream, which every other stream
        // is ordered against through events of its own
        for (int round = 0; round < 2 && ok; ++round) {
            ok = hipDeviceSynchronize() == hipSuccess && hipEventRecord(e0, st[0]) == hipSuccess;
            std::vector<hipEvent_t> done(size_t(streams), nullptr);
            for (int i = 0; i < streams && ok; ++i) {
                if (i > 0) ok = hipStreamWaitEvent(st[size_t(i)], e0, 0) == hipSuccess;
                ok = ok && launch_spin((long long)microseconds * 100, st[size_t(i)]) == hipSuccess;
                if (i > 0) ok = ok && hipEventCreateWithFlags(&done[size_t(i)], hipEventDisableTiming) == hipSuccess && hipEventRecord(done[size_t(i)], st[size_t(i)]) == hipSuccess &&
                                hipStreamWaitEvent(st[0], done[size_t(i)], 0) == hipSuccess;
            }
            ok = ok && hipEventRecord(e1, st[0]) == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            for (auto e : done) if (e) (void)hipEventDestroy(e);
        }
    }
    if (!stream_handles)
        for (auto s : st) if (s) (void)hipStreamDestroy(s);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (!ok || ms <= 0.f) return fail(FVSRN_ERR_DEVICE, "stream concurrency probe failed");
    *concurrent = float(streams) * float(microseconds) * 1e-3f / ms;
    return FVSRN_OK;
}

// ---------------------------------------------------------------------------------------------- network
int fvsrn_network_create_from_volnet(const void* bytes, size_t len, fvsrn_network** out) {
    return guarded([&] {
        if (!bytes || !out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        auto n = std::make_unique<fvsrn_network>();
        n->net = SceneNetwork::load(bytes, len);
        *out = n.release();
        return FVSRN_OK;
    });
}

int fvsrn_network_create(fvsrn_network** out) {
    return guarded([&] {
        if (!out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        *out = new fvsrn_network();
        return FVSRN_OK;
    });
}

void fvsrn_network_destroy(fvsrn_network* net) { delete net; }

int fvsrn_network_set_input(fvsrn_network* net, int has_time, int has_direction, const float* fourier_matrix,
                            int num_fourier, int fourier_cols, int premultiplied) {
    return guarded([&] {
        if (!net) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null network");
        std::lock_guard<std::mutex> lock(net->mu);
        net->net->input.hasTime = has_time != 0;
        net->net->input.hasDirection = has_direction != 0;
        if (num_fourier > 0) {
            if (!fourier_matrix) return fail(FVSRN_ERR_INVALID_ARGUMENT, "fourier matrix is null");
            net->net->setFourierMatrix(fourier_matrix, num_fourier, fourier_cols, premultiplied != 0);
        } else {  // disableFourierFeatures, volume_interpolation_network.cpp:158-163
            net->net->input.numFourierFeatures = 0;
            net->net->input.useDirectionInFourierFeatures = false;
            net->net->input.fourierMatrix.clear();
        }
        net->invalidate();
        return FVSRN_OK;
    });
}

int fvsrn_network_set_output_mode(fvsrn_network* net, fvsrn_output_mode mode) {
    return guarded([&] {
        if (!net || int(mode) < 0 || int(mode) > 8) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad output mode");
        std::lock_guard<std::mutex> lock(net->mu);
        net->net->outputMode = mode;
        net->invalidate();
        return FVSRN_OK;
    });
}

int fvsrn_network_add_layer(fvsrn_network* net, const float* weights, const float* bias, int channels_out,
                            int channels_in, fvsrn_activation act, float act_param) {
    return guarded([&] {
        if (!net || !weights || !bias || channels_out <= 0 || channels_in <= 0 || int(act) < 0 || int(act) > 5)
            return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad layer arguments");
        std::lock_guard<std::mutex> lock(net->mu);
        net->net->addLayerFromFloat(weights, bias, channels_out, channels_in, act, act_param);
        net->invalidate();
        return FVSRN_OK;
    });
}

int fvsrn_network_set_box(fvsrn_network* net, const float box_min[3], const float box_size[3]) {
    return guarded([&] {
        if (!net || !box_min || !box_size) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> lock(net->mu);
        for (int i = 0; i < 3; ++i) {
            net->net->boxMin[i] = box_min[i];
            net->net->boxSize[i] = box_size[i];
        }
        net->invalidate();
        return FVSRN_OK;
    });
}

int fvsrn_network_set_latent_grid_layout(fvsrn_network* net, int time_min, int time_num, int time_step,
                                         int ensemble_min, int ensemble_num) {
    return guarded([&] {
        if (!net || time_num < 0 || ensemble_num < 0 || time_step == 0)
            return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad latent grid layout");
        std::lock_guard<std::mutex> lock(net->mu);
        auto g = std::make_shared<LatentGridTimeAndEnsemble>();
        g->timeMin = time_min; g->timeNum = time_num; g->timeStep = time_step;
        g->ensembleMin = ensemble_min; g->ensembleNum = ensemble_num;
        g->timeGrids.resize(size_t(time_num));
        g->ensembleGrids.resize(size_t(ensemble_num));
        net->net->latentGrid = g;
        net->invalidate();
        return FVSRN_OK;
    });
}

int fvsrn_network_set_latent_grid(fvsrn_network* net, int is_ensemble, int index, const float* grid, int C, int Z,
                                  int Y, int X, fvsrn_grid_encoding enc, double* encoding_error) {
    return guarded([&] {
        if (!net || !grid || C <= 0 || Z <= 0 || Y <= 0 || X <= 0 || int(enc) < 0 || int(enc) > 2)
            return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad latent grid arguments");
        std::lock_guard<std::mutex> lock(net->mu);
        if (!net->net->latentGrid) return fail(FVSRN_ERR_INVALID_ARGUMENT, "set the latent grid layout first");
        auto& list = is_ensemble ? net->net->latentGrid->ensembleGrids : net->net->latentGrid->timeGrids;
        if (index < 0 || size_t(index) >= list.size()) return fail(FVSRN_ERR_INVALID_ARGUMENT, "index out of bounds!");
        list[size_t(index)] = LatentGrid::fromFloat(grid, C, Z, Y, X, enc, encoding_error);
        net->invalidate();
        return FVSRN_OK;
    });
}

int fvsrn_network_valid(const fvsrn_network* net) {
    if (!net) return 0;
    std::string why;
    const bool ok = net->net->valid(&why);
    g_lastError = ok ? "" : why;
    return ok ? 1 : 0;
}

int fvsrn_network_save_volnet(const fvsrn_network* net, void* buf, size_t cap, size_t* len) {
    return guarded([&] {
        if (!net || !len) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        const std::vector<char> bytes = net->net->save();
        *len = bytes.size();
        if (buf) {
            if (cap < bytes.size()) return fail(FVSRN_ERR_INVALID_ARGUMENT, "buffer too small");
            std::memcpy(buf, bytes.data(), bytes.size());
        }
        return FVSRN_OK;
    });
}

int fvsrn_network_set_time_and_ensemble(fvsrn_network* net, float time, int ensemble) {
    return guarded([&] {
        if (!net) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null network");
        std::lock_guard<std::mutex> lock(net->mu);
        net->net->setTimeAndEnsemble(time, ensemble);
        net->timeDirty = true;  // key frames stay resident; the next launch re-blends on the device
        return FVSRN_OK;
    });
}

int fvsrn_network_prepare(fvsrn_network* net, void* stream) {
    return guarded([&] {
        if (!net) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null network");
        std::lock_guard<std::mutex> lock(net->mu);
        try {
            hipStream_t s = static_cast<hipStream_t>(stream);
            net->ensureDevice(s);
            net->syncTime(s);
            return FVSRN_OK;
        } catch (const DeviceError& e) {
            return fail(fvsrn_device_count() == 0 ? FVSRN_ERR_NO_DEVICE : FVSRN_ERR_DEVICE, e.what());
        }
    });
}

int fvsrn_network_clear_gpu_resources(fvsrn_network* net) {
    return guarded([&] {
        if (!net) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null network");
        std::lock_guard<std::mutex> lock(net->mu);
        net->invalidate();
        net->releaseDevice();
        return FVSRN_OK;
    });
}

int fvsrn_network_get_info(const fvsrn_network* netc, fvsrn_network_info* info) {
    return guarded([&] {
        if (!netc || !info) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> lock(netc->mu);
        if (netc->infoValid) {  // bindings ask per frame (output channels, FLOP counts): nothing is recomputed until the network changes
            *info = netc->info;
            return FVSRN_OK;
        }
        const SceneNetwork& n = *netc->net;
        std::memset(info, 0, sizeof(*info));
        info->num_layers = int(n.hidden.size());
        info->num_fourier = n.input.numFourierFeatures;
        info->has_direction = n.input.hasDirection;
        info->has_time = n.input.hasTime;
        info->use_direction_in_fourier = n.input.useDirectionInFourierFeatures;
        info->output_mode = int(n.outputMode);
        info->output_channels = n.outputChannels();
        std::string why;
        // a network under construction may hold unset grids (fvsrn_network_set_latent_grid_layout before the grids): no grid data
        if (n.latentGrid && n.latentGrid->isValid(&why)) {
            info->grid_channels = n.latentGrid->totalChannels();
            info->grid_encoding = int(n.latentGrid->commonEncoding());
            const LatentGrid* g = n.latentGrid->hasTimeGrids() ? n.latentGrid->timeGrids[0].get()
                                                               : (n.latentGrid->hasEnsembleGrids() ? n.latentGrid->ensembleGrids[0].get() : nullptr);
            if (g) { info->grid_res[0] = g->gridSizeX; info->grid_res[1] = g->gridSizeY; info->grid_res[2] = g->gridSizeZ; }
        }
        if (n.latentGrid) {
            info->time_num = n.latentGrid->timeNum;
            info->ensemble_num = n.latentGrid->ensembleNum;
        }
        for (int i = 0; i < 3; ++i) { info->box_min[i] = n.boxMin[i]; info->box_size[i] = n.boxSize[i]; }
        if (n.valid(&why)) {
            try {  // getDefines-level checks (uniform activation parameter, layer shapes) are stricter than valid(): fields stay 0
                info->num_parameters = n.numParameters();
                info->max_warps_shared = n.computeMaxWarps(true, false);
                info->max_warps_mixed = n.computeMaxWarps(false, false);
                const NetworkConfig c = n.config();
                info->hidden_channels = c.hiddenChannels;
                info->activation = int(c.activation);
                info->activation_param = c.activationParam;
                info->flops_per_sample = n.flopsPerSample();
                info->mfma_flops_per_sample = mfmaFlopsPerSample(c, int(n.hidden.size()));  // no packing: bindings call this once per frame
            } catch (const Unsupported&) {
                info->mfma_flops_per_sample = 0;
            } catch (const InvalidNetwork&) {
                info->hidden_channels = 0;
            }
        }
        netc->info = *info;
        netc->infoValid = true;
        return FVSRN_OK;
    });
}

int fvsrn_network_get_layer(const fvsrn_network* net, int index, int* channels_out, int* channels_in, int* activation,
                            float* act_param, uint16_t* weights, uint16_t* bias) {
    return guarded([&] {
        if (!net || index < 0 || size_t(index) >= net->net->hidden.size())
            return fail(FVSRN_ERR_INVALID_ARGUMENT, "layer index out of bounds");
        const Layer& l = net->net->hidden[size_t(index)];
        if (channels_out) *channels_out = l.channelsOut;
        if (channels_in) *channels_in = l.channelsIn;
        if (activation) *activation = int(l.activation);
        if (act_param) *act_param = l.activationParameter;
        if (weights) std::memcpy(weights, l.weights.data(), 2 * l.weights.size());
        if (bias) std::memcpy(bias, l.bias.data(), 2 * l.bias.size());
        return FVSRN_OK;
    });
}

int fvsrn_network_get_fourier(const fvsrn_network* net, uint16_t* matrix, int cap, int* count) {
    return guarded([&] {
        if (!net || !count) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        const auto& m = net->net->input.fourierMatrix;
        *count = int(m.size());
        if (matrix) {
            if (cap < int(m.size())) return fail(FVSRN_ERR_INVALID_ARGUMENT, "buffer too small");
            std::memcpy(matrix, m.data(), 2 * m.size());
        }
        return FVSRN_OK;
    });
}

int fvsrn_network_kernel_name(fvsrn_network* net, int render, char* buf, size_t cap) {
    return guarded([&] {
        if (!net || !buf || cap == 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> lock(net->mu);
        if (!net->deviceValid) net->pack();  // (a live network keeps its packed state: it carries the device pointers)
        std::string name = render ? net->kinfoScaled.renderName : net->kinfoScaled.evalName;  // (evaluate_points runs the re-scaled image too, r03)
        if (!render && net->keyScaled.act == ACT_RELU01 && net->keyScaled.grid == 2) name = net->kinfo.evalName;  // (... but for BYTE_GAUSSIAN grids: evaluateImpl)
        if (render) {  // the register-resident kernel takes over for scenes with an Identity / Texture TF and no shading (renderImpl)
            const NetParams& P = net->packed.params;
            const VariantKey& k = net->keyScaled;
            const bool scalarNet = P.outputMode == FVSRN_OUT_DENSITY || P.outputMode == FVSRN_OUT_DENSITY_DIRECT;
            const bool colourNet = P.outputMode == FVSRN_OUT_RGBO || P.outputMode == FVSRN_OUT_RGBO_DIRECT;
            const bool folded = P.bias0Folded && (!net->scaledImage || net->packed.scaledBias0Exact);
            const bool cells = k.grid == 1 && net->opts[FVSRN_OPT_CELL_TABLE] != 0 && P.gridX >= 2 && P.gridY >= 2 && P.gridZ >= 2 &&
                               double(P.gridX + 1) * (P.gridY + 1) * (P.gridZ + 1) * 512.0 * ((net->packed.cfg.hiddenChannels + 31) / 32) <= 1073741824.0;  // (ensureDevice: cellTableBytes)
            const int smallGrid = k.grid == 0 ? 0 : (k.grid == 1 && folded ? (cells ? 2 : (P.gridK == 1 ? 1 : 3)) : 3);
            // (the latent-grid path is chosen per launch: FVSRN_OPT_CELL_TABLE = -1 takes the table by the footprint of a pixel tile, renderImpl)
            const std::string byFootprint = cells && net->opts[FVSRN_OPT_CELL_TABLE] == -1 ? "; cells or gathers by footprint" : "";
            if (net->opts[FVSRN_OPT_SMALL_KERNEL] != 0 && k.CD == 2 && smallGrid <= 2 && !P.noFourier && !P.fourierNeedsFractPlain && !P.fourierClampPos && (scalarNet || colourNet) &&
                (render_small_fn(k.act, k.dir, P.numLayers, colourNet ? 3 : 1, smallGrid) ||
                 (smallGrid == 2 && P.gridK == 1 && render_small_fn(k.act, k.dir, P.numLayers, colourNet ? 3 : 1, 1))))
                name = "render_small_kernel<act " + std::to_string(k.act) + "," + (k.dir ? "true" : "false") + "," + std::to_string(P.numLayers) +
                       ",SGRID=" + std::to_string(smallGrid) + "> (unshaded" + byFootprint + "; else " + name + ")";
            else if (cells && render_cells_fn(k))  // the decoded latent grid through the cell table (renderImpl)
                name = "render_cells_kernel<" + std::to_string(k.CD) + ",act " + std::to_string(k.act) + "," + (k.dir ? "true" : "false") + "> (unshaded" + byFootprint + "; else " + name + ")";
        }
        std::strncpy(buf, name.c_str(), cap - 1);
        buf[cap - 1] = 0;
        return FVSRN_OK;
    });
}

// --------------------------------------------------------------------------------------------- options
static int setOption(Options& o, int option, int value) {
    if (option < 0 || option >= FVSRN_OPT_COUNT_) return fail(FVSRN_ERR_INVALID_ARGUMENT, "unknown option");
    const std::string why = Options::check(option, value);
    if (!why.empty()) return fail(FVSRN_ERR_INVALID_ARGUMENT, "option " + std::to_string(option) + ": " + why);
    o.v[option] = value;
    return FVSRN_OK;
}

int fvsrn_network_set_option(fvsrn_network* net, int option, int value) {
    return guarded([&] {
        if (!net) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null network");
        std::lock_guard<std::mutex> lock(net->mu);
        const int old = option >= 0 && option < FVSRN_OPT_COUNT_ ? net->opts[option] : 0;
        const int rc = setOption(net->opts, option, value);
        // the weight image / the key-frame residency are part of the device state
        if (rc == FVSRN_OK && (option == FVSRN_OPT_RELU_CLAMP || option == FVSRN_OPT_KEYFRAME_SLOTS || option == FVSRN_OPT_WORKING_GRIDS ||
                                 (option == FVSRN_OPT_CELL_TABLE && (old == 0) != (value == 0))) && old != value) net->invalidate();  // (cell table: built or not)
        return rc;
    });
}

int fvsrn_scene_set_option(fvsrn_scene* scene, int option, int value) {
    return guarded([&] {
        if (!scene) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null scene");
        std::lock_guard<std::mutex> lock(scene->mu);
        return setOption(scene->opts, option, value);
    });
}

int fvsrn_network_get_option(const fvsrn_network* net, int option, int* value) {
    if (!net || !value || option < 0 || option >= FVSRN_OPT_COUNT_) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad option query");
    std::lock_guard<std::mutex> lock(net->mu);
    *value = net->opts[option];
    return FVSRN_OK;
}

int fvsrn_network_keyframe_stats(const fvsrn_network* net, unsigned long long out[6]) {
    if (!net || !out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(net->mu);
    const KeyframeStore& k = net->keyStore;
    out[0] = (unsigned long long)k.numKeys; out[1] = (unsigned long long)k.slots;
    for (int i = 0; i < 4; ++i) out[2 + i] = k.stats[i];
    return FVSRN_OK;
}

int fvsrn_network_cell_table_stats(const fvsrn_network* net, unsigned long long out[4]) {
    if (!net || !out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(net->mu);
    out[0] = (unsigned long long)net->cellTableBytes;
    out[1] = net->cellTableBuilds[0];
    out[2] = net->cellTableBuilds[1];
    out[3] = 0;
    for (const auto& w : net->workGrid) out[3] += (unsigned long long)(w.cells.cap + w.cellsPlain.cap);
    return FVSRN_OK;
}

int fvsrn_scene_get_option(fvsrn_scene* scene, int option, int* value) {
    if (!scene || !value || option < 0 || option >= FVSRN_OPT_COUNT_) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad option query");
    std::lock_guard<std::mutex> lock(scene->mu);
    *value = scene->opts[option];
    return FVSRN_OK;
}

// ------------------------------------------------------------------------------------------- evaluation
static int evaluateImpl(fvsrn_network* net, const float* d_positions, const float* d_directions, size_t n, float* d_out, int flags,
                        bool adjoint, float adjointGridStep, void* stream, bool halfIO = false) {
    return guarded([&] {
        if (!net || (n > 0 && (!d_positions || !d_out))) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> lock(net->mu);
        try {
            hipStream_t s = static_cast<hipStream_t>(stream);
            net->ensureDevice(s);
            net->syncTime(s);
            if (n == 0) return FVSRN_OK;
            net->beginUse(s);
            struct Done { fvsrn_network* n; hipStream_t s; ~Done() { try { n->endUse(s); } catch (...) {} } } done{net, s};
            if (net->key.dir && !d_directions)
                return fail(FVSRN_ERR_INVALID_ARGUMENT, "the network uses the view direction, but no directions were given");
            EvalArgs a{net->packed.params, d_positions, d_directions, n, d_out, net->net->outputChannels()};
            // The re-scaled weight images (pack.cpp) serve evaluate_points too: SnakeAlt with its affine part in the next layer is exact
            // algebra for any input; the [0,1]-scaled ReLU image is bounded for points inside the unit box, so the kernel checks every
            // batch of 64 points and runs a batch with a point outside through the same image with the unclamped activation (kernels.hpp,
            // eval_batch_outside; r05: one launch for every n -- r03 / r04 deferred such batches to a second launch through a per-call list,
            // ~9 us per call, and therefore took the plain image below 2^23 points).
            VariantKey evalKey = net->key;
            a.P.evalHalfIO = halfIO ? 1 : 0;
            if (halfIO && (adjoint || (flags & FVSRN_EVAL_WITH_PREDICTED_CURVATURE)))
                return fail(FVSRN_ERR_INVALID_ARGUMENT, "fp16 positions / values: plain evaluation and predicted gradients only");
            // (BYTE_GAUSSIAN grids keep the plain ReLU image: their kernels decode the grid themselves and have no registers left for a second pass)
            if (net->scaledImage && !adjoint && !(net->keyScaled.act == ACT_RELU01 && net->keyScaled.grid == 2)) {
                evalKey = net->keyScaled;
                a.P.ldsImage = net->scaledImage;
                if (!net->packed.scaledBias0Exact) a.P.bias0Folded = 0;  // (a residue of the folded bias sits in the fp32 block: pack.cpp)
                a.P.reluClamp = net->keyScaled.act == ACT_RELU01 ? 1 : 0;
            }
            void* evalTmp = nullptr;
            struct FreeTmp { void*& p; hipStream_t s; ~FreeTmp() { if (p) (void)hipFreeAsync(p, s); } } freeTmp{evalTmp, s};
            const bool curvature = (flags & FVSRN_EVAL_WITH_PREDICTED_CURVATURE) != 0;
            if (curvature) {
                // evalCurvature (renderer_volume_tensorcores.cuh:1541-1556): only networks that estimate it, GRADIENT_MODE_OFF_OR_DIRECT
                const int om = a.P.outputMode;
                if (adjoint || (om != FVSRN_OUT_DENSITY_CURVATURE && om != FVSRN_OUT_DENSITY_CURVATURE_DIRECT) || !net->curvatureImage)
                    return fail(FVSRN_ERR_INVALID_ARGUMENT, "curvature is only available from networks that predict it (output mode densitycurvature*)");
                // two passes over the points -- value + predicted gradient, then the same layers with the last one computing the two
                // curvature outputs -- into a temporary (n,4) + (n,4), combined into d_out (n,6) by two strided copies
                // (allocated and freed in stream order, per call: two calls on different streams share nothing)
                evalTmp = g_temporaries.alloc(net->device, n * 8 * sizeof(float), s);
                a.out = static_cast<float*>(evalTmp);
                a.outChannels = 4;
            } else if (flags & FVSRN_EVAL_WITH_PREDICTED_GRADIENT) {
                const int om = a.P.outputMode;
                if (om < FVSRN_OUT_DENSITY_GRADIENT || om > FVSRN_OUT_DENSITY_CURVATURE_DIRECT)
                    return fail(FVSRN_ERR_INVALID_ARGUMENT, "the network does not predict gradients (output mode densitygrad* / densitycurvature*)");
                a.outChannels = 4;
            }
            if (!(flags & FVSRN_EVAL_WORLD_POSITIONS))  // volume_interpolation.cpp:46-49: box := [0,1]^3
                for (int i = 0; i < 3; ++i) { a.P.boxMin[i] = 0.f; a.P.boxSize[i] = 1.f; a.P.invBoxSize[i] = 1.f; }
            const size_t batches = (n + 63) / 64;
            const size_t wpb = size_t(wavesPerBlockFor(size_t(net->packed.params.ldsBytes), net->opts));
            const size_t blocks = (batches + wpb - 1) / wpb;
            if (adjoint) {
                // evalNormal in GRADIENT_MODE_ADJOINT_METHOD differentiates output 0 of a scalar network
                if (flags & FVSRN_EVAL_WITH_PREDICTED_GRADIENT) return fail(FVSRN_ERR_INVALID_ARGUMENT, "predicted and adjoint gradients exclude each other");
                if (a.P.outputMode == FVSRN_OUT_RGBO || a.P.outputMode == FVSRN_OUT_RGBO_DIRECT)
                    return fail(FVSRN_ERR_INVALID_ARGUMENT, "gradients can only be evaluated for scalar networks");
                if (!(adjointGridStep >= 0.f)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "adjoint_grid_stepsize must not be negative");
                const float gridStep = adjointGridStep > 0.f ? adjointGridStep : 1.0f / (float(std::max(1, a.P.gridX)) * 4.0f);
                const unsigned gridG = unsigned(std::min<size_t>(blocks, size_t(net->numCUs) * 8 / wpb));  // (two waves per SIMD where the variant fits 256 registers)
                const hipError_t e = launch_eval_gradient(net->key, a, gridStep, gridG, unsigned(64 * wpb), size_t(net->packed.params.ldsBytes), s);
                if (e == hipErrorInvalidDeviceFunction) return fail(FVSRN_ERR_UNSUPPORTED, "this network variant has no gradient kernel");
                if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("Error during evaluation! ") + hipGetErrorString(e));
                return FVSRN_OK;
            }
            const unsigned grid = unsigned(std::min<size_t>(blocks, size_t(net->numCUs) * 32 / wpb));
            // small networks in registers (evaluate_small_kernel): see renderImpl; the plain weight image, any output mode
            hipError_t e = hipErrorInvalidDeviceFunction;
            {
                const VariantKey& k = evalKey;
                const int smallGrid = k.grid == 0 ? 0 : (k.grid == 1 && a.P.gridK == 1 && a.P.bias0Folded ? 1 : 2);  // one decoded 16-channel chunk, as in renderImpl
                // (no v_fract in the register-resident kernels: the bound on the phases that applies -- the in-box one for the scaled ReLU image)
                const int needsFract = k.act == ACT_RELU01 ? a.P.fourierNeedsFractPlain : a.P.fourierNeedsFractEval;
                if (net->opts[FVSRN_OPT_SMALL_KERNEL] != 0 && k.CD == 2 && smallGrid <= 1 && !a.P.noFourier && !needsFract && a.P.numLayers >= 1 &&
                    a.P.numLayers <= 3) {
                    // (the Fourier-only kernels need 156 registers: three waves per SIMD fit, and the grid-stride loop profits from them)
                    // (FVSRN_OPT_MAX_BLOCKS_PER_CU on the network: waves per CU of this launch, for occupancy experiments)
                    const int wavesPerCU = net->opts[FVSRN_OPT_MAX_BLOCKS_PER_CU] > 0 ? net->opts[FVSRN_OPT_MAX_BLOCKS_PER_CU] : (smallGrid == 0 ? 12 : 8);
                    // four waves per workgroup: the network lives in registers, a workgroup only shares the copy of the image it is loaded from
                    // (a quarter of the workgroups and LDS copies of a launch: 2^20 points 78 -> 83, 2^22 102 -> 107 G points/s, r04)
                    const size_t wpbS = net->opts[FVSRN_OPT_WAVES_PER_BLOCK] ? wpb : 4;
                    const unsigned gridSmall = unsigned(std::min<size_t>((batches + wpbS - 1) / wpbS, size_t(net->numCUs) * size_t(wavesPerCU) / wpbS));
                    e = launch_eval_small(k.act, k.dir, a.P.numLayers, smallGrid, a, gridSmall, unsigned(64 * wpbS), size_t(net->packed.params.ldsBytes), s);
                }
            }
            if (e == hipErrorInvalidDeviceFunction)
                e = launch_eval(evalKey, a, grid, unsigned(64 * wpb), size_t(net->packed.params.ldsBytes), s);
            if (e == hipSuccess && curvature) {
                EvalArgs c = a;
                c.P = net->packed.params;  // (the curvature image is a variant of the plain image)
                if (!(flags & FVSRN_EVAL_WORLD_POSITIONS))
                    for (int i = 0; i < 3; ++i) { c.P.boxMin[i] = 0.f; c.P.boxSize[i] = 1.f; c.P.invBoxSize[i] = 1.f; }
                c.P.ldsImage = net->curvatureImage;
                c.P.outputMode = FVSRN_OUT_DENSITY_GRADIENT_DIRECT;  // rows 0, 1 of the last layer, raw
                c.out = a.out + 4 * n;
                e = launch_eval(net->key, c, grid, unsigned(64 * wpb), size_t(net->packed.params.ldsBytes), s);
                if (e == hipSuccess)
                    e = hipMemcpy2DAsync(d_out, 6 * sizeof(float), a.out, 4 * sizeof(float), 4 * sizeof(float), n, hipMemcpyDeviceToDevice, s);
                if (e == hipSuccess)
                    e = hipMemcpy2DAsync(d_out + 4, 6 * sizeof(float), c.out, 4 * sizeof(float), 2 * sizeof(float), n, hipMemcpyDeviceToDevice, s);
            }
            if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("Error during evaluation! ") + hipGetErrorString(e));
            return FVSRN_OK;
        } catch (const DeviceError& e) {
            return fail(fvsrn_device_count() == 0 ? FVSRN_ERR_NO_DEVICE : FVSRN_ERR_DEVICE, e.what());
        }
    });
}

int fvsrn_evaluate_points(fvsrn_network* net, const float* d_positions, const float* d_directions, size_t n, float* d_out, int flags,
                          void* stream) {
    return evaluateImpl(net, d_positions, d_directions, n, d_out, flags, false, 0.f, stream);
}

int fvsrn_evaluate_points_half(fvsrn_network* net, const void* d_positions_f16, const void* d_directions_f16, size_t n, void* d_out_f16, int flags,
                               void* stream) {
    return evaluateImpl(net, static_cast<const float*>(d_positions_f16), static_cast<const float*>(d_directions_f16), n, static_cast<float*>(d_out_f16), flags, false,
                        0.f, stream, true);
}

int fvsrn_evaluate_points_adjoint(fvsrn_network* net, const float* d_positions, const float* d_directions, size_t n, float* d_out4,
                                  float adjoint_grid_stepsize, int flags, void* stream) {
    return evaluateImpl(net, d_positions, d_directions, n, d_out4, flags, true, adjoint_grid_stepsize, stream);
}

// ------------------------------------------------------------------------------------------------ scene
static int sceneValidate(const fvsrn_scene_desc* d) {
    if (!d) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null scene description");
    if (d->tf_kind < FVSRN_TF_NONE || d->tf_kind > FVSRN_TF_TEXTURE) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad tf_kind");
    if (tfCols(d->tf_kind) > 0 && (!d->tf_table || d->tf_rows <= 0))
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "this transfer function needs a table");
    if (d->tf_kind == FVSRN_TF_PIECEWISE && d->tf_rows < 2)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "a piecewise transfer function needs at least two control points");
    if (tfCols(d->tf_kind) * d->tf_rows > 1024)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "transfer function table too large (max 1024 floats)");
    if (!(d->stepsize > 0)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "stepsize must be positive");
    if (d->gradient_mode < FVSRN_GRADIENT_OFF_OR_DIRECT || d->gradient_mode > FVSRN_GRADIENT_ADJOINT_METHOD)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad gradient mode");
    if (!(d->adjoint_grid_stepsize >= 0)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "adjoint_grid_stepsize must not be negative");
    if (d->gradient_mode == FVSRN_GRADIENT_FINITE_DIFFERENCES && !(d->finite_differences_stepsize > 0))
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "finite_differences_stepsize must be positive");
    if (d->brdf_light_type != FVSRN_LIGHT_POINT && d->brdf_light_type != FVSRN_LIGHT_DIRECTIONAL)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad brdf_light_type");
    if (d->tf_preintegration < FVSRN_PREINTEGRATE_NONE || d->tf_preintegration > FVSRN_PREINTEGRATE_2D)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad tf_preintegration");
    if (d->tf_preintegration != FVSRN_PREINTEGRATE_NONE && d->tf_kind != FVSRN_TF_TEXTURE)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "pre-integration is a mode of the Texture transfer function");
    if (d->blend_mode != FVSRN_BLEND_ALPHA && d->blend_mode != FVSRN_BLEND_BEER_LAMBERT)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad blend mode");
    if (d->tf_gaussian_mode < FVSRN_TF_GAUSSIAN_PLAIN || d->tf_gaussian_mode > FVSRN_TF_GAUSSIAN_ANALYTIC)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad tf_gaussian_mode");
    if (d->tf_gaussian_mode != FVSRN_TF_GAUSSIAN_PLAIN && d->tf_kind != FVSRN_TF_GAUSSIAN)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "tf_gaussian_mode is a mode of the Gaussian transfer function");
    return FVSRN_OK;
}

int fvsrn_scene_update(fvsrn_scene* scene, const fvsrn_scene_desc* desc) {
    return guarded([&] {
        if (!scene) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null scene");
        if (int r = sceneValidate(desc)) return r;
        std::lock_guard<std::mutex> lock(scene->mu);
        const size_t n = size_t(tfCols(desc->tf_kind)) * size_t(std::max(desc->tf_rows, 0));
        std::vector<float> table(desc->tf_table ? desc->tf_table : nullptr, desc->tf_table ? desc->tf_table + n : nullptr);
        if (table != scene->tfTable) scene->tfDirty = true;
        scene->tfTable = std::move(table);
        scene->tfOpacityNonNegative = true;
        if (desc->tf_kind == FVSRN_TF_TEXTURE)
            for (size_t i = 3; i < scene->tfTable.size(); i += 4)
                if (!(scene->tfTable[i] >= 0.f)) scene->tfOpacityNonNegative = false;
        scene->desc = *desc;
        scene->desc.tf_table = nullptr;
        return FVSRN_OK;
    });
}

int fvsrn_scene_create(const fvsrn_scene_desc* desc, fvsrn_scene** out) {
    return guarded([&] {
        if (!out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        if (int r = sceneValidate(desc)) return r;
        auto s = std::make_unique<fvsrn_scene>();
        const int r = fvsrn_scene_update(s.get(), desc);
        if (r != FVSRN_OK) return r;
        *out = s.release();
        return FVSRN_OK;
    });
}

void fvsrn_scene_destroy(fvsrn_scene* scene) { delete scene; }

int fvsrn_camera_on_a_sphere(int orientation, const double center[3], double pitch, double yaw, double distance,
                             float eye[3], float right[3], float up[3]) {
    return guarded([&] {
        if (orientation < 0 || orientation > 5 || !center || !eye || !right || !up)
            return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad camera arguments");
        // tables of renderer/camera.cpp:17-35
        static const double kUp[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
        static const int kPerm[6][3] = {{2, -1, -3}, {-2, 1, 3}, {1, 2, 3}, {-1, -2, -3}, {-3, -1, 2}, {3, 1, -2}};
        static const bool kInvertYaw[6] = {false, true, true, false, true, false};
        // eulerToCartesian, camera.cpp:553-569
        const double y2 = !kInvertYaw[orientation] ? -yaw : +yaw;
        const double p2 = -pitch;  // OrientationInvertPitch is false for every orientation
        const double pos[3] = {std::cos(p2) * std::cos(y2) * distance, std::sin(p2) * distance,
                               std::cos(p2) * std::sin(y2) * distance};
        double origin[3];
        for (int i = 0; i < 3; ++i) {
            const int p = kPerm[orientation][i];
            origin[i] = pos[std::abs(p) - 1] * (p > 0 ? 1 : -1) + center[i];
        }
        // look-at frame, camera.cpp:484-490
        auto norm = [](double v[3]) {
            const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            v[0] /= l; v[1] /= l; v[2] /= l;
        };
        auto cross = [](const double a[3], const double b[3], double o[3]) {
            o[0] = a[1] * b[2] - a[2] * b[1];
            o[1] = a[2] * b[0] - a[0] * b[2];
            o[2] = a[0] * b[1] - a[1] * b[0];
        };
        double front[3] = {center[0] - origin[0], center[1] - origin[1], center[2] - origin[2]};
        norm(front);
        double r[3], u[3];
        cross(front, kUp[orientation], r);
        norm(r);
        cross(r, front, u);
        norm(u);
        for (int i = 0; i < 3; ++i) { eye[i] = float(origin[i]); right[i] = float(r[i]); up[i] = float(u[i]); }
        return FVSRN_OK;
    });
}

// camera, ray stepping, TF and BRDF constants of a scene (device tables must be uploaded: fvsrn_scene::uploadTf)
}  // extern "C"

// (shared with launch_plan.cpp: C++ linkage)
int extractImpl(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping, float max_exposure,
                float* d_out4, unsigned int* d_out8, void* stream, const float* d_range3) {
    if (!d_raw8 || (!d_out4 && !d_out8)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null image pointer");
    if (width <= 0 || height <= 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad image size");
    if (channel_mode < FVSRN_CHANNEL_MASK || channel_mode > FVSRN_CHANNEL_COLOR) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad channel mode");
    if (use_tonemapping && !(max_exposure > 0)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "max_exposure must be positive");
    return guarded([&]() -> int {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(FVSRN_ERR_NO_DEVICE, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
        // 3 words of scratch for the depth range, allocated per call in stream order on the CURRENT device (r06, ADVICE r05: a thread_local buffer was shared
        // by every stream a thread drives -- two DEPTH extracts on two streams raced on it -- and lived on the device of its first use)
        hipStream_t st = static_cast<hipStream_t>(stream);
        int dev = 0;
        void* scratch = nullptr;
        try {
            HIP_CHECK(hipGetDevice(&dev));
            scratch = g_temporaries.alloc(dev, 16, st);
        } catch (const DeviceError& e) {
            return fail(FVSRN_ERR_DEVICE, e.what());
        }
        struct Free { void* p; hipStream_t s; ~Free() { (void)hipFreeAsync(p, s); } } freeScratch{scratch, st};
        ExtractParams p{};
        p.raw = d_raw8; p.out4 = d_out4; p.out8 = d_out8;
        p.minmax = static_cast<float*>(scratch);
        p.range3 = channel_mode == FVSRN_CHANNEL_DEPTH ? d_range3 : nullptr;
        p.pixels = (unsigned long long)width * (unsigned long long)height;
        p.mode = channel_mode; p.tonemap = use_tonemapping; p.maxExposure = max_exposure;
        const hipError_t e = launch_extract_color(p, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("extract_color failed: ") + hipGetErrorString(e));
        return FVSRN_OK;
    });
}

extern "C" {

int fvsrn_extract_color(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping, float max_exposure,
                        float* d_out4, void* stream) {
    return extractImpl(d_raw8, width, height, channel_mode, use_tonemapping, max_exposure, d_out4, nullptr, stream);
}

int fvsrn_extract_color_rgba8(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping,
                              float max_exposure, unsigned int* d_out, void* stream) {
    return extractImpl(d_raw8, width, height, channel_mode, use_tonemapping, max_exposure, nullptr, d_out, stream);
}

int fvsrn_extract_color_ranged(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping, float max_exposure,
                               const float* d_range3, float* d_out4, unsigned int* d_out8, void* stream) {
    if ((d_out4 != nullptr) == (d_out8 != nullptr)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "exactly one of d_out4 / d_out8 must be given");
    return extractImpl(d_raw8, width, height, channel_mode, use_tonemapping, max_exposure, d_out4, d_out8, stream, d_range3);
}

int fvsrn_depth_range(const float* d_raw8, int width, int height, float* d_range3, void* stream) {
    if (!d_raw8 || !d_range3) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null pointer");
    if (width <= 0 || height <= 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad image size");
    return guarded([&]() -> int {
        if (fvsrn_device_count() == 0) return fail(FVSRN_ERR_NO_DEVICE, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
        hipStream_t st = static_cast<hipStream_t>(stream);
        int dev = 0;
        void* scratch = nullptr;  // (per call, stream-ordered, on the current device: see extractImpl)
        try {
            HIP_CHECK(hipGetDevice(&dev));
            scratch = g_temporaries.alloc(dev, 16, st);
        } catch (const DeviceError& e) {
            return fail(FVSRN_ERR_DEVICE, e.what());
        }
        struct Free { void* p; hipStream_t s; ~Free() { (void)hipFreeAsync(p, s); } } freeScratch{scratch, st};
        const unsigned long long pixels = (unsigned long long)width * (unsigned long long)height;
        const hipError_t e = launch_depth_range(d_raw8 + 7 * pixels, pixels, static_cast<float*>(scratch), d_range3, st);
        if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("depth_range failed: ") + hipGetErrorString(e));
        return FVSRN_OK;
    });
}

int fvsrn_generate_rays(const float eye[3], const float right[3], const float up[3], float fov_y_radians, int width, int height,
                        float* d_ray_start, float* d_ray_dir, void* stream) {
    if (!eye || !right || !up || !d_ray_start || !d_ray_dir) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null pointer");
    if (width <= 0 || height <= 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad image size");
    return guarded([&]() -> int {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(FVSRN_ERR_NO_DEVICE, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
        SceneParams S{};
        for (int i = 0; i < 3; ++i) { S.eye[i] = eye[i]; S.right[i] = right[i]; S.up[i] = up[i]; }
        S.front[0] = S.up[1] * S.right[2] - S.up[2] * S.right[1];  // cross(up, right), renderer_camera.cuh:47
        S.front[1] = S.up[2] * S.right[0] - S.up[0] * S.right[2];
        S.front[2] = S.up[0] * S.right[1] - S.up[1] * S.right[0];
        S.tanFovY = std::tan(fov_y_radians / 2);
        S.tanFovX = S.tanFovY * (float(width) / float(height));
        S.width = width; S.height = height;
        const hipError_t e = launch_generate_rays(S, d_ray_start, d_ray_dir, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("generate_rays failed: ") + hipGetErrorString(e));
        return FVSRN_OK;
    });
}

int fvsrn_scene_evaluate_tf(fvsrn_scene* scene, const float* d_density, const float* d_previous_density, size_t n, float density_min,
                            float density_max, float stepsize, float* d_colors, void* stream) {
    if (!scene || !d_density || !d_colors) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null pointer");
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> lock(scene->mu);
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(FVSRN_ERR_NO_DEVICE, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
        const fvsrn_scene_desc& d = scene->desc;
        if (d.tf_kind == FVSRN_TF_NONE) return fail(FVSRN_ERR_INVALID_ARGUMENT, "the scene has no transfer function");
        if (n == 0) return FVSRN_OK;
        hipStream_t s = static_cast<hipStream_t>(stream);
        // evaluate(): no previous density, step size 1; evaluate_with_previous(): the caller's step size
        const float step = d_previous_density ? stepsize : 1.0f;
        if (const int rc = scene->uploadTf(step, s)) return rc;
        SceneParams S{};
        S.stepsize = step;
        S.densityMin = density_min;
        S.divDensityRange = 1.0f / (density_max - density_min);
        S.tfKind = d.tf_kind; S.tfRows = d.tf_rows; S.tfRowsF = float(d.tf_rows);
        S.tfScaleAbsorption = d.tf_scale_absorption; S.tfScaleEmission = d.tf_scale_emission;
        S.tfTable = static_cast<const float*>(scene->dTf.ptr);
        S.tfPreintegration = d.tf_preintegration;
        S.tfPreintegrated = static_cast<const float*>(scene->dPreint.ptr);
        S.tfGaussianMode = d.tf_gaussian_mode;
        const hipError_t e = launch_evaluate_tf(S, d_density, d_previous_density, n, d_colors, s);
        if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("evaluate_tf failed: ") + hipGetErrorString(e));
        return FVSRN_OK;
    });
}

}  // extern "C"
