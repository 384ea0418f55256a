// Device state of a network handle: the key-frame store (pinned host key frames, device slots, copy stream), uploads of the weight images, the key-frame
// blend into the working grids and the lazily built cell tables.  Declarations and the small members: api_internal.hpp.
#include "api_internal.hpp"

void KeyframeStore::release() {
    for (StreamOrder& o : order) o.release();
    order.clear();
    if (copyStream) { (void)hipStreamSynchronize(copyStream); (void)hipStreamDestroy(copyStream); copyStream = nullptr; }
    if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; }
    dSlots.release();
    numKeys = slots = 0;
}

void KeyframeStore::init(const std::vector<char>& data, int keys, int budget) {
    release();
    if (keys <= 0 || data.empty()) return;
    numKeys = keys;
    bytesPerKey = data.size() / size_t(keys);
    slots = budget <= 0 ? keys : std::min(keys, std::max(budget, 2));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&pinned), data.size(), hipHostMallocDefault));
    std::memcpy(pinned, data.data(), data.size());
    dSlots.ensure(size_t(slots) * bytesPerKey);
    HIP_CHECK(hipStreamCreateWithFlags(&copyStream, hipStreamNonBlocking));
    keyOfSlot.assign(size_t(slots), -1);
    slotOfKey.assign(size_t(keys), -1);
    lastUse.assign(size_t(slots), 0);
    order.assign(size_t(slots), StreamOrder{});
    tick = 0;
    lastTime = -1.f;
    if (slots == keys)  // everything resident: upload now, asynchronously, the first blend waits for what it needs
        for (int k = 0; k < keys; ++k) upload(k, k, false);
}

void KeyframeStore::upload(int key, int slot, bool prefetch) {
    order[size_t(slot)].beginWrite(copyStream);  // every blend kernel that read the old content, on whatever stream
    HIP_CHECK(hipMemcpyAsync(const_cast<char*>(slotPtr(slot)), pinned + size_t(key) * bytesPerKey, bytesPerKey, hipMemcpyHostToDevice, copyStream));
    order[size_t(slot)].endWrite(copyStream);
    if (keyOfSlot[size_t(slot)] >= 0) slotOfKey[size_t(keyOfSlot[size_t(slot)])] = -1;
    keyOfSlot[size_t(slot)] = key;
    slotOfKey[size_t(key)] = slot;
    ++stats[0];
    ++stats[prefetch ? 2 : 1];
    stats[3] += bytesPerKey;
}

int KeyframeStore::victim(int keepA, int keepB) const {  // least recently used slot that holds neither key
    int best = -1;
    for (int i = 0; i < slots; ++i) {
        const int k = keyOfSlot[size_t(i)];
        if (k >= 0 && (k == keepA || k == keepB)) continue;
        if (k < 0) return i;
        if (best < 0 || lastUse[size_t(i)] < lastUse[size_t(best)]) best = i;
    }
    return best;
}

void KeyframeStore::acquire(int lo, int hi, float time, hipStream_t stream, const void** pLo, const void** pHi) {
    ++tick;
    for (int key : {lo, hi}) {
        if (slotOfKey[size_t(key)] < 0) upload(key, victim(lo, hi), false);
        const int s = slotOfKey[size_t(key)];
        lastUse[size_t(s)] = tick;
        order[size_t(s)].beginRead(stream);
    }
    *pLo = slotPtr(slotOfKey[size_t(lo)]);
    *pHi = slotPtr(slotOfKey[size_t(hi)]);
    // prefetch the key frame the time is moving towards, if a slot is free of this frame's two
    if (slots >= 3 && slots < numKeys && lastTime >= 0.f && time != lastTime) {
        const int next = time > lastTime ? hi + 1 : lo - 1;
        if (next >= 0 && next < numKeys && slotOfKey[size_t(next)] < 0) {
            const int v = victim(lo, hi);
            if (v >= 0) upload(next, v, true);
        }
    }
    lastTime = time;
}

void KeyframeStore::released(int lo, int hi, hipStream_t stream) {
    for (int key : {lo, hi}) {
        order[size_t(slotOfKey[size_t(key)])].endRead(stream);
        if (hi == lo) break;
    }
}

void fvsrn_network::pack() {  // host part only (no GPU needed): variant selection + LDS image
    packed = packNetwork(*net);
    key.CD = packed.cfg.hiddenChannels / 16;
    key.act = actIndex(packed.cfg.activation);
    key.grid = packed.cfg.gridChannels == 0 ? 0 : (packed.cfg.gridEncoding == FVSRN_GRID_BYTE_GAUSSIAN ? 2 : 1);
    key.dir = packed.cfg.directionMode > 0;
    if (!kernel_info(key, &kinfo))
        throw Unsupported("no ahead-of-time kernel for hidden width " + std::to_string(packed.cfg.hiddenChannels) +
                          ", activation " + activationName(packed.cfg.activation) +
                          (key.dir ? ", with view direction" : "") + " (compiled: widths 16 .. 128 in steps of 16)");
    keyScaled = key;
    kinfoScaled = kinfo;
    if (!packed.ldsImageScaled.empty() && opts[FVSRN_OPT_RELU_CLAMP]) {  // ACT_RELU01 / ACT_SNAKEALT0 image (pack.cpp)
        keyScaled.act = packed.scaledAct;
        if (!kernel_info(keyScaled, &kinfoScaled)) throw Unsupported("kernel variant of the re-scaled weight image missing");
    }
}

void fvsrn_network::ensureDevice(hipStream_t stream) {
    if (deviceValid) return bindOrCheckDevice(device, "the network");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        throw DeviceError("no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
    if (!dLds.ptr) device = -1;  // nothing resident (new handle or after clear_gpu_resources): bind to the current device
    bindOrCheckDevice(device, "the network");
    pack();
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    numCUs = prop.multiProcessorCount;
    // a changed network (rare): kernels on the streams that used it may still read the old images -- wait for exactly those streams
    // (no event is recorded behind the launches: that record cost 3 % of a 0.3 ms frame, r03)
    if (dLds.ptr) {
        bool unknown = imageReaders.size() >= 64;  // (the list is bounded: beyond it, or for a stream that no longer exists, the device)
        for (hipStream_t rs : imageReaders)
            if (rs != stream && hipStreamSynchronize(rs) != hipSuccess) { (void)hipGetLastError(); unknown = true; }
        if (unknown) HIP_CHECK(hipDeviceSynchronize());
    }
    imageReaders.clear();
    dLds.ensure(packed.ldsImage.size());
    HIP_CHECK(hipMemcpyAsync(dLds.ptr, packed.ldsImage.data(), packed.ldsImage.size(), hipMemcpyHostToDevice, stream));
    packed.params.ldsImage = dLds.ptr;
    packed.params.reluClamp = 0;
    scaledImage = nullptr;
    if (keyScaled.act != key.act) {
        dLdsScaled.ensure(packed.ldsImageScaled.size());
        HIP_CHECK(hipMemcpyAsync(dLdsScaled.ptr, packed.ldsImageScaled.data(), packed.ldsImageScaled.size(), hipMemcpyHostToDevice, stream));
        scaledImage = dLdsScaled.ptr;
    }
    curvatureImage = nullptr;
    if (!packed.ldsImageCurvature.empty()) {  // densitycurvature networks: last layer = the two curvature outputs (pack.cpp)
        dLdsCurvature.ensure(packed.ldsImageCurvature.size());
        HIP_CHECK(hipMemcpyAsync(dLdsCurvature.ptr, packed.ldsImageCurvature.data(), packed.ldsImageCurvature.size(), hipMemcpyHostToDevice, stream));
        curvatureImage = dLdsCurvature.ptr;
    }
    // latent key frames: uploaded once and kept resident; the working grid is blended from them on the device
    const GridKeyframes& K = packed.keys;
    if (K.records) {
        keyStore.init(K.timeData, K.timeNum, opts[FVSRN_OPT_KEYFRAME_SLOTS]);
        if (!K.ensData.empty()) {
            dKeysEns.ensure(K.ensData.size());
            HIP_CHECK(hipMemcpyAsync(dKeysEns.ptr, K.ensData.data(), K.ensData.size(), hipMemcpyHostToDevice, stream));
        }
        std::vector<float> coeffs;
        for (const auto* v : {&K.timeOffset, &K.timeScale, &K.ensOffset, &K.ensScale}) coeffs.insert(coeffs.end(), v->begin(), v->end());
        dCoeffs.ensure(std::max<size_t>(coeffs.size(), 1) * 4);
        if (!coeffs.empty()) HIP_CHECK(hipMemcpyAsync(dCoeffs.ptr, coeffs.data(), coeffs.size() * 4, hipMemcpyHostToDevice, stream));
        const int wantGrids = opts[FVSRN_OPT_WORKING_GRIDS] ? opts[FVSRN_OPT_WORKING_GRIDS] : (K.timeNum > 1 || K.ensNum > 1 ? 2 : 1);
        numWorkGrids = wantGrids;
        curWorkGrid = 0;
        // Cell table (device_params.hpp): 512 bytes per cell and M tile; grids whose table would pass 1 GiB keep the gather path
        cellTableBytes = 0;
        {
            const NetParams& np = packed.params;
            const double cells = double(np.gridX + 1) * double(np.gridY + 1) * double(np.gridZ + 1);  // (the grid extended by one ghost cell per side: cell_tap)
            const int MT = (packed.cfg.hiddenChannels + 31) / 32;
            if (opts[FVSRN_OPT_CELL_TABLE] != 0 && K.enc != FVSRN_GRID_BYTE_GAUSSIAN && np.gridX >= 2 && np.gridY >= 2 && np.gridZ >= 2 &&
                np.numLayers >= 1 && cells * 512.0 * MT <= 1073741824.0)
                cellTableBytes = size_t(cells) * 512 * size_t(MT);
            // (the shaded kernels' table keeps the corner form over the grid's own cells: buildCellTable(plain))
            cellTableBytesCorners = cellTableBytes ? size_t(np.gridX - 1) * size_t(np.gridY - 1) * size_t(np.gridZ - 1) * 512 * size_t(MT) : 0;
        }
        for (int i = 0; i < 2; ++i) {
            if (i >= numWorkGrids) { workGrid[i].a.release(); workGrid[i].b.release(); workGrid[i].cells.release(); workGrid[i].cellsPlain.release(); continue; }
            workGrid[i].a.ensure(K.records * size_t(K.Gt + K.Ge) * 2 * 2);
            if (K.enc == FVSRN_GRID_BYTE_GAUSSIAN) workGrid[i].b.ensure(K.records * size_t(K.Gt + K.Ge) * 2 * 2);
            // (cell tables: allocated by the first launch that uses them, ensureCellTable; a re-pack drops what the old state held)
            workGrid[i].cells.release(); workGrid[i].cellsPlain.release();
            workGrid[i].cellsValid = workGrid[i].cellsPlainValid = false;
        }
        packed.params.grid = workGrid[0].a.ptr;
        packed.params.gridB = K.enc == FVSRN_GRID_BYTE_GAUSSIAN ? workGrid[0].b.ptr : nullptr;
        packed.params.gridEncoding = int(K.enc);
        packed.params.gridTimeChannels = K.Gt;
    } else {
        packed.params.grid = nullptr;
        cellTableBytes = 0;
    }
    packed.params.cellTable = nullptr;  // (set per launch: renderImpl -> ensureCellTable)
    {
        const int MT = (packed.cfg.hiddenChannels + 31) / 32;
        packed.params.cellStride = cellTableBytes ? unsigned(512 * MT) : 0u;
        packed.params.cellCount = cellTableBytes ? unsigned(cellTableBytes / (512 * size_t(MT))) : 0u;
    }
    cellsWanted = cellsPlainWanted = false;
    cellTableBuilds[0] = cellTableBuilds[1] = 0;
    timeDirty = true;
    imagesOrder.endWrite(stream);  // launches on other streams wait for the uploads above (beginUse)
    // the staging vectors are pageable: the copies above complete before hipMemcpyAsync returns
    const size_t maxLds = packed.ldsImage.size() + 4096 + 256 * 6 * 4;
    HIP_CHECK(hipFuncSetAttribute(kinfo.evalFn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    if (kinfoScaled.evalFn != kinfo.evalFn) HIP_CHECK(hipFuncSetAttribute(kinfoScaled.evalFn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    HIP_CHECK(hipFuncSetAttribute(kinfoScaled.renderFn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    HIP_CHECK(hipFuncSetAttribute(kinfo.renderShadedFn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    if (const void* fn = render_stripe_fn(keyScaled)) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    if (const void* fn = render_cells_fn(keyScaled)) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    if (const void* fn = render_shaded_cells_fn(key)) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    if (const void* fn = render_adjoint_fn(key)) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(maxLds)));
    deviceValid = true;
}

void fvsrn_network::buildCellTable(WorkingGrid& W, bool plain, hipStream_t stream) {
    const bool own = plain;  // (r06: the shaded renderer's table is its own since it keeps the corner form)
    DeviceBuffer& buf = own ? W.cellsPlain : W.cells;
    bool& valid = own ? W.cellsPlainValid : W.cellsValid;
    if (valid) return;
    buf.ensure(plain ? cellTableBytesCorners : cellTableBytes);
    const NetParams& np = packed.params;
    const int MT = (packed.cfg.hiddenChannels + 31) / 32, KS = packed.cfg.hiddenChannels / 16;
    CellTableParams ct{};
    ct.grid = W.a.ptr;
    // the unshaded renderer runs the [0,1]-scaled image where the network has it
    ct.latentFrags = static_cast<const char*>(!plain && scaledImage ? scaledImage : dLds.ptr) + np.offLayer0 + size_t(MT) * KS * kFragBytes;
    ct.out = buf.ptr;
    ct.X = np.gridX; ct.Y = np.gridY; ct.Z = np.gridZ; ct.G = np.gridC; ct.MT = MT;
    ct.corners = plain ? 1 : 0;
    HIP_CHECK(launch_grid_cell_table(ct, stream));
    valid = true;
    ++cellTableBuilds[own ? 1 : 0];
}

const void* fvsrn_network::ensureCellTable(bool plain, hipStream_t stream) {
    WorkingGrid& W = workGrid[curWorkGrid];
    const bool own = plain;
    if (!(own ? W.cellsPlainValid : W.cellsValid)) {
        imagesOrder.beginRead(stream);  // the latent fragments of the weight image
        W.order.beginWrite(stream);
        buildCellTable(W, plain, stream);
        W.order.endWrite(stream);
    }
    (plain ? cellsPlainWanted : cellsWanted) = true;
    return own ? W.cellsPlain.ptr : W.cells.ptr;
}

void fvsrn_network::syncTime(hipStream_t stream) {
    if (!timeDirty) return;
    const GridKeyframes& K = packed.keys;
    if (K.records) {
        const GridSelection g = selectGrid(*net);
        BlendParams b{};
        b.ensData = dKeysEns.ptr;
        imagesOrder.beginRead(stream);  // ensemble key frames + coefficients
        const int next = numWorkGrids > 1 ? (curWorkGrid + 1) % numWorkGrids : 0;
        WorkingGrid& W = workGrid[next];
        W.order.beginWrite(stream);  // every kernel that still reads this grid, on whatever stream
        if (keyStore.active()) keyStore.acquire(g.lo, g.hi, g.timeIndex, stream, &b.timeLo, &b.timeHi);
        const float* c = static_cast<const float*>(dCoeffs.ptr);
        b.timeOffset = c; b.timeScale = c + K.timeOffset.size();
        b.ensOffset = c + 2 * K.timeOffset.size(); b.ensScale = b.ensOffset + K.ensOffset.size();
        b.out = W.a.ptr; b.outB = W.b.ptr; b.records = K.records; b.enc = int(K.enc); b.Gt = K.Gt; b.Ge = K.Ge;
        b.lo = g.lo; b.hi = g.hi; b.ens = g.ens; b.frac = g.frac;
        HIP_CHECK(launch_grid_blend(b, stream));
        W.cellsValid = W.cellsPlainValid = false;
        // the tables the launches before this blend went through are rebuilt with it (same stream, same write bracket: fvsrn_network_prepare puts
        // both on its side stream); any other is built by the launch that first wants it
        if (cellTableBytes && cellsWanted) buildCellTable(W, false, stream);
        if (cellTableBytes && cellsPlainWanted) buildCellTable(W, true, stream);
        W.order.endWrite(stream);
        curWorkGrid = next;
        packed.params.grid = W.a.ptr;
        packed.params.gridB = K.enc == FVSRN_GRID_BYTE_GAUSSIAN ? W.b.ptr : nullptr;
        if (keyStore.active()) keyStore.released(g.lo, g.hi, stream);
        // decode coefficients of the selected key frames (BYTE_GAUSSIAN decodes inside the render kernel)
        packed.params.gridFrac = g.frac;
        packed.params.gridMeanTime = b.timeOffset + size_t(g.lo) * K.Gt;
        packed.params.gridStdTime = b.timeScale + size_t(g.lo) * K.Gt;
        packed.params.gridMeanEns = b.ensOffset + size_t(g.ens) * K.Ge;
        packed.params.gridStdEns = b.ensScale + size_t(g.ens) * K.Ge;
        // Networks that take the time as an input: the fp16 time entry of the phase fragment is a KERNEL ARGUMENT (every
        // kernel patches its LDS copy of the image, load_network_to_lds) -- no write to the shared device images, so frames
        // at different times can be in flight at once and every image (plain, scaled, curvature) sees the same time.
        packed.params.timeSlotOffset = packed.timeSlotOffset;
        packed.params.timeSlotBits = packed.timeSlotOffset >= 0 ? float_to_half_bits(g.timeIndex) : 0;
    }
    timeDirty = false;
}

void fvsrn_network::releaseDevice() {
    dLds.release();
    dLdsScaled.release();
    dLdsCurvature.release();
    for (WorkingGrid& w : workGrid) { w.a.release(); w.b.release(); w.cells.release(); w.cellsPlain.release(); w.cellsValid = w.cellsPlainValid = false; w.order.release(); }
    imagesOrder.release();
    imageReaders.clear();
    keyStore.release();
    dKeysEns.release();
    dCoeffs.release();
    scaledImage = curvatureImage = nullptr;
}
