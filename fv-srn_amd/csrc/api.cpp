// C ABI of libfvsrn.so (see include/fvsrn.h for the reference interfaces each entry replaces).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>

#include "../../include/fvsrn.h"
#include "half.hpp"
#include "launch.hpp"
#include "launch_host.hpp"
#include "srn_device_enums.hpp"
#include "pack.hpp"
#include "scene_network.hpp"
#include "grid_volume.hpp"
#include <fstream>
#include <vector>

using namespace fvsrn;

// Tuning / developer switches of a handle (include/fvsrn.h, fvsrn_option).  Every new handle starts from the process defaults,
// which are read from the environment ONCE (FVSRN_SMALL_KERNEL, FVSRN_PERSISTENT, FVSRN_SEGMENTS, FVSRN_FOURIER_RESYNC,
// FVSRN_UNIT_QUOTA, FVSRN_TILE_ORDER, FVSRN_WAVES_PER_BLOCK, FVSRN_MAX_BLOCKS_PER_CU, FVSRN_DISABLE_RELU_CLAMP); nothing on the
// per-frame path calls getenv.
struct Options {
    int v[FVSRN_OPT_COUNT_];
    Options() {
        v[FVSRN_OPT_SMALL_KERNEL] = -1; v[FVSRN_OPT_PERSISTENT] = -1; v[FVSRN_OPT_DEPTH_SEGMENTS] = 0; v[FVSRN_OPT_FOURIER_RESYNC] = 0;
        v[FVSRN_OPT_UNIT_QUOTA] = -1; v[FVSRN_OPT_TILE_ORDER] = -1; v[FVSRN_OPT_WAVES_PER_BLOCK] = 0; v[FVSRN_OPT_MAX_BLOCKS_PER_CU] = 0;
        v[FVSRN_OPT_RELU_CLAMP] = 1; v[FVSRN_OPT_KEYFRAME_SLOTS] = 0; v[FVSRN_OPT_WORKING_GRIDS] = 0; v[FVSRN_OPT_OVERLAP_KERNEL] = -1;
        v[FVSRN_OPT_PERSISTENT_RESERVE] = -1; v[FVSRN_OPT_CELL_TABLE] = -1;
    }
    int operator[](int i) const { return v[i]; }
    // empty string = valid
    static std::string check(int opt, int value) {
        switch (opt) {
            case FVSRN_OPT_SMALL_KERNEL: case FVSRN_OPT_PERSISTENT: case FVSRN_OPT_TILE_ORDER: case FVSRN_OPT_OVERLAP_KERNEL: case FVSRN_OPT_CELL_TABLE:
                return value >= -1 && value <= 1 ? "" : "value must be -1 (automatic), 0 or 1";
            case FVSRN_OPT_DEPTH_SEGMENTS: return value >= 0 && value <= 64 ? "" : "segments must be 0 (automatic) .. 64";
            case FVSRN_OPT_FOURIER_RESYNC:
                return value == 0 || (value >= 1 && value <= 4096 && (value & (value - 1)) == 0) ? "" : "resync period must be 0 (default) or a power of two <= 4096";
            case FVSRN_OPT_UNIT_QUOTA: return value >= -1 && value <= 1024 ? "" : "unit quota must be -1 (automatic) .. 1024";
            case FVSRN_OPT_WAVES_PER_BLOCK: return value == 0 || value == 1 || value == 2 || value == 4 ? "" : "waves per workgroup must be 0 (automatic), 1, 2 or 4";
            case FVSRN_OPT_MAX_BLOCKS_PER_CU: return value >= 0 && value <= 32 ? "" : "workgroups per CU must be 0 (no limit) .. 32";
            case FVSRN_OPT_RELU_CLAMP: return value == 0 || value == 1 ? "" : "value must be 0 or 1";
            case FVSRN_OPT_KEYFRAME_SLOTS: return value == 0 || (value >= 2 && value <= 65536) ? "" : "key-frame slots must be 0 (all resident) or >= 2";
            case FVSRN_OPT_WORKING_GRIDS: return value >= 0 && value <= 2 ? "" : "working grids must be 0 (automatic), 1 or 2";
            case FVSRN_OPT_PERSISTENT_RESERVE: return value >= -1 && value <= 4096 ? "" : "reserved workgroup slots must be -1 (automatic) .. 4096";
            default: return "unknown option";
        }
    }
};

static const Options& defaultOptions() {
    static const Options defaults = [] {
        Options o;
        static const struct { const char* name; int opt; } kEnv[] = {
            {"FVSRN_SMALL_KERNEL", FVSRN_OPT_SMALL_KERNEL}, {"FVSRN_PERSISTENT", FVSRN_OPT_PERSISTENT}, {"FVSRN_SEGMENTS", FVSRN_OPT_DEPTH_SEGMENTS},
            {"FVSRN_FOURIER_RESYNC", FVSRN_OPT_FOURIER_RESYNC}, {"FVSRN_UNIT_QUOTA", FVSRN_OPT_UNIT_QUOTA}, {"FVSRN_TILE_ORDER", FVSRN_OPT_TILE_ORDER},
            {"FVSRN_WAVES_PER_BLOCK", FVSRN_OPT_WAVES_PER_BLOCK}, {"FVSRN_MAX_BLOCKS_PER_CU", FVSRN_OPT_MAX_BLOCKS_PER_CU},
            {"FVSRN_KEYFRAME_SLOTS", FVSRN_OPT_KEYFRAME_SLOTS}, {"FVSRN_WORKING_GRIDS", FVSRN_OPT_WORKING_GRIDS}, {"FVSRN_OVERLAP_KERNEL", FVSRN_OPT_OVERLAP_KERNEL},
            {"FVSRN_PERSISTENT_RESERVE", FVSRN_OPT_PERSISTENT_RESERVE}, {"FVSRN_CELL_TABLE", FVSRN_OPT_CELL_TABLE}};
        for (const auto& e : kEnv)
            if (const char* t = std::getenv(e.name)) {
                const int val = std::atoi(t);
                if (Options::check(e.opt, val).empty()) o.v[e.opt] = val;
            }
        if (std::getenv("FVSRN_DISABLE_RELU_CLAMP")) o.v[FVSRN_OPT_RELU_CLAMP] = 0;
        return o;
    }();
    return defaults;
}

// Waves per workgroup: as few as the LDS budget allows.  16 waves per CU (4 per SIMD) must fit their network copies
// into the 160 KiB of LDS; a workgroup's slot is only recycled when its slowest wave is done, so fewer waves per
// workgroup = better balance between long and empty pixel tiles (measured r01: 1 wave 110.8, 4 waves 92.3 Gsamples/s).
#ifndef FVSRN_IDENTITY_TAIL
#define FVSRN_IDENTITY_TAIL 4  // TAIL_SCALAR_IDENTITY (1 = the Identity TF through TAIL_SCALAR_TABLE: A/B builds)
#endif
static int wavesPerBlockFor(size_t ldsBytesPerBlock, const Options& o) {
    if (o[FVSRN_OPT_WAVES_PER_BLOCK]) return o[FVSRN_OPT_WAVES_PER_BLOCK];
    const size_t budget = 160 * 1024;
    for (int w : {1, 2, 4})
        if (size_t(16 / w) * ldsBytesPerBlock <= budget) return w;
    return 4;
}

namespace {
thread_local std::string g_lastError;

int fail(int code, const std::string& msg) {
    g_lastError = msg;
    return code;
}

struct WrongDeviceBase : std::runtime_error { using std::runtime_error::runtime_error; };

template <class F>
int guarded(F&& f) {
    try {
        g_lastError.clear();
        return f();
    } catch (const FormatError& e) {
        return fail(FVSRN_ERR_FORMAT, e.what());
    } catch (const InvalidNetwork& e) {
        return fail(FVSRN_ERR_INVALID_NETWORK, e.what());
    } catch (const Unsupported& e) {
        return fail(FVSRN_ERR_UNSUPPORTED, e.what());
    } catch (const WrongDeviceBase& e) {
        return fail(FVSRN_ERR_WRONG_DEVICE, e.what());
    } catch (const std::bad_alloc&) {
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "out of host memory");
    } catch (const std::exception& e) {
        return fail(FVSRN_ERR_INVALID_ARGUMENT, e.what());
    }
}

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw DeviceError(std::string(#expr) + " failed: " + hipGetErrorString(_e));                  \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        HIP_CHECK(hipMalloc(&ptr, bytes));
        cap = bytes;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

// Cross-stream ordering of one device resource that is written rarely and read by kernels on any stream (a working grid, a
// key-frame slot, the weight images): the writer records an event, every reader on ANOTHER stream waits for it; every reader
// records an event of its own stream, and the next writer waits for all of them.  One event per (resource, stream): a reader
// on a second stream does not overwrite the first stream's mark.  Waiting on an event that has completed costs a
// microsecond of host time and nothing on the device.
struct StreamOrder {
    struct Reader { hipStream_t stream; hipEvent_t done; bool pending; };
    std::vector<Reader> readers;
    hipEvent_t written = nullptr;
    hipStream_t writer = nullptr;
    bool haveWrite = false;
    void beginWrite(hipStream_t s) {
        for (Reader& r : readers)
            if (r.pending && r.stream != s) HIP_CHECK(hipStreamWaitEvent(s, r.done, 0));
        if (haveWrite && writer != s) HIP_CHECK(hipStreamWaitEvent(s, written, 0));
    }
    void endWrite(hipStream_t s) {
        if (!written) HIP_CHECK(hipEventCreateWithFlags(&written, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(written, s));
        writer = s;
        haveWrite = true;
        // readers of the old content on the writer's own stream are ordered by the stream; the others were waited for
        for (Reader& r : readers) r.pending = false;
    }
    void beginRead(hipStream_t s) {
        if (haveWrite && writer != s) HIP_CHECK(hipStreamWaitEvent(s, written, 0));
    }
    void endRead(hipStream_t s) {
        for (Reader& r : readers)
            if (r.stream == s) {
                HIP_CHECK(hipEventRecord(r.done, s));
                r.pending = true;
                return;
            }
        // a stream handle seen for the first time: drop the entries of streams whose last read has completed (a caller that creates a
        // stream per frame would otherwise grow this list by one event per stream)
        if (readers.size() >= 8) {
            size_t keep = 0;
            for (Reader& r : readers) {
                if (r.pending && hipEventQuery(r.done) == hipErrorNotReady) readers[keep++] = r;
                else (void)hipEventDestroy(r.done);
            }
            readers.resize(keep);
        }
        Reader r{s, nullptr, true};
        HIP_CHECK(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(r.done, s));
        readers.push_back(r);
    }
    void release() {
        for (Reader& r : readers) (void)hipEventDestroy(r.done);
        readers.clear();
        if (written) (void)hipEventDestroy(written);
        written = nullptr;
        haveWrite = false;
    }
};

// Per-call temporaries of evaluate_points (the list of deferred batches, the two passes of the curvature evaluation) are allocated
// and freed in stream order from a pool of the library's own (one per device, created on first use): the device's default pool
// hands memory back to the driver at every synchronisation, which turned a 134 MB temporary into 1.5 ms of host time per call;
// this one keeps up to 512 MiB cached between calls.
struct TemporaryPools {
    std::mutex mu;
    hipMemPool_t pool[16] = {};
    hipMemPool_t get(int device) {
        if (device < 0 || device >= 16) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        if (!pool[device]) {
            hipMemPoolProps props{};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = device;
            HIP_CHECK(hipMemPoolCreate(&pool[device], &props));
            uint64_t keep = uint64_t(512) << 20;
            HIP_CHECK(hipMemPoolSetAttribute(pool[device], hipMemPoolAttrReleaseThreshold, &keep));
        }
        return pool[device];
    }
    void* alloc(int device, size_t bytes, hipStream_t s) {
        void* p = nullptr;
        hipMemPool_t mp = get(device);
        if (mp) HIP_CHECK(hipMallocFromPoolAsync(&p, bytes, mp, s));
        else HIP_CHECK(hipMallocAsync(&p, bytes, s));
        return p;
    }
};
TemporaryPools g_temporaries;

struct WrongDevice : WrongDeviceBase { using WrongDeviceBase::WrongDeviceBase; };
// A handle's device state lives on the device that was current at its first use; every later call must run there.
void bindOrCheckDevice(int& bound, const char* what) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (bound < 0) bound = dev;
    else if (bound != dev)
        throw WrongDevice(std::string(what) + " holds resources on HIP device " + std::to_string(bound) + ", but the current device is " +
                          std::to_string(dev) + " (hipSetDevice before the call, or use one handle per device)");
}

int actIndex(fvsrn_activation a) {
    switch (a) {
        case FVSRN_ACT_RELU: return 0;
        case FVSRN_ACT_SINE: return 1;
        case FVSRN_ACT_SNAKE: return 2;
        case FVSRN_ACT_SNAKEALT: return 3;
        case FVSRN_ACT_SIGMOID: return 5;  // ACT_SIGMOID (4 is the scaled-ReLU image)
        default: return -1;
    }
}
}  // namespace

// Time key frames of a latent grid on the device (BASELINE.json configs[4]; reference: LatentGrid textures uploaded lazily by a
// synchronous cudaMemcpy3D at first use and kept forever, volume_interpolation_network.cpp:482-488,524-535,1308-1315).
// Here every key frame sits in PINNED host memory in device layout and `slots` of them are resident in HBM (all of them by
// default, FVSRN_OPT_KEYFRAME_SLOTS bounds it; >= 2).  Uploads run on a copy stream of the store: a slot is overwritten once every
// blend kernel that read it is done (one event per slot and reading stream, StreamOrder), the blend of a frame waits for the
// uploads it needs (event) -- and for nothing else, so the copy of frame i+1's key frame overlaps the render of frame i (the render kernel reads the blended working grid,
// not the key frames).  With >= 3 slots the key frame the time is moving towards is prefetched one interval ahead.
struct KeyframeStore {
    char* pinned = nullptr;      // [numKeys][bytesPerKey]
    size_t bytesPerKey = 0;
    int numKeys = 0, slots = 0;
    DeviceBuffer dSlots;         // [slots][bytesPerKey]
    std::vector<int> keyOfSlot, slotOfKey;
    std::vector<unsigned long long> lastUse;
    std::vector<StreamOrder> order;  // per slot: upload (copy stream) <-> blend kernels (any stream)
    hipStream_t copyStream = nullptr;
    unsigned long long tick = 0;
    float lastTime = -1.f;
    unsigned long long stats[4] = {0, 0, 0, 0};  // uploads, of which on demand (a blend waited for them), prefetched, bytes

    bool active() const { return numKeys > 0; }
    void release() {
        for (StreamOrder& o : order) o.release();
        order.clear();
        if (copyStream) { (void)hipStreamSynchronize(copyStream); (void)hipStreamDestroy(copyStream); copyStream = nullptr; }
        if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; }
        dSlots.release();
        numKeys = slots = 0;
    }
    // host data of all key frames -> pinned memory; `budget` = 0 (all resident) or the number of device slots
    void init(const std::vector<char>& data, int keys, int budget) {
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
    const char* slotPtr(int slot) const { return static_cast<const char*>(dSlots.ptr) + size_t(slot) * bytesPerKey; }
    void upload(int key, int slot, bool prefetch) {
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
    int victim(int keepA, int keepB) const {  // least recently used slot that holds neither key
        int best = -1;
        for (int i = 0; i < slots; ++i) {
            const int k = keyOfSlot[size_t(i)];
            if (k >= 0 && (k == keepA || k == keepB)) continue;
            if (k < 0) return i;
            if (best < 0 || lastUse[size_t(i)] < lastUse[size_t(best)]) best = i;
        }
        return best;
    }
    // device pointers of key frames lo / hi for a blend enqueued on `stream` (which is made to wait for their uploads)
    void acquire(int lo, int hi, float time, hipStream_t stream, const void** pLo, const void** pHi) {
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
    // call after the blend kernel has been enqueued on `stream`
    void released(int lo, int hi, hipStream_t stream) {
        for (int key : {lo, hi}) {
            order[size_t(slotOfKey[size_t(key)])].endRead(stream);
            if (hi == lo) break;
        }
    }
};

struct fvsrn_network {
    std::shared_ptr<SceneNetwork> net = std::make_shared<SceneNetwork>();
    // device image (lazy; invalidated by any mutation)
    bool deviceValid = false;
    PackedNetwork packed;
    DeviceBuffer dLds, dLdsScaled, dLdsCurvature, dKeysEns, dCoeffs;
    // Working grids: the fp16 x-pair records the kernels read, blended from the key frames when the time / ensemble changes
    // (grid_blend_kernel).  Two of them for networks with more than one key frame: the blend of frame i + 1 writes the grid that
    // frame i does NOT read, so a caller may keep two frames in flight on two streams (tiles.StripeRenderer, BASELINE.json
    // configs[4]); `order` makes a blend wait for every kernel that still reads the grid it overwrites, on whatever stream.
    // cells: the grid's cell table (NetParams::cellTable) for the image the unshaded renderer runs; cellsPlain: the one of the plain
    // image for the shaded renderer, where the network has a re-scaled image as well (otherwise the two are one)
    // Tables are built LAZILY (r05): allocated and filled by the first launch that runs a cell-table kernel on this working grid
    // (ensureCellTable), and rebuilt together with a blend only while the previous launches used them (cellsWanted / cellsPlainWanted) -- a time-animated
    // 64^3 .. 128^3 grid whose frames take the gathers (footprint rule, adjoint mode) no longer writes 0.25 .. 2 GB of table per frame nor holds up to
    // 4 GiB of HBM for a path it never takes, and nothing builds the plain-image table unless something renders shaded.
    struct WorkingGrid { DeviceBuffer a, b, cells, cellsPlain; bool cellsValid = false, cellsPlainValid = false; StreamOrder order; };
    bool cellsWanted = false, cellsPlainWanted = false;  // the last unshaded / shaded launch went through the table
    unsigned long long cellTableBuilds[2] = {0, 0};       // table builds since the device state was created: unshaded-image table, plain-image table
    WorkingGrid workGrid[2];
    int numWorkGrids = 1, curWorkGrid = 0;
    size_t cellTableBytes = 0;  // 0: no cell table (no grid, BYTE_GAUSSIAN, a resolution below 2, above the size cap, FVSRN_OPT_CELL_TABLE = 0)
    StreamOrder imagesOrder;  // weight images, ensemble key frames, decode coefficients: written at first use
    // streams that have launched kernels reading the images (handles only: recording an event behind every launch cost 3 % of a 0.3 ms
    // frame, r03): a re-pack of a live network waits for THESE streams, not for the device (hipDeviceSynchronize stalled the
    // collective's and every other pipeline's streams, and is illegal during stream capture)
    std::vector<hipStream_t> imageReaders;
    KeyframeStore keyStore;  // time key frames
    const void* scaledImage = nullptr;
    const void* curvatureImage = nullptr;
    bool timeDirty = true;  // working grid / time slot do not match net->currentTime yet
    VariantKey key{};       // plain image
    VariantKey keyScaled{};  // ReLU networks: [0,1]-scaled image (render only)
    KernelInfo kinfo{}, kinfoScaled{};
    int numCUs = 0;
    int device = -1;  // HIP device of the buffers above (-1: none yet)
    Options opts = defaultOptions();
    mutable std::mutex mu;
    // fvsrn_network_get_info is called per frame by bindings (output channels, FLOP counts): computed once per network state
    mutable bool infoValid = false;
    mutable fvsrn_network_info info{};

    void invalidate() { deviceValid = false; occKey = 0; infoValid = false; }

    // resident workgroups per CU of the render kernel for (blockDim, dynamic LDS); cached
    unsigned long long occKey = 0;
    const void* occFn = nullptr;
    int occBlocks = 0;
    // smallFn: render_small_kernel variant to use instead of render_kernel (nullptr = none)
    int renderBlocksPerCU(unsigned blockDim, size_t ldsBytes, bool shaded, const void* smallFn, int maxBlocks) {
        const unsigned long long k = (static_cast<unsigned long long>(ldsBytes) << 24) | (static_cast<unsigned long long>(maxBlocks) << 18) | (blockDim << 2) |
                                     (smallFn ? 2u : 0u) | (shaded ? 1u : 0u);
        if (smallFn != occFn) occKey = 0;  // another render_small_kernel variant
        occFn = smallFn;
        if (k != occKey) {
            int n = 0;
            const void* fn = smallFn ? smallFn : (shaded ? kinfo.renderShadedFn : kinfoScaled.renderFn);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, int(blockDim), ldsBytes) != hipSuccess) n = 0;
            if (maxBlocks >= 1 && maxBlocks < n) n = maxBlocks;  // FVSRN_OPT_MAX_BLOCKS_PER_CU: occupancy experiments
            occBlocks = n;
            occKey = k;
        }
        return occBlocks;
    }

    void pack() {  // host part only (no GPU needed): variant selection + LDS image
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

    void ensureDevice(hipStream_t stream) {
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
                const double cells = double(np.gridX - 1) * double(np.gridY - 1) * double(np.gridZ - 1);
                const int MT = (packed.cfg.hiddenChannels + 31) / 32;
                if (opts[FVSRN_OPT_CELL_TABLE] != 0 && K.enc != FVSRN_GRID_BYTE_GAUSSIAN && np.gridX >= 2 && np.gridY >= 2 && np.gridZ >= 2 &&
                    np.numLayers >= 1 && cells * 512.0 * MT <= 1073741824.0)
                    cellTableBytes = size_t(cells) * 512 * size_t(MT);
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

    // Fills the cell table of working grid W from its blended records (grid_cell_table_kernel); the caller holds W's write bracket.  plain: the table
    // of the plain weight image (the shaded renderer's) -- the same buffer as the unshaded one where the network has no re-scaled image.
    void buildCellTable(WorkingGrid& W, bool plain, hipStream_t stream) {
        const bool own = plain && scaledImage != nullptr;
        DeviceBuffer& buf = own ? W.cellsPlain : W.cells;
        bool& valid = own ? W.cellsPlainValid : W.cellsValid;
        if (valid) return;
        buf.ensure(cellTableBytes);
        const NetParams& np = packed.params;
        const int MT = (packed.cfg.hiddenChannels + 31) / 32, KS = packed.cfg.hiddenChannels / 16;
        CellTableParams ct{};
        ct.grid = W.a.ptr;
        // the unshaded renderer runs the [0,1]-scaled image where the network has it
        ct.latentFrags = static_cast<const char*>(!plain && scaledImage ? scaledImage : dLds.ptr) + np.offLayer0 + size_t(MT) * KS * kFragBytes;
        ct.out = buf.ptr;
        ct.X = np.gridX; ct.Y = np.gridY; ct.Z = np.gridZ; ct.G = np.gridC; ct.MT = MT;
        HIP_CHECK(launch_grid_cell_table(ct, stream));
        valid = true;
        ++cellTableBuilds[own ? 1 : 0];
    }
    // The table of the CURRENT working grid for a launch on `stream` (after syncTime, before beginUse): built now if no launch has needed it since the
    // last blend.  Readers of the grid on other streams are waited for like by a blend, later readers wait for this write.
    const void* ensureCellTable(bool plain, hipStream_t stream) {
        WorkingGrid& W = workGrid[curWorkGrid];
        const bool own = plain && scaledImage != nullptr;
        if (!(own ? W.cellsPlainValid : W.cellsValid)) {
            imagesOrder.beginRead(stream);  // the latent fragments of the weight image
            W.order.beginWrite(stream);
            buildCellTable(W, plain, stream);
            W.order.endWrite(stream);
        }
        (plain ? cellsPlainWanted : cellsWanted) = true;
        return own ? W.cellsPlain.ptr : W.cells.ptr;
    }

    // Brings the working grid and the time input of the network in line with net->currentTime/currentEnsemble:
    // one small kernel + (networks that take the time as input) a 2-byte patch, both stream-ordered -- no host
    // synchronisation, no re-upload (the reference re-fills its constant block and lazily uploads textures with a
    // synchronous cudaMemcpy3D, volume_interpolation_network.cpp:482-488,923-938,1308-1315).
    void syncTime(hipStream_t stream) {
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

    // Brackets of every kernel launch that reads the network's device state on `stream`
    void beginUse(hipStream_t stream) {
        imagesOrder.beginRead(stream);
        if (std::find(imageReaders.begin(), imageReaders.end(), stream) == imageReaders.end()) {
            if (imageReaders.size() < 64) imageReaders.push_back(stream);  // (bounded: a full list makes the next re-pack wait for the device)
        }
        if (packed.keys.records) workGrid[curWorkGrid].order.beginRead(stream);
    }
    void endUse(hipStream_t stream) {
        if (packed.keys.records) workGrid[curWorkGrid].order.endRead(stream);
    }

    void releaseDevice() {
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
    ~fvsrn_network() { releaseDevice(); }
};

struct BoxCenter { float c[3]; };
static BoxCenter P_boxCenter(const NetParams& P) {
    return {{P.boxMin[0] + 0.5f * P.boxSize[0], P.boxMin[1] + 0.5f * P.boxSize[1], P.boxMin[2] + 0.5f * P.boxSize[2]}};
}

// Live scene handles, for fvsrn_debug_state (a watchdog thread asks what the library last launched when a caller hangs).  Leaked on purpose: handles
// may be destroyed during static destruction.
struct SceneRegistry { std::mutex mu; std::vector<fvsrn_scene*> live; };
static SceneRegistry& sceneRegistry() { static SceneRegistry* r = new SceneRegistry; return *r; }

struct fvsrn_scene {
    fvsrn_scene() { SceneRegistry& r = sceneRegistry(); std::lock_guard<std::mutex> l(r.mu); r.live.push_back(this); }
    fvsrn_scene(const fvsrn_scene&) = delete;
    fvsrn_scene_desc desc{};
    std::vector<float> tfTable;
    DeviceBuffer dTf, dOrder, dCounters, dPartial, dPreint;
    bool tfOpacityNonNegative = true;  // Texture TF: no negative opacity in the table (fvsrn_scene_update)
    int preintMode = 0;          // what dPreint holds
    float preintStepsize = -1.f;
    unsigned launches = 0;  // parity selects which of the two tile counters a launch uses (the kernel zeroes the other)
    int lastInfo[4] = {0, 0, 0, 0};  // fvsrn_scene_last_render_info
    std::string lastKernel;          // fvsrn_scene_last_kernel_name: the kernel the last render launched
    bool tfDirty = true;
    int device = -1;  // HIP device of the buffers above (-1: none yet)
    Options opts = defaultOptions();
    std::mutex mu;
    // cached launch order of the 8x8 pixel tiles
    struct OrderKey { int tilesX = -1, tilesY = -1, cx = 0, cy = 0, y0 = 0, stripeRows = 0, stripeRank = 0, stripeWorld = 0; } orderKey;
    std::vector<int> order;
    ~fvsrn_scene() {
        { SceneRegistry& r = sceneRegistry(); std::lock_guard<std::mutex> l(r.mu); r.live.erase(std::remove(r.live.begin(), r.live.end(), this), r.live.end()); }
        dTf.release(); dOrder.release(); dCounters.release(); dPartial.release(); dPreint.release();
    }
    // what the last launch of this scene was, for fvsrn_debug_state: written under `mu` by renderImpl
    struct LastLaunch { unsigned grid = 0, block = 0; long long units = 0; int width = 0, height = 0, rows = 0, stripeWorld = 1, persistent = 0, frames = 1; void* stream = nullptr; unsigned long long count = 0; } lastLaunch;

    // TF table (and, for pre-integrated Texture TFs, its tables) on the device, for step size `stepsize`
    int uploadTf(float stepsize, hipStream_t s) {
        bindOrCheckDevice(device, "the scene");
        const fvsrn_scene_desc& d = desc;
        const size_t tfFloats = tfTable.size();
        const bool tfChanged = tfDirty;
        try {
            if (tfDirty) {
                if (tfFloats) {
                    dTf.ensure(tfFloats * 4);
                    HIP_CHECK(hipMemcpyAsync(dTf.ptr, tfTable.data(), tfFloats * 4, hipMemcpyHostToDevice, s));
                }
                tfDirty = false;
            }
            // TransferFunctionTexture::updatePreintegrationTable (transfer_function_texture.cpp:364-379): rebuilt when the
            // texture or (2D) the step size changed
            if (d.tf_preintegration != FVSRN_PREINTEGRATE_NONE &&
                (tfChanged || preintMode != d.tf_preintegration || (d.tf_preintegration == FVSRN_PREINTEGRATE_2D && preintStepsize != stepsize))) {
                const int R = d.tf_rows;
                dPreint.ensure(size_t(d.tf_preintegration == FVSRN_PREINTEGRATE_2D ? R : 1) * R * 4 * sizeof(float));
                const hipError_t e = launch_tf_preintegration(static_cast<const float*>(dTf.ptr), static_cast<float*>(dPreint.ptr), R,
                                                              d.tf_preintegration, stepsize, 256, s);
                if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("pre-integration failed: ") + hipGetErrorString(e));
                preintMode = d.tf_preintegration;
                preintStepsize = stepsize;
            }
        } catch (const DeviceError& e) {
            return fail(FVSRN_ERR_DEVICE, e.what());
        }
        return FVSRN_OK;
    }

    // Two work counters for the persistent render waves.  Launches of one scene must be ordered on one stream
    // (like everything else a scene owns: TF table, tile order).
    bool tileCounters(hipStream_t stream, int** cur, int** next) {
        if (!dCounters.ptr) {
            dCounters.ensure(2 * sizeof(int));
            if (hipMemsetAsync(dCounters.ptr, 0, 2 * sizeof(int), stream) != hipSuccess) return false;
        }
        int* c = static_cast<int*>(dCounters.ptr);
        *cur = c + (launches & 1u);
        *next = c + ((launches + 1u) & 1u);
        ++launches;
        return true;
    }

    // Tiles sorted by distance from the projection of the box centre: rays through the middle of the box are the
    // longest, rays that miss it cost one iteration.  The hardware dispatches workgroups in index order, so the
    // expensive tiles start first and the cheap ones fill the tail.  Pure scheduling: any order gives the same image.
    const int* tileOrder(const SceneParams& S, const BoxCenter& bc, int tilesX, int tilesY, hipStream_t stream) {
        const float v[3] = {bc.c[0] - S.eye[0], bc.c[1] - S.eye[1], bc.c[2] - S.eye[2]};
        const float zf = v[0] * S.front[0] + v[1] * S.front[1] + v[2] * S.front[2];
        float px = 0.5f * S.width, py = 0.5f * S.height;
        if (zf > 1e-6f) {
            const float xr = v[0] * S.right[0] + v[1] * S.right[1] + v[2] * S.right[2];
            const float yu = v[0] * S.up[0] + v[1] * S.up[1] + v[2] * S.up[2];
            px = (xr / (zf * S.tanFovX) + 1.f) * 0.5f * S.width;
            py = (yu / (zf * S.tanFovY) + 1.f) * 0.5f * S.height;
        }
        OrderKey k;
        k.tilesX = tilesX; k.tilesY = tilesY;
        k.cx = int(std::floor(px / 8.f)); k.cy = int(std::floor(py / 8.f));
        k.y0 = S.y0; k.stripeRows = S.stripeRows; k.stripeRank = S.stripeRank; k.stripeWorld = S.stripeWorld;
        if (std::memcmp(&k, &orderKey, sizeof(k)) != 0 || order.empty()) {
            const int n = tilesX * tilesY;
            std::vector<std::pair<float, int>> keyed(static_cast<size_t>(n));
            for (int ty = 0; ty < tilesY; ++ty) {
                const int l = ty * 8;  // first local row of the tile -> image row (same mapping as the kernel)
                const int y = S.y0 + ((l / S.stripeRows) * S.stripeWorld + S.stripeRank) * S.stripeRows + l % S.stripeRows;
                const float dy = (float(y) + 4.f) - py;
                for (int tx = 0; tx < tilesX; ++tx) {
                    const float dx = (float(tx * 8) + 4.f) - px;
                    keyed[size_t(ty * tilesX + tx)] = {dx * dx + dy * dy, ty * tilesX + tx};
                }
            }
            std::sort(keyed.begin(), keyed.end());
            order.resize(size_t(n));
            for (int i = 0; i < n; ++i) order[size_t(i)] = keyed[size_t(i)].second;
            dOrder.ensure(size_t(n) * sizeof(int));
            if (hipMemcpyAsync(dOrder.ptr, order.data(), size_t(n) * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess)
                return nullptr;
            orderKey = k;
        }
        return static_cast<const int*>(dOrder.ptr);
    }
};

static int tfCols(int kind) {
    switch (kind) {
        case FVSRN_TF_GAUSSIAN: return 6;
        case FVSRN_TF_PIECEWISE: return 5;
        case FVSRN_TF_TEXTURE: return 4;
        default: return 0;
    }
}

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
            if (L.count) out += std::string(", stream ") + (hipStreamQuery(static_cast<hipStream_t>(L.stream)) == hipSuccess ? "idle" : "BUSY");
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

int fvsrn_network_kernel_name(fvsrn_network* net, int render, char* buf, size_t cap) {
    return guarded([&] {
        if (!net || !buf || cap == 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> lock(net->mu);
        if (!net->deviceValid) net->pack();  // (a live network keeps its packed state: it carries the device pointers)
        std::string name = render ? net->kinfoScaled.renderName : net->kinfoScaled.evalName;  // (evaluate_points runs the re-scaled image too, r03)
        if (render) {  // the register-resident kernel takes over for scenes with an Identity / Texture TF and no shading (renderImpl)
            const NetParams& P = net->packed.params;
            const VariantKey& k = net->keyScaled;
            const bool scalarNet = P.outputMode == FVSRN_OUT_DENSITY || P.outputMode == FVSRN_OUT_DENSITY_DIRECT;
            const bool colourNet = P.outputMode == FVSRN_OUT_RGBO || P.outputMode == FVSRN_OUT_RGBO_DIRECT;
            const bool folded = P.bias0Folded && (!net->scaledImage || net->packed.scaledBias0Exact);
            const bool cells = k.grid == 1 && net->opts[FVSRN_OPT_CELL_TABLE] != 0 && P.gridX >= 2 && P.gridY >= 2 && P.gridZ >= 2 &&
                               double(P.gridX - 1) * (P.gridY - 1) * (P.gridZ - 1) * 512.0 * ((net->packed.cfg.hiddenChannels + 31) / 32) <= 1073741824.0;  // (ensureDevice: cellTableBytes)
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
                        bool adjoint, float adjointGridStep, void* stream) {
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
            // batch of 64 points and evaluates a batch with a point outside from the plain image in global memory (kernels.hpp).
            VariantKey evalKey = net->key;
            a.P.evalTodo = nullptr;
            void* evalTodo = nullptr;
            struct FreeTodo { void*& p; hipStream_t s; ~FreeTodo() { if (p) (void)hipFreeAsync(p, s); } } freeTodo{evalTodo, s};
            // The [0,1]-scaled ReLU image costs a second launch and a per-call list (below): ~9 us.  Measured r04 (32x4, best of interleaved
            // repetitions, tools/dev/eval_knobs.py): 2^20 points 48 G points/s with it against 83 G on the plain image, 2^22: 93 against 107,
            // 2^24: 119 against 116, 2^26: 117 against 110 -- the plain image below 2^23 points.
            const bool scaledPays = net->keyScaled.act != ACT_RELU01 || n >= (size_t(1) << 23);
            if (net->scaledImage && !adjoint && scaledPays) {
                evalKey = net->keyScaled;
                a.P.ldsImage = net->scaledImage;
                if (!net->packed.scaledBias0Exact) a.P.bias0Folded = 0;  // (a residue of the folded bias sits in the fp32 block: pack.cpp)
                a.P.reluClamp = net->keyScaled.act == ACT_RELU01 ? 1 : 0;
                if (net->keyScaled.act == ACT_RELU01) {  // two launches, see kernels.hpp (eval_batch_deferred); stream-ordered scratch per call
                    evalTodo = g_temporaries.alloc(net->device, (1 + (n + 63) / 64) * sizeof(unsigned), s);
                    HIP_CHECK(hipMemsetAsync(evalTodo, 0, sizeof(unsigned), s));
                    a.P.evalTodo = static_cast<unsigned*>(evalTodo);
                }
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
            if (e == hipSuccess && a.P.evalTodo) {  // the batches the scaled-image launch deferred (as a rule: none), from the plain image
                EvalArgs b2 = a;
                // (one workgroup per CU walks the list: as a rule it is empty and the launch costs a few microseconds)
                const unsigned gridTodo = unsigned(std::min<size_t>(blocks, size_t(net->numCUs)));
                b2.P.ldsImage = net->packed.params.ldsImage;
                b2.P.reluClamp = 0;
                hipError_t e2 = hipErrorInvalidDeviceFunction;
                const VariantKey& k = net->key;
                const int smallGrid = k.grid == 0 ? 0 : (k.grid == 1 && b2.P.gridK == 1 && b2.P.bias0Folded ? 1 : 2);
                if (net->opts[FVSRN_OPT_SMALL_KERNEL] != 0 && k.CD == 2 && smallGrid <= 1 && !b2.P.noFourier && !b2.P.fourierNeedsFractEval && b2.P.numLayers >= 1 &&
                    b2.P.numLayers <= 3)
                    e2 = launch_eval_small(k.act, k.dir, b2.P.numLayers, smallGrid, b2, gridTodo, unsigned(64 * wpb), size_t(net->packed.params.ldsBytes), s);
                if (e2 == hipErrorInvalidDeviceFunction) e2 = launch_eval(k, b2, gridTodo, unsigned(64 * wpb), size_t(net->packed.params.ldsBytes), s);
                e = e2;
            }
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
static void fillSceneParams(fvsrn_scene* scene, const fvsrn_scene_desc& d, int width, int height, SceneParams& S) {
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
            // the fragment-major variant of the wide latent-grid renderers (render_stripe_kernel, kernels.hpp: no register spills, 1 % slower):
            // on request only (FVSRN_OPT_OVERLAP_KERNEL = 1) since the launch-to-launch differences it was built around turned out to be a
            // hazard in the tap arithmetic (srn_device.hpp, grid_tap) and not concurrent scratch use
            const void* stripeFn = nullptr;
            if (!smallFn && !a.shaded && O[FVSRN_OPT_OVERLAP_KERNEL] == 1)
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
            if ((smallFn && smallGrid == 2) || cellsFn) a.P.cellTable = net->ensureCellTable(a.shaded, s);
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
                else if (stripeFn) scene->lastKernel = "render_stripe_kernel<" + v + ">";
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

static int extractImpl(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping, float max_exposure,
                       float* d_out4, unsigned int* d_out8, void* stream, const float* d_range3 = nullptr) {
    if (!d_raw8 || (!d_out4 && !d_out8)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null image pointer");
    if (width <= 0 || height <= 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad image size");
    if (channel_mode < FVSRN_CHANNEL_MASK || channel_mode > FVSRN_CHANNEL_COLOR) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad channel mode");
    if (use_tonemapping && !(max_exposure > 0)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "max_exposure must be positive");
    return guarded([&]() -> int {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(FVSRN_ERR_NO_DEVICE, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
        // 3 words of scratch for the depth range: one small allocation per thread, reused
        thread_local DeviceBuffer scratch;
        try {
            scratch.ensure(16);
        } catch (const DeviceError& e) {
            return fail(FVSRN_ERR_DEVICE, e.what());
        }
        ExtractParams p{};
        p.raw = d_raw8; p.out4 = d_out4; p.out8 = d_out8;
        p.minmax = static_cast<float*>(scratch.ptr);
        p.range3 = channel_mode == FVSRN_CHANNEL_DEPTH ? d_range3 : nullptr;
        p.pixels = (unsigned long long)width * (unsigned long long)height;
        p.mode = channel_mode; p.tonemap = use_tonemapping; p.maxExposure = max_exposure;
        const hipError_t e = launch_extract_color(p, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("extract_color failed: ") + hipGetErrorString(e));
        return FVSRN_OK;
    });
}

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
        thread_local DeviceBuffer scratch;
        try {
            scratch.ensure(16);
        } catch (const DeviceError& e) {
            return fail(FVSRN_ERR_DEVICE, e.what());
        }
        const unsigned long long pixels = (unsigned long long)width * (unsigned long long)height;
        const hipError_t e = launch_depth_range(d_raw8 + 7 * pixels, pixels, static_cast<float*>(scratch.ptr), d_range3, static_cast<hipStream_t>(stream));
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
    if (stripe_rows <= 0 || stripe_rows % 8 != 0 || world <= 0 || rank < 0 || rank >= world)
        return fail(FVSRN_ERR_INVALID_ARGUMENT, "stripe_rows must be a positive multiple of 8 and 0 <= rank < world");
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
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ dense grid volumes
struct fvsrn_volume {
    std::mutex mu;
    std::vector<float> host;  // x fastest: x + X (y + Y z), like Volume::MipmapLevel::idx (volume.h:126-132)
    int res[3] = {0, 0, 0};
    float boxMin[3] = {0, 0, 0}, boxSize[3] = {1, 1, 1};
    DeviceBuffer dData;
    bool deviceValid = false;
    ~fvsrn_volume() { dData.release(); }
    void ensureDevice(hipStream_t s) {  // upload in 4x4x4 bricks (grid_volume.hpp)
        if (deviceValid) return;
        const size_t bx = size_t(res[0] + 3) / 4, by = size_t(res[1] + 3) / 4, bz = size_t(res[2] + 3) / 4;
        std::vector<float> bricked(bx * by * bz * 64, 0.f);
        for (int z = 0; z < res[2]; ++z)
            for (int y = 0; y < res[1]; ++y) {
                const float* row = host.data() + size_t(res[0]) * (size_t(y) + size_t(res[1]) * size_t(z));
                const size_t base = ((size_t(z >> 2) * by + size_t(y >> 2)) * bx) * 64 + size_t(((z & 3) << 4) | ((y & 3) << 2));
                for (int x = 0; x < res[0]; ++x) bricked[base + size_t(x >> 2) * 64 + size_t(x & 3)] = row[x];
            }
        dData.ensure(bricked.size() * sizeof(float));
        HIP_CHECK(hipMemcpyAsync(dData.ptr, bricked.data(), bricked.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        deviceValid = true;
    }
    VolumeParams params(int source, int interpolation, int newBehavior, int provideNormals = 0) const {
        VolumeParams V{};
        V.data = static_cast<const float*>(dData.ptr);
        for (int i = 0; i < 3; ++i) { V.res[i] = res[i]; V.boxMin[i] = boxMin[i]; V.boxSize[i] = boxSize[i]; }
        V.bricks[0] = (res[0] + 3) / 4; V.bricks[1] = (res[1] + 3) / 4;
        V.source = source; V.interpolation = interpolation; V.newBehavior = newBehavior; V.provideNormals = provideNormals;
        return V;
    }
};

namespace {
// u8 / u16 voxels are read as normalised floats like the reference's textures (cudaReadModeNormalizedFloat, volume.cpp:109-167)
void convertVoxels(const void* src, int dtype, size_t n, float* dst) {
    switch (dtype) {
        case FVSRN_VOLUME_U8: { const unsigned char* p = static_cast<const unsigned char*>(src); for (size_t i = 0; i < n; ++i) dst[i] = float(p[i]) / 255.0f; } break;
        case FVSRN_VOLUME_U16: { const unsigned short* p = static_cast<const unsigned short*>(src); for (size_t i = 0; i < n; ++i) dst[i] = float(p[i]) / 65535.0f; } break;
        default: std::memcpy(dst, src, n * sizeof(float)); break;
    }
}
int checkVolumeModes(int source, int interpolation) {
    if (source != FVSRN_VOLUME_SOURCE_TEXTURE && source != FVSRN_VOLUME_SOURCE_TENSOR) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad volume source");
    if (interpolation < FVSRN_VOLUME_NEAREST || interpolation > FVSRN_VOLUME_TRICUBIC) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad volume interpolation");
    return FVSRN_OK;
}
}  // namespace

extern "C" {

int fvsrn_volume_create(const void* host_data, int dtype, int sx, int sy, int sz, int x_fastest, const float box_min[3],
                        const float box_size[3], fvsrn_volume** out) {
    return guarded([&] {
        if (!host_data || !out || !box_min || !box_size) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        if (sx <= 0 || sy <= 0 || sz <= 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad volume resolution");
        if (dtype < FVSRN_VOLUME_U8 || dtype > FVSRN_VOLUME_F32) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad volume data type");
        for (int i = 0; i < 3; ++i)
            if (!(box_size[i] > 0)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "box size must be positive");
        auto v = std::make_unique<fvsrn_volume>();
        const size_t n = size_t(sx) * sy * sz;
        v->host.resize(n);
        if (x_fastest) {
            convertVoxels(host_data, dtype, n, v->host.data());
        } else {  // contiguous (X,Y,Z) tensor: z fastest
            std::vector<float> tmp(n);
            convertVoxels(host_data, dtype, n, tmp.data());
            for (int x = 0; x < sx; ++x)
                for (int y = 0; y < sy; ++y)
                    for (int z = 0; z < sz; ++z) v->host[size_t(x) + size_t(sx) * (size_t(y) + size_t(sy) * z)] = tmp[(size_t(x) * sy + y) * sz + z];
        }
        v->res[0] = sx; v->res[1] = sy; v->res[2] = sz;
        for (int i = 0; i < 3; ++i) { v->boxMin[i] = box_min[i]; v->boxSize[i] = box_size[i]; }
        *out = v.release();
        return FVSRN_OK;
    });
}

int fvsrn_volume_destroy(fvsrn_volume* volume) {
    delete volume;
    return FVSRN_OK;
}

int fvsrn_volume_get_data(fvsrn_volume* volume, float* out, size_t count) {
    if (!volume || !out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    if (count != volume->host.size()) return fail(FVSRN_ERR_INVALID_ARGUMENT, "count must be the number of voxels (fvsrn_volume_info)");
    std::memcpy(out, volume->host.data(), count * sizeof(float));
    return FVSRN_OK;
}

int fvsrn_volume_info(fvsrn_volume* volume, int resolution[3], float box_min[3], float box_size[3]) {
    if (!volume) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    for (int i = 0; i < 3; ++i) {
        if (resolution) resolution[i] = volume->res[i];
        if (box_min) box_min[i] = volume->boxMin[i];
        if (box_size) box_size[i] = volume->boxSize[i];
    }
    return FVSRN_OK;
}

// Volume::save / Volume::Volume(filename) (volume.cpp:623-668, 685-740), Feature::save / load (:278-332, 346-385)
// One LZ4 block (the published block format) for `n` bytes at `src`: greedy matcher with a 4-byte hash table, matches inside the block only
// (an independent block is a valid message of the dependent stream the reader decodes), the format's end-of-block rules: the last sequence is
// literals only, its last five bytes are literals, no match starts in the last twelve bytes.
static void lz4CompressBlock(const unsigned char* src, size_t n, std::vector<char>& out) {
    auto emit = [&](const unsigned char* lit, size_t litLen, size_t matchLen, size_t offset) {
        const size_t ml = matchLen ? matchLen - 4 : 0;
        out.push_back(char(((litLen >= 15 ? 15 : litLen) << 4) | (ml >= 15 ? 15 : ml)));
        if (litLen >= 15) { size_t r = litLen - 15; for (; r >= 255; r -= 255) out.push_back(char(255)); out.push_back(char(r)); }
        out.insert(out.end(), lit, lit + litLen);
        if (matchLen) {
            out.push_back(char(offset & 255)); out.push_back(char(offset >> 8));
            if (ml >= 15) { size_t r = ml - 15; for (; r >= 255; r -= 255) out.push_back(char(255)); out.push_back(char(r)); }
        }
    };
    std::vector<int> table(1 << 13, -1);
    const size_t matchStartLimit = n >= 12 ? n - 12 : 0, matchEndLimit = n >= 5 ? n - 5 : 0;
    size_t i = 0, anchor = 0;
    while (i < matchStartLimit) {
        unsigned v;
        std::memcpy(&v, src + i, 4);
        const unsigned h = (v * 2654435761u) >> 19;
        const int cand = table[h];
        table[h] = int(i);
        if (cand >= 0 && i - size_t(cand) <= 65535 && std::memcmp(src + cand, src + i, 4) == 0) {
            size_t len = 4;
            while (i + len < matchEndLimit && src[size_t(cand) + len] == src[i + len]) ++len;
            emit(src + anchor, i - anchor, len, i - size_t(cand));
            i += len;
            anchor = i;
        } else {
            ++i;
        }
    }
    emit(src + anchor, n - anchor, 0, 0);
}

// Volume::save (volume.cpp:623-682): the version-1 container; compression > 0 sets Flag_Compressed and writes every feature body as LZ4
// messages in the framing lz4ReadMessages documents (int32 size + one block per <= 64 KiB of input; the reference's levels 1 .. 9 select
// LZ4 / LZ4-HC effort, here every level is the greedy matcher: the format is the same, the files are larger than LZ4-HC's)
int fvsrn_cvol_write(const char* path, const float world_size[3], int num_features, const fvsrn_cvol_feature* features, const void* const* data,
                     int compression) {
    return guarded([&] {
        if (!path || !world_size || num_features < 0 || (num_features > 0 && (!features || !data))) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        if (compression < 0 || compression > 9) return fail(FVSRN_ERR_INVALID_ARGUMENT, "Illegal compression factor");  // volume.cpp:634-635
        static const size_t bytesPerType[3] = {1, 2, 4};
        for (int i = 0; i < num_features; ++i) {
            const fvsrn_cvol_feature& ft = features[i];
            if (!data[i] || ft.resolution[0] <= 0 || ft.resolution[1] <= 0 || ft.resolution[2] <= 0 || ft.channels <= 0 || ft.dtype < FVSRN_VOLUME_U8 ||
                ft.dtype > FVSRN_VOLUME_F32 || !std::memchr(ft.name, 0, sizeof(ft.name)))
                return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad feature " + std::to_string(i));
        }
        std::ofstream f(path, std::ios::binary);
        if (!f) return fail(FVSRN_ERR_IO, std::string("cannot write ") + path);
        const int version = 1, flags = compression > 0 ? 1 : 0;
        const char pad[4] = {0, 0, 0, 0};
        f.write("CVOL", 4);
        f.write(reinterpret_cast<const char*>(&version), 4);
        f.write(reinterpret_cast<const char*>(world_size), 12);
        f.write(reinterpret_cast<const char*>(&num_features), 4);
        f.write(reinterpret_cast<const char*>(&flags), 4);
        f.write(pad, 4);
        std::vector<char> block;
        for (int i = 0; i < num_features; ++i) {
            const fvsrn_cvol_feature& ft = features[i];
            const int lenName = int(std::strlen(ft.name));
            const unsigned long long X = ft.resolution[0], Y = ft.resolution[1], Z = ft.resolution[2];
            f.write(reinterpret_cast<const char*>(&lenName), 4);
            f.write(ft.name, lenName);
            f.write(reinterpret_cast<const char*>(&X), 8);
            f.write(reinterpret_cast<const char*>(&Y), 8);
            f.write(reinterpret_cast<const char*>(&Z), 8);
            f.write(reinterpret_cast<const char*>(&ft.channels), 4);
            f.write(reinterpret_cast<const char*>(&ft.dtype), 4);
            const size_t bytes = bytesPerType[ft.dtype] * size_t(ft.channels) * X * Y * Z;
            if (compression > 0) {
                const unsigned char* p = static_cast<const unsigned char*>(data[i]);
                for (size_t pos = 0; pos < bytes; pos += 65536) {
                    block.clear();
                    lz4CompressBlock(p + pos, std::min<size_t>(65536, bytes - pos), block);
                    const int size = int(block.size());
                    f.write(reinterpret_cast<const char*>(&size), 4);
                    f.write(block.data(), size);
                }
            } else {
                f.write(static_cast<const char*>(data[i]), std::streamsize(bytes));
            }
        }
        if (!f) return fail(FVSRN_ERR_IO, std::string("error while writing ") + path);
        return FVSRN_OK;
    });
}

static int saveCvol(const char* path, const char* feature_name, const void* host_data, int dtype, int sx, int sy, int sz, float world_x, float world_y,
                    float world_z, int compression) {
    if (!path || !feature_name || !host_data) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    if (sx <= 0 || sy <= 0 || sz <= 0 || dtype < FVSRN_VOLUME_U8 || dtype > FVSRN_VOLUME_F32) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad volume");
    if (std::strlen(feature_name) >= sizeof(fvsrn_cvol_feature{}.name)) return fail(FVSRN_ERR_INVALID_ARGUMENT, "feature name too long");
    fvsrn_cvol_feature ft{};
    std::strcpy(ft.name, feature_name);
    ft.index = 0; ft.num_features = 1; ft.dtype = dtype; ft.channels = 1;
    ft.resolution[0] = sx; ft.resolution[1] = sy; ft.resolution[2] = sz;
    const float world[3] = {world_x, world_y, world_z};
    const void* ptrs[1] = {host_data};
    return fvsrn_cvol_write(path, world, 1, &ft, ptrs, compression);
}

int fvsrn_volume_save_cvol(const char* path, const char* feature_name, const void* host_data, int dtype, int sx, int sy, int sz,
                           float world_x, float world_y, float world_z) {
    return saveCvol(path, feature_name, host_data, dtype, sx, sy, sz, world_x, world_y, world_z, 0);
}

int fvsrn_volume_save_cvol_compressed(const char* path, const char* feature_name, const void* host_data, int dtype, int sx, int sy, int sz,
                                      float world_x, float world_y, float world_z, int compression) {
    return saveCvol(path, feature_name, host_data, dtype, sx, sy, sz, world_x, world_y, world_z, compression);
}

// ---- LZ4 framing of compressed .cvol bodies -------------------------------------------------------------------------------------
// The reference compresses through its `lz4cpp` wrapper (LZ4Compressor / LZ4Decompressor, third-party/lz4cpp: an EMPTY submodule in the
// reference snapshot), in streaming mode: messages of at most 64 KiB (LZ4Compressor::MAX_CHUNK_SIZE), each stored as
//     int32 compressed size | one LZ4 block (the published block format: token, literals, 16-bit offset, match length)
// whose matches may reach back up to 64 KiB into the messages before it (dependent blocks, one stream for the whole file).  The framing
// is recovered from the one volume the snapshot holds, applications/volumes/RichtmyerMeshkov/ppm-t0020.cvol: 256 messages of exactly
// 65 536 bytes decode to 256^3 bytes and consume the file to its last byte (tests/test_volnet_format.py keeps its histogram).
// Decodes messages until `want` bytes are appended to `out` (`out` may already hold earlier features of the same stream: the history).
static const char* lz4ReadMessages(std::istream& f, std::vector<unsigned char>& out, size_t want) {
    const size_t end = out.size() + want;
    std::vector<unsigned char> src;
    while (out.size() < end) {
        int csize = 0;
        f.read(reinterpret_cast<char*>(&csize), 4);
        if (!f || csize <= 0 || csize > (1 << 24)) return "corrupt LZ4 message header";
        src.resize(size_t(csize));
        f.read(reinterpret_cast<char*>(src.data()), csize);
        if (!f) return "unexpected end of file inside an LZ4 message";
        size_t i = 0;
        const size_t n = src.size();
        while (i < n) {
            const unsigned tok = src[i++];
            size_t lit = tok >> 4;
            if (lit == 15) {
                unsigned b;
                do {
                    if (i >= n) return "corrupt LZ4 block (literal length)";
                    b = src[i++];
                    lit += b;
                } while (b == 255);
            }
            if (lit > n - i || lit > end - out.size()) return "corrupt LZ4 block (literals overrun)";
            out.insert(out.end(), src.begin() + long(i), src.begin() + long(i + lit));
            i += lit;
            if (i >= n) break;  // the last sequence of a block has no match
            if (n - i < 2) return "corrupt LZ4 block (offset)";
            const size_t off = size_t(src[i]) | (size_t(src[i + 1]) << 8);
            i += 2;
            size_t len = tok & 15;
            if (len == 15) {
                unsigned b;
                do {
                    if (i >= n) return "corrupt LZ4 block (match length)";
                    b = src[i++];
                    len += b;
                } while (b == 255);
            }
            len += 4;
            if (off == 0 || off > out.size() || len > end - out.size()) return "corrupt LZ4 block (match outside the stream)";
            const size_t start = out.size() - off;
            out.resize(out.size() + len);
            unsigned char* d = out.data() + start + off;
            const unsigned char* sp = out.data() + start;
            for (size_t k = 0; k < len; ++k) d[k] = sp[k];  // (overlapping matches repeat their pattern: byte by byte)
        }
    }
    return nullptr;
}

int fvsrn_cvol_read(const char* path, float world_size[3], fvsrn_cvol_feature_callback on_feature, void* user) {
    return guarded([&] {
        if (!path || !on_feature) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        std::ifstream f(path, std::ios::binary);
        if (!f) return fail(FVSRN_ERR_IO, std::string("Unable to open file ") + path);
        char magic[4] = {0, 0, 0, 0};
        f.read(magic, 4);
        static const size_t bytesPerType[3] = {1, 2, 4};
        constexpr size_t kMaxBytes = size_t(1) << 34;  // 16 GiB of decoded host data per feature
        fvsrn_cvol_feature info;
        std::memset(&info, 0, sizeof info);
        if (f && std::memcmp(magic, "cvol", 4) == 0) {
            // the old format: one density feature (Volume::Volume(filename), volume.cpp:741-793)
            unsigned long long X = 0, Y = 0, Z = 0;
            double voxel[3] = {0, 0, 0};
            unsigned type = 0;
            char useCompression = 0;
            f.read(reinterpret_cast<char*>(&X), 8); f.read(reinterpret_cast<char*>(&Y), 8); f.read(reinterpret_cast<char*>(&Z), 8);
            f.read(reinterpret_cast<char*>(voxel), 24);
            f.read(reinterpret_cast<char*>(&type), 4);
            f.read(&useCompression, 1);
            f.ignore(7);
            if (!f || type > 2 || X == 0 || Y == 0 || Z == 0 || X > 65536 || Y > 65536 || Z > 65536 || !(voxel[0] > 0) || !(voxel[1] > 0) || !(voxel[2] > 0))
                return fail(FVSRN_ERR_FORMAT, "corrupt header of a legacy 'cvol' file");
            const size_t bytes = bytesPerType[type] * X * Y * Z;
            if (bytes > kMaxBytes) return fail(FVSRN_ERR_UNSUPPORTED, "volume too large");
            std::vector<unsigned char> raw;
            if (useCompression) {
                raw.reserve(bytes);
                if (const char* why = lz4ReadMessages(f, raw, bytes)) return fail(FVSRN_ERR_FORMAT, why);
            } else {
                raw.resize(bytes);
                f.read(reinterpret_cast<char*>(raw.data()), std::streamsize(bytes));
                if (!f) return fail(FVSRN_ERR_FORMAT, "unexpected end of file");
            }
            if (world_size) { world_size[0] = float(voxel[0] * double(X)); world_size[1] = float(voxel[1] * double(Y)); world_size[2] = float(voxel[2] * double(Z)); }
            std::snprintf(info.name, sizeof info.name, "density");
            info.index = 0; info.num_features = 1; info.dtype = int(type); info.channels = 1;
            info.resolution[0] = int(X); info.resolution[1] = int(Y); info.resolution[2] = int(Z);
            on_feature(user, &info, raw.data(), raw.size());
            return FVSRN_OK;
        }
        int version = 0, numFeatures = 0, flags = 0;
        float world[3];
        if (!f || std::memcmp(magic, "CVOL", 4) != 0) return fail(FVSRN_ERR_FORMAT, "Illegal magic number");
        f.read(reinterpret_cast<char*>(&version), 4);
        if (version != 1) return fail(FVSRN_ERR_FORMAT, "Unknown file version!");
        f.read(reinterpret_cast<char*>(world), 12);
        f.read(reinterpret_cast<char*>(&numFeatures), 4);
        f.read(reinterpret_cast<char*>(&flags), 4);
        f.ignore(4);
        if (!f || numFeatures < 0 || numFeatures > 1024) return fail(FVSRN_ERR_FORMAT, "corrupt .cvol header");
        if (world_size) for (int i = 0; i < 3; ++i) world_size[i] = world[i];
        const bool compressed = (flags & 1) != 0;  // Flag_Compressed: every feature body is a run of LZ4 messages of ONE stream (Volume::save :647-664)
        std::vector<unsigned char> stream;  // compressed files: <= 64 KiB of history + the current feature (a match may reach into the previous feature)
        for (int i = 0; i < numFeatures; ++i) {
            int lenName = 0, channels = 0, type = 0;
            unsigned long long X = 0, Y = 0, Z = 0;
            f.read(reinterpret_cast<char*>(&lenName), 4);
            if (!f || lenName < 0 || lenName > 4096) return fail(FVSRN_ERR_FORMAT, "corrupt feature header");
            std::string name(size_t(lenName), ' ');
            f.read(name.data(), lenName);
            f.read(reinterpret_cast<char*>(&X), 8);
            f.read(reinterpret_cast<char*>(&Y), 8);
            f.read(reinterpret_cast<char*>(&Z), 8);
            f.read(reinterpret_cast<char*>(&channels), 4);
            f.read(reinterpret_cast<char*>(&type), 4);
            if (!f || type < 0 || type > 2 || channels <= 0 || channels > 64 || X == 0 || Y == 0 || Z == 0 || X > 65536 || Y > 65536 || Z > 65536)
                return fail(FVSRN_ERR_FORMAT, "corrupt feature header");
            const size_t bytes = bytesPerType[type] * X * Y * Z * size_t(channels);
            if (bytes > kMaxBytes) return fail(FVSRN_ERR_UNSUPPORTED, "volume too large");
            if (compressed) {
                if (stream.size() > (size_t(1) << 16)) stream.erase(stream.begin(), stream.end() - (1 << 16));  // only the last 64 KiB can be referenced
                if (const char* why = lz4ReadMessages(f, stream, bytes)) return fail(FVSRN_ERR_FORMAT, why);
            } else {
                stream.resize(bytes);
                f.read(reinterpret_cast<char*>(stream.data()), std::streamsize(bytes));
                if (!f) return fail(FVSRN_ERR_FORMAT, "unexpected end of file");
            }
            std::snprintf(info.name, sizeof info.name, "%s", name.c_str());
            info.index = i; info.num_features = numFeatures; info.dtype = type; info.channels = channels;
            info.resolution[0] = int(X); info.resolution[1] = int(Y); info.resolution[2] = int(Z);
            if (on_feature(user, &info, stream.data() + (stream.size() - bytes), bytes) != 0) break;  // (non-zero: the caller has what it wants)
        }
        return FVSRN_OK;
    });
}

int fvsrn_volume_load_cvol(const char* path, int feature_index, fvsrn_volume** out) {
    if (!path || !out) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
    struct Pick { int want; int rc; bool seen; int numFeatures; float world[3]; fvsrn_volume** out; } pick{feature_index, FVSRN_OK, false, 0, {1, 1, 1}, out};
    // (the world size is known before the first callback: fvsrn_cvol_read fills it from the header)
    const int rc = fvsrn_cvol_read(path, pick.world, [](void* user, const fvsrn_cvol_feature* info, const void* data, size_t) -> int {
        Pick& p = *static_cast<Pick*>(user);
        p.numFeatures = info->num_features;
        if (info->index != p.want) return 0;
        p.seen = true;
        if (info->channels != 1) { p.rc = fail(FVSRN_ERR_UNSUPPORTED, "only scalar (1-channel) features can be rendered as densities"); return 1; }
        const float boxMin[3] = {-p.world[0] / 2, -p.world[1] / 2, -p.world[2] / 2};  // VolumeInterpolationGrid::setSource, :193-198
        p.rc = fvsrn_volume_create(data, info->dtype, info->resolution[0], info->resolution[1], info->resolution[2], 1, boxMin, p.world, p.out);
        return 1;
    }, &pick);
    if (rc != FVSRN_OK) return rc;
    if (!pick.seen) return fail(FVSRN_ERR_INVALID_ARGUMENT, "no such feature in the volume");
    return pick.rc;
}

int fvsrn_volume_evaluate_points(fvsrn_volume* volume, int source, int interpolation, int grid_resolution_new_behavior,
                                 const float* d_positions, size_t n, float* d_out, void* stream) {
    return guarded([&] {
        if (!volume || (n > 0 && (!d_positions || !d_out))) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        if (const int rc = checkVolumeModes(source, interpolation)) return rc;
        std::lock_guard<std::mutex> lock(volume->mu);
        try {
            hipStream_t s = static_cast<hipStream_t>(stream);
            volume->ensureDevice(s);
            const hipError_t e = launch_volume_evaluate(volume->params(source, interpolation, grid_resolution_new_behavior), d_positions, n, d_out, s);
            if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("Error during evaluation! ") + hipGetErrorString(e));
            return FVSRN_OK;
        } catch (const DeviceError& e) {
            return fail(fvsrn_device_count() == 0 ? FVSRN_ERR_NO_DEVICE : FVSRN_ERR_DEVICE, e.what());
        }
    });
}

int fvsrn_render_volume(fvsrn_scene* scene, fvsrn_volume* volume, int source, int interpolation, int grid_resolution_new_behavior,
                        int provide_normals, int width, int height, float* d_out8, unsigned long long* d_stats, void* stream) {
    return guarded([&] {
        if (!scene || !volume || !d_out8) return fail(FVSRN_ERR_INVALID_ARGUMENT, "null argument");
        if (width <= 0 || height <= 0) return fail(FVSRN_ERR_INVALID_ARGUMENT, "bad image size");
        if (const int rc = checkVolumeModes(source, interpolation)) return rc;
        std::lock_guard<std::mutex> lockV(volume->mu);
        std::lock_guard<std::mutex> lockS(scene->mu);
        try {
            hipStream_t s = static_cast<hipStream_t>(stream);
            const fvsrn_scene_desc& d = scene->desc;
            if (d.tf_kind == FVSRN_TF_NONE) return fail(FVSRN_ERR_INVALID_ARGUMENT, "a grid volume holds densities; the scene needs a transfer function");
            // (fvsrn_scene_desc::gradient_mode configures network volumes; a grid always differentiates by central differences)
            const int normals = provide_normals || d.brdf_enable_phong || d.brdf_enable_magnitude_scaling ||  // brdf.cpp:40,279
                                d.tf_gaussian_mode == FVSRN_TF_GAUSSIAN_SCALE_WITH_GRADIENT;                 // transfer_function_gaussian.cpp:271-272
            volume->ensureDevice(s);
            const size_t tfFloats = scene->tfTable.size();
            if (const int rc = scene->uploadTf(d.stepsize, s)) return rc;
            SceneParams S{};
            fillSceneParams(scene, d, width, height, S);
            S.width = width; S.height = height; S.y0 = 0; S.y1 = height;
            // depth segments (see renderImpl): enough waves to cover the gather latency of small images, >= 48 steps per segment
            // (early-out then works per segment; not with a pre-integrated TF, which looks at the previous sample)
            int K = 1;
            {
                static int numCUs = 0;  // hipGetDeviceProperties costs ~0.1 ms: once
                if (numCUs == 0) {
                    hipDeviceProp_t prop;
                    int dev = 0;
                    numCUs = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
                }
                const double waves = double(((width + 15) / 16) * ((height + 15) / 16)) * 4.0, slots = double(numCUs) * 32.0;
                const float* bs = volume->boxSize;
                const double maxSteps = std::sqrt(double(bs[0]) * bs[0] + double(bs[1]) * bs[1] + double(bs[2]) * bs[2]) / d.stepsize;
                const bool looksBack = d.tf_preintegration != FVSRN_PREINTEGRATE_NONE || d.tf_gaussian_mode == FVSRN_TF_GAUSSIAN_ANALYTIC;
                while (!looksBack && K < 8 && waves * K < slots / 2 && maxSteps / (2 * K) >= 48.0) K *= 2;  // r01, 256^2: K = 1 / 2 / 4 / 8 -> 0.34 / 0.19 / 0.17 / 0.24 ms
                if (scene->opts[FVSRN_OPT_DEPTH_SEGMENTS] >= 1 && !looksBack) K = scene->opts[FVSRN_OPT_DEPTH_SEGMENTS];
            }
            S.segments = K;
            const size_t plane = size_t(width) * size_t(height);
            if (K > 1) {
                scene->dPartial.ensure(size_t(K) * 8 * plane * sizeof(float));
                S.partial = static_cast<float*>(scene->dPartial.ptr);
            }
            hipError_t e = launch_volume_render(volume->params(source, interpolation, grid_resolution_new_behavior, normals), S, d_out8, d_stats, tfFloats, s);
            if (e == hipSuccess && K > 1) e = launch_composite(S.partial, d_out8, K, plane, S, s);
            if (e != hipSuccess) return fail(FVSRN_ERR_DEVICE, std::string("Error during rendering! ") + hipGetErrorString(e));
            return FVSRN_OK;
        } catch (const DeviceError& e) {
            return fail(fvsrn_device_count() == 0 ? FVSRN_ERR_NO_DEVICE : FVSRN_ERR_DEVICE, e.what());
        }
    });
}

}  // extern "C"
