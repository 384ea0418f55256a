// Shared host-side state of the C ABI (include/fvsrn.h): options, error reporting, device buffers, cross-stream ordering, the device state of a network
// handle (weight images, key-frame store, working grids, cell tables) and of a scene handle.  Included by the translation units of the ABI:
//   api.cpp          handles, options, point evaluation, ExtractColor, ray / TF tensor APIs
//   launch_plan.cpp  fvsrn_render / _stripes / _stripes_batch: kernel selection and launch shape (the scheduler heuristics), debug report
//   keyframes.cpp    KeyframeStore and the device state of a network (uploads, key-frame blend, lazy cell tables)
//   cvol_io.cpp      grid volumes: the .cvol container (LZ4 reader / writer), fvsrn_volume_*, fvsrn_render_volume
#pragma once
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

inline const Options& defaultOptions() {
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
inline int wavesPerBlockFor(size_t ldsBytesPerBlock, const Options& o) {
    if (o[FVSRN_OPT_WAVES_PER_BLOCK]) return o[FVSRN_OPT_WAVES_PER_BLOCK];
    const size_t budget = 160 * 1024;
    for (int w : {1, 2, 4})
        if (size_t(16 / w) * ldsBytesPerBlock <= budget) return w;
    return 4;
}

inline thread_local std::string g_lastError;

inline int fail(int code, const std::string& msg) {
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
inline TemporaryPools g_temporaries;

struct WrongDevice : WrongDeviceBase { using WrongDeviceBase::WrongDeviceBase; };
// A handle's device state lives on the device that was current at its first use; every later call must run there.
// FVSRN_DEBUG_DEVICE_SKEW=k (developer / test switch, read once): every check AFTER the binding call compares the current device with `bound + k`, i.e. a handle
// behaves as if it had been bound k devices further on -- its next call meets the refusal below on a one-GPU box (tests/test_gpu_parity.py
// test_wrong_device_check_fires_on_one_gpu; a second device cannot be had on a one-GPU lease).  The binding call itself and the stored index are untouched.
inline int debugDeviceSkew() {
    static const int skew = [] { const char* e = std::getenv("FVSRN_DEBUG_DEVICE_SKEW"); return e ? std::atoi(e) : 0; }();
    return skew;
}
inline void bindOrCheckDevice(int& bound, const char* what) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (bound < 0) bound = dev;
    else if (bound + debugDeviceSkew() != dev)
        throw WrongDevice(std::string(what) + " holds resources on HIP device " + std::to_string(bound + debugDeviceSkew()) + ", but the current device is " +
                          std::to_string(dev) + " (hipSetDevice before the call, or use one handle per device)");
}

inline int actIndex(fvsrn_activation a) {
    switch (a) {
        case FVSRN_ACT_RELU: return 0;
        case FVSRN_ACT_SINE: return 1;
        case FVSRN_ACT_SNAKE: return 2;
        case FVSRN_ACT_SNAKEALT: return 3;
        case FVSRN_ACT_SIGMOID: return 5;  // ACT_SIGMOID (4 is the scaled-ReLU image)
        default: return -1;
    }
}

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
    void release();
    // host data of all key frames -> pinned memory; `budget` = 0 (all resident) or the number of device slots
    void init(const std::vector<char>& data, int keys, int budget);
    const char* slotPtr(int slot) const { return static_cast<const char*>(dSlots.ptr) + size_t(slot) * bytesPerKey; }
    void upload(int key, int slot, bool prefetch);
    int victim(int keepA, int keepB) const;
    // device pointers of key frames lo / hi for a blend enqueued on `stream` (which is made to wait for their uploads)
    void acquire(int lo, int hi, float time, hipStream_t stream, const void** pLo, const void** pHi);
    // call after the blend kernel has been enqueued on `stream`
    void released(int lo, int hi, hipStream_t stream);
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
    size_t cellTableBytesCorners = 0;  // the shaded renderer's table (corner form over the grid's own cells)
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

    void pack();

    void ensureDevice(hipStream_t stream);

    // Fills the cell table of working grid W from its blended records (grid_cell_table_kernel); the caller holds W's write bracket.  plain: the table
    // of the plain weight image (the shaded renderer's) -- the same buffer as the unshaded one where the network has no re-scaled image.
    void buildCellTable(WorkingGrid& W, bool plain, hipStream_t stream);
    // The table of the CURRENT working grid for a launch on `stream` (after syncTime, before beginUse): built now if no launch has needed it since the
    // last blend.  Readers of the grid on other streams are waited for like by a blend, later readers wait for this write.
    const void* ensureCellTable(bool plain, hipStream_t stream);

    // Brings the working grid and the time input of the network in line with net->currentTime/currentEnsemble:
    // one small kernel + (networks that take the time as input) a 2-byte patch, both stream-ordered -- no host
    // synchronisation, no re-upload (the reference re-fills its constant block and lazily uploads textures with a
    // synchronous cudaMemcpy3D, volume_interpolation_network.cpp:482-488,923-938,1308-1315).
    void syncTime(hipStream_t stream);

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

    void releaseDevice();
    ~fvsrn_network() { releaseDevice(); }
};

struct BoxCenter { float c[3]; };
inline BoxCenter P_boxCenter(const NetParams& P) {
    return {{P.boxMin[0] + 0.5f * P.boxSize[0], P.boxMin[1] + 0.5f * P.boxSize[1], P.boxMin[2] + 0.5f * P.boxSize[2]}};
}

// Live scene handles, for fvsrn_debug_state (a watchdog thread asks what the library last launched when a caller hangs).  Leaked on purpose: handles
// may be destroyed during static destruction.
struct SceneRegistry { std::mutex mu; std::vector<fvsrn_scene*> live; };
inline SceneRegistry& sceneRegistry() { static SceneRegistry* r = new SceneRegistry; return *r; }

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

inline int tfCols(int kind) {
    switch (kind) {
        case FVSRN_TF_GAUSSIAN: return 6;
        case FVSRN_TF_PIECEWISE: return 5;
        case FVSRN_TF_TEXTURE: return 4;
        default: return 0;
    }
}

// shared between the translation units of the ABI
int extractImpl(const float* d_raw8, int width, int height, int channel_mode, int use_tonemapping, float max_exposure, float* d_out4, unsigned int* d_out8,
                void* stream, const float* d_range3 = nullptr);                                             // api.cpp
void fillSceneParams(fvsrn_scene* scene, const fvsrn_scene_desc& d, int width, int height, fvsrn::SceneParams& S);  // launch_plan.cpp
