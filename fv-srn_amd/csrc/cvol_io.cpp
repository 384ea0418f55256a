// Dense grid volumes behind the C ABI: fvsrn_volume_*, the .cvol container (old and new format, LZ4 messages; reader and writer) and fvsrn_render_volume.
#include "api_internal.hpp"

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
        // the header's sizes are not trusted before a body is read -- an uncompressed body cannot be longer than what is left of the file, an
        // LZ4 body decodes to at most 255 x its bytes (a match token extends by 255 per byte): a 40-byte file that declares 16 GiB fails here instead
        // of zero-filling them
        f.seekg(0, std::ios::end);
        const unsigned long long fileSize = f.tellg() < 0 ? 0ull : (unsigned long long)f.tellg();
        f.seekg(4, std::ios::beg);
        auto bodyFits = [&](size_t bytes, bool lz4) {
            const std::streamoff at = f.tellg();
            const unsigned long long left = at < 0 || (unsigned long long)at > fileSize ? 0ull : fileSize - (unsigned long long)at;
            return lz4 ? bytes / 255 <= left : bytes <= left;
        };
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
            if (!bodyFits(bytes, useCompression != 0)) return fail(FVSRN_ERR_FORMAT, "the header declares more voxels than the file can hold");
            std::vector<unsigned char> raw;
            if (useCompression) {
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
            if (!bodyFits(bytes, compressed)) return fail(FVSRN_ERR_FORMAT, "a feature header declares more voxels than the file can hold");
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
