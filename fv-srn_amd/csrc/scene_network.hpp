// Host model of the scene representation network and the .volnet container.
//
// Mirrors the *behaviour* of renderer::SceneNetwork and friends
// (reference renderer/volume_interpolation_network.{h,cpp}); storage is plain std::vector,
// no torch, no CUDA types.  Field semantics and on-disk layout are cited per function in the .cpp.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fvsrn.h"

namespace fvsrn {

struct FormatError : std::runtime_error { using std::runtime_error::runtime_error; };
struct InvalidNetwork : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

struct InputParametrization {
    bool hasTime = false;
    bool hasDirection = false;
    int numFourierFeatures = 0;
    bool useDirectionInFourierFeatures = false;
    // half bits, feature-fastest: F[cout + numFourier*cin], premultiplied by 2*pi
    std::vector<uint16_t> fourierMatrix;

    int channelsOut() const;
    bool valid(std::string* why) const;
};

struct Layer {
    int channelsIn = 0, channelsOut = 0;
    std::vector<uint16_t> weights;  // half bits, as stored after addLayer (see SceneNetwork::addLayer)
    std::vector<uint16_t> bias;     // half bits
    fvsrn_activation activation = FVSRN_ACT_NONE;
    float activationParameter = 1.f;
    bool valid(bool isOutputLayer) const;
};

struct LatentGrid {
    fvsrn_grid_encoding encoding = FVSRN_GRID_FLOAT;
    int gridChannels = 0, gridSizeZ = 0, gridSizeY = 0, gridSizeX = 0;
    std::vector<char> grid;  // [C/4][Z][Y][X][4] of float or uint8
    std::vector<float> gridOffsetOrMean, gridScaleOrStd;

    size_t bytesPerEntry() const { return encoding == FVSRN_GRID_FLOAT ? 4 : 1; }
    size_t idx(int cHigh, int z, int y, int x, int cLow) const {
        return cLow + 4 * (x + size_t(gridSizeX) * (y + size_t(gridSizeY) * (z + size_t(gridSizeZ) * cHigh)));
    }
    bool isValid(std::string* why) const;
    // returns the average absolute encoding error
    static std::shared_ptr<LatentGrid> fromFloat(const float* czyx, int C, int Z, int Y, int X,
                                                 fvsrn_grid_encoding enc, double* encodingError);
    // value of channel c at voxel (z,y,x) *before* decoding: float value, or byte/255
    float raw(int c, int z, int y, int x) const;
};

struct LatentGridTimeAndEnsemble {
    int timeMin = 0, timeNum = 0, timeStep = 1;
    std::vector<std::shared_ptr<LatentGrid>> timeGrids;
    int ensembleMin = 0, ensembleNum = 0;
    std::vector<std::shared_ptr<LatentGrid>> ensembleGrids;

    bool hasTimeGrids() const { return timeNum > 0; }
    bool hasEnsembleGrids() const { return ensembleNum > 0; }
    int timeMaxInclusive() const { return timeMin + (timeNum - 1) * timeStep; }
    int ensembleMaxInclusive() const { return ensembleMin + ensembleNum - 1; }
    float interpolateTime(float time) const;
    int interpolateEnsemble(int ensemble) const;
    bool isValid(std::string* why) const;
    fvsrn_grid_encoding commonEncoding() const;
    int timeChannels() const { return timeGrids.empty() ? 0 : timeGrids[0]->gridChannels; }
    int ensembleChannels() const { return ensembleGrids.empty() ? 0 : ensembleGrids[0]->gridChannels; }
    int totalChannels() const { return timeChannels() + ensembleChannels(); }
};

// Compile-time configuration the reference derives in SceneNetwork::getDefines
// (volume_interpolation_network.cpp:1139-1219); here it selects the AOT kernel variant.
struct NetworkConfig {
    int hiddenChannels = 0;   // HIDDEN_CHANNELS
    int numHiddenLayers = 0;  // NUM_HIDDEN_LAYERS (CxC layers after the first/grid layer)
    bool hasFourier = false;
    int numFourier = 0;
    int directionMode = 0;  // USE_DIRECTION
    fvsrn_activation activation = FVSRN_ACT_NONE;
    float activationParam = 1.f;
    fvsrn_output_mode outputMode = FVSRN_OUT_DENSITY;
    int gridChannels = 0;  // 16*LATENT_GRID_CHANNELS_DIV16
    fvsrn_grid_encoding gridEncoding = FVSRN_GRID_FLOAT;
    bool passTime = false;
};

class SceneNetwork {
public:
    InputParametrization input;
    fvsrn_output_mode outputMode = FVSRN_OUT_DENSITY;
    std::vector<Layer> hidden;  // all Linear layers incl. the last, as stored
    float boxMin[3] = {-5.f, -5.f, -5.f};  // reference default, volume_interpolation_network.cpp:799-800
    float boxSize[3] = {1.f, 1.f, 1.f};
    std::shared_ptr<LatentGridTimeAndEnsemble> latentGrid;
    float currentTime = 0.f;
    int currentEnsemble = 0;

    static std::shared_ptr<SceneNetwork> load(const void* bytes, size_t len);
    std::vector<char> save() const;

    void setFourierMatrix(const float* m, int numFourier, int cols, bool premultiplied);
    void addLayer(Layer layer);
    void addLayerFromFloat(const float* w, const float* b, int cout, int cin, fvsrn_activation act, float param);
    void setTimeAndEnsemble(float time, int ensemble);
    bool valid(std::string* why) const;
    int numParameters() const;
    int computeMaxWarps(bool onlySharedMemory, bool adjoint) const;
    NetworkConfig config() const;  // throws InvalidNetwork / Unsupported like getDefines throws
    int outputChannels() const;
    int outputChannelsIn() const;
    double flopsPerSample() const;
};

const char* activationName(fvsrn_activation a);
fvsrn_activation activationFromString(const std::string& s);
const char* outputModeName(fvsrn_output_mode m);
fvsrn_output_mode outputModeFromString(const std::string& s);

}  // namespace fvsrn
