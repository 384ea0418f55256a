// Re-layout of a SceneNetwork into the LDS image consumed by the MFMA kernels.
#pragma once
#include <vector>

#include "device_params.hpp"
#include "scene_network.hpp"

namespace fvsrn {

// all latent key frames in device layout (see packLatentGrid)
struct GridKeyframes {
    fvsrn_grid_encoding enc = FVSRN_GRID_FLOAT;
    int X = 0, Y = 0, Z = 0, Gt = 0, Ge = 0, timeNum = 0, ensNum = 0;
    size_t records = 0;                    // Z*Y*(X+1)
    std::vector<char> timeData, ensData;   // [key][record][Gc][2] fp32 | uint8
    std::vector<float> timeOffset, timeScale, ensOffset, ensScale;  // [key][Gc]
};
struct GridSelection {
    int lo = 0, hi = 0, ens = 0;
    float frac = 0.f, timeIndex = 0.f;
};
GridSelection selectGrid(const SceneNetwork& net);

struct PackedNetwork {
    NetworkConfig cfg;
    int MT = 0, KS = 0, KS0 = 0, NL = 0;
    std::vector<char> ldsImage;  // see device_params.hpp
    std::vector<char> ldsImageScaled;  // second image for the renderer (empty if not applicable): ReLU networks with activations scaled
                                       // into [0,1], SnakeAlt networks with the 1/(2p) factor folded into the next layer
    bool scaledBias0Exact = true;      // ldsImageScaled: the folded first-layer bias is exact in fp16 (else: residue in the fp32 bias block, no resident SGRID kernel)
    int scaledAct = -1;                // ACT_RELU01 / ACT_SNAKEALT0: the kernel variant that goes with ldsImageScaled
    std::vector<char> ldsImageCurvature;  // densitycurvature networks: the plain image with the last layer computing outputs 4, 5 in rows 0, 1
    std::vector<int> reluExponents;    // e_l of the scaled image
    std::vector<float> gridMaxAbs;     // per latent channel
    NetParams params{};          // pointers left null (filled by the device layer)
    double mfmaFlopsPerSample = 0;
    GridKeyframes keys;
    int gridX = 0, gridY = 0, gridZ = 0, gridC = 0;
    int timeSlotOffset = -1;  // see pack.cpp
};

// K slot k of a chained 32x32x16 MFMA -> row (= channel) of the producing accumulator tile
inline int chiOfSlot(int k) {
    const int s = k >> 4, h = (k >> 3) & 1, j = k & 7;
    return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
}

// Throws Unsupported for networks outside the compiled variant set, InvalidNetwork like getDefines.
PackedNetwork packNetwork(const SceneNetwork& net);
// PackedNetwork::mfmaFlopsPerSample from the configuration alone; numLinearLayers = SceneNetwork::hidden.size()
double mfmaFlopsPerSample(const NetworkConfig& c, int numLinearLayers);
void packLatentGrid(const SceneNetwork& net, PackedNetwork& out);

}  // namespace fvsrn
