// Re-layout of a SceneNetwork into the LDS image consumed by the MFMA kernels.
#pragma once
#include <vector>

#include "device_params.hpp"
#include "scene_network.hpp"

namespace fvsrn {

struct PackedNetwork {
    NetworkConfig cfg;
    int MT = 0, KS = 0, KS0 = 0, NL = 0;
    std::vector<char> ldsImage;  // see device_params.hpp
    NetParams params{};          // pointers left null (filled by the device layer)
    double mfmaFlopsPerSample = 0;
    // working latent grid, f16 bits [Z][Y][X][G] (empty without a grid)
    std::vector<uint16_t> grid;
    int gridX = 0, gridY = 0, gridZ = 0, gridC = 0;
};

// K slot k of a chained 32x32x16 MFMA -> row (= channel) of the producing accumulator tile
inline int chiOfSlot(int k) {
    const int s = k >> 4, h = (k >> 3) & 1, j = k & 7;
    return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
}

// Throws Unsupported for networks outside the compiled variant set, InvalidNetwork like getDefines.
PackedNetwork packNetwork(const SceneNetwork& net);
// (re)build only the blended working grid for net.currentTime / currentEnsemble
void packLatentGrid(const SceneNetwork& net, PackedNetwork& out);

}  // namespace fvsrn
