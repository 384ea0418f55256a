// render_small_kernel<.., ADVANCE = false>: the register-resident kernels (Fourier-only and cell-table variants) for FVSRN_OPT_FOURIER_RESYNC = 1 -- every
// step derives its features from the fp16 position like the reference, so the per-step rotation is compiled out (kernels.hpp)
#include "kernels.hpp"
#include "launch.hpp"

namespace fvsrn {

#define FVSRN_EXACT_VARIANTS(X) \
    X(ACT_RELU01, false) X(ACT_RELU01, true) X(ACT_SINE, false) X(ACT_SINE, true) \
    X(ACT_SNAKE, false) X(ACT_SNAKE, true) X(ACT_SNAKEALT, false) X(ACT_SNAKEALT, true) X(ACT_SIGMOID, false) X(ACT_SIGMOID, true) \
    X(ACT_SNAKEALT0, false) X(ACT_SNAKEALT0, true)
#define FVSRN_EXACT_LAYERS(A, D) \
    Y(A, D, 1, 1) Y(A, D, 2, 1) Y(A, D, 3, 1) Y(A, D, 1, 2) Y(A, D, 2, 2) Y(A, D, 3, 2) Y(A, D, 1, 3) Y(A, D, 2, 3) Y(A, D, 3, 3) \
    Y(A, D, 1, 4) Y(A, D, 2, 4) Y(A, D, 3, 4) Y(A, D, 1, 5) Y(A, D, 2, 5) Y(A, D, 3, 5)
#define FVSRN_EXACT_CELL_VARIANTS(X) X(ACT_RELU01, false) X(ACT_SNAKEALT, false) X(ACT_SNAKEALT0, false) X(ACT_SINE, false) X(ACT_SNAKE, false)
#define FVSRN_EXACT_CELL_LAYERS(A, D) \
    G(A, D, 1, 1) G(A, D, 2, 1) G(A, D, 3, 1) G(A, D, 1, 4) G(A, D, 2, 4) G(A, D, 3, 4) G(A, D, 1, 5) G(A, D, 2, 5) G(A, D, 3, 5)

const void* render_small_exact_fn(int act, bool dir, int numLayers, int tail, int grid) {
    if (grid == 2) {
#define G(A, D, N, L) \
        if (act == A && dir == D && numLayers == N && tail == L) return reinterpret_cast<const void*>(&render_small_kernel<A, D, N, L, 2, false>);
        FVSRN_EXACT_CELL_VARIANTS(FVSRN_EXACT_CELL_LAYERS)
#undef G
        return nullptr;
    }
    if (grid != 0) return nullptr;
#define Y(A, D, N, L) \
    if (act == A && dir == D && numLayers == N && tail == L) return reinterpret_cast<const void*>(&render_small_kernel<A, D, N, L, 0, false>);
    FVSRN_EXACT_VARIANTS(FVSRN_EXACT_LAYERS)
#undef Y
    return nullptr;
}

hipError_t launch_render_small_exact(int act, bool dir, int numLayers, int tail, int grid, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    if (grid == 2) {
#define G(A, D, N, L)                                                                                             \
        if (act == A && dir == D && numLayers == N && tail == L) {                                                \
            hipLaunchKernelGGL((render_small_kernel<A, D, N, L, 2, false>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.S, a.out, a.stats); \
            return hipGetLastError();                                                                             \
        }
        FVSRN_EXACT_CELL_VARIANTS(FVSRN_EXACT_CELL_LAYERS)
#undef G
        return hipErrorInvalidDeviceFunction;
    }
#define Y(A, D, N, L)                                                                                         \
    if (act == A && dir == D && numLayers == N && tail == L) {                                                \
        hipLaunchKernelGGL((render_small_kernel<A, D, N, L, 0, false>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.S, a.out, a.stats); \
        return hipGetLastError();                                                                             \
    }
    FVSRN_EXACT_VARIANTS(FVSRN_EXACT_LAYERS)
#undef Y
    return hipErrorInvalidDeviceFunction;
}

}  // namespace fvsrn
