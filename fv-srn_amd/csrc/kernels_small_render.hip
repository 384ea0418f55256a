// render_small_kernel: 32-wide Fourier-only scalar networks resident in registers (kernels.hpp, srn_device.hpp)
#include "kernels.hpp"
#include "launch.hpp"

namespace fvsrn {

#define FVSRN_SMALL_VARIANTS(X) \
    X(ACT_RELU01, false) X(ACT_RELU01, true) X(ACT_SINE, false) X(ACT_SINE, true) \
    X(ACT_SNAKE, false) X(ACT_SNAKE, true) X(ACT_SNAKEALT, false) X(ACT_SNAKEALT, true) X(ACT_SIGMOID, false) X(ACT_SIGMOID, true) \
    X(ACT_SNAKEALT0, false) X(ACT_SNAKEALT0, true)
#define FVSRN_SMALL_LAYERS(A, D) \
    Y(A, D, 1, 1) Y(A, D, 2, 1) Y(A, D, 3, 1) Y(A, D, 1, 2) Y(A, D, 2, 2) Y(A, D, 3, 2) Y(A, D, 1, 3) Y(A, D, 2, 3) Y(A, D, 3, 3) \
    Y(A, D, 1, 4) Y(A, D, 2, 4) Y(A, D, 3, 4) Y(A, D, 1, 5) Y(A, D, 2, 5) Y(A, D, 3, 5)

// grid: 0 none, 1 one 16-channel chunk of decoded latent values (scalar networks behind an Identity / Texture TF only: tails 4 / 5 / 1)
#define FVSRN_SMALL_GRID_VARIANTS(X) X(ACT_RELU01, false) X(ACT_SNAKEALT, false) X(ACT_SNAKEALT0, false) X(ACT_SINE, false) X(ACT_SNAKE, false)
#define FVSRN_SMALL_GRID_LAYERS(A, D) \
    G(A, D, 1, 1) G(A, D, 2, 1) G(A, D, 3, 1) G(A, D, 1, 4) G(A, D, 2, 4) G(A, D, 3, 4) G(A, D, 1, 5) G(A, D, 2, 5) G(A, D, 3, 5)

const void* render_small_fn(int act, bool dir, int numLayers, int tail, int grid) {
    if (grid == 2) return render_small_cells_fn(act, dir, numLayers, tail);
    if (grid == 1) {
#define G(A, D, N, L) \
        if (act == A && dir == D && numLayers == N && tail == L) return reinterpret_cast<const void*>(&render_small_kernel<A, D, N, L, 1>);
        FVSRN_SMALL_GRID_VARIANTS(FVSRN_SMALL_GRID_LAYERS)
#undef G
        return nullptr;
    }
    if (grid != 0) return nullptr;
#define Y(A, D, N, L) \
    if (act == A && dir == D && numLayers == N && tail == L) return reinterpret_cast<const void*>(&render_small_kernel<A, D, N, L>);
    FVSRN_SMALL_VARIANTS(FVSRN_SMALL_LAYERS)
#undef Y
    return nullptr;
}

hipError_t launch_render_small(int act, bool dir, int numLayers, int tail, int grid, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    if (grid == 2) return launch_render_small_cells(act, dir, numLayers, tail, a, gridDim, blockDim, ldsBytes, s);
    if (grid == 1) {
#define G(A, D, N, L)                                                                                             \
        if (act == A && dir == D && numLayers == N && tail == L) {                                                \
            hipLaunchKernelGGL((render_small_kernel<A, D, N, L, 1>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.S, a.out, a.stats); \
            return hipGetLastError();                                                                             \
        }
        FVSRN_SMALL_GRID_VARIANTS(FVSRN_SMALL_GRID_LAYERS)
#undef G
        return hipErrorInvalidDeviceFunction;
    }
#define Y(A, D, N, L)                                                                                         \
    if (act == A && dir == D && numLayers == N && tail == L) {                                                \
        hipLaunchKernelGGL((render_small_kernel<A, D, N, L>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.S, a.out, a.stats); \
        return hipGetLastError();                                                                             \
    }
    FVSRN_SMALL_VARIANTS(FVSRN_SMALL_LAYERS)
#undef Y
    return hipErrorInvalidDeviceFunction;
}

// ---- evaluate_small_kernel (plain weight image: evaluate takes positions outside the box) ---------------------------------------
#define FVSRN_SMALL_EVAL_VARIANTS(X) \
    X(ACT_RELU, false) X(ACT_RELU, true) X(ACT_SINE, false) X(ACT_SINE, true) X(ACT_SNAKE, false) X(ACT_SNAKE, true) \
    X(ACT_SNAKEALT, false) X(ACT_SNAKEALT, true) X(ACT_SIGMOID, false) X(ACT_SIGMOID, true) \
    X(ACT_RELU01, false) X(ACT_RELU01, true) X(ACT_SNAKEALT0, false) X(ACT_SNAKEALT0, true)
#define FVSRN_SMALL_EVAL_LAYERS(A, D) Z(A, D, 1) Z(A, D, 2) Z(A, D, 3)

#define FVSRN_SMALL_EVAL_GRID_VARIANTS(X) X(ACT_RELU, false) X(ACT_SINE, false) X(ACT_SNAKE, false) X(ACT_SNAKEALT, false) X(ACT_RELU01, false) X(ACT_SNAKEALT0, false)

hipError_t launch_eval_small(int act, bool dir, int numLayers, int grid, const EvalArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
    if (grid == 1) {
#define Z(A, D, N)                                                                                                  \
        if (act == A && dir == D && numLayers == N) {                                                               \
            if (a.P.evalHalfIO)                                                                                     \
                hipLaunchKernelGGL((evaluate_small_kernel<A, D, N, 1, true>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.pos, a.dir, a.n, a.out, \
                                   a.outChannels);                                                                  \
            else                                                                                                    \
                hipLaunchKernelGGL((evaluate_small_kernel<A, D, N, 1>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.pos, a.dir, a.n, a.out, \
                                   a.outChannels);                                                                  \
            return hipGetLastError();                                                                               \
        }
        FVSRN_SMALL_EVAL_GRID_VARIANTS(FVSRN_SMALL_EVAL_LAYERS)
#undef Z
        return hipErrorInvalidDeviceFunction;
    }
    if (grid != 0) return hipErrorInvalidDeviceFunction;
#define Z(A, D, N)                                                                                                  \
    if (act == A && dir == D && numLayers == N) {                                                                   \
        if (a.P.evalHalfIO)                                                                                         \
            hipLaunchKernelGGL((evaluate_small_kernel<A, D, N, 0, true>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.pos, a.dir, a.n, a.out, \
                               a.outChannels);                                                                      \
        else                                                                                                        \
            hipLaunchKernelGGL((evaluate_small_kernel<A, D, N>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.pos, a.dir, a.n, a.out, \
                               a.outChannels);                                                                      \
        return hipGetLastError();                                                                                   \
    }
    FVSRN_SMALL_EVAL_VARIANTS(FVSRN_SMALL_EVAL_LAYERS)
#undef Z
    return hipErrorInvalidDeviceFunction;
}

}  // namespace fvsrn
