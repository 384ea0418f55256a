// render_small_kernel<.., SGRID = 2>: 32-wide scalar networks resident in registers whose latent grid enters through the cell table
// (srn_forward_resident_cells, srn_device.hpp; grid_cell_table_kernel, launch.hip)
#include "kernels.hpp"
#include "launch.hpp"

namespace fvsrn {

// the variant set of SGRID = 1 (kernels_small_render.hip): scalar networks behind an Identity / Texture TF (tails 4 / 5 / 1), no view direction
#define FVSRN_CELL_VARIANTS(X) X(ACT_RELU01, false) X(ACT_SNAKEALT, false) X(ACT_SNAKEALT0, false) X(ACT_SINE, false) X(ACT_SNAKE, false)
#define FVSRN_CELL_LAYERS(A, D) \
    G(A, D, 1, 1) G(A, D, 2, 1) G(A, D, 3, 1) G(A, D, 1, 4) G(A, D, 2, 4) G(A, D, 3, 4) G(A, D, 1, 5) G(A, D, 2, 5) G(A, D, 3, 5)

const void* render_small_cells_fn(int act, bool dir, int numLayers, int tail) {
#define G(A, D, N, L) \
    if (act == A && dir == D && numLayers == N && tail == L) return reinterpret_cast<const void*>(&render_small_kernel<A, D, N, L, 2>);
    FVSRN_CELL_VARIANTS(FVSRN_CELL_LAYERS)
#undef G
    return nullptr;
}

hipError_t launch_render_small_cells(int act, bool dir, int numLayers, int tail, const RenderArgs& a, unsigned gridDim, unsigned blockDim, size_t ldsBytes, hipStream_t s) {
#define G(A, D, N, L)                                                                                             \
    if (act == A && dir == D && numLayers == N && tail == L) {                                                    \
        hipLaunchKernelGGL((render_small_kernel<A, D, N, L, 2>), dim3(gridDim), dim3(blockDim), ldsBytes, s, a.P, a.S, a.out, a.stats); \
        return hipGetLastError();                                                                                 \
    }
    FVSRN_CELL_VARIANTS(FVSRN_CELL_LAYERS)
#undef G
    return hipErrorInvalidDeviceFunction;
}

}  // namespace fvsrn
