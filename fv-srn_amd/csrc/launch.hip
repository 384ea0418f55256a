#include "launch.hpp"

namespace fvsrn {

#define FVSRN_DISPATCH_CD(expr_prefix, ...)          \
    switch (k.CD) {                                  \
        case 2: return expr_prefix<2>(__VA_ARGS__);  \
        case 3: return expr_prefix<3>(__VA_ARGS__);  \
        case 4: return expr_prefix<4>(__VA_ARGS__);  \
        case 6: return expr_prefix<6>(__VA_ARGS__);  \
        case 8: return expr_prefix<8>(__VA_ARGS__);  \
        default: break;                              \
    }

bool kernel_info(const VariantKey& k, KernelInfo* info) {
    FVSRN_DISPATCH_CD(kernel_info_cd, k, info)
    return false;
}
hipError_t launch_eval(const VariantKey& k, const EvalArgs& a, unsigned gridDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_eval_cd, k, a, gridDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}
hipError_t launch_render(const VariantKey& k, const RenderArgs& a, unsigned gridDim, size_t ldsBytes, hipStream_t s) {
    FVSRN_DISPATCH_CD(launch_render_cd, k, a, gridDim, ldsBytes, s)
    return hipErrorInvalidDeviceFunction;
}

}  // namespace fvsrn
