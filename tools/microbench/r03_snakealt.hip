// Round-3 microbenchmark: the SnakeAlt activation x - cos(2 p x) of a pair of fp32 accumulator values -> one packed fp16
// register, as instruction skeletons on gfx950.  Three formulations, 16 independent pairs per loop iteration, 1 / 2 / 4 waves per SIMD:
//   A  shipped (ACT_SNAKEALT0, srn_device.hpp): v_pk_mul_f32 (phase in revolutions), 2 x v_cos_f32, v_pk_add_f32 (x - c), v_cvt_pk_f16_f32
//   B  half-precision transcendentals like the reference's hcos (renderer_activations.cuh): v_cvt_pk_f16_f32, v_pk_mul_f16,
//      2 x v_cos_f16 (low / high half by SDWA selects: gfx950 has no op_sel on VOP1), v_pk_add_f16
//   C  packed fp16 polynomial: convert, v_pk_mul_f16, range reduction by the 1.5 * 2^10 trick (2 x v_pk_add_f16 + 1), r^2, a degree-4 even
//      polynomial (4 x v_pk_fma_f16), v_pk_add_f16
// r05: two more skeletons for "one packed instruction per pair besides the cosines and the convert"
//   D  the phase u = x p / pi arrives in the accumulator (scale carried by the previous layer's weights): 2 x v_cos_f32, v_pk_fma_f32 (pi/p u - c), convert
//   E  D plus what makes D EXACT: u from a hi + lo copy of the weight image is two more K passes of the layer -- per 16 pairs (two tile-layers of a
//      32-wide network) eight more v_mfma_f32_32x32x16_f16
// build: hipcc --offload-arch=gfx950 -O3 -o r03_snakealt r03_snakealt.hip ; results are garbage by design (raw asm, no data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
             "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
             "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", \
             "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", \
             "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "vcc"
// pair i uses the register pair v[8+4i : 9+4i] as its fp32 input / scratch and v[10+4i] / v[11+4i] as temporaries
#define PAIRS(L) L(8) L(12) L(16) L(20) L(24) L(28) L(32) L(36) L(40) L(44) L(48) L(52) L(56) L(60) L(64) L(68)
#define A_(b)                                                              \
    "v_pk_mul_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[0:1]\n"     \
    "v_cos_f32 v[" #b "+2], v[" #b "+2]\n"                                     \
    "v_cos_f32 v[" #b "+3], v[" #b "+3]\n"                                     \
    "v_pk_add_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[" #b "+2:" #b "+3] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "v_cvt_pk_f16_f32 v[" #b "+2], v[" #b "+2], v[" #b "+3]\n"
#define B_(b)                                                              \
    "v_cvt_pk_f16_f32 v[" #b "+2], v[" #b "], v[" #b "+1]\n"               \
    "v_pk_mul_f16 v[" #b "+3], v[" #b "+2], v0\n"                          \
    "v_cos_f16_sdwa v[" #b "], v[" #b "+3] dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n" \
    "v_cos_f16_sdwa v[" #b "], v[" #b "+3] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n" \
    "v_pk_add_f16 v[" #b "+2], v[" #b "+2], v[" #b "] neg_lo:[0,1] neg_hi:[0,1]\n"
#define C_(b)                                                              \
    "v_cvt_pk_f16_f32 v[" #b "+2], v[" #b "], v[" #b "+1]\n"                     \
    "v_pk_mul_f16 v[" #b "+3], v[" #b "+2], v0\n"                              \
    "v_pk_add_f16 v[" #b "], v[" #b "+3], v1\n"                                \
    "v_pk_add_f16 v[" #b "], v[" #b "], v1 neg_lo:[0,1] neg_hi:[0,1]\n"        \
    "v_pk_add_f16 v[" #b "+3], v[" #b "+3], v[" #b "] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "v_pk_mul_f16 v[" #b "], v[" #b "+3], v[" #b "+3]\n"                         \
    "v_pk_fma_f16 v[" #b "+1], v[" #b "], v2, v3\n"                            \
    "v_pk_fma_f16 v[" #b "+1], v[" #b "+1], v[" #b "], v4\n"                     \
    "v_pk_fma_f16 v[" #b "+1], v[" #b "+1], v[" #b "], v5\n"                     \
    "v_pk_fma_f16 v[" #b "+1], v[" #b "+1], v[" #b "], v6\n"                     \
    "v_pk_add_f16 v[" #b "+2], v[" #b "+2], v[" #b "+1] neg_lo:[0,1] neg_hi:[0,1]\n"
#define D_(b)                                                              \
    "v_cos_f32 v[" #b "+2], v[" #b "]\n"                                       \
    "v_cos_f32 v[" #b "+3], v[" #b "+1]\n"                                     \
    "v_pk_fma_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[0:1], v[" #b "+2:" #b "+3] neg_lo:[0,0,1] neg_hi:[0,0,1]\n" \
    "v_cvt_pk_f16_f32 v[" #b "+2], v[" #b "+2], v[" #b "+3]\n"
#define MFMA8                                                              \
    "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n" "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n" \
    "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n" "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n" \
    "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n" "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n" \
    "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n" "v_mfma_f32_32x32x16_f16 v[72:87], v[88:91], v[92:95], v[72:87]\n"
// the two fp32 -> fp16 steps every formulation shares with ReLU (the floor of any activation): convert only
#define Z_(b) "v_cvt_pk_f16_f32 v[" #b "+2], v[" #b "], v[" #b "+1]\n"

#define DEF(NAME, BODY)                                                                                    \
    __global__ void __launch_bounds__(256) k_##NAME(float* out, int iters, long long* clk) {               \
        const long long c0 = clock64();                                                                    \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY ::: CLOB); }                                \
        const long long c1 = clock64();                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;                                         \
        out[threadIdx.x] = 0;                                                                              \
    }
DEF(A, PAIRS(A_)) DEF(B, PAIRS(B_)) DEF(C, PAIRS(C_)) DEF(Z, PAIRS(Z_)) DEF(D, PAIRS(D_)) DEF(E, PAIRS(D_) MFMA8) DEF(AM, PAIRS(A_) "s_nop 0\n")

template <class K>
double run(K k, int wavesPerSimd) {
    float* out; long long* clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 16);
    const int iters = 20000;
    // one workgroup per CU on every CU: wavesPerSimd * 4 waves of 64 lanes
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    hipLaunchKernelGGL(k, dim3(p.multiProcessorCount), dim3(64 * 4 * wavesPerSimd), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(p.multiProcessorCount), dim3(64 * 4 * wavesPerSimd), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    hipFree(out); hipFree(clk);
    return double(c) / iters / 16.0 / wavesPerSimd;  // clock64 ticks per pair and wave (SIMD time)
}

int main() {
    printf("SnakeAlt activation of one pair of values, cycles of SIMD issue time per pair (clock64 ticks / 16 pairs / waves per SIMD)\n");
    printf("%-44s %8s %8s %8s\n", "formulation", "1 wave", "2 waves", "4 waves");
    struct { const char* name; double v[3]; } rows[6] = {
        {"Z convert only (the ReLU floor)", {run(k_Z, 1), run(k_Z, 2), run(k_Z, 4)}},
        {"A shipped: pk_mul_f32, 2 v_cos_f32, pk_add, cvt", {run(k_A, 1), run(k_A, 2), run(k_A, 4)}},
        {"B cvt, pk_mul_f16, 2 v_cos_f16, pk_add_f16", {run(k_B, 1), run(k_B, 2), run(k_B, 4)}},
        {"C cvt + packed fp16 polynomial (10 pk ops)", {run(k_C, 1), run(k_C, 2), run(k_C, 4)}},
        {"D 2 v_cos_f32, pk_fma_f32, cvt (phase given)", {run(k_D, 1), run(k_D, 2), run(k_D, 4)}},
        {"E D + 8 MFMA per 16 pairs (exact hi+lo phase)", {run(k_E, 1), run(k_E, 2), run(k_E, 4)}}};
    for (auto& r : rows) printf("%-44s %8.2f %8.2f %8.2f\n", r.name, r.v[0], r.v[1], r.v[2]);
    printf("r05: D saves %.1f %% of A per pair; with the MFMAs that make its phase exact (E) it costs %.1f %% MORE than A (2 waves per SIMD)\n",
           100 * (1 - rows[4].v[1] / rows[1].v[1]), 100 * (rows[5].v[1] / rows[1].v[1] - 1));
    printf("per wave step of the 32x4 network (96 hidden activations = 48 pairs), 2 waves per SIMD: A %.0f, B %.0f, C %.0f, convert only %.0f cycles\n",
           48 * rows[1].v[1], 48 * rows[2].v[1], 48 * rows[3].v[1], 48 * rows[0].v[1]);
    return 0;
}
