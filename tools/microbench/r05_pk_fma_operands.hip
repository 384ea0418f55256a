// Round-5 microbenchmark: what does ONE packed-fp32 instruction cost the SIMD's issue port, by opcode and by where its operands live?
// (r05 found the fused SnakeAlt step -- 2 v_cos_f32 + v_pk_fma_f32 + convert -- SLOWER than the shipped five instructions: a v_pk_fma_f32 with three
// register-pair operands issued in ~11 cycles there, against 4.6 for v_pk_mul_f32 / v_pk_add_f32.  Is that the opcode or the operand pattern?)
// 16 independent instructions per loop iteration on register pairs v[8+4i : 9+4i] (banks 0,1) / v[10+4i : 11+4i] (banks 2,3), two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o r05_pk_fma_operands r05_pk_fma_operands.hip ; results are garbage by design (raw asm, no data).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
             "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
             "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", \
             "v68", "v69", "v70", "v71", "vcc"
#define X16(L) L(8) L(12) L(16) L(20) L(24) L(28) L(32) L(36) L(40) L(44) L(48) L(52) L(56) L(60) L(64) L(68)
#define MUL_(b)   "v_pk_mul_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[0:1]\n"
#define ADD_(b)   "v_pk_add_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[0:1]\n"
#define FMA_A(b)  "v_pk_fma_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[0:1], v[" #b "+2:" #b "+3]\n"                 /* src0, src1 on banks 0,1; src2 = dst on 2,3 */
#define FMA_B(b)  "v_pk_fma_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[2:3], v[" #b "+2:" #b "+3]\n"                 /* src1 on banks 2,3 like src2 */
#define FMA_C(b)  "v_pk_fma_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], s[4:5], v[" #b "+2:" #b "+3]\n"                 /* src1 in scalar registers */
#define FMA_D(b)  "v_pk_fma_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[" #b ":" #b "+1], v[" #b "+2:" #b "+3]\n"     /* src0 = src1: two distinct register pairs */
#define FMA_E(b)  "v_pk_fma_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[0:1], 1.0\n"                                  /* src2 an inline constant */
#define FMA_R(b)  "v_pk_fma_f32 v[" #b ":" #b "+1], v[" #b ":" #b "+1], v[0:1], v[" #b "+2:" #b "+3] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n"  /* the feature rotation's second instruction */
#define SFMA_(b)  "v_fma_f32 v[" #b "+2], v[" #b "], v0, v[" #b "+2]\n" "v_fma_f32 v[" #b "+3], v[" #b "+1], v1, v[" #b "+3]\n"  /* two scalar FMAs */
#define MULADD_(b) "v_pk_mul_f32 v[" #b "+2:" #b "+3], v[" #b ":" #b "+1], v[0:1]\n" "v_pk_add_f32 v[" #b "+2:" #b "+3], v[" #b "+2:" #b "+3], v[2:3]\n"

#define DEF(NAME, BODY)                                                                                    \
    __global__ void __launch_bounds__(512) k_##NAME(float* out, int iters, long long* clk) {               \
        const long long c0 = clock64();                                                                    \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY ::: CLOB); }                                \
        const long long c1 = clock64();                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;                                         \
        out[threadIdx.x] = 0;                                                                              \
    }
DEF(mul, X16(MUL_)) DEF(add, X16(ADD_)) DEF(fmaA, X16(FMA_A)) DEF(fmaB, X16(FMA_B)) DEF(fmaC, X16(FMA_C)) DEF(fmaD, X16(FMA_D)) DEF(fmaE, X16(FMA_E))
DEF(fmaR, X16(FMA_R)) DEF(sfma, X16(SFMA_)) DEF(muladd, X16(MULADD_))

template <class K>
double run(K k, int wavesPerSimd) {
    float* out; long long* clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 16);
    const int iters = 20000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    hipLaunchKernelGGL(k, dim3(p.multiProcessorCount), dim3(64 * 4 * wavesPerSimd), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(p.multiProcessorCount), dim3(64 * 4 * wavesPerSimd), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    hipFree(out); hipFree(clk);
    return double(c) / iters / 16.0 / wavesPerSimd;  // clock64 ticks per instruction (group) and wave: SIMD issue time
}

int main() {
    printf("packed fp32 on gfx950: SIMD issue time per instruction (group), clock64 ticks / 16 / waves per SIMD\n%-78s %8s %8s\n", "form", "1 wave", "2 waves");
#define ROW(NAME, TEXT) printf("%-78s %8.2f %8.2f\n", TEXT, run(k_##NAME, 1), run(k_##NAME, 2));
    ROW(mul, "v_pk_mul_f32 d, v(banks 0,1), v[0:1]")
    ROW(add, "v_pk_add_f32 d, v(banks 0,1), v[0:1]")
    ROW(fmaA, "v_pk_fma_f32 d, v(0,1), v[0:1], d(2,3)              (the fused SnakeAlt step)")
    ROW(fmaB, "v_pk_fma_f32 d, v(0,1), v[2:3], d(2,3)")
    ROW(fmaC, "v_pk_fma_f32 d, v(0,1), s[4:5], d(2,3)              (scalar src1)")
    ROW(fmaD, "v_pk_fma_f32 d, v(0,1), same pair, d(2,3)           (two distinct pairs)")
    ROW(fmaE, "v_pk_fma_f32 d, v(0,1), v[0:1], 1.0                 (constant src2)")
    ROW(fmaR, "v_pk_fma_f32 with the op_sel / neg of the feature rotation")
    ROW(sfma, "2 x v_fma_f32                                        (scalar pair)")
    ROW(muladd, "v_pk_mul_f32 + v_pk_add_f32                          (unfused pair)")
    return 0;
}
