// Microbenchmark: issue cost (cycles per wave64 instruction, one wave per SIMD, independent operations) of the VALU
// instructions the SRN kernels are made of, and of MFMA + VALU mixes, on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_throughput valu_throughput.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 32 independent instances per loop iteration, destination registers v[8..39], sources v[0..7]
#define DEFINE_KERNEL(NAME, ASM_LINE)                                                             \
    __global__ void __launch_bounds__(64) k_##NAME(float* out, int iters, long long* clk) {       \
        const long long c0 = clock64(), w0 = wall_clock64();                                                           \
        for (int it = 0; it < iters; ++it) {                                                      \
            asm volatile(ASM_LINE(8) ASM_LINE(9) ASM_LINE(10) ASM_LINE(11) ASM_LINE(12) ASM_LINE(13) ASM_LINE(14) ASM_LINE(15) \
                         ASM_LINE(16) ASM_LINE(17) ASM_LINE(18) ASM_LINE(19) ASM_LINE(20) ASM_LINE(21) ASM_LINE(22) ASM_LINE(23) \
                         ASM_LINE(24) ASM_LINE(25) ASM_LINE(26) ASM_LINE(27) ASM_LINE(28) ASM_LINE(29) ASM_LINE(30) ASM_LINE(31) \
                         ASM_LINE(32) ASM_LINE(33) ASM_LINE(34) ASM_LINE(35) ASM_LINE(36) ASM_LINE(37) ASM_LINE(38) ASM_LINE(39) \
                         ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", \
                         "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", \
                         "v38", "v39", "v40", "vcc");                                             \
        }                                                                                         \
        const long long c1 = clock64();                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }                                \
        out[threadIdx.x] = 0;                                                                     \
    }

#define L_FMA(d) "v_fma_f32 v" #d ", v0, v1, v2\n"
#define L_MUL(d) "v_mul_f32 v" #d ", v0, v1\n"
#define L_PKFMA(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[0:1], v[2:3], v[4:5]\n"
#define L_PKMUL(d) "v_pk_mul_f32 v[" #d ":" #d "+1], v[0:1], v[2:3]\n"
#define L_CVTPK(d) "v_cvt_pk_f16_f32 v" #d ", v0, v1\n"
#define L_CVTPKC(d) "v_cvt_pk_f16_f32 v" #d ", v0, v1 clamp\n"
#define L_CVTRTZ(d) "v_cvt_pkrtz_f16_f32 v" #d ", v0, v1\n"
#define L_CVT1(d) "v_cvt_f16_f32 v" #d ", v0\n"
#define L_COS(d) "v_cos_f32 v" #d ", v0\n"
#define L_EXP(d) "v_exp_f32 v" #d ", v0\n"
#define L_RCP(d) "v_rcp_f32 v" #d ", v0\n"
#define L_FRACT(d) "v_fract_f32 v" #d ", v0\n"
#define L_DOT2(d) "v_dot2_f32_f16 v" #d ", v0, v1, v2\n"
#define L_DOT2C(d) "v_dot2c_f32_f16 v" #d ", v0, v1\n"
#define L_PKMAXH(d) "v_pk_max_f16 v" #d ", v0, v1\n"
#define L_PKFMAH(d) "v_pk_fma_f16 v" #d ", v0, v1, v2\n"
#define L_CNDMASK(d) "v_cndmask_b32 v" #d ", v0, v1, vcc\n"
#define L_MOV(d) "v_mov_b32 v" #d ", v0\n"
#define L_ADDU(d) "v_add_u32 v" #d ", v0, v1\n"
#define L_LSHLADD(d) "v_lshl_add_u32 v" #d ", v0, 2, v1\n"
#define L_MAX(d) "v_max_f32 v" #d ", v0, v1\n"
#define L_MED3(d) "v_med3_f32 v" #d ", v0, v1, v2\n"
#define L_PERM(d) "v_perm_b32 v" #d ", v0, v1, v2\n"
#define L_MULLO(d) "v_mul_lo_u32 v" #d ", v0, v1\n"
#define L_FLOOR(d) "v_floor_f32 v" #d ", v0\n"
#define L_CVTI(d) "v_cvt_i32_f32 v" #d ", v0\n"
#define L_PERMLANE(d) "v_permlane32_swap_b32 v" #d ", v40\n"

DEFINE_KERNEL(fma, L_FMA)
DEFINE_KERNEL(mul, L_MUL)
DEFINE_KERNEL(cvtpk, L_CVTPK)
DEFINE_KERNEL(cvtpkc, L_CVTPKC)
DEFINE_KERNEL(cvtrtz, L_CVTRTZ)
DEFINE_KERNEL(cvt1, L_CVT1)
DEFINE_KERNEL(cos, L_COS)
DEFINE_KERNEL(exp, L_EXP)
DEFINE_KERNEL(rcp, L_RCP)
DEFINE_KERNEL(fract, L_FRACT)
DEFINE_KERNEL(dot2, L_DOT2)
DEFINE_KERNEL(dot2c, L_DOT2C)
DEFINE_KERNEL(pkmaxh, L_PKMAXH)
DEFINE_KERNEL(pkfmah, L_PKFMAH)
DEFINE_KERNEL(cndmask, L_CNDMASK)
DEFINE_KERNEL(mov, L_MOV)
DEFINE_KERNEL(addu, L_ADDU)
DEFINE_KERNEL(lshladd, L_LSHLADD)
DEFINE_KERNEL(maxf, L_MAX)
DEFINE_KERNEL(med3, L_MED3)
DEFINE_KERNEL(perm, L_PERM)
DEFINE_KERNEL(mullo, L_MULLO)
DEFINE_KERNEL(floorf, L_FLOOR)
DEFINE_KERNEL(cvti, L_CVTI)
DEFINE_KERNEL(permlane, L_PERMLANE)

// even destinations only for 64-bit results
#define DEFINE_KERNEL64(NAME, ASM_LINE)                                                           \
    __global__ void __launch_bounds__(64) k_##NAME(float* out, int iters, long long* clk) {       \
        const long long c0 = clock64(), w0 = wall_clock64();                                                           \
        for (int it = 0; it < iters; ++it) {                                                      \
            asm volatile(ASM_LINE(8) ASM_LINE(10) ASM_LINE(12) ASM_LINE(14) ASM_LINE(16) ASM_LINE(18) ASM_LINE(20) ASM_LINE(22) \
                         ASM_LINE(24) ASM_LINE(26) ASM_LINE(28) ASM_LINE(30) ASM_LINE(32) ASM_LINE(34) ASM_LINE(36) ASM_LINE(38) \
                         ASM_LINE(8) ASM_LINE(10) ASM_LINE(12) ASM_LINE(14) ASM_LINE(16) ASM_LINE(18) ASM_LINE(20) ASM_LINE(22) \
                         ASM_LINE(24) ASM_LINE(26) ASM_LINE(28) ASM_LINE(30) ASM_LINE(32) ASM_LINE(34) ASM_LINE(36) ASM_LINE(38) \
                         ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", \
                         "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", \
                         "v38", "v39");                                                           \
        }                                                                                         \
        const long long c1 = clock64();                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }                                \
        out[threadIdx.x] = 0;                                                                     \
    }
DEFINE_KERNEL64(pkfma, L_PKFMA)
DEFINE_KERNEL64(pkmul, L_PKMUL)

// ---- MFMA kinds and MFMA + VALU mixes (raw asm: no compiler-inserted wait states; results are garbage by design) ----
// accumulators v[64:79], v[80:95]; A/B in v[0:3], v[4:7]
#define MFMA_A "v_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\n"
#define MFMA_B "v_mfma_f32_32x32x16_f16 v[80:95], v[0:3], v[4:7], v[80:95]\n"
#define MFMA_A0 "v_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], 0\n"
#define MFMA_B0 "v_mfma_f32_32x32x16_f16 v[80:95], v[0:3], v[4:7], 0\n"
#define MFMA16_A "v_mfma_f32_16x16x32_f16 v[64:67], v[0:3], v[4:7], v[64:67]\n"
#define MFMA16_B "v_mfma_f32_16x16x32_f16 v[80:83], v[0:3], v[4:7], v[80:83]\n"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
             "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
             "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"

#define DEFINE_MIX(NAME, BODY)                                                                    \
    __global__ void __launch_bounds__(64) m_##NAME(float* out, int iters, long long* clk) {       \
        const long long c0 = clock64(), w0 = wall_clock64();                                                           \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY ::: CLOB); }                       \
        const long long c1 = clock64();                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }                                \
        out[threadIdx.x] = 0;                                                                     \
    }
// each body: 4 MFMAs (2 chains) + VALU in between
DEFINE_MIX(mfma, MFMA_A MFMA_B MFMA_A MFMA_B)
DEFINE_MIX(mfma_c0, MFMA_A0 MFMA_B0 MFMA_A0 MFMA_B0)
DEFINE_MIX(mfma16, MFMA16_A MFMA16_B MFMA16_A MFMA16_B)
#define V4_FMA L_FMA(8) L_FMA(9) L_FMA(10) L_FMA(11)
#define V4_CVT L_CVTPK(8) L_CVTPK(9) L_CVTPK(10) L_CVTPK(11)
// converts reading the OTHER chain's accumulator (completed one MFMA earlier)
#define V4_CVT_B "v_cvt_pk_f16_f32 v8, v80, v81\nv_cvt_pk_f16_f32 v9, v82, v83\nv_cvt_pk_f16_f32 v10, v84, v85\nv_cvt_pk_f16_f32 v11, v86, v87\n"
#define V4_CVT_A "v_cvt_pk_f16_f32 v12, v64, v65\nv_cvt_pk_f16_f32 v13, v66, v67\nv_cvt_pk_f16_f32 v14, v68, v69\nv_cvt_pk_f16_f32 v15, v70, v71\n"
#define V4_COS L_COS(8) L_COS(9) L_COS(10) L_COS(11)
DEFINE_MIX(mfma_fma4, MFMA_A V4_FMA MFMA_B V4_FMA MFMA_A V4_FMA MFMA_B V4_FMA)
DEFINE_MIX(mfma_fma8, MFMA_A V4_FMA V4_FMA MFMA_B V4_FMA V4_FMA MFMA_A V4_FMA V4_FMA MFMA_B V4_FMA V4_FMA)
DEFINE_MIX(mfma_cvt4, MFMA_A V4_CVT MFMA_B V4_CVT MFMA_A V4_CVT MFMA_B V4_CVT)
DEFINE_MIX(mfma_cvt4dep, MFMA_A V4_CVT_B MFMA_B V4_CVT_A MFMA_A V4_CVT_B MFMA_B V4_CVT_A)
DEFINE_MIX(mfma_cos4, MFMA_A V4_COS MFMA_B V4_COS MFMA_A V4_COS MFMA_B V4_COS)
DEFINE_MIX(mfma16_fma4, MFMA16_A V4_FMA MFMA16_B V4_FMA MFMA16_A V4_FMA MFMA16_B V4_FMA)

static int g_wavesPerSimd = 1;
// SIMD time per instruction: event-timed kernel duration x measured shader clock / (iterations x waves per SIMD)
template <class K>
double timeit(K kernel, float* out, long long* clk, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kernel<<<1024 * g_wavesPerSimd, 64>>>(out, 10, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kernel<<<1024 * g_wavesPerSimd, 64>>>(out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = double(h[0]) / (double(h[1]) * 10.0);
    return double(ms) * 1e6 * ghz / iters / g_wavesPerSimd;
}

int main(int argc, char** argv) {
    float* out; long long* clk;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&clk, 64);
    const int it = 4000;
    if (argc > 1) g_wavesPerSimd = atoi(argv[1]);
    printf("== %d wave(s) per SIMD (cycles are SIMD cycles per instruction: wave time / waves)\n", g_wavesPerSimd);
#define RUN(NAME) printf("  %-24s %5.2f cycles per instruction\n", #NAME, timeit(k_##NAME, out, clk, it) / 32);
    printf("VALU issue cost, one wave per SIMD, 32 independent instructions per iteration:\n");
    RUN(fma) RUN(mul) RUN(pkfma) RUN(pkmul) RUN(cvtpk) RUN(cvtpkc) RUN(cvtrtz) RUN(cvt1) RUN(cos) RUN(exp) RUN(rcp) RUN(fract) RUN(dot2) RUN(dot2c)
    RUN(pkmaxh) RUN(pkfmah) RUN(cndmask) RUN(mov) RUN(addu) RUN(lshladd) RUN(maxf) RUN(med3) RUN(perm) RUN(mullo) RUN(floorf) RUN(cvti) RUN(permlane)
#define RUNM(NAME, NOTE) printf("  %-24s %6.1f cycles per MFMA   %s\n", #NAME, timeit(m_##NAME, out, clk, it) / 4, NOTE);
    printf("MFMA (+ VALU between MFMAs), one wave per SIMD:\n");
    RUNM(mfma, "32x32x16 f16, C = D")
    RUNM(mfma_c0, "32x32x16 f16, C = 0")
    RUNM(mfma16, "16x16x32 f16")
    RUNM(mfma_fma4, "+ 4 v_fma per MFMA")
    RUNM(mfma_fma8, "+ 8 v_fma per MFMA")
    RUNM(mfma_cvt4, "+ 4 v_cvt_pk (independent registers) per MFMA")
    RUNM(mfma_cvt4dep, "+ 4 v_cvt_pk reading the other chain's accumulator per MFMA")
    RUNM(mfma_cos4, "+ 4 v_cos per MFMA")
    RUNM(mfma16_fma4, "16x16x32 + 4 v_fma per MFMA")
    return 0;
}
