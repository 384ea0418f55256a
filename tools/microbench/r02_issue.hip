// Round-2 microbenchmarks for the SRN kernels on gfx950 (MI355X): what a SIMD does with the instruction mix of a wave step.
//   part 1: issue cost of the convert / packed instructions at 1, 2 and 4 waves per SIMD
//   part 2: how many VALU instructions of the SAME wave hide behind one MFMA (32x32x16 and 16x16x32), by instruction kind
//   part 3: two waves on one SIMD, A = the MFMA chain of a step (MFMA + 4 converts), B = the vector phase of a step
//           (packed fp32 rotations), with wave priorities: who gets the issue slots
//   part 4: the whole wave step of the 32x4 kernel as an instruction skeleton (62-instruction vector phase, 12 + 4 MFMAs with
//           their converts), 2 waves per SIMD running the same program, by priority policy / start offset / convert kind
// build: hipcc --offload-arch=gfx950 -O3 -o r02_issue r02_issue.hip ; results are garbage by design (raw asm, no data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define CLOB_V "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
               "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "vcc"
#define CLOB_M "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
               "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"

// ---------------------------------------------------------------------------------------------------------------------------
// part 1: 32 independent instances per iteration
#define X32(L) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) \
               L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31) L(32) L(33) L(34) L(35) L(36) L(37) L(38) L(39)
#define X16E(L) L(8) L(10) L(12) L(14) L(16) L(18) L(20) L(22) L(24) L(26) L(28) L(30) L(32) L(34) L(36) L(38)
#define L_FMA(d) "v_fma_f32 v" #d ", v0, v1, v2\n"
#define L_CVTPK(d) "v_cvt_pk_f16_f32 v" #d ", v0, v1\n"
#define L_CVTRTZ(d) "v_cvt_pkrtz_f16_f32 v" #d ", v0, v1\n"
#define L_CVTRTZC(d) "v_cvt_pkrtz_f16_f32_e64 v" #d ", v0, v1 clamp\n"
#define L_CVT1(d) "v_cvt_f16_f32 v" #d ", v0\n"
#define L_PKMULH(d) "v_pk_mul_f16 v" #d ", v0, v1\n"
#define L_PKFMAH(d) "v_pk_fma_f16 v" #d ", v0, v1, v2\n"
#define L_PKADDH(d) "v_pk_add_f16 v" #d ", v0, v1\n"
#define L_PKFMA(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[0:1], v[2:3], v[4:5]\n"
#define L_PKMUL(d) "v_pk_mul_f32 v[" #d ":" #d "+1], v[0:1], v[2:3]\n"
#define L_PKADD(d) "v_pk_add_f32 v[" #d ":" #d "+1], v[0:1], v[2:3]\n"
#define L_COS(d) "v_cos_f32 v" #d ", v0\n"
#define L_DOT2(d) "v_dot2_f32_f16 v" #d ", v0, v1, v2\n"
#define L_SUBCL(d) "v_sub_f32_e64 v" #d ", 1.0, |v0| clamp\n"
#define L_PERM(d) "v_perm_b32 v" #d ", v0, v1, v2\n"

#define DEF1(NAME, BODY)                                                                                   \
    __global__ void __launch_bounds__(64) k_##NAME(float* out, int iters, long long* clk) {                \
        const long long c0 = clock64(), w0 = wall_clock64();                                               \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY ::: CLOB_V); }                              \
        const long long c1 = clock64();                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }       \
        out[threadIdx.x] = 0;                                                                              \
    }
DEF1(fma, X32(L_FMA)) DEF1(cvtpk, X32(L_CVTPK)) DEF1(cvtrtz, X32(L_CVTRTZ)) DEF1(cvtrtzc, X32(L_CVTRTZC)) DEF1(cvt1, X32(L_CVT1))
DEF1(pkmulh, X32(L_PKMULH)) DEF1(pkfmah, X32(L_PKFMAH)) DEF1(pkaddh, X32(L_PKADDH)) DEF1(cos, X32(L_COS)) DEF1(dot2, X32(L_DOT2))
DEF1(subcl, X32(L_SUBCL)) DEF1(perm, X32(L_PERM))
DEF1(pkfma, X16E(L_PKFMA) X16E(L_PKFMA)) DEF1(pkmul, X16E(L_PKMUL) X16E(L_PKMUL)) DEF1(pkadd, X16E(L_PKADD) X16E(L_PKADD))

// ---------------------------------------------------------------------------------------------------------------------------
// part 2: MFMA + N VALU of the same wave.  Accumulators v[64:79] / v[80:95]; converts read the OTHER chain's accumulator.
#define MA "v_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\n"
#define MB "v_mfma_f32_32x32x16_f16 v[80:95], v[0:3], v[4:7], v[80:95]\n"
#define M16A "v_mfma_f32_16x16x32_f16 v[64:67], v[0:3], v[4:7], v[64:67]\n"
#define M16B "v_mfma_f32_16x16x32_f16 v[80:83], v[0:3], v[4:7], v[80:83]\n"
#define CV(op, d, s) op " v" #d ", v" #s ", v" #s "+1\n"
#define PK "v_cvt_pk_f16_f32"
#define RZ "v_cvt_pkrtz_f16_f32"
// n converts of accumulator base s into v8..
#define CV2(op, s) "" op " v8, v[" #s "], v[" #s "+1]\n" op " v9, v[" #s "+2], v[" #s "+3]\n"
#define CV4(op, s) CV2(op, s) "" op " v10, v[" #s "+4], v[" #s "+5]\n" op " v11, v[" #s "+6], v[" #s "+7]\n"
#define CV6(op, s) CV4(op, s) "" op " v12, v[" #s "+8], v[" #s "+9]\n" op " v13, v[" #s "+10], v[" #s "+11]\n"
#define CV8(op, s) CV6(op, s) "" op " v14, v[" #s "+12], v[" #s "+13]\n" op " v15, v[" #s "+14], v[" #s "+15]\n"
#define F2 L_FMA(16) L_FMA(17)
#define F4 F2 L_FMA(18) L_FMA(19)
#define F6 F4 L_FMA(20) L_FMA(21)
#define F8 F6 L_FMA(22) L_FMA(23)
#define P2 L_PKFMA(24) L_PKFMA(26)
#define P4 P2 L_PKFMA(28) L_PKFMA(30)
#define P6 P4 L_PKFMA(32) L_PKFMA(34)
#define C2 L_COS(36) L_COS(37)
#define C4 C2 L_COS(38) L_COS(39)
#define DEF2(NAME, BODY)                                                                                   \
    __global__ void __launch_bounds__(64) m_##NAME(float* out, int iters, long long* clk) {                \
        const long long c0 = clock64(), w0 = wall_clock64();                                               \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY ::: CLOB_V, CLOB_M); }                      \
        const long long c1 = clock64();                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }       \
        out[threadIdx.x] = 0;                                                                              \
    }
DEF2(bare, MA MB MA MB)
DEF2(pk4, MA CV4(PK, 80) MB CV4(PK, 64) MA CV4(PK, 80) MB CV4(PK, 64))
DEF2(pk6, MA CV6(PK, 80) MB CV6(PK, 64) MA CV6(PK, 80) MB CV6(PK, 64))
DEF2(pk8, MA CV8(PK, 80) MB CV8(PK, 64) MA CV8(PK, 80) MB CV8(PK, 64))
DEF2(rz4, MA CV4(RZ, 80) MB CV4(RZ, 64) MA CV4(RZ, 80) MB CV4(RZ, 64))
DEF2(rz6, MA CV6(RZ, 80) MB CV6(RZ, 64) MA CV6(RZ, 80) MB CV6(RZ, 64))
DEF2(rz8, MA CV8(RZ, 80) MB CV8(RZ, 64) MA CV8(RZ, 80) MB CV8(RZ, 64))
DEF2(rz8f2, MA CV8(RZ, 80) F2 MB CV8(RZ, 64) F2 MA CV8(RZ, 80) F2 MB CV8(RZ, 64) F2)
DEF2(pk4f2, MA CV4(PK, 80) F2 MB CV4(PK, 64) F2 MA CV4(PK, 80) F2 MB CV4(PK, 64) F2)
DEF2(pk4f4, MA CV4(PK, 80) F4 MB CV4(PK, 64) F4 MA CV4(PK, 80) F4 MB CV4(PK, 64) F4)
DEF2(pk4p2, MA CV4(PK, 80) P2 MB CV4(PK, 64) P2 MA CV4(PK, 80) P2 MB CV4(PK, 64) P2)
DEF2(rz4p2, MA CV4(RZ, 80) P2 MB CV4(RZ, 64) P2 MA CV4(RZ, 80) P2 MB CV4(RZ, 64) P2)
DEF2(rz4p4, MA CV4(RZ, 80) P4 MB CV4(RZ, 64) P4 MA CV4(RZ, 80) P4 MB CV4(RZ, 64) P4)
DEF2(f6, MA F6 MB F6 MA F6 MB F6)
DEF2(f8, MA F8 MB F8 MA F8 MB F8)
DEF2(p4, MA P4 MB P4 MA P4 MB P4)
DEF2(p6, MA P6 MB P6 MA P6 MB P6)
DEF2(c2pk2, MA C2 CV2(PK, 80) MB C2 CV2(PK, 64) MA C2 CV2(PK, 80) MB C2 CV2(PK, 64))
DEF2(c4, MA C4 MB C4 MA C4 MB C4)
DEF2(m16bare, M16A M16B M16A M16B)
DEF2(m16pk2, M16A CV2(PK, 80) M16B CV2(PK, 64) M16A CV2(PK, 80) M16B CV2(PK, 64))
DEF2(m16rz2, M16A CV2(RZ, 80) M16B CV2(RZ, 64) M16A CV2(RZ, 80) M16B CV2(RZ, 64))
DEF2(m16rz4, M16A CV4(RZ, 80) M16B CV4(RZ, 64) M16A CV4(RZ, 80) M16B CV4(RZ, 64))
DEF2(m16pk4, M16A CV4(PK, 80) M16B CV4(PK, 64) M16A CV4(PK, 80) M16B CV4(PK, 64))

static int g_wavesPerSimd = 1;
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
    return double(ms) * 1e6 * ghz / iters / g_wavesPerSimd;  // SIMD cycles per loop body
}

// ---------------------------------------------------------------------------------------------------------------------------
// part 3: two roles on one SIMD (waves w and w + 4 of a 512-thread workgroup share a SIMD)
//   A: the MFMA chain of a step: { MFMA ; 4 converts of the other accumulator } x 8 per iteration
//   B: the vector phase: 32 independent packed fp32 FMAs per iteration, runs until all A waves are done
template <int PRIO_A, int PRIO_B, int KIND_B>
__global__ void __launch_bounds__(512) two_roles(float* out, int iters, long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ int doneA;
    if (threadIdx.x == 0) doneA = 0;
    __syncthreads();
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(PRIO_A);
        const long long c0 = clock64();
        for (int it = 0; it < iters; ++it) {
            asm volatile(MA CV4(PK, 80) MB CV4(PK, 64) MA CV4(PK, 80) MB CV4(PK, 64) MA CV4(PK, 80) MB CV4(PK, 64) MA CV4(PK, 80) MB CV4(PK, 64) ::: CLOB_V, CLOB_M);
        }
        const long long c1 = clock64();
        if (lane == 0) { atomicAdd(&doneA, 1); if (blockIdx.x == 0 && wave == 0) clk[0] = c1 - c0; }
    } else {
        __builtin_amdgcn_s_setprio(PRIO_B);
        long long n = 0;
        while (__builtin_amdgcn_readfirstlane(*(volatile int*)&doneA) < 4) {
            if constexpr (KIND_B == 0) asm volatile(X16E(L_PKFMA) X16E(L_PKFMA) ::: CLOB_V);
            else if constexpr (KIND_B == 1) asm volatile(X32(L_FMA) ::: CLOB_V);
            else asm volatile(X32(L_COS) ::: CLOB_V);
            ++n;
        }
        if (blockIdx.x == 0 && wave == 4 && lane == 0) clk[1] = n;
    }
    out[blockIdx.x * 512 + threadIdx.x] = 0;
}
template <int PA, int PB, int KB>
void run3(float* out, long long* clk) {
    const int iters = 2000;
    two_roles<PA, PB, KB><<<256, 512>>>(out, 10, clk);
    (void)hipDeviceSynchronize();
    two_roles<PA, PB, KB><<<256, 512>>>(out, iters, clk);
    (void)hipDeviceSynchronize();
    long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("  prio A %d / B %d, B = %-9s: A %6.1f cycles per MFMA (+4 cvt); B issued %5.2f instructions per MFMA of A\n", PA, PB,
           KB == 0 ? "pk_fma" : (KB == 1 ? "v_fma" : "v_cos"), double(h[0]) / iters / 8, double(h[1]) * 32 / (double(iters) * 8));
}

// ---------------------------------------------------------------------------------------------------------------------------
// part 4: skeleton of the wave step of render_small_kernel (32x4 ReLU): vector phase = 16 feature converts + 30 packed
// rotations + 16 scalar fp32; chain = 12 x (32x32x16 MFMA + 4 converts) + 2 x (16x16x32 + 4 converts) + 2 x 16x16x32.
// All waves run the same program; POLICY: 0 no priorities, 1 vector phase prio 3 / chain 0 (shipped), 2 reversed,
// 3 = policy 1 + waves 4..7 start half a step later (they run one chain first); CVT: 0 v_cvt_pk_f16_f32, 1 v_cvt_pkrtz
#define VPHASE(op) CV8(op, 64) CV8(op, 80) X16E(L_PKFMA) X16E(L_PKMUL) F8 F8
#define SLOT(op) MA CV4(op, 80) MA CV4(op, 80) MB CV4(op, 64) MB CV4(op, 64)
#define CHAIN(op) MA MA SLOT(op) SLOT(op) MB CV4(op, 64) MB CV4(op, 64) M16A CV4(op, 80) M16A CV4(op, 80) M16B M16B
template <int POLICY, int CVT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) step_skeleton(float* out, int iters, long long* clk) {
    const int wave = threadIdx.x >> 6;
    const long long c0 = clock64(), w0 = wall_clock64();
    if (POLICY == 3 && wave >= 4) {
        if constexpr (CVT == 0) asm volatile(CHAIN(PK) ::: CLOB_V, CLOB_M); else asm volatile(CHAIN(RZ) ::: CLOB_V, CLOB_M);
    }
    for (int it = 0; it < iters; ++it) {
        if (POLICY == 1 || POLICY == 3) __builtin_amdgcn_s_setprio(3);
        if (POLICY == 2) __builtin_amdgcn_s_setprio(0);
        if constexpr (CVT == 0) asm volatile(VPHASE(PK) ::: CLOB_V, CLOB_M); else asm volatile(VPHASE(RZ) ::: CLOB_V, CLOB_M);
        if (POLICY == 1 || POLICY == 3) __builtin_amdgcn_s_setprio(0);
        if (POLICY == 2) __builtin_amdgcn_s_setprio(3);
        if constexpr (CVT == 0) asm volatile(CHAIN(PK) ::: CLOB_V, CLOB_M); else asm volatile(CHAIN(RZ) ::: CLOB_V, CLOB_M);
    }
    const long long c1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = 0;
}
template <int POLICY, int CVT, int WAVES>
void run4(float* out, long long* clk, const char* note) {
    const int iters = 3000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    step_skeleton<POLICY, CVT, WAVES><<<256, WAVES * 64>>>(out, 10, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    step_skeleton<POLICY, CVT, WAVES><<<256, WAVES * 64>>>(out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = double(h[0]) / (double(h[1]) * 10.0);
    const double simdCyclesPerStep = double(ms) * 1e6 * ghz / iters / (WAVES / 4);
    printf("  %d waves/SIMD, policy %d, %s: %7.1f SIMD cycles per wave step (%.2f GHz) -> %.1f Gsamples/s equivalent   %s\n", WAVES / 4, POLICY,
           CVT ? "pkrtz " : "cvt_pk", simdCyclesPerStep, ghz, 64.0 * 1024.0 * ghz / simdCyclesPerStep, note);
}

// ---------------------------------------------------------------------------------------------------------------------------
// part 5: do transcendentals of one wave run beside plain VALU work of the other wave of the SIMD?  (waves w / w + 4)
//   KA / KB: 0 = 32 v_cos, 1 = 32 v_fma, 2 = 32 v_cvt_pk, 3 = 16 v_pk_fma_f32; both waves loop a fixed number of iterations
template <int KA, int KB>
__global__ void __launch_bounds__(512) pair_kinds(float* out, int iters, long long* clk) {
    const int wave = threadIdx.x >> 6;
    const long long c0 = clock64();
    auto body = [&](auto kind) {
        constexpr int K = decltype(kind)::value;
        for (int it = 0; it < iters; ++it) {
            if constexpr (K == 0) asm volatile(X32(L_COS) ::: CLOB_V);
            else if constexpr (K == 1) asm volatile(X32(L_FMA) ::: CLOB_V);
            else if constexpr (K == 2) asm volatile(X32(L_CVTPK) ::: CLOB_V);
            else asm volatile(X16E(L_PKFMA) ::: CLOB_V);
        }
    };
    if (wave < 4) body(std::integral_constant<int, KA>{}); else body(std::integral_constant<int, KB>{});
    const long long c1 = clock64();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) clk[wave >> 2] = c1 - c0;
    out[blockIdx.x * 512 + threadIdx.x] = 0;
}
template <int KA, int KB>
void run5(float* out, long long* clk, const char* note) {
    const int iters = 4000;
    pair_kinds<KA, KB><<<256, 512>>>(out, 10, clk);
    (void)hipDeviceSynchronize();
    pair_kinds<KA, KB><<<256, 512>>>(out, iters, clk);
    (void)hipDeviceSynchronize();
    long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("  %-34s wave A %6.1f cycles per iteration, wave B %6.1f\n", note, double(h[0]) / iters, double(h[1]) / iters);
}

// ---------------------------------------------------------------------------------------------------------------------------
// part 6: wave-step skeletons of the other two kernels, 2 waves per SIMD running the same program
//   SnakeAlt 32x4 (register-resident kernel): the chain carries, per MFMA, 4 x { pk_mul, 2 cos, pk_add, [pk_fma], cvt }
//   (FOLD: the 1/(2p) scale folded into the next layer's weights: no pk_fma)
#define SA_PAIR(d) "v_pk_mul_f32 v[" #d ":" #d "+1], v[64:65], v[2:3]\nv_cos_f32 v[" #d "], v[" #d "]\nv_cos_f32 v[" #d "+1], v[" #d "+1]\nv_pk_add_f32 v[" #d ":" #d "+1], v[64:65], v[" #d ":" #d "+1]\n"
#define SA_FMA(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #d ":" #d "+1], v[2:3], v[2:3]\n"
#define SA_CVT(d) "v_cvt_pk_f16_f32 v[" #d "], v[" #d "], v[" #d "+1]\n"
#define SA4(F) SA_PAIR(8) F(8) SA_CVT(8) SA_PAIR(10) F(10) SA_CVT(10) SA_PAIR(12) F(12) SA_CVT(12) SA_PAIR(14) F(14) SA_CVT(14)
#define NOFMA(d) ""
#define SA_SLOT(F) MA SA4(F) MA SA4(F) MB SA4(F) MB SA4(F)
#define SA_CHAIN(F) MA MA SA_SLOT(F) SA_SLOT(F) MB SA4(F) MB SA4(F) M16A SA4(F) M16A SA4(F) M16B M16B
template <int FOLD>
__global__ void __launch_bounds__(512) snakealt_skeleton(float* out, int iters, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(VPHASE(PK) ::: CLOB_V, CLOB_M);
        if constexpr (FOLD) asm volatile(SA_CHAIN(NOFMA) ::: CLOB_V, CLOB_M); else asm volatile(SA_CHAIN(SA_FMA) ::: CLOB_V, CLOB_M);
    }
    const long long c1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }
    out[blockIdx.x * 512 + threadIdx.x] = 0;
}
//   32x4 + 16-channel latent grid (LDS kernel, no rotation): per wave step 2 phase MFMAs, 30 v_cos, ~16 + 3 converts, 3 + 8 lane
//   swaps, ~40 scalar ops of tap arithmetic, 64 v_dot2 + 8 converts (the loads are left out), then the layer chain (18 MFMAs with
//   48 converts) and the 16-instruction tail.  BLOCK = 1: the trilinear fetch as 6 more MFMAs on per-cell-block matrices: 3 swaps,
//   ~2 x (15 scalar + 27 packed) for the vertex weights instead of taps / swaps / dot products.
#define X30COS L_COS(8) L_COS(9) L_COS(10) L_COS(11) L_COS(12) L_COS(13) L_COS(14) L_COS(15) L_COS(16) L_COS(17) L_COS(18) L_COS(19) L_COS(20) L_COS(21) L_COS(22) \
               L_COS(23) L_COS(24) L_COS(25) L_COS(26) L_COS(27) L_COS(28) L_COS(29) L_COS(30) L_COS(31) L_COS(32) L_COS(33) L_COS(34) L_COS(35) L_COS(36) L_COS(37)
#define L_SWAP(d) "v_permlane32_swap_b32 v" #d ", v40\n"
#define SWAP3 L_SWAP(8) L_SWAP(9) L_SWAP(10)
#define SWAP8 SWAP3 L_SWAP(11) L_SWAP(12) L_SWAP(13) L_SWAP(14) L_SWAP(15)
#define F16x F8 F8
#define GRID_GATHER SWAP8 F16x F16x F8 X32(L_DOT2) X32(L_DOT2) CV8(PK, 64)
#define GRID_BLOCK F16x F8 F4 F2 X16E(L_PKMUL) X16E(L_PKMUL) CV8(PK, 64) CV8(PK, 80) CV8(PK, 64) L_PKMUL(8) L_PKMUL(10) L_PKMUL(12) MA MB MA MB MA MB
#define LAYERS MA MA SLOT(PK) SLOT(PK) MB CV4(PK, 64) MB CV4(PK, 64) M16A CV4(PK, 80) M16A CV4(PK, 80) M16B M16B
template <int BLOCK>
__global__ void __launch_bounds__(768) grid_skeleton(float* out, int iters, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(SWAP3 CV2(PK, 64) L_CVTPK(10) MA MB X30COS CV8(PK, 64) CV8(PK, 80) ::: CLOB_V, CLOB_M);
        if constexpr (BLOCK) asm volatile(GRID_BLOCK ::: CLOB_V, CLOB_M); else asm volatile(GRID_GATHER MA MB ::: CLOB_V, CLOB_M);
        asm volatile(LAYERS F8 F8 ::: CLOB_V, CLOB_M);
    }
    const long long c1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = wall_clock64() - w0; }
    out[blockIdx.x * 768 + threadIdx.x] = 0;
}
template <class K>
void run6(K kernel, int threads, float* out, long long* clk, const char* note) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kernel<<<256, threads>>>(out, 10, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kernel<<<256, threads>>>(out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = double(h[0]) / (double(h[1]) * 10.0);
    const double cyc = double(ms) * 1e6 * ghz / iters / (threads / 256);
    printf("  %-58s %7.1f SIMD cycles per wave step (%.2f GHz): %.1f Gsamples/s at this clock, %.1f at 2.04 GHz\n", note, cyc, ghz, 64.0 * 1024.0 * ghz / cyc,
           64.0 * 1024.0 * 2.04 / cyc);
}

int main(int argc, char** argv) {
    float* out; long long* clk;
    (void)hipMalloc(&out, 1024 * 512 * 4);
    (void)hipMalloc(&clk, 64);
    const int it = 4000;
    for (int w : {1, 2, 4}) {
        g_wavesPerSimd = w;
        printf("== part 1: %d wave(s) per SIMD, SIMD cycles per instruction\n", w);
#define RUN(NAME) printf("  %-10s %5.2f\n", #NAME, timeit(k_##NAME, out, clk, it) / 32);
        RUN(fma) RUN(cvtpk) RUN(cvtrtz) RUN(cvtrtzc) RUN(cvt1) RUN(pkmulh) RUN(pkfmah) RUN(pkaddh) RUN(pkfma) RUN(pkmul) RUN(pkadd) RUN(cos) RUN(dot2) RUN(subcl) RUN(perm)
    }
    for (int w : {1, 2}) {
        g_wavesPerSimd = w;
        printf("== part 2: %d wave(s) per SIMD (same program), SIMD cycles per MFMA\n", w);
#define RUNM(NAME, NOTE) printf("  %-10s %6.1f   %s\n", #NAME, timeit(m_##NAME, out, clk, it) / 4, NOTE);
        RUNM(bare, "32x32x16 alone")
        RUNM(pk4, "+4 cvt_pk") RUNM(pk6, "+6 cvt_pk") RUNM(pk8, "+8 cvt_pk")
        RUNM(rz4, "+4 pkrtz") RUNM(rz6, "+6 pkrtz") RUNM(rz8, "+8 pkrtz") RUNM(rz8f2, "+8 pkrtz +2 fma")
        RUNM(pk4f2, "+4 cvt_pk +2 fma") RUNM(pk4f4, "+4 cvt_pk +4 fma") RUNM(pk4p2, "+4 cvt_pk +2 pk_fma_f32") RUNM(rz4p2, "+4 pkrtz +2 pk_fma_f32")
        RUNM(rz4p4, "+4 pkrtz +4 pk_fma_f32")
        RUNM(f6, "+6 fma") RUNM(f8, "+8 fma") RUNM(p4, "+4 pk_fma_f32") RUNM(p6, "+6 pk_fma_f32") RUNM(c2pk2, "+2 cos +2 cvt_pk") RUNM(c4, "+4 cos")
        RUNM(m16bare, "16x16x32 alone") RUNM(m16pk2, "16x16x32 +2 cvt_pk") RUNM(m16rz2, "16x16x32 +2 pkrtz") RUNM(m16rz4, "16x16x32 +4 pkrtz")
        RUNM(m16pk4, "16x16x32 +4 cvt_pk")
    }
    printf("== part 3: two roles on one SIMD\n");
    run3<0, 0, 0>(out, clk); run3<0, 3, 0>(out, clk); run3<3, 0, 0>(out, clk); run3<0, 1, 0>(out, clk);
    run3<0, 0, 1>(out, clk); run3<0, 3, 1>(out, clk); run3<0, 0, 2>(out, clk); run3<0, 3, 2>(out, clk);
    printf("== part 4: wave-step skeleton of the 32x4 kernel (62 VALU vector phase + 16 MFMA chain with 56 converts)\n");
    run4<0, 0, 4>(out, clk, "one wave per SIMD");
    run4<0, 1, 4>(out, clk, "one wave per SIMD");
    run4<0, 0, 8>(out, clk, ""); run4<1, 0, 8>(out, clk, "shipped policy"); run4<2, 0, 8>(out, clk, ""); run4<3, 0, 8>(out, clk, "staggered start");
    run4<0, 1, 8>(out, clk, ""); run4<1, 1, 8>(out, clk, ""); run4<2, 1, 8>(out, clk, ""); run4<3, 1, 8>(out, clk, "staggered start");
    run4<0, 0, 12>(out, clk, "3 waves"); run4<1, 0, 12>(out, clk, "3 waves"); run4<1, 1, 12>(out, clk, "3 waves");
    printf("== part 5: two waves on one SIMD, each with one instruction kind (alone = the other wave idle)\n");
    run5<0, 0>(out, clk, "A cos, B cos"); run5<1, 1>(out, clk, "A fma, B fma"); run5<0, 1>(out, clk, "A cos, B fma"); run5<0, 2>(out, clk, "A cos, B cvt_pk");
    run5<0, 3>(out, clk, "A cos, B pk_fma_f32 (16 per iteration)"); run5<2, 1>(out, clk, "A cvt_pk, B fma"); run5<2, 2>(out, clk, "A cvt_pk, B cvt_pk");
    printf("== part 6: wave-step skeletons, 2 (SnakeAlt) / 3 (latent grid) waves per SIMD\n");
    run6(snakealt_skeleton<0>, 512, out, clk, "32x4 SnakeAlt, as shipped in r01");
    run6(snakealt_skeleton<1>, 512, out, clk, "32x4 SnakeAlt, 1/(2p) folded into the next layer");
    run6(grid_skeleton<0>, 768, out, clk, "32x4 + latent grid, gather + v_dot2 (loads left out)");
    run6(grid_skeleton<1>, 768, out, clk, "32x4 + latent grid, trilinear fetch as 6 MFMAs on cell-block matrices");
    return 0;
}
