// Round-4 reproducer, part 3 (profiles/r04/nondeterminism_r04.md): the read-after-write matrix of packed-fp32 instructions next to MFMA waves.
//   producer:  v_pk_fma_f32 v[20:21], %[p], %[k], %[c]        (writes the pair R = v[20:21]; inputs long settled)
//   <GAP independent plain VALU instructions>
//   consumer:  v_pk_mul_f32 v[10:11], R, %[q] op_sel:[A,B] ...   (R as src0)   or   v_pk_mul_f32 v[10:11], %[q], R op_sel:[A,B] ...  (R as src1)
// for the four selections [A,B] of the consumer's LOW pass (high pass: default) and gaps 0 .. 3; the result is compared with plain, spaced
// fp32 instructions.  Neighbours: waves of a second kernel on a second stream issuing MFMAs (two per SIMD).
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/r04_pk_raw_matrix.hip -o tools/microbench/bin/r04_pk_raw_matrix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) neighbour(long long ticks, unsigned* sink) {
    unsigned pad[56];
#pragma unroll
    for (int i = 0; i < 56; ++i) { pad[i] = threadIdx.x + i; asm volatile("" : "+v"(pad[i])); }
    const long long t0 = wall_clock64();  // 100 MHz
    floatx16 acc = {0};
    floatx4 acc4 = {0, 0, 0, 0};
    half8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    while (wall_clock64() - t0 < ticks) {
        if constexpr (KIND == 0) { __builtin_amdgcn_s_sleep(8); }
        if constexpr (KIND == 1) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));) }
        if constexpr (KIND == 2) { REP8(asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b));) }
    }
    unsigned r = __float_as_uint(acc[0] + acc[5] + acc4[0] + acc4[3]);
#pragma unroll
    for (int i = 0; i < 56; ++i) { asm volatile("" : "+v"(pad[i])); r ^= pad[i]; }
    if (r == 0xdeadbeefu) *sink = r;
}
static const char* kNeighbour[] = {"s_sleep", "mfma 32x32x16 f16", "mfma 16x16x32 f16"};

__device__ __forceinline__ float safe_mul(float a, float b) { float r; asm volatile("s_nop 3\n\t v_mul_f32 %0, %1, %2\n\t s_nop 3" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float safe_fma(float a, float b, float c) { float r; asm volatile("s_nop 3\n\t v_fma_f32 %0, %1, %2, %3\n\t s_nop 3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

#define GAPS0 ""
#define GAPS1 "v_mov_b32 v12, %[t]\n "
#define GAPS2 GAPS1 "v_mov_b32 v13, %[t]\n "
#define GAPS3 GAPS2 "v_mov_b32 v12, %[t]\n "
#define GAPS4 "s_nop 0\n "
#define GAPS5 "s_nop 1\n "
#define PRODUCER "s_nop 7\n v_pk_fma_f32 v[20:21], %[p], %[k], %[c]\n "
// PROD 0: the packed producer; 1: two plain v_fma_f32 (low, then high)
#define PRODUCER_PLAIN "s_nop 7\n v_fma_f32 v20, %[p0], %[k0], %[c0]\n v_fma_f32 v21, %[p1], %[k1], %[c1]\n "

template <int SRC, int SEL, int GAP, int PROD>
__global__ void __launch_bounds__(256) victim(const float* in, unsigned* bad, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float2_t p = {in[i & 4095] + 0.5f, in[(i + 1) & 4095] + 0.25f}, q = {in[(i + 2) & 4095] + 1.f, in[(i + 3) & 4095] + 2.f};
    const float2_t k = {1.25f, 0.75f}, c = {0.125f, 0.375f};
    float t = in[(i + 5) & 4095] + 0.75f;
    unsigned wl = 0, wh = 0;
    for (int it = 0; it < iters; ++it) {
        float2_t r, R, sc;
#define RUN(GAPTXT, CONS)                                                                                                             \
        if constexpr (PROD == 0) asm volatile(PRODUCER GAPTXT CONS "\n s_nop 7" : "=&{v[10:11]}"(r), "=&{v[20:21]}"(R), "=&{v[12:13]}"(sc)    \
                                              : [p] "v"(p), [q] "v"(q), [k] "v"(k), [c] "v"(c), [t] "v"(t));                                \
        else asm volatile(PRODUCER_PLAIN GAPTXT CONS "\n s_nop 7" : "=&{v[10:11]}"(r), "=&{v[20:21]}"(R), "=&{v[12:13]}"(sc)                \
                          : [p0] "v"(p[0]), [p1] "v"(p[1]), [k0] "v"(k[0]), [k1] "v"(k[1]), [c0] "v"(c[0]), [c1] "v"(c[1]), [q] "v"(q), [t] "v"(t));
#define RUNG(CONS)                                                                                                                    \
        if constexpr (GAP == 0) { RUN(GAPS0, CONS) } else if constexpr (GAP == 1) { RUN(GAPS1, CONS) } else if constexpr (GAP == 2) { RUN(GAPS2, CONS) } \
        else if constexpr (GAP == 3) { RUN(GAPS3, CONS) } else if constexpr (GAP == 4) { RUN(GAPS4, CONS) } else { RUN(GAPS5, CONS) }
#define CONS0(A, B) "v_pk_mul_f32 v[10:11], v[20:21], %[q] op_sel:[" #A "," #B "]"
#define CONS1(A, B) "v_pk_mul_f32 v[10:11], %[q], v[20:21] op_sel:[" #A "," #B "]"
        if constexpr (SRC == 0) {
            if constexpr (SEL == 0) { RUNG(CONS0(0, 0)) } else if constexpr (SEL == 1) { RUNG(CONS0(0, 1)) } else if constexpr (SEL == 2) { RUNG(CONS0(1, 0)) } else { RUNG(CONS0(1, 1)) }
        } else {
            if constexpr (SEL == 0) { RUNG(CONS1(0, 0)) } else if constexpr (SEL == 1) { RUNG(CONS1(0, 1)) } else if constexpr (SEL == 2) { RUNG(CONS1(1, 0)) } else { RUNG(CONS1(1, 1)) }
        }
        const float R0 = safe_fma(p[0], k[0], c[0]), R1 = safe_fma(p[1], k[1], c[1]);
        const int a = SEL >> 1, b = SEL & 1;
        float el, eh;
        if constexpr (SRC == 0) { el = safe_mul(a ? R1 : R0, q[b]); eh = safe_mul(R1, q[1]); }
        else { el = safe_mul(q[a], b ? R1 : R0); eh = safe_mul(q[1], R1); }
        wl += __float_as_uint(r[0]) != __float_as_uint(el);
        wh += __float_as_uint(r[1]) != __float_as_uint(eh);
        p = p * 0.9993f + float2_t{0.0011f, 0.0023f};
        q = q * 1.0002f - float2_t{0.0003f, 0.0001f};
        t = t * 0.9998f + 0.0002f;
    }
    const int qd = (threadIdx.x & 63) >> 4;
    if (wl) atomicAdd(&bad[qd * 2 + 0], wl);
    if (wh) atomicAdd(&bad[qd * 2 + 1], wh);
}

template <int SRC, int SEL, int PROD>
static void run_g(int gap, const float* in, unsigned* bad, int iters, hipStream_t s) {
    switch (gap) {
        case 0: hipLaunchKernelGGL((victim<SRC, SEL, 0, PROD>), dim3(1024), dim3(256), 0, s, in, bad, iters); break;
        case 1: hipLaunchKernelGGL((victim<SRC, SEL, 1, PROD>), dim3(1024), dim3(256), 0, s, in, bad, iters); break;
        case 2: hipLaunchKernelGGL((victim<SRC, SEL, 2, PROD>), dim3(1024), dim3(256), 0, s, in, bad, iters); break;
        case 3: hipLaunchKernelGGL((victim<SRC, SEL, 3, PROD>), dim3(1024), dim3(256), 0, s, in, bad, iters); break;
        case 4: hipLaunchKernelGGL((victim<SRC, SEL, 4, PROD>), dim3(1024), dim3(256), 0, s, in, bad, iters); break;
        default: hipLaunchKernelGGL((victim<SRC, SEL, 5, PROD>), dim3(1024), dim3(256), 0, s, in, bad, iters); break;
    }
}
template <int PROD>
static void run(int src, int sel, int gap, const float* in, unsigned* bad, int iters, hipStream_t s) {
#define C(S, L) if (src == S && sel == L) run_g<S, L, PROD>(gap, in, bad, iters, s);
    C(0, 0) C(0, 1) C(0, 2) C(0, 3) C(1, 0) C(1, 1) C(1, 2) C(1, 3)
#undef C
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (i * 2654435761u % 10007) / 10007.0f;
    float* d; unsigned *bad, *sink;
    hipMalloc(&d, 4096 * 4); hipMalloc(&bad, 32); hipMalloc(&sink, 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    static const char* gaps[] = {"none", "1 v_mov", "2 v_mov", "3 v_mov", "s_nop 0", "s_nop 1"};
    printf("%llu checks per cell and quarter of the wave; per gap: wrong low results in lanes 0-47 + lanes 48-63 / wrong high results (all lanes)\n", 1024ull * 64 * iters);
    for (int kind = 0; kind < 3; ++kind)
        for (int prod = 0; prod < 2; ++prod) {
            printf("---- neighbour waves: %s; producer of v[20:21]: %s\n", kNeighbour[kind], prod ? "two v_fma_f32" : "v_pk_fma_f32");
            for (int src = 0; src < 2; ++src)
                for (int sel = 0; sel < 4; ++sel) {
                    printf("  consumer v_pk_mul_f32 reads v[20:21] as src%d, op_sel:[%d,%d] (low pass reads v%d)", src, sel >> 1, sel & 1, 20 + (src == 0 ? sel >> 1 : sel & 1));
                    for (int gap = 0; gap < 6; ++gap) {
                        hipMemset(bad, 0, 32);
                        hipDeviceSynchronize();
                        switch (kind) {
                            case 0: hipLaunchKernelGGL(neighbour<0>, dim3(512), dim3(256), 0, sb, 100ll * 1000 * 30, sink); break;
                            case 1: hipLaunchKernelGGL(neighbour<1>, dim3(512), dim3(256), 0, sb, 100ll * 1000 * 30, sink); break;
                            default: hipLaunchKernelGGL(neighbour<2>, dim3(512), dim3(256), 0, sb, 100ll * 1000 * 30, sink); break;
                        }
                        if (prod) run<1>(src, sel, gap, d, bad, iters, sa); else run<0>(src, sel, gap, d, bad, iters, sa);
                        hipDeviceSynchronize();
                        unsigned b[8]; hipMemcpy(b, bad, 32, hipMemcpyDeviceToHost);
                        printf(" | %s: %u + %u / %u", gaps[gap], b[0] + b[2] + b[4], b[6], b[1] + b[3] + b[5] + b[7]);
                    }
                    printf("\n");
                }
        }
    return 0;
}
