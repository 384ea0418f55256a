// Round-4 reproducer, part 2 (profiles/r04/nondeterminism_r04.md): which source selections (op_sel / op_sel_hi) of a packed-fp32 instruction go
// wrong next to MFMA waves.  All 16 selections of v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (third operand of the fma: default selection),
// each checked against plain spaced fp32 instructions; neighbours as in r04_pk_mfma_neighbour.hip.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/r04_pk_opsel_sweep.hip -o tools/microbench/bin/r04_pk_opsel_sweep
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
__device__ __forceinline__ float safe_add(float a, float b) { float r; asm volatile("s_nop 3\n\t v_add_f32 %0, %1, %2\n\t s_nop 3" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float safe_fma(float a, float b, float c) { float r; asm volatile("s_nop 3\n\t v_fma_f32 %0, %1, %2, %3\n\t s_nop 3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

#define COMBOS(X) X(0,0,0,0) X(0,0,0,1) X(0,0,1,0) X(0,0,1,1) X(0,1,0,0) X(0,1,0,1) X(0,1,1,0) X(0,1,1,1) \
                  X(1,0,0,0) X(1,0,0,1) X(1,0,1,0) X(1,0,1,1) X(1,1,0,0) X(1,1,0,1) X(1,1,1,0) X(1,1,1,1)

// OP: 0 mul, 1 add, 2 fma (r = p * q + t2, t2 = {t, t + 1} with its default selection), 3 fma with the third operand's halves crossed
template <int OP, int COMBO>
__global__ void __launch_bounds__(256) victim(const float* in, unsigned* bad, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float2_t p = {in[i & 4095] + 0.5f, in[(i + 1) & 4095] + 0.25f}, q = {in[(i + 2) & 4095] + 1.f, in[(i + 3) & 4095] + 2.f};
    float t = in[(i + 5) & 4095] + 0.75f;
    unsigned wl = 0, wh = 0, cz = 0, cprev = 0, cdef = 0, cin = 0;
    float prevl = -1.f;
    for (int it = 0; it < iters; ++it) {
        float2_t r;
        const float2_t t2 = {t, t + 1.f};
        int a = 0, b = 0, c = 0, d = 0;
#define X(A, B, C, D)                                                                                                                  \
        if constexpr (COMBO == A * 8 + B * 4 + C * 2 + D) {                                                                             \
            a = A; b = B; c = C; d = D;                                                                                                 \
            if constexpr (OP == 0) asm volatile("v_pk_mul_f32 v[10:11], %[p], %[q] op_sel:[" #A "," #B "] op_sel_hi:[" #C "," #D "]\n s_nop 7" : "=&{v[10:11]}"(r) : [p] "v"(p), [q] "v"(q)); \
            if constexpr (OP == 1) asm volatile("v_pk_add_f32 v[10:11], %[p], %[q] op_sel:[" #A "," #B "] op_sel_hi:[" #C "," #D "]\n s_nop 7" : "=&{v[10:11]}"(r) : [p] "v"(p), [q] "v"(q)); \
            if constexpr (OP == 2) asm volatile("v_pk_fma_f32 v[10:11], %[p], %[q], %[t] op_sel:[" #A "," #B ",0] op_sel_hi:[" #C "," #D ",1]\n s_nop 7" : "=&{v[10:11]}"(r) : [p] "v"(p), [q] "v"(q), [t] "v"(t2)); \
            if constexpr (OP == 3) asm volatile("v_pk_fma_f32 v[10:11], %[p], %[q], %[t] op_sel:[" #A "," #B ",1] op_sel_hi:[" #C "," #D ",0]\n s_nop 7" : "=&{v[10:11]}"(r) : [p] "v"(p), [q] "v"(q), [t] "v"(t2)); \
            if constexpr (OP == 4) asm volatile("v_pk_mov_b32 v[10:11], %[p], %[q] op_sel:[" #A "," #B "]\n s_nop 7" : "=&{v[10:11]}"(r) : [p] "v"(p), [q] "v"(q)); \
        }
        COMBOS(X)
#undef X
        const float pl = p[a], ql = q[b], ph = p[c], qh = q[d];
        float el, eh;
        if constexpr (OP == 0) { el = safe_mul(pl, ql); eh = safe_mul(ph, qh); }
        else if constexpr (OP == 1) { el = safe_add(pl, ql); eh = safe_add(ph, qh); }
        else if constexpr (OP == 2) { el = safe_fma(pl, ql, t2[0]); eh = safe_fma(ph, qh, t2[1]); }
        else if constexpr (OP == 3) { el = safe_fma(pl, ql, t2[1]); eh = safe_fma(ph, qh, t2[0]); }
        else { el = p[a]; eh = q[b]; }  // v_pk_mov_b32: D.lo = src0[op_sel[0]], D.hi = src1[op_sel[1]]
        if (__float_as_uint(r[0]) != __float_as_uint(el)) {  // what IS the wrong low result?
            ++wl;
            float dflt;  // the default selection {p0, q0}
            if constexpr (OP == 0) dflt = safe_mul(p[0], q[0]); else if constexpr (OP == 1) dflt = safe_add(p[0], q[0]); else if constexpr (OP == 4) dflt = p[0]; else dflt = safe_fma(p[0], q[0], t2[OP == 3]);
            cz += r[0] == 0.f;
            cprev += __float_as_uint(r[0]) == __float_as_uint(prevl);
            cdef += __float_as_uint(r[0]) == __float_as_uint(dflt);
            cin += r[0] == pl || r[0] == ql || r[0] == t2[0];  // (one operand passed through: the other read as 0 / 1)
        }
        prevl = el;
        wh += __float_as_uint(r[1]) != __float_as_uint(eh);
        p = p * 0.9993f + float2_t{0.0011f, 0.0023f};
        q = q * 1.0002f - float2_t{0.0003f, 0.0001f};
        t = t * 0.9998f + 0.0002f;
    }
    const int qd = (threadIdx.x & 63) >> 4;
    if (wl) { atomicAdd(&bad[qd * 2 + 0], wl); atomicAdd(&bad[8], cz); atomicAdd(&bad[9], cprev); atomicAdd(&bad[10], cdef); atomicAdd(&bad[11], cin); }
    if (wh) atomicAdd(&bad[qd * 2 + 1], wh);
}

template <int OP>
static void run_victim(int combo, const float* in, unsigned* bad, int iters, hipStream_t s) {
    switch (combo) {
#define X(A, B, C, D) case A * 8 + B * 4 + C * 2 + D: hipLaunchKernelGGL((victim<OP, A * 8 + B * 4 + C * 2 + D>), dim3(1024), dim3(256), 0, s, in, bad, iters); break;
        COMBOS(X)
#undef X
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (i * 2654435761u % 10007) / 10007.0f;
    float* d; unsigned *bad, *sink;
    hipMalloc(&d, 4096 * 4); hipMalloc(&bad, 64); hipMalloc(&sink, 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    static const char* ops[] = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_fma_f32 (src2 crossed: op_sel[2] = 1, op_sel_hi[2] = 0)", "v_pk_mov_b32 (op_sel only; op_sel_hi ignored)"};
    printf("%llu checks per cell and quarter of the wave; wrong low / high results in lanes 0-47 and in lanes 48-63\n", 1024ull * 64 * iters);
    for (int kind = 0; kind < 3; ++kind) {
        printf("---- neighbour waves: %s\n", kNeighbour[kind]);
        for (int op = 0; op < 5; ++op)
            for (int combo = 0; combo < 16; ++combo) {
                hipMemset(bad, 0, 64);
                hipDeviceSynchronize();
                switch (kind) {
                    case 0: hipLaunchKernelGGL(neighbour<0>, dim3(512), dim3(256), 0, sb, 100ll * 1000 * 30, sink); break;
                    case 1: hipLaunchKernelGGL(neighbour<1>, dim3(512), dim3(256), 0, sb, 100ll * 1000 * 30, sink); break;
                    default: hipLaunchKernelGGL(neighbour<2>, dim3(512), dim3(256), 0, sb, 100ll * 1000 * 30, sink); break;
                }
                if (op == 0) run_victim<0>(combo, d, bad, iters, sa); else if (op == 1) run_victim<1>(combo, d, bad, iters, sa); else if (op == 2) run_victim<2>(combo, d, bad, iters, sa); else if (op == 3) run_victim<3>(combo, d, bad, iters, sa); else run_victim<4>(combo, d, bad, iters, sa);
                hipDeviceSynchronize();
                unsigned b[16]; hipMemcpy(b, bad, 64, hipMemcpyDeviceToHost);
                const unsigned lo47 = b[0] + b[2] + b[4], hi47 = b[1] + b[3] + b[5];
                if (kind == 0 && (lo47 | hi47 | b[6] | b[7]) == 0) continue;  // (quiet rows of the control run are not printed)
                printf("  %s op_sel:[%d,%d] op_sel_hi:[%d,%d]   lanes 0-47: %u / %u   lanes 48-63: %u / %u", ops[op], combo >> 3, (combo >> 2) & 1, (combo >> 1) & 1, combo & 1,
                       lo47, hi47, b[6], b[7]);
                if (lo47 | hi47 | b[6] | b[7]) printf("   <-- WRONG: of the low results %u are 0, %u the previous iteration's, %u the default selection's, %u one operand passed through", b[8], b[9], b[10], b[11]);
                printf("\n");
            }
    }
    return 0;
}
