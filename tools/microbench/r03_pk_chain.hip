// Round-3 microbenchmark: is a back-to-back dependent chain of packed-fp32 instructions, in the shape hipcc generated inside grid_tap
// (profiles/r03/nondeterminism_r03.md), safe?  Every lane runs the five-instruction sequence twice per iteration on the same inputs -- once
// as emitted (no gaps), once with `s_nop 4` between the instructions -- and counts the iterations whose results differ bitwise.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/r03_pk_chain.hip -o tools/microbench/bin/r03_pk_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));

#define SEQ(GAP)                                                                                        \
    asm volatile("v_sub_f32 %[w], %[f], %[fl]\n\t" GAP                                                  \
                 "v_cvt_u32_f32 %[u], %[c]\n\t" GAP                                                     \
                 "v_sub_f32 %[uw], 1.0, %[w]\n\t" GAP                                                   \
                 "v_pk_mul_f32 %[r], %[p], %[p] op_sel:[0,1] op_sel_hi:[0,1]\n\t" GAP                   \
                 "v_pk_mul_f32 %[r], %[q], %[r]\n\t"                                                    \
                 : [w] "=&v"(w), [u] "=&v"(u), [uw] "=&v"(uw), [r] "=&v"(r)                              \
                 : [f] "v"(f), [fl] "v"(fl), [c] "v"(c), [p] "v"(p), [q] "v"(q))

__global__ void __launch_bounds__(256, 2) chain(const float* in, unsigned* bad, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float f = in[i & 4095] * 7.3f + 3.1f, c = in[(i + 7) & 4095] * 100.f + 5.f;
    float2_t p = {in[(i + 1) & 4095], in[(i + 2) & 4095]};
    unsigned mism = 0;
    for (int it = 0; it < iters; ++it) {
        const float fl = floorf(f);
        float w, uw; unsigned u; float2_t r;
        float2_t q = {0.f, 0.f};
        // the second instruction pair of the chain reads {uw, w} as a register pair: emulate by building q behind the sequence's own w / uw
        // (the generated code had them in adjacent registers; here the pair is a copy, the dependent chain on r is what is under test)
        q[0] = 1.0f - (f - fl); q[1] = f - fl;
        SEQ("");
        const float2_t r0 = r; const float w0 = w; const unsigned u0 = u;
        SEQ("s_nop 4\n\t");
        if (__float_as_uint(r0[0]) != __float_as_uint(r[0]) || __float_as_uint(r0[1]) != __float_as_uint(r[1]) || __float_as_uint(w0) != __float_as_uint(w) || u0 != u) ++mism;
        f = f * 1.0001f + 0.37f; if (f > 900.f) f -= 890.f;
        c += 1.5f; p = p * 0.999f + float2_t{0.001f, 0.002f};
    }
    if (mism) atomicAdd(bad, mism);
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (i * 2654435761u % 10007) / 10007.0f;
    float* d; unsigned* bad;
    hipMalloc(&d, 4096 * 4); hipMalloc(&bad, 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(chain, dim3(256 * 8), dim3(256), 0, 0, d, bad, 20000);
        unsigned b = 0;
        hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("launch %d: %u of %llu sequence pairs differ between the back-to-back and the spaced form\n", rep, b, 256ull * 8 * 256 * 20000);
    }
    return 0;
}
