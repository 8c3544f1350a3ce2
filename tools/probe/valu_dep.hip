// valu_dep.hip — does a DEPENDENT chain of VALU instructions issue as fast as independent ones when other waves are
// ready on the SIMD?  (tools/probe/valu_dep.py)  CH = number of independent chains a wave interleaves (1, 2, 4, 16).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float F2 __attribute__((ext_vector_type(2)));
template <int PK, int CH> __global__ void __launch_bounds__(256) dep(float *out, int iters, float seed) {
    float a[CH];
    F2 p[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + i; p[i] = F2{seed + i, seed - i}; }
    const float m = 1.0000001f, c = 1e-9f;
    const F2 m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32 / CH; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < CH; i++) s += a[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
extern "C" int valu_dep(int pk, int ch, int blocks, int iters, float *ms_out) {
    float *d = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
#define L(P, C) if (pk == P && ch == C) hipLaunchKernelGGL((dep<P, C>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
        L(0, 1) L(0, 2) L(0, 4) L(0, 16) L(1, 1) L(1, 2) L(1, 4) L(1, 16)
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    *ms_out = best;
    (void)hipFree(d);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
