// clock_probe.hip — what VALU issue rate does an MI355X sustain under an all-VALU load?  (tools/probe/clock_probe.py)
// Every wave runs `iters` x 32 independent v_pk_fma_f32 (no dependent-issue stalls), 8 waves per SIMD on every SIMD of
// the chip.  A wave64 VALU instruction occupies its SIMD16 for 4 cycles, so the effective engine clock is
//   waves x iters x 32 x 4 cycles / (SIMDs x seconds).
// Variants: 0 = v_pk_fma_f32 only, 1 = one v_exp_f32 per 8 packed FMAs (quarter-rate transcendental), 2 = plain v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float F2 __attribute__((ext_vector_type(2)));
template <int VAR> __global__ void __launch_bounds__(256) probe(float *out, int iters, float seed) {
    F2 a[16];
    for (int i = 0; i < 16; i++) a[i] = F2{seed + i, seed - i};
    const F2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    float e = seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (VAR == 2) { a[i][0] = __builtin_fmaf(a[i][0], m[0], c[0]); }
                else a[i] = __builtin_elementwise_fma(a[i], m, c);
            }
            if (VAR == 1) { e = __builtin_amdgcn_exp2f(e); e = __builtin_amdgcn_exp2f(e); e = __builtin_amdgcn_exp2f(e); e = __builtin_amdgcn_exp2f(e); }
        }
    }
    F2 s = {0, 0};
    for (int i = 0; i < 16; i++) s += a[i];
    if (s[0] + s[1] + e == 12345.678f) out[threadIdx.x] = s[0];
}
extern "C" int clock_probe(int variant, int blocks, int iters, float *ms_out) {
    float *d = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        if (variant == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
        else if (variant == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
        else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    hipEventElapsedTime(ms_out, e0, e1);
    hipFree(d);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
