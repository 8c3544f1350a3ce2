// valu_rate.hip — issue cost of individual gfx950 VALU / LDS instructions, in SIMD cycles per wave64 instruction
// (tools/probe/valu_rate.py).  Each variant runs `iters` x 32 independent copies of one instruction (16 accumulators, no
// dependent-issue stalls) in every wave, 8 waves per SIMD on every SIMD; the cost is reported relative to v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int VAR> __global__ void __launch_bounds__(256) rate(float *out, int iters, float seed) {
    float a[16], b[16];
    typedef float F2 __attribute__((ext_vector_type(2)));
    F2 p[16];
    extern __shared__ float lds[];
    for (int i = 0; i < 16; i++) { a[i] = seed + i; b[i] = seed - i; p[i] = F2{seed + i, seed - i}; }
    lds[threadIdx.x * 4] = seed;
    const float m = 1.0000001f, c = 1e-9f;
    const F2 m2 = {m, m}, c2 = {c, c};
    const F2 sm2 = {__builtin_amdgcn_readfirstlane(__float_as_int(m)) ? m : m, m};
    const uint64_t mask = 0x5555555555555555ull;
    typedef float F4 __attribute__((ext_vector_type(4)));
    F4 q[4];
    const uint32_t ldsaddr = threadIdx.x * 16;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
#define ONE(i)                                                                                                            \
    if (VAR == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                                  \
    else if (VAR == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));                        \
    else if (VAR == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                         \
    else if (VAR == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                                         \
    else if (VAR == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));                                     \
    else if (VAR == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));                                          \
    else if (VAR == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                      \
    else if (VAR == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                         \
    else if (VAR == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));                             \
    else if (VAR == 9) asm volatile("v_lshl_add_u32 %0, %1, 31, %0" : "+v"(a[i]) : "v"(b[i]));                             \
    else if (VAR == 10) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                     \
    else if (VAR == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");                         \
    else if (VAR == 12) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                                     \
    else if (VAR == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                                     \
    else if (VAR == 14) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));                                                   \
    else if (VAR == 15) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                   \
    else if (VAR == 16) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));                                                 \
    else if (VAR == 17) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                        \
    else if (VAR == 18) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                     \
    else if (VAR == 19) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));                                    \
    else if (VAR == 20) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*(double *)&p[i]) : "v"(1.0000001), "v"(1e-9));     \
    else if (VAR == 21) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                         \
    else if (VAR == 22) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(a[i]));                                         \
    else if (VAR == 23) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(m));                                        \
    else if (VAR == 24) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(sm2), "v"(c2));                      \
    else if (VAR == 25) asm volatile("v_fmamk_f32 %0, %0, 0x3f800001, %1" : "+v"(a[i]) : "v"(c));                            \
    else if (VAR == 26) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));                                                \
    else if (VAR == 27) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c));                            \
    else if (VAR == 28) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                        \
    else if (VAR == 29) asm volatile("v_mul_f32_e64 %0, -%0, |%1|" : "+v"(a[i]) : "v"(m));                                 \
    else if (VAR == 30) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask));              \
    else if (VAR == 31) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(ldsaddr) : "memory");                    \
    else if (VAR == 32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                     \
    else if (VAR == 33) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                               \
    else if (VAR == 34) asm volatile("v_lshlrev_b32 %0, 31, %0" : "+v"(a[i]));                                             \
    else if (VAR == 35) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                      \
    else if (VAR == 36) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                                                    \
    else if (VAR == 37) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));                                                     \
    else if (VAR == 38) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&p[i]) : "v"(1.0000001));                    \
    else if (VAR == 39) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(p[i]) : "v"(p[(i + 1) & 15]));
            REP16(ONE)
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + p[i][0] + p[i][1];
    if (VAR == 31) { asm volatile("s_waitcnt lgkmcnt(0)"); for (int i = 0; i < 4; i++) s += q[i][0]; }
    if (s == 12345.678f) out[threadIdx.x] = s;
}
extern "C" int valu_rate(int variant, int blocks, int iters, float *ms_out) {
    float *d = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        switch (variant) {
#define L(V) case V: hipLaunchKernelGGL(rate<V>, dim3(blocks), dim3(256), 4096, 0, d, iters, 1.5f); break;
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31) L(32) L(33) L(34) L(35) L(36) L(37) L(38) L(39)
        default: return 3;
        }
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
