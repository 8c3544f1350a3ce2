// div_probe.hip — is a SHORTER Float32 division sequence still correctly rounded on gfx950?  
//
// The exact-mode division of the eval kernel (csrc/de_kernels.hip div_safe) is LLVM's expansion of `/` without the
// scale / fixup steps, valid for operands in [2^-40, 2^40]:
//     y0 = v_rcp_f32(d); e = fma(-d, y0, 1); y = fma(e, y0, y0); q0 = n*y; r0 = fma(-d, q0, n); q1 = fma(r0, y, q0);
//     r1 = fma(-d, q1, n); q2 = fma(r1, y, q1)                                                            (7 FMA-class ops)
// Markstein's theorem (Cornea, Harrison, Tang: "Scientific Computing on Itanium-based Systems", Thm. 8.? / Markstein 1990):
// if y = RN(1/d) and q0 is within 1 ulp of n/d, then q1 = RN(q0 + r0*y) with r0 = n - d*q0 (exact in an FMA) IS RN(n/d).
// So SEQ5 = the first five operations is correctly rounded whenever the Newton-refined y equals RN(1/d).
// This probe
//   A. checks y == RN(1/d) for ALL 2^23 significands of d (the hardware's v_rcp_f32 is the unknown) and lists the exceptions;
//   B. compares SEQ5 (and, for a constant divisor with a host-made y = RN(1/c), SEQ3 = q0, r0, q1) with the IEEE quotient on
//        B1. 2^N random (n, d) pairs with exponents in [-40, 40],
//        B2. for every d significand, 64 quotients placed at ROUNDING BOUNDARIES (n = RN(d * (q + ulp/2)) and its two
//            neighbours: the cases a division algorithm gets wrong first),
//        B3. for every exception of A, ALL 2^23 numerator significands.
// Prints one JSON object.   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_probe.bin div_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ float seq_y(float d) {
    float y = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float seq3(float n, float d, float y) { // y = (supposedly) RN(1/d)
    const float q0 = n * y;
    const float r0 = __builtin_fmaf(-d, q0, n);
    return __builtin_fmaf(r0, y, q0);
}
__device__ __forceinline__ float seq5(float n, float d) { return seq3(n, d, seq_y(d)); }
__device__ __forceinline__ float seq7(float n, float d) { // what the kernel runs today
    const float y = seq_y(d);
    float q = n * y;
    float r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float ieee_div(float n, float d) { return __fdiv_rn(n, d); }
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float mk(uint32_t sig23, int e, uint32_t sign) { return __uint_as_float((sign << 31) | ((uint32_t)(e + 127) << 23) | (sig23 & 0x7FFFFFu)); }

struct Counters { unsigned long long rcp_bad, b1_5, b1_3, b1_7, b2_5, b2_3, b2_7, b3_5, b3_n; uint32_t n_list; uint32_t list[4096]; };

__global__ void check_rcp(Counters *c) { // A: every significand (the exponent only shifts: rcp, fma are exact in the scaling)
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (1u << 23)) return;
    for (int e = -1; e <= 1; e++) { // and two neighbouring binades, to be sure the scaling argument holds on this hardware
        const float d = mk(m, e, 0);
        if (seq_y(d) != ieee_div(1.0f, d)) {
            atomicAdd(&c->rcp_bad, 1ull);
            if (e == 0) { const uint32_t k = atomicAdd(&c->n_list, 1u); if (k < 4096) c->list[k] = m; }
        }
    }
}
__global__ void check_random(Counters *c, uint64_t seed, int per_thread) { // B1
    uint64_t s = mix(seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) << 20));
    unsigned bad5 = 0, bad3 = 0, bad7 = 0;
    for (int i = 0; i < per_thread; i++) {
        s = mix(s);
        const uint64_t t = mix(s ^ 0x1234567ull);
        const float n = mk((uint32_t)s, (int)((s >> 32) % 81) - 40, (uint32_t)(s >> 63));
        const float d = mk((uint32_t)t, (int)((t >> 32) % 81) - 40, (uint32_t)(t >> 63));
        const float q = ieee_div(n, d);
        bad5 += seq5(n, d) != q;
        bad7 += seq7(n, d) != q;
        bad3 += seq3(n, d, ieee_div(1.0f, d)) != q;
    }
    if (bad5) atomicAdd(&c->b1_5, (unsigned long long)bad5);
    if (bad3) atomicAdd(&c->b1_3, (unsigned long long)bad3);
    if (bad7) atomicAdd(&c->b1_7, (unsigned long long)bad7);
}
__global__ void check_boundaries(Counters *c, uint64_t seed, int per_d) { // B2: quotients at rounding boundaries
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (1u << 23)) return;
    const float d = mk(m, 0, 0);
    const float yr = ieee_div(1.0f, d);
    uint64_t s = mix(seed ^ m);
    unsigned bad5 = 0, bad3 = 0, bad7 = 0;
    for (int i = 0; i < per_d; i++) {
        s = mix(s);
        const double q = (double)mk((uint32_t)s, 0, 0) + 0x1p-24; // halfway between two floats of [1, 2)
        const float n0 = (float)((double)d * q);                  // 24 x 25 bits: exact in double, rounded once
        for (int k = -1; k <= 1; k++) {
            const float n = __uint_as_float(__float_as_uint(n0) + (uint32_t)k);
            const float ref = ieee_div(n, d);
            bad5 += seq5(n, d) != ref;
            bad7 += seq7(n, d) != ref;
            bad3 += seq3(n, d, yr) != ref;
        }
    }
    if (bad5) atomicAdd(&c->b2_5, (unsigned long long)bad5);
    if (bad3) atomicAdd(&c->b2_3, (unsigned long long)bad3);
    if (bad7) atomicAdd(&c->b2_7, (unsigned long long)bad7);
}
__global__ void check_all_numerators(Counters *c, uint32_t dm) { // B3: one divisor, every numerator significand, two binades
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (1u << 23)) return;
    const float d = mk(dm, 0, 0);
    unsigned bad = 0;
    for (int e = 0; e <= 1; e++) {
        const float n = mk(m, e, 0);
        bad += seq5(n, d) != ieee_div(n, d);
    }
    if (bad) atomicAdd(&c->b3_5, (unsigned long long)bad);
    atomicAdd(&c->b3_n, 2ull);
}

int main(int argc, char **argv) {
    const int log2_random = argc > 1 ? atoi(argv[1]) : 36;
    Counters *d_c, h;
    CK(hipMalloc(&d_c, sizeof(Counters)));
    CK(hipMemset(d_c, 0, sizeof(Counters)));
    const dim3 blk(256), grid23((1u << 23) / 256);
    hipLaunchKernelGGL(check_rcp, grid23, blk, 0, 0, d_c);
    CK(hipDeviceSynchronize());
    const int per_thread = 1 << 12;
    const unsigned long long threads = 1ull << (log2_random - 12);
    hipLaunchKernelGGL(check_random, dim3((unsigned)(threads / 256)), blk, 0, 0, d_c, 0xD1B54A32D192ED03ull, per_thread);
    CK(hipDeviceSynchronize());
    const int per_d = 64;
    hipLaunchKernelGGL(check_boundaries, grid23, blk, 0, 0, d_c, 0x2545F4914F6CDD1Dull, per_d);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, d_c, sizeof(Counters), hipMemcpyDeviceToHost));
    const uint32_t n_exc = h.n_list < 4096 ? h.n_list : 4096;
    for (uint32_t i = 0; i < n_exc && i < 256; i++) hipLaunchKernelGGL(check_all_numerators, grid23, blk, 0, 0, d_c, h.list[i]);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, d_c, sizeof(Counters), hipMemcpyDeviceToHost));
    printf("{\n \"A_refined_reciprocal_not_RN\": {\"checked\": %llu, \"mismatches\": %llu, \"exception_significands_binade0\": %u, \"first\": [",
           3ull << 23, h.rcp_bad, h.n_list);
    for (uint32_t i = 0; i < n_exc && i < 16; i++) printf("%s\"0x%06x\"", i ? ", " : "", h.list[i]);
    printf("]},\n \"B1_random_pairs\": {\"pairs\": %llu, \"exponents\": \"[-40, 40]\", \"seq5_mismatches\": %llu, \"seq3_hostreciprocal_mismatches\": %llu, \"seq7_mismatches\": %llu},\n",
           threads * (unsigned long long)per_thread, h.b1_5, h.b1_3, h.b1_7);
    printf(" \"B2_rounding_boundaries\": {\"cases\": %llu, \"seq5_mismatches\": %llu, \"seq3_hostreciprocal_mismatches\": %llu, \"seq7_mismatches\": %llu},\n",
           (unsigned long long)(1u << 23) * per_d * 3, h.b2_5, h.b2_3, h.b2_7);
    printf(" \"B3_all_numerators_of_exception_divisors\": {\"divisors\": %u, \"cases\": %llu, \"seq5_mismatches\": %llu}\n}\n",
           n_exc < 256 ? n_exc : 256, h.b3_n, h.b3_5);
    return 0;
}
