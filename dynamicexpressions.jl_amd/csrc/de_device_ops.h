// de_device_ops.h — scalar operator semantics on gfx950 (device side).
//
// Same contract as oracle/de_oracle_ops.h (the CPU restatement): Julia Base
// semantics for every opcode of include/de_opcodes.h; NaN where Julia throws
// DomainError.  IEEE-exact operators are plain VALU instructions (compiled with
// -ffp-contract=off so a*b+c is never fused: Julia does not contract);
// transcendentals are the ROCm device-library (OCML) implementations, which are
// accurate to 1-2 ulp — inside the 1e-5 (f32) relative tolerance of the path.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/de_opcodes.h"
#include "de_program.h"

namespace de {

template <typename T> struct M; // math traits

template <> struct M<float> {
    using T = float;
    static __device__ __forceinline__ T abs(T x) { return fabsf(x); }
    static __device__ __forceinline__ T sqrt(T x) { return sqrtf(x); }
    static __device__ __forceinline__ T cbrt(T x) { return cbrtf(x); }
    static __device__ __forceinline__ T exp(T x) { return expf(x); }
    static __device__ __forceinline__ T exp2(T x) { return exp2f(x); }
    static __device__ __forceinline__ T log(T x) { return logf(x); }
    static __device__ __forceinline__ T log2(T x) { return log2f(x); }
    static __device__ __forceinline__ T log10(T x) { return log10f(x); }
    static __device__ __forceinline__ T log1p(T x) { return log1pf(x); }
    static __device__ __forceinline__ T sin(T x) { return sinf(x); }
    static __device__ __forceinline__ T cos(T x) { return cosf(x); }
    static __device__ __forceinline__ T tan(T x) { return tanf(x); }
    static __device__ __forceinline__ T sinh(T x) { return sinhf(x); }
    static __device__ __forceinline__ T cosh(T x) { return coshf(x); }
    static __device__ __forceinline__ T tanh(T x) { return tanhf(x); }
    static __device__ __forceinline__ T asin(T x) { return asinf(x); }
    static __device__ __forceinline__ T acos(T x) { return acosf(x); }
    static __device__ __forceinline__ T atan(T x) { return atanf(x); }
    static __device__ __forceinline__ T asinh(T x) { return asinhf(x); }
    static __device__ __forceinline__ T acosh(T x) { return acoshf(x); }
    static __device__ __forceinline__ T atanh(T x) { return atanhf(x); }
    static __device__ __forceinline__ T tgamma(T x) { return tgammaf(x); }
    static __device__ __forceinline__ T pow(T x, T y) { return powf(x, y); }
    static __device__ __forceinline__ T fmod(T x, T y) { return fmodf(x, y); }
    static __device__ __forceinline__ T rint(T x) { return rintf(x); }
    static __device__ __forceinline__ T floor(T x) { return floorf(x); }
    static __device__ __forceinline__ T ceil(T x) { return ceilf(x); }
    static __device__ __forceinline__ T trunc(T x) { return truncf(x); }
    static __device__ __forceinline__ T fma(T x, T y, T z) { return fmaf(x, y, z); }
    static __device__ __forceinline__ T copysign(T x, T y) { return copysignf(x, y); }
    static __device__ __forceinline__ T inf() { return __builtin_inff(); }
    static __device__ __forceinline__ T nan() { return __builtin_nanf(""); }
    static __device__ __forceinline__ bool isfinite(T x) { return __builtin_isfinite(x); }
    static __device__ __forceinline__ bool signbit(T x) { return __builtin_signbit(x); }
};

template <> struct M<double> {
    using T = double;
    static __device__ __forceinline__ T abs(T x) { return ::fabs(x); }
    static __device__ __forceinline__ T sqrt(T x) { return ::sqrt(x); }
    static __device__ __forceinline__ T cbrt(T x) { return ::cbrt(x); }
    static __device__ __forceinline__ T exp(T x) { return ::exp(x); }
    static __device__ __forceinline__ T exp2(T x) { return ::exp2(x); }
    static __device__ __forceinline__ T log(T x) { return ::log(x); }
    static __device__ __forceinline__ T log2(T x) { return ::log2(x); }
    static __device__ __forceinline__ T log10(T x) { return ::log10(x); }
    static __device__ __forceinline__ T log1p(T x) { return ::log1p(x); }
    static __device__ __forceinline__ T sin(T x) { return ::sin(x); }
    static __device__ __forceinline__ T cos(T x) { return ::cos(x); }
    static __device__ __forceinline__ T tan(T x) { return ::tan(x); }
    static __device__ __forceinline__ T sinh(T x) { return ::sinh(x); }
    static __device__ __forceinline__ T cosh(T x) { return ::cosh(x); }
    static __device__ __forceinline__ T tanh(T x) { return ::tanh(x); }
    static __device__ __forceinline__ T asin(T x) { return ::asin(x); }
    static __device__ __forceinline__ T acos(T x) { return ::acos(x); }
    static __device__ __forceinline__ T atan(T x) { return ::atan(x); }
    static __device__ __forceinline__ T asinh(T x) { return ::asinh(x); }
    static __device__ __forceinline__ T acosh(T x) { return ::acosh(x); }
    static __device__ __forceinline__ T atanh(T x) { return ::atanh(x); }
    static __device__ __forceinline__ T tgamma(T x) { return ::tgamma(x); }
    static __device__ __forceinline__ T pow(T x, T y) { return ::pow(x, y); }
    static __device__ __forceinline__ T fmod(T x, T y) { return ::fmod(x, y); }
    static __device__ __forceinline__ T rint(T x) { return ::rint(x); }
    static __device__ __forceinline__ T floor(T x) { return ::floor(x); }
    static __device__ __forceinline__ T ceil(T x) { return ::ceil(x); }
    static __device__ __forceinline__ T trunc(T x) { return ::trunc(x); }
    static __device__ __forceinline__ T fma(T x, T y, T z) { return ::fma(x, y, z); }
    static __device__ __forceinline__ T copysign(T x, T y) { return ::copysign(x, y); }
    static __device__ __forceinline__ T inf() { return __builtin_inf(); }
    static __device__ __forceinline__ T nan() { return __builtin_nan(""); }
    static __device__ __forceinline__ bool isfinite(T x) { return __builtin_isfinite(x); }
    static __device__ __forceinline__ bool signbit(T x) { return __builtin_signbit(x); }
};

// ---- fast Float32 transcendentals for the hot operators ------------------------
// OCML's cosf/sinf cost ~125 VALU instructions (both reduction paths are inlined and
// both polynomials evaluated); cos is >50 % of the VALU work of the headline workload.
// These versions are ~22 VALU on the fast path and keep <=1.6 ulp (measured against a
// correctly rounded reference over |x| <= 1e5, tests/test_gpu_ops.py):
//   k = rint(x*2/pi);  r = x - k*pi/2 with pi/2 = C1+C2+C3 (3 FMAs: the first is exact,
//   the other two round relative to the already-small r, so r keeps full relative
//   accuracy next to the zeros of cos/sin);  Cephes minimax sin/cos polynomials on
//   [-pi/4, pi/4];  quadrant select.  |x| > 1e5 (and Inf) takes the OCML Payne-Hanek
//   path under a divergent branch; NaN flows through the fast path.
constexpr float DE_TRIG_FAST_BOUND = 1.0e5f;
template <bool SIN> __device__ __forceinline__ float fast_trig_f32(float x) {
    const float t = x * 0x1.45f306p-1f; // 2/pi
    const float k = __builtin_rintf(t);
    float r = __builtin_fmaf(-k, 0x1.921fb6p+0f, x);
    r = __builtin_fmaf(-k, -0x1.777a5cp-25f, r);
    r = __builtin_fmaf(-k, -0x1.ee59dap-50f, r);
    const int q = (int)k;
    const float r2 = r * r;
    float c = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    c = __builtin_fmaf(r2, c, 4.166664568298827e-2f);
    c = __builtin_fmaf(r2, c, -0.5f);
    c = __builtin_fmaf(r2, c, 1.0f);
    float p = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    p = __builtin_fmaf(r2, p, -1.6666654611e-1f);
    const float s = __builtin_fmaf(r * r2, p, r);
    // cos: q=0:c 1:-s 2:-c 3:s      sin: q=0:s 1:c 2:-s 3:-c
    const float v = (q & 1) ? (SIN ? c : s) : (SIN ? s : c);
    const unsigned sign = (SIN ? ((unsigned)q << 30) : (((unsigned)q << 30) + 0x40000000u)) & 0x80000000u;
    return __uint_as_float(__float_as_uint(v) ^ sign);
}
// sin and cos of the same argument (value + derivative of cos/sin in the gradient kernel):
// one reduction, both polynomials, two quadrant selects.  Same accuracy as fast_trig_f32.
__device__ __forceinline__ void fast_sincos_f32(float x, float *sn, float *cs) {
    const float t = x * 0x1.45f306p-1f;
    const float k = __builtin_rintf(t);
    float r = __builtin_fmaf(-k, 0x1.921fb6p+0f, x);
    r = __builtin_fmaf(-k, -0x1.777a5cp-25f, r);
    r = __builtin_fmaf(-k, -0x1.ee59dap-50f, r);
    const int q = (int)k;
    const float r2 = r * r;
    float c = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    c = __builtin_fmaf(r2, c, 4.166664568298827e-2f);
    c = __builtin_fmaf(r2, c, -0.5f);
    c = __builtin_fmaf(r2, c, 1.0f);
    float p = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    p = __builtin_fmaf(r2, p, -1.6666654611e-1f);
    const float s = __builtin_fmaf(r * r2, p, r);
    const bool odd = q & 1;
    const unsigned ssign = ((unsigned)q << 30) & 0x80000000u;                 // sin: negative in quadrants 2,3
    const unsigned csign = (((unsigned)q << 30) + 0x40000000u) & 0x80000000u; // cos: negative in quadrants 1,2
    *sn = __uint_as_float(__float_as_uint(odd ? c : s) ^ ssign);
    *cs = __uint_as_float(__float_as_uint(odd ? s : c) ^ csign);
}
// exp(x) = 2^(x*log2(e)): k = rint(x*L), r = x*L - k in two FMAs (hi/lo split of L),
// hardware v_exp_f32 on r in [-0.5, 0.5], v_ldexp_f32 for the 2^k scaling (gradual
// underflow and overflow to Inf come from ldexp).  ~10 VALU vs 15; <= 2 ulp.
__device__ __forceinline__ float fast_exp_f32(float x) {
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -105.0f), 89.0f);
    const float k = __builtin_rintf(xc * 0x1.715476p+0f);
    float r = __builtin_fmaf(xc, 0x1.715476p+0f, -k);
    r = __builtin_fmaf(xc, 0x1.4ae0c0p-26f, r);
    const float e = __builtin_amdgcn_exp2f(r);
    const float y = __builtin_amdgcn_ldexpf(e, (int)k);
    return x != x ? x : y;
}

// Julia max/min: NaN-propagating, -0 < +0.
template <typename T> __device__ __forceinline__ T jl_max(T x, T y) {
    if (x != x) return x;
    if (y != y) return y;
    return (y > x || (M<T>::signbit(x) && !M<T>::signbit(y))) ? y : x;
}
template <typename T> __device__ __forceinline__ T jl_min(T x, T y) {
    if (x != x) return x;
    if (y != y) return y;
    return (y < x || (M<T>::signbit(y) && !M<T>::signbit(x))) ? y : x;
}
template <typename T> __device__ __forceinline__ T jl_mod(T x, T y) { // Base float.jl mod
    T r = M<T>::fmod(x, y);
    if (r == T(0)) return M<T>::copysign(r, y);
    if ((r > T(0)) != (y > T(0))) return r + y;
    return r;
}
template <typename T> __device__ __forceinline__ T jl_sign(T x) {
    return x > T(0) ? T(1) : (x < T(0) ? T(-1) : x);
}
template <typename T> __device__ __forceinline__ T jl_pow_abs2(T x, T y) {
    T l = M<T>::log(M<T>::abs(x));
    T m = y * l;
    return M<T>::exp(m);
}

// digamma for d gamma / dx (SpecialFunctions.digamma; unpinned)
template <typename T> __device__ inline T dev_digamma(T x) {
    T r = T(0);
    if (x <= T(0)) {
        if (x == M<T>::floor(x)) return M<T>::nan();
        const T pi = T(3.14159265358979323846);
        r = -pi / M<T>::tan(pi * x); // reflection: psi(x) = psi(1-x) - pi*cot(pi*x)
        x = T(1) - x;
    }
    while (x < T(10)) { r -= T(1) / x; x += T(1); }
    T f = T(1) / (x * x);
    T t = f * (T(-1.0 / 12) + f * (T(1.0 / 120) + f * (T(-1.0 / 252) + f * (T(1.0 / 240) + f * T(-1.0 / 132)))));
    return r + M<T>::log(x) - T(0.5) / x + t;
}

// Sum over the 64 lanes of a wavefront; the total is valid in lane 63.  Float32: DPP row
// operations fused into the adds (no LDS traffic); Float64: cross-lane shuffles.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define DE_DPP_ADD(CTRL, ROWMASK)                                                                          \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
    DE_DPP_ADD(0x111, 0xf) // row_shr:1
    DE_DPP_ADD(0x112, 0xf) // row_shr:2
    DE_DPP_ADD(0x114, 0xf) // row_shr:4   (bound_ctrl off + old = 0: lanes without a source add 0)
    DE_DPP_ADD(0x118, 0xf) // row_shr:8   -> lane 15 of every row holds the row sum
    DE_DPP_ADD(0x142, 0xa) // row_bcast:15 into rows 1 and 3
    DE_DPP_ADD(0x143, 0xc) // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
#undef DE_DPP_ADD
    return v;
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
    _Pragma("unroll") for (int m = 1; m < 64; m <<= 1) {
        const double o = __shfl_up(v, m, 64);
        if ((int)(threadIdx.x & 63) >= m) v += o;
    }
    return v;
}


} // namespace de
