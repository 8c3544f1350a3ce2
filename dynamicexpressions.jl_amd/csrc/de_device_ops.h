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

// One flag of the early exit (protocol = GArgs / KArgs skip_flagged).  Protocol 2, the default: flags are stored
// and loaded THROUGH THE CACHES — plain byte accesses, coherent within an XCD through its L2 — and the workgroups of every 32nd
// tile of an XCD ("refreshers") carry them between the XCDs through memory with agent-scope accesses: a flag that is down in
// memory but up in this XCD's L2 is stored into the L2 (pull), one that is down in the L2 but up in memory is written through
// (push).  Why not simply agent scope for every access (protocol 1, the first version): every uncached access of a chunk's flag
// line is serialised at that line's memory channel — ONE tree x 5e7 samples (the reference's own call shape: 195 000 workgroups on
// one byte) 0.34 instead of 0.17 ms, 64 trees 4.2 instead of 2.8 ms, and the stores count as much as the loads.  Why not the caches
// alone: an XCD never sees what another one found unless the line happens to leave both L2s — a kernel that writes little
// keeps it forever (fused loss: 10.7 instead of 8.4 ms: the late-flagged trees stayed alive on seven XCDs).
__device__ __forceinline__ uint8_t skip_flag_load(uint8_t *q, int protocol, int64_t tile) {
    if (protocol == 1) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // protocol 3 (experiment): as 2, but the load is non-temporal — the vector L1 does not keep the line, so a CU cannot sit on a stale
    // copy in a kernel that writes little; the XCD's L2 answers
    uint8_t f = protocol == 3 ? __builtin_nontemporal_load(q) : *q;
    if (((tile >> 3) & 31) == 0) { // (tiles of one XCD are 8 apart: map_block / gmap_block)
        const uint8_t m = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m == 0 && f != 0) { *q = 0; f = 0; }
        else if (f == 0 && m != 0) __hip_atomic_store(q, (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return f;
}


template <typename T> struct M; // math traits

__device__ __forceinline__ float fast_exp_f32(float x); // below: exp2 of the reduced argument + ldexp, rounds into the subnormal range

template <> struct M<float> {
    using T = float;
    static __device__ __forceinline__ T abs(T x) { return fabsf(x); }
    static __device__ __forceinline__ T sqrt(T x) { return sqrtf(x); }
    static __device__ __forceinline__ T cbrt(T x) { return cbrtf(x); }
    // not OCML's expf: that returns 0 below log(2^-149) where the correctly rounded result is still 2^-149 (results in
    // (1/2, 1) * 2^-149; Julia and glibc round them up), which `safe_log(pow_abs2(..))` turns into NaN against a finite value
    static __device__ __forceinline__ T exp(T x) { return fast_exp_f32(x); }
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
    static __device__ __forceinline__ T pow(T x, T y) {
        T r = powf(x, y);
        // OCML's powf has the same cut as its expf: results in (1/2, 1) * 2^-149 come back as 0 instead of 2^-149
        if (r == 0.0f && x != 0.0f && __builtin_isfinite(x) && __builtin_isfinite(y)) {
            const T e = fast_exp_f32(y * logf(fabsf(x))); // 0 or 2^-149 here
            r = copysignf(e, r);
        }
        return r;
    }
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

// ---- attribution (third party, NOT /root/reference) ------------------------------------------------------------------------------
// de_atan_f64, de_tan_f64 (__kernel_tan + the Cody-Waite rem_pio2) and de_pow_f64 (the __ieee754_pow core) below restate the algorithms
// of FreeBSD msun / fdlibm — s_atan.c, k_tan.c, e_rem_pio2.c, e_pow.c — the library Julia's Base ports: the same constants in the same
// operation order, so that the device returns the reference's bits.  fdlibm's notice, which its licence requires to be preserved:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//   Copyright (C) 2004 by Sun Microsystems, Inc. All rights reserved.   (k_tan.c, e_pow.c)
//
//   Developed at SunSoft/SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
//
// (tools/fit/ holds the bit-exact Python prototypes that were checked against msun's hex words; the oracle calls glibc and restates none of it.)
// Float64 atan: the algorithm Julia's Base.atan uses (the FreeBSD msun / fdlibm scheme: argument reduction at 7/16, 11/16, 19/16,
// 39/16 to atan(0.5), atan(1), atan(1.5), atan(inf) in hi + lo parts, odd/even split of a degree-11 polynomial in x^2).  Same
// constants, same operation order, no contraction (-ffp-contract=off): the reference's bits, 0.85 ulp worst case — OCML's atan
// measured 1.36 ulp, outside north_star's 1 ulp (tests/test_gpu_ulp_f64.py).  Divisions are IEEE (v_div_* sequence).
static __device__ __noinline__ double de_atan_f64(double x) {
    const double ax = __builtin_fabs(x);
    if (!(ax < 0x1p+66)) return x != x ? x + x : __builtin_copysign(1.57079632679489655800e+00 + 6.12323399573676603587e-17, x);
    double hi = 0.0, lo = 0.0, t = x;
    bool reduced = true;
    if (ax < 0.4375) {
        if (ax < 0x1p-29) return x;
        reduced = false;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; t = (2.0 * ax - 1.0) / (2.0 + ax); }
        else { hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; t = (ax - 1.0) / (ax + 1.0); }
    } else {
        if (ax < 2.4375) { hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; t = (ax - 1.5) / (1.0 + 1.5 * ax); }
        else { hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; t = -1.0 / ax; }
    }
    const double z = t * t, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                      w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                      w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    if (!reduced) return t - t * (s1 + s2);
    const double r = hi - ((t * (s1 + s2) - lo) - t);
    return x < 0.0 ? -r : r;
}

// Float64 tan: the algorithm of the reference's Base.tan (Julia base/special/trig.jl `tan_kernel` + `rem_pio2_kernel`, themselves
// the FreeBSD msun __kernel_tan / __ieee754_rem_pio2): Cody-Waite reduction by pi/2 in up to three steps of 33 + 53 bits
// (pio2_1 .. pio2_3t; good to 151 bits, covers |x| < 2^20 pi/2), then on |r| <= pi/4 the odd degree-27 polynomial T[] — for
// |r| >= 0.6744 after the reflection r -> pi/4 - r — and for odd multiples -1/tan(r) from a reciprocal corrected in two parts.
// Same constants (checked against msun's hex words), same operation order, no contraction: < 1 ulp (OCML's tan measured 1.03,
// outside north_star's 1 ulp; tests/test_gpu_ulp_f64.py).  |x| >= 2^20 pi/2, Inf and NaN keep OCML's Payne-Hanek path.
static __device__ __noinline__ double de_tan_f64(double x) {
    const double ax = __builtin_fabs(x);
    if (!(ax < 1647099.0)) return ::tan(x); // 2^20 * pi/2 = 1647099.33: the medium reduction's domain
    double y0 = x, y1 = 0.0;
    int n = 0;
    if (ax > 7.85398163397448278999e-01) {
        const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
        n = (int)fn;
        double r = x - fn * 1.57079632673412561417e+00;
        double w = fn * 6.07710050650619224932e-11; // first step: good to 85 bits
        y0 = r - w;
        const int ex = (int)((__double_as_longlong(ax) >> 52) & 0x7ff);
        if (ex - (int)((__double_as_longlong(y0) >> 52) & 0x7ff) > 16) { // cancellation: second step, 118 bits
            double t = r;
            w = fn * 6.07710050630396597660e-11;
            r = t - w;
            w = fn * 2.02226624879595063154e-21 - ((t - r) - w);
            y0 = r - w;
            if (ex - (int)((__double_as_longlong(y0) >> 52) & 0x7ff) > 49) { // third step, 151 bits: covers every case
                t = r;
                w = fn * 2.02226624871116645580e-21;
                r = t - w;
                w = fn * 8.47842766036889956997e-32 - ((t - r) - w);
                y0 = r - w;
            }
        }
        y1 = (r - y0) - w;
    } else if (ax < 0x1p-27) {
        return x; // tan(x) = x to the last bit (and keeps -0, subnormals)
    }
    // __kernel_tan(y0, y1, iy), iy = 1: tan, -1: -1/tan
    const int iy = 1 - ((n & 1) << 1);
    double xx = y0, yy = y1;
    const bool big = __builtin_fabs(xx) >= 0.6743354797363281; // msun compares the high word with 0x3FE59428
    const bool negx = xx < 0.0;
    if (big) {
        if (negx) { xx = -xx; yy = -yy; }
        const double z0 = 7.85398163397448278999e-01 - xx, w0 = 3.06161699786838301793e-17 - yy;
        xx = z0 + w0;
        yy = 0.0;
    }
    double z = xx * xx, w = z * z;
    double r = 1.33333333333201242699e-01 + w * (2.18694882948595424599e-02 + w * (3.59207910759131235356e-03 +
               w * (5.88041240820264096874e-04 + w * (7.81794442939557092300e-05 + w * -1.85586374855275456654e-05))));
    double v = z * (5.39682539762260521377e-02 + w * (8.86323982359930005737e-03 + w * (1.45620945432529025516e-03 +
               w * (2.46463134818469906812e-04 + w * (7.14072491382608190305e-05 + w * 2.59073051863633712884e-05)))));
    double sx = z * xx;
    r = yy + z * (sx * (r + v) + yy);
    r += 3.33333333333334091986e-01 * sx;
    w = xx + r;
    if (big) {
        v = (double)iy;
        return (negx ? -1.0 : 1.0) * (v - 2.0 * (xx - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1 / (xx + r), accurately: the reciprocal and w split at 32 bits
    const double zt = __longlong_as_double(__double_as_longlong(w) & ~0xFFFFFFFFLL);
    v = r - (zt - xx); // zt + v = r + xx
    const double a = -1.0 / w;
    const double t = __longlong_as_double(__double_as_longlong(a) & ~0xFFFFFFFFLL);
    sx = 1.0 + t * zt;
    return t + a * (sx + t * v);
}

// Float64 x^y: the FreeBSD msun __ieee754_pow core (error < 0.70 ulp; OCML's pow measured 1.26 ulp, outside north_star's 1 ulp,
// tests/test_gpu_ulp_f64.py).  log2|x| = n + dp_h + z_h + z_l from s = (m - bp)/(m + bp), bp = 1 or 1.5, carried as head + tail
// (s_h + s_l, the degree-6 series L1..L6 in s^2, 2/(3 ln 2) = cp_h + cp_l), y log2|x| formed from the split operands as p_h + p_l,
// 2^(p_h + p_l) by the exp kernel P1..P5 after removing the integer part.  Constants checked against msun's hex words.  The main
// path: x finite and non-zero (negative with an integral y), y finite and non-zero, |y| <= 2^31; every other case (signed zeros,
// Inf, NaN, huge |y|, negative base with fractional exponent) keeps OCML's case analysis — those results are exact or NaN.
#define DE_TRUNC32(v) __longlong_as_double(__double_as_longlong(v) & ~0xFFFFFFFFLL)
// log2(ax) = t1 + t2 for a finite ax > 0 (msun pow's logarithm: t1 carries the leading 21 bits, the sum is good to ~2^-68)
static __device__ __forceinline__ void pow_log2_core(double ax, double &t1_out, double &t2_out) {
    int n = 0;
    long long bits = __double_as_longlong(ax);
    if ((bits >> 52) == 0) { ax *= 0x1p53; n = -53; bits = __double_as_longlong(ax); } // subnormal base
    int ix = (int)(bits >> 32);
    n += (ix >> 20) - 0x3ff;
    const int j = ix & 0x000fffff;
    ix = j | 0x3ff00000; // mantissa in [1, 2)
    int k;
    if (j <= 0x3988E) k = 0;       // m < sqrt(3/2)
    else if (j < 0xBB67A) k = 1;   // m < sqrt(3)
    else { k = 0; n += 1; ix -= 0x00100000; }
    ax = __longlong_as_double(((long long)ix << 32) | (bits & 0xFFFFFFFFLL));
    const double bp = k ? 1.5 : 1.0, dp_h = k ? 5.84962487220764160156e-01 : 0.0, dp_l = k ? 1.35003920212974897128e-08 : 0.0;
    // ss = s_h + s_l = (m - bp) / (m + bp)
    double u = ax - bp, v = 1.0 / (ax + bp);
    const double ss = u * v;
    const double s_h = DE_TRUNC32(ss);
    double t_h = __longlong_as_double((long long)(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18)) << 32); // high part of m + bp
    double t_l = ax - (t_h - bp);
    const double s_l = v * ((u - s_h * t_h) - s_h * t_l);
    // log(m)
    double s2 = ss * ss;
    double r = s2 * s2 * (5.99999999999994648725e-01 + s2 * (4.28571428578550184252e-01 + s2 * (3.33333329818377432918e-01 +
               s2 * (2.72728123808534006489e-01 + s2 * (2.30660745775561754067e-01 + s2 * 2.06975017800338417784e-01)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = DE_TRUNC32(3.0 + s2 + r);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    const double p_h = DE_TRUNC32(u + v);
    const double p_l = v - (p_h - u);
    const double z_h = 9.61796700954437255859e-01 * p_h; // cp_h + cp_l = 2 / (3 ln 2)
    const double z_l = -7.02846165095275826516e-09 * p_h + p_l * 9.61796693925975554329e-01 + dp_l;
    const double t = (double)n;
    const double t1 = DE_TRUNC32(((z_h + z_l) + dp_h) + t);
    t1_out = t1;
    t2_out = z_l - (((t1 - t) - dp_h) - z_h);
}
// 2^(p_h + p_l), |p_h + p_l| inside the exponent range (the callers test): integer part out, msun's exp kernel on the rest
static __device__ __forceinline__ double pow_exp2_core(double p_h, double p_l) {
    double z = p_l + p_h;
    int ni = 0;
    if (__builtin_fabs(z) > 0.5) {
        const double zi = __builtin_rint(z);
        ni = (int)zi;
        p_h -= zi;
    }
    double t = DE_TRUNC32(p_l + p_h);
    const double u = t * 6.93147182464599609375e-01;
    const double v = (p_l - (t - p_h)) * 6.93147180559945286227e-01 + t * -1.90465429995776804525e-09;
    z = u + v;
    const double w = v - (z - u);
    t = z * z;
    const double tt = z - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
                      t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    const double r = (z * tt) / (tt - 2.0) - (w + z * w);
    z = 1.0 - (r - z);
    return ::ldexp(z, ni);
}
static __device__ __noinline__ double de_pow_f64(double x, double y) {
    const bool yint = y == __builtin_rint(y);
    if (!(__builtin_isfinite(x) && __builtin_isfinite(y) && x != 0.0 && y != 0.0 && __builtin_fabs(y) <= 0x1p31 && (x > 0.0 || yint))) return ::pow(x, y);
    if (y == 1.0) return x;
    if (y == 2.0) return x * x;
    if (y == -1.0) return 1.0 / x;
    if (y == 0.5 && x > 0.0) return __builtin_sqrt(x);
    const double sgn = (x < 0.0 && ((long long)y & 1LL)) ? -1.0 : 1.0; // (-|x|)^(odd integer)
    const double ax = __builtin_fabs(x);
    if (ax == 1.0) return sgn;
    double t1, t2; // log2|x| = t1 + t2
    pow_log2_core(ax, t1, t2);
    // y * log2|x| = p_h + p_l with y split as y1 + y2
    const double y1 = DE_TRUNC32(y);
    const double p_l = (y - y1) * t1 + y * t2;
    const double p_h = y1 * t1;
    const double z = p_l + p_h;
    if (z >= 1024.0) { // overflow unless the sum is a hair below 1024
        if (z > 1024.0 || p_l + 8.0085662595372944372e-17 > z - p_h) return sgn * __builtin_inf();
    } else if (z <= -1075.0) {
        if (z < -1075.0 || p_l <= z - p_h) return sgn * 0.0;
    }
    return sgn * pow_exp2_core(p_h, p_l);
}

// Float64 gamma: 2^E with E accumulated in head + tail and ONE final rounding in pow's exp kernel (OCML's tgamma measured 4.3 ulp,
// outside north_star's 1 ulp: tests/test_gpu_ulp_f64.py; this one 0.77 over (-170, 171.6) incl. the neighbourhoods of the poles,
// tools/fit/gamma_proto.py against mpmath).  x is shifted up to y = x + n >= 16 with the product P = x (x + 1) .. (x + n - 1) in
// double-double (every factor an exact two_sum: relative accuracy survives next to a pole), Gamma(x) = Gamma(y) / P and
//   log2 Gamma(y) = (y - 1/2) log2 y - y log2 e + log2(2 pi) / 2 + log2 e * (1/(12 y) - 1/(360 y^3) + .. - 3617/(122400 y^15))
// (Stirling; the tail is < 2^-63 at y = 16), the logarithms from pow_log2_core.  Zero, the negative integers, NaN, Inf and
// x < -180 (the result underflows) keep OCML's results.
static __device__ __forceinline__ void dd_two_sum(double a, double b, double &s, double &e) {
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
static __device__ __forceinline__ void dd_fast_two_sum(double a, double b, double &s, double &e) {
    s = a + b;
    e = b - (s - a);
}
static __device__ __forceinline__ void dd_mul(double &ah, double &al, double bh, double bl) { // a *= b
    const double p = ah * bh;
    double e = __builtin_fma(ah, bh, -p);
    e += ah * bl + al * bh;
    dd_fast_two_sum(p, e, ah, al);
}
static __device__ __forceinline__ void dd_add(double &ah, double &al, double bh, double bl) { // a += b
    double s, e;
    dd_two_sum(ah, bh, s, e);
    e += al + bl;
    dd_fast_two_sum(s, e, ah, al);
}
static __device__ __noinline__ double de_gamma_f64(double x) {
    if (!__builtin_isfinite(x) || x == 0.0 || (x < 0.0 && x == __builtin_floor(x)) || x < -180.0) return ::tgamma(x);
    if (__builtin_fabs(x) < 0x1p-55) return 1.0 / x; // Gamma(x) = 1/x - gamma + O(x)
    const int n = x >= 16.0 ? 0 : (int)__builtin_ceil(16.0 - x);
    double ph = 1.0, pl = 0.0;
    int esum = 0;
    for (int k = 0; k < n; k++) {
        double fh, fl;
        dd_two_sum(x, (double)k, fh, fl);
        dd_mul(ph, pl, fh, fl);
        if (__builtin_fabs(ph) > 0x1p500) { ph *= 0x1p-500; pl *= 0x1p-500; esum += 500; }
        else if (__builtin_fabs(ph) < 0x1p-500) { ph *= 0x1p500; pl *= 0x1p500; esum -= 500; }
    }
    double yh = x, yl = 0.0;
    if (n) dd_two_sum(x, (double)n, yh, yl);
    const double INV_LN2 = 0x1.71547652b82fep+0, LOG2E_LO = 0x1.777d0ffda0d24p-56;
    double t1, t2;
    pow_log2_core(yh, t1, t2);
    const double a_h = yh - 0.5, a1 = DE_TRUNC32(a_h); // (exact: yh >= 16)
    const double p_l = (a_h - a1) * t1 + a_h * t2 + yl * (t1 + t2) + a_h * (yl / yh) * INV_LN2; // log2(yh + yl) = log2 yh + yl / (yh ln 2)
    double eh, el;
    dd_fast_two_sum(a1 * t1, p_l, eh, el);
    { // - y log2 e
        const double qh = yh * INV_LN2;
        const double ql = __builtin_fma(yh, INV_LN2, -qh) + (yh * LOG2E_LO + yl * INV_LN2);
        double qs, qe;
        dd_fast_two_sum(qh, ql, qs, qe);
        dd_add(eh, el, -qs, -qe);
    }
    dd_add(eh, el, 0x1.536439a4c6efcp+0, -0x1.49e49a361efebp-54); // log2(2 pi) / 2
    {
        const double r = 1.0 / yh, r2 = r * r;
        double s = -0x1.e4286cb0f5398p-6;
        s = s * r2 + 0x1.a41a41a41a41ap-8;
        s = s * r2 + -0x1.f6ab0d9993c7dp-10;
        s = s * r2 + 0x1.b951e2b18ff23p-11;
        s = s * r2 + -0x1.3813813813814p-11;
        s = s * r2 + 0x1.a01a01a01a01ap-11;
        s = s * r2 + -0x1.6c16c16c16c17p-9;
        s = s * r2 + 0x1.5555555555555p-4;
        el += s * r * INV_LN2;
    }
    double sign = 1.0;
    if (n) { // - log2 |P|
        if (ph < 0.0) { sign = -1.0; ph = -ph; pl = -pl; }
        double l1, l2;
        pow_log2_core(ph, l1, l2);
        l2 += (pl / ph) * INV_LN2;
        dd_add(eh, el, -l1 - (double)esum, -l2);
    }
    const double z = eh + el;
    if (z >= 1024.0) return sign * __builtin_inf();
    if (z <= -1075.0) return sign * 0.0;
    return sign * pow_exp2_core(eh, el);
}
#undef DE_TRUNC32

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
    static __device__ __forceinline__ T tan(T x) { return de_tan_f64(x); }
    static __device__ __forceinline__ T sinh(T x) { return ::sinh(x); }
    static __device__ __forceinline__ T cosh(T x) { return ::cosh(x); }
    static __device__ __forceinline__ T tanh(T x) { return ::tanh(x); }
    static __device__ __forceinline__ T asin(T x) { return ::asin(x); }
    static __device__ __forceinline__ T acos(T x) { return ::acos(x); }
    static __device__ __forceinline__ T atan(T x) { return de_atan_f64(x); }
    static __device__ __forceinline__ T asinh(T x) { return ::asinh(x); }
    static __device__ __forceinline__ T acosh(T x) { return ::acosh(x); }
    static __device__ __forceinline__ T atanh(T x) { return ::atanh(x); }
    static __device__ __forceinline__ T tgamma(T x) { return de_gamma_f64(x); }
    static __device__ __forceinline__ T pow(T x, T y) { return de_pow_f64(x, y); }
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
// both polynomials evaluated); cos is the largest single share of the VALU work of the headline
// workload.  One odd polynomial serves both functions:
//   sin(x) = (-1)^n sin(r),  r = x - n*pi,          n = rint(x/pi)
//   cos(x) = (-1)^n sin(r),  r = x - (n - 1/2)*pi,  n = rint(x/pi + 1/2)
// so r lies in [-pi/2, pi/2] and is SMALL next to the zeros of either function.  pi = P1+P2+P3
// (3 FMAs: the first is exact — (n - 1/2)*P1 has its last bit at 2^-23 and |r| < 2 — the other
// two round relative to the already-small r, which keeps full relative accuracy at the zeros).
// n comes from the 1.5*2^23 magic add (packable; its low mantissa bit is the parity of n, i.e.
// the sign flip).  sin(r) = r + r^3 Q(r^2), Q = degree-3 minimax of the relative error on
// |r| <= pi/2 + 0.02 (6.9e-9, tools/fit/trig_fit.py: the Float32 Horner steps and the final r + r^3 Q cancellation
// near |r| = pi/2 dominate the error, so the degree-4 Q of round 1 (2.8e-11) bought nothing: 1.97 -> 2.03 ulp worst
// case in the emulation).  10 VALU per element (two elements share every v_pk_* op) + 1 for the exact-extremum
// test below; <= 2.1 ulp over |x| <= 1e5 (measured: tests/test_gpu_ops.py).  |x| > 1e5 and
// Inf take the OCML Payne-Hanek path under a divergent branch; NaN flows through the fast path.
constexpr float DE_TRIG_FAST_BOUND = 1.0e5f;
constexpr float DE_TRIG_FAST_BOUND_M = 31829.5f; // < rint(1e5/pi + 1/2) = 31830 <= rint(|x|/pi [+ 1/2]) for |x| > 1e5: the wave-uniform pre-test on the multiple of pi
#define DE_TRIG_INV_PI 0x1.45f306p-2f
#define DE_TRIG_MAGIC 12582912.0f
// pi rounded DOWN term by term: all three positive, so that for x = -0 every step is (-0) * P + (-0) = -0 — with a negative
// P2 the second step was (+0) + (-0) = +0 and sin(-0) came out +0 (1 / sin(-0) = -Inf in the reference; tests/fuzz/fuzz_gpu.py 102)
#define DE_TRIG_P1 0x1.921fb4p+1f
#define DE_TRIG_P2 0x1.4442d0p-23f
#define DE_TRIG_P3 0x1.846988p-47f
#define DE_TRIG_S0 -0x1.55554ap-3f
#define DE_TRIG_S1 0x1.110ea0p-7f // 3 ulp below the minimax coefficient: the worst case of the Float32 EVALUATION drops from 2.03 to 1.75 ulp (tools/fit/trig_fit.py)
#define DE_TRIG_S2 -0x1.9f6716p-13f
#define DE_TRIG_S3 0x1.5d3a4ep-19f
// Within 2^-12 of +-pi/2 the correctly rounded sine IS +-1 (1 - d^2/2 with d^2/2 <= 2^-25): return it
// exactly, so that cos(0) == 1, cos(2k*pi) == 1, sin(pi/2 + k*pi) == +-1 hold bit for bit (the
// polynomial's 2-ulp worst case sits exactly there, where r + r^3 Q cancels from 1.57 to 1).
__device__ __forceinline__ float trig_extremum_fix(float r, float s) {
    const float one = __uint_as_float((__float_as_uint(r) & 0x80000000u) | 0x3f800000u);
    // two-sided: for |x| near 1e5 the rounding of x/pi can leave |r| up to 0.006 beyond pi/2
    return __builtin_fabsf(__builtin_fabsf(r) - 0x1.921fb6p+0f) < 0x1.fep-13f ? one : s;
}
template <bool SIN> __device__ __forceinline__ float fast_trig_f32(float x) {
    const float t = SIN ? x * DE_TRIG_INV_PI : __builtin_fmaf(x, DE_TRIG_INV_PI, 0.5f);
    const float kk = t + DE_TRIG_MAGIC;
    const float n = kk - DE_TRIG_MAGIC;
    const float m = SIN ? n : n - 0.5f;
    float r = __builtin_fmaf(-m, DE_TRIG_P1, x);
    r = __builtin_fmaf(-m, DE_TRIG_P2, r);
    r = __builtin_fmaf(-m, DE_TRIG_P3, r);
    const float z = r * r;
    float p = __builtin_fmaf(z, DE_TRIG_S3, DE_TRIG_S2);
    p = __builtin_fmaf(z, p, DE_TRIG_S1);
    p = __builtin_fmaf(z, p, DE_TRIG_S0);
    float s = __builtin_fmaf(r * z, p, r);
    if constexpr (SIN) s = __builtin_copysignf(s, r); // sin(-0) = -0: the sum of the -0 argument and its +0 correction term is +0
#ifndef DE_TRIG_NO_EXTREMUM_FIX
    s = trig_extremum_fix(r, s);
#endif
    return __uint_as_float(__float_as_uint(s) ^ (__float_as_uint(kk) << 31));
}
// Two elements at a time: every multiply/add/fma is one v_pk_*_f32.
typedef float DeF2 __attribute__((ext_vector_type(2)));
typedef unsigned DeU2 __attribute__((ext_vector_type(2)));
#define DE_F2(c) (DeF2{(c), (c)})
// The two-element core: reduced argument r, its square z, sin(r) and the magic-add word (parity of n).
struct DeTrig2 { DeF2 r, z, s, kk, m; };
template <bool SIN> __device__ __forceinline__ DeTrig2 fast_trig_core_f32x2(DeF2 x) {
    DeTrig2 o;
    const DeF2 t = SIN ? x * DE_F2(DE_TRIG_INV_PI) : __builtin_elementwise_fma(x, DE_F2(DE_TRIG_INV_PI), DE_F2(0.5f));
    o.kk = t + DE_F2(DE_TRIG_MAGIC);
    const DeF2 n = o.kk - DE_F2(DE_TRIG_MAGIC);
    const DeF2 m = SIN ? n : n - DE_F2(0.5f);
    o.m = n; // the wave-uniform range pre-test reads the multiple itself
    DeF2 r = __builtin_elementwise_fma(-m, DE_F2(DE_TRIG_P1), x);
    r = __builtin_elementwise_fma(-m, DE_F2(DE_TRIG_P2), r);
    r = __builtin_elementwise_fma(-m, DE_F2(DE_TRIG_P3), r);
    const DeF2 z = r * r;
    DeF2 p = __builtin_elementwise_fma(z, DE_F2(DE_TRIG_S3), DE_F2(DE_TRIG_S2));
    p = __builtin_elementwise_fma(z, p, DE_F2(DE_TRIG_S1));
    p = __builtin_elementwise_fma(z, p, DE_F2(DE_TRIG_S0));
    o.r = r;
    o.z = z;
    o.s = __builtin_elementwise_fma(r * z, p, r);
    // sin(-0) = -0 (IEEE, Julia): r = -0 there, but r + r^3 Q adds a +0 correction to it and the sum of opposite zeros is +0.
    // |s| <= |r| and their signs agree everywhere else, so copying r's sign (v_bfi_b32) changes nothing but that zero.
    if constexpr (SIN) { o.s[0] = __builtin_copysignf(o.s[0], r[0]); o.s[1] = __builtin_copysignf(o.s[1], r[1]); }
    return o;
}
// The same arithmetic in two pieces, for handlers that test the multiple n before anything else (h_un_fast, de_kernels.hip):
// the reduced argument r(x, n) and sin(r).  TB = the turbo mode's two-term pi.
template <bool SIN, bool TB> __device__ __forceinline__ DeF2 trig_reduced_f32x2(DeF2 x, DeF2 n) {
    const DeF2 m = SIN ? n : n - DE_F2(0.5f);
    DeF2 r = __builtin_elementwise_fma(-m, DE_F2(DE_TRIG_P1), x);
    r = __builtin_elementwise_fma(-m, DE_F2(DE_TRIG_P2), r);
    if constexpr (!TB) r = __builtin_elementwise_fma(-m, DE_F2(DE_TRIG_P3), r);
    return r;
}
template <bool SIN, bool TB> __device__ __forceinline__ DeF2 trig_poly_f32x2(DeF2 x, DeF2 n) {
    const DeF2 r = trig_reduced_f32x2<SIN, TB>(x, n);
    const DeF2 z = r * r;
    DeF2 p = __builtin_elementwise_fma(z, DE_F2(DE_TRIG_S3), DE_F2(DE_TRIG_S2));
    p = __builtin_elementwise_fma(z, p, DE_F2(DE_TRIG_S1));
    p = __builtin_elementwise_fma(z, p, DE_F2(DE_TRIG_S0));
    DeF2 s = __builtin_elementwise_fma(r * z, p, r);
    if constexpr (SIN) { s[0] = __builtin_copysignf(s[0], r[0]); s[1] = __builtin_copysignf(s[1], r[1]); } // sin(-0) = -0
    return s;
}
__device__ __forceinline__ DeF2 fast_trig_sign_f32x2(DeF2 s, DeF2 kk) {
    // (-1)^n: the parity of n is the low mantissa bit of the magic sum; ADDING it at bit 31 flips the sign bit exactly like
    // the xor would (the carry leaves the word) and is one v_lshl_add_u32 instead of a shift and a xor
    DeF2 y;
    y[0] = __uint_as_float((__float_as_uint(kk[0]) << 31) + __float_as_uint(s[0]));
    y[1] = __uint_as_float((__float_as_uint(kk[1]) << 31) + __float_as_uint(s[1]));
    return y;
}
// "Does any of the four arguments exceed `bound` in magnitude?" — v_max3_f32 + v_max_f32 on |.| + v_cmp (3 VALU for four
// elements).  NaN arguments are NOT reported (maxnum drops them): every caller's fast path propagates NaN by itself.
// (Round 2 first tried the OR of the four bit patterns against bound's: the OR of the exponent fields of 1.5 and 2.5 is
// already 0x7F8 — practically every wavefront took the slow path's per-element tests.)
// Callers pass RESULTS of arithmetic (the multiple of pi, x log2 e): on a raw input the compiler must first quiet a
// signalling NaN (one v_max_f32 x, x per operand).
__device__ __forceinline__ bool any_abs_exceeds_f32x4(float a, float b, float c, float d, float bound) {
    // spelled out: from fmaxf(fabsf(.)) the compiler sometimes adds a v_max_f32 |x|, |x| per operand (sNaN quieting)
    float m;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(m) : "v"(m), "v"(d));
    return m > bound;
}
// Four elements with a wave-uniform short cut for the extremum select: |r| within 2^-12 of pi/2 (or beyond it: |r| never
// exceeds pi/2 by more than 0.006) means r^2 > pi^2/4 - 8e-4; the one-sided test on the already computed r^2 is a
// v_max3 + v_max + v_cmp per FOUR elements and the select itself (4 per element) runs only in the ~4 % of wavefronts
// where some lane is that close.
template <bool SIN> __device__ __forceinline__ bool fast_trig_f32x4(const float (&x)[4], float (&y)[4]) {
    DeTrig2 a = fast_trig_core_f32x2<SIN>(DeF2{x[0], x[1]}), b = fast_trig_core_f32x2<SIN>(DeF2{x[2], x[3]});
#ifndef DE_TRIG_NO_EXTREMUM_FIX
    // NaN: fmax drops it (the select would leave a NaN alone anyway)
    const bool near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a.z[0], a.z[1]), b.z[0]), b.z[1]) > (0x1.3bd3ccp+1f - 8.0e-4f); // pi^2/4
    if (__ballot(near) != 0ull) {
        a.s[0] = trig_extremum_fix(a.r[0], a.s[0]);
        a.s[1] = trig_extremum_fix(a.r[1], a.s[1]);
        b.s[0] = trig_extremum_fix(b.r[0], b.s[0]);
        b.s[1] = trig_extremum_fix(b.r[1], b.s[1]);
    }
#endif
    const DeF2 ya = fast_trig_sign_f32x2(a.s, a.kk), yb = fast_trig_sign_f32x2(b.s, b.kk);
    y[0] = ya[0]; y[1] = ya[1]; y[2] = yb[0]; y[3] = yb[1];
    // |x| > 1e5 (or Inf) somewhere in the wavefront?  Tested on the multiple of pi the reduction found — |x| > 1e5 means
    // |n| >= 31830 — the caller then applies the exact per-element condition on x.
    return __ballot(any_abs_exceeds_f32x4(a.m[0], a.m[1], b.m[0], b.m[1], DE_TRIG_FAST_BOUND_M)) != 0ull;
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
    const float sn_ = __uint_as_float(__float_as_uint(odd ? c : s) ^ ssign);
    *sn = x == 0.0f ? x : sn_; // sin(-0) = -0: the reduction (k = -0: +0 * P + -0) and r + r^3 Q both turn the zero positive
    *cs = __uint_as_float(__float_as_uint(odd ? s : c) ^ csign);
}
// exp(x) = 2^(x*log2(e)).  ldexp form: k = rint(x*L), r = x*L - k in two FMAs (hi/lo split of L), hardware v_exp_f32 on
// r in [-0.5, 0.5], v_ldexp_f32 for the 2^k scaling (gradual underflow and overflow to Inf come from ldexp); <= 2 ulp.
// Results in the normal range (|x log2 e| <= 125.9) come from the direct form 2^RN(x log2 e) * (1 + e ln 2), e = the
// rounding error of the product (two FMAs) — one v_exp_f32, no rint / clamp / ldexp, 1.3 ulp; everything else (gradual
// underflow, overflow, Inf) from the ldexp form.  The choice is PER ELEMENT, so every kernel (the packed handlers of the
// threaded interpreter take the ldexp branch wave-uniformly and select per element) returns the same bits.
constexpr float DE_EXP_DIRECT_BOUND_T = 125.9f;
#define DE_LN2_HI 0x1.62e430p-1f  // ln 2 = HI + LO to 2^-50: t * (HI + LO) misses t ln 2 by < 2^-43 for |t| <= 126
#define DE_LN2_LO -0x1.05c610p-29f
__device__ __forceinline__ float fast_exp_ldexp_f32(float x) {
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -105.0f), 89.0f);
    const float k = __builtin_rintf(xc * 0x1.715476p+0f);
    float r = __builtin_fmaf(xc, 0x1.715476p+0f, -k);
    r = __builtin_fmaf(xc, 0x1.4ae0c0p-26f, r);
    const float e = __builtin_amdgcn_exp2f(r);
    const float y = __builtin_amdgcn_ldexpf(e, (int)k);
    return x != x ? x : y;
}
__device__ __forceinline__ float fast_exp_f32(float x) {
    const float t = x * 0x1.715476p+0f;
    if (__builtin_fabsf(t) > DE_EXP_DIRECT_BOUND_T) return fast_exp_ldexp_f32(x);
    float r = __builtin_fmaf(-t, DE_LN2_HI, x); // x - t ln 2 = (x log2 e - t) ln 2: what 2^t still lacks, in natural units
    r = __builtin_fmaf(-t, DE_LN2_LO, r);
    const float v = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(v, r, v); // NaN: t is NaN, the comparison above false, v NaN
}

// Two elements at a time (v_pk_mul/v_pk_fma for the reduction; exp2/ldexp/rint stay per element).
__device__ __forceinline__ DeF2 fast_exp_f32x2(DeF2 x) {
    // Only the integer part k is clamped (to the ldexp range): x itself stays as it is, so NaN flows through
    // the FMAs, +-Inf and huge arguments give r = +-Inf / far outside [-1/2, 1/2] exactly where the result is
    // Inf or 0 anyway, and no NaN fix-up select is needed (2 VALU per element less than clamping x).
    const DeF2 t = x * DE_F2(0x1.715476p+0f);
    const DeF2 k = {__builtin_amdgcn_fmed3f(__builtin_rintf(t[0]), -152.0f, 130.0f),
                    __builtin_amdgcn_fmed3f(__builtin_rintf(t[1]), -152.0f, 130.0f)};
    DeF2 r = __builtin_elementwise_fma(x, DE_F2(0x1.715476p+0f), -k);
    r = __builtin_elementwise_fma(x, DE_F2(0x1.4ae0c0p-26f), r);
    DeF2 y;
    y[0] = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(r[0]), (int)k[0]);
    y[1] = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(r[1]), (int)k[1]);
    return y;
}

// ---- DE_OPT_TURBO: relaxed-accuracy Float32 operators (the reference's `turbo` = LoopVectorization/SLEEF path,
// ext/DynamicExpressionsLoopVectorizationExt.jl:24-43, whose results drift from Base's by design,
// test/test_supposition_consistency.jl:106-108).  Contract: <= 1e-6 relative on ordinary arguments — an order of
// magnitude inside north_star's 1e-5 — with these documented domain edges, all far from symbolic-regression data:
//   cos/sin : two-term pi (48 bits): absolute error <= 2e-7 up to |x| = 1e5 (the relative error next to a zero of the
//             function grows as |x| * 1e-15 / distance); beyond 1e5, and for Inf, the OCML path under a wave-uniform
//             branch as in the exact mode (cos(exp(exp(x))) is common in random trees, and a value outside [-1, 1]
//             would change flags downstream); no exact-extremum select (cos(0) may be 1 - 2^-24, so `x ^ cos(0)`
//             with x < 0 is NaN).
//   exp     : results below 2^-126 flush to 0 (v_exp_f32 has no denormal results); an overflowing result is Inf or NaN
//             (non-finite either way) and exp(-Inf) is NaN instead of 0 — visible only with early_exit=false, the
//             argument of exp is validity-tested otherwise.
//   /       : x * v_rcp_f32(y), <= 1.5 ulp; |y| < 2^-126 divides like 0 and |y| >= 2^126 like Inf (v_rcp_f32 flushes
//             denormal inputs and results).
// Flags: identical to the exact mode except through those edges (a value that is Inf/NaN/0 only in one mode).
constexpr float DE_TURBO_TRIG_BOUND = 1.0e5f; // = DE_TRIG_FAST_BOUND: beyond it x/pi rounded in Float32 no longer picks the right n
#define DE_TURBO_S0 DE_TRIG_S0 // the exact mode's polynomial (degree 9)
#define DE_TURBO_S1 DE_TRIG_S1
#define DE_TURBO_S2 DE_TRIG_S2
#define DE_TURBO_S3 DE_TRIG_S3
template <bool SIN> __device__ __forceinline__ DeF2 turbo_trig_f32x2(DeF2 x, DeF2 &m) {
    const DeF2 t = SIN ? x * DE_F2(DE_TRIG_INV_PI) : __builtin_elementwise_fma(x, DE_F2(DE_TRIG_INV_PI), DE_F2(0.5f));
    const DeF2 kk = t + DE_F2(DE_TRIG_MAGIC);
    const DeF2 n = kk - DE_F2(DE_TRIG_MAGIC);
    const DeF2 m_ = SIN ? n : n - DE_F2(0.5f);
    m = n; // out: the multiple, for the caller's range pre-test
    DeF2 r = __builtin_elementwise_fma(-m_, DE_F2(DE_TRIG_P1), x);
    r = __builtin_elementwise_fma(-m_, DE_F2(DE_TRIG_P2), r);
    const DeF2 z = r * r;
    DeF2 p = __builtin_elementwise_fma(z, DE_F2(DE_TURBO_S3), DE_F2(DE_TURBO_S2));
    p = __builtin_elementwise_fma(z, p, DE_F2(DE_TURBO_S1));
    p = __builtin_elementwise_fma(z, p, DE_F2(DE_TURBO_S0));
    DeF2 s = __builtin_elementwise_fma(r * z, p, r);
    if constexpr (SIN) { s[0] = __builtin_copysignf(s[0], r[0]); s[1] = __builtin_copysignf(s[1], r[1]); } // sin(-0) = -0, as in the exact mode
    // (-1)^n: the low mantissa bit of the magic sum is the parity of n; adding it at bit 31 flips the sign (v_lshl_add_u32)
    DeF2 y;
    y[0] = __uint_as_float((__float_as_uint(kk[0]) << 31) + __float_as_uint(s[0]));
    y[1] = __uint_as_float((__float_as_uint(kk[1]) << 31) + __float_as_uint(s[1]));
    return y;
}
// exp(x) = 2^t * (1 + r), r = x - t ln 2: t = RN(x log2 e) goes to v_exp_f32, r is what the rounded product left out
__device__ __forceinline__ DeF2 turbo_exp_f32x2(DeF2 x, DeF2 t);
__device__ __forceinline__ DeF2 turbo_exp_f32x2(DeF2 x) { return turbo_exp_f32x2(x, x * DE_F2(0x1.715476p+0f)); }
__device__ __forceinline__ DeF2 turbo_exp_f32x2(DeF2 x, DeF2 t) { // t = x * float(log2 e)
    DeF2 r = __builtin_elementwise_fma(-t, DE_F2(DE_LN2_HI), x); // x - t ln 2 in two FMAs (the first cancels to ~|x| 2^-24: exact)
    r = __builtin_elementwise_fma(-t, DE_F2(DE_LN2_LO), r);
    const DeF2 v = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    return __builtin_elementwise_fma(v, r, v); // overflow: Inf or NaN (both non-finite); exp(-Inf) = NaN, not 0 (r = Inf - Inf)
}
__device__ __forceinline__ DeF2 turbo_div_f32x2(DeF2 n, DeF2 d) {
    const DeF2 y = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return n * y;
}

// A class id outside [class_base, class_base + n_classes) is a caller error (the reference asserts on the host,
// src/ParametricExpression.jl:378-379, and so do the shims); the device clamps the id so that such a call stays
// memory-safe (the values of those samples are then those of the first / last class).
__device__ __forceinline__ int64_t clamp_class(int64_t cl, int64_t n_classes) {
    return cl < 0 ? 0 : (cl >= n_classes ? n_classes - 1 : cl);
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
#ifdef DE_STUB_WAVE_SUM // MEASUREMENT ONLY (wrong results): what the kernels would cost without their wave reductions (profiles/r5_rev_bound.md)
    return v;
#endif
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
__device__ __forceinline__ double wave_sum_to_lane63(double v, int lane) { // lane = this thread's lane (a handler derives it from its LDS address: no work-item id input)
    _Pragma("unroll") for (int m = 1; m < 64; m <<= 1) {
        const double o = __shfl_up(v, m, 64);
        if (lane >= m) v += o;
    }
    return v;
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) { return wave_sum_to_lane63(v, (int)(threadIdx.x & 63)); }
__device__ __forceinline__ float wave_sum_to_lane63(float v, int) { return wave_sum_to_lane63(v); }


} // namespace de
