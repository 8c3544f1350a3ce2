/*
 * de_oracle_ops.h — scalar operator semantics of the CPU oracle (TEST
 * INFRASTRUCTURE, not product code).  Included twice by de_oracle.c, once with
 * T=float and once with T=double.
 *
 * The reference's operators are Julia Base functions (a pure-Julia libm that is
 * NOT under /root/reference — third-party arithmetic).  What is
 * restated here:
 *   - IEEE-exact operators (+ - * / sqrt abs neg max min fma floor ceil round
 *     sign rem mod) are computed natively in T: bit-identical to Julia given
 *     -ffp-contract=off.
 *   - transcendental operators are evaluated in the next wider format (double
 *     for T=float, x87 long double for T=double) and rounded once to T, i.e. a
 *     (practically) correctly rounded value.  Julia Base documents <1 ulp for
 *     its own implementations, so the reference's value is within 1 ulp of
 *     this one.  Bit-level parity of transcendentals is UNPINNED (no Julia in
 *     this container) — see the header of de_oracle.c.
 *   - where Julia throws DomainError (sqrt/log of a negative number, acosh(x<1),
 *     negative^fractional ...) the value here is NaN; a C ABI cannot throw.
 *
 * Derivative rules restate ChainRules.jl's @scalar_rule table as used through
 * Zygote (reference ext/DynamicExpressionsZygoteExt.jl:12-15: `nothing`
 * partials become zero).  ChainRules is third-party and unpinned; tie/edge
 * conventions chosen here are listed in DESIGN.md §"Derivative conventions".
 */

#ifndef OT
#error "define OT (float|double), OW (wider type), OSUF (f|/**/), WSUF (/**/|l)"
#endif

#define OCAT_(a, b) a##b
#define OCAT(a, b) OCAT_(a, b)
#define ON(name) OCAT(name, ONAME)           /* per-type symbol suffix  */
#define WF(fn) OCAT(fn, WSUF)                /* wide libm function       */
#define NF(fn) OCAT(fn, OSUF)                /* native-T libm function   */
#define WIDE1(fn, x) ((OT)WF(fn)((OW)(x)))
#define WIDE2(fn, x, y) ((OT)WF(fn)((OW)(x), (OW)(y)))

static inline int ON(o_isvalid)(OT x) { return isfinite(x); }

/* Julia max/min: NaN-propagating, signed zeros ordered (-0 < +0). */
static inline OT ON(o_jlmax)(OT x, OT y) {
    if (isnan(x)) return x;
    if (isnan(y)) return y;
    if (y > x || (signbit(x) && !signbit(y))) return y;
    return x;
}
static inline OT ON(o_jlmin)(OT x, OT y) {
    if (isnan(x)) return x;
    if (isnan(y)) return y;
    if (y < x || (signbit(y) && !signbit(x))) return y;
    return x;
}
/* Julia mod(x,y) for floats (Base float.jl): r = rem(x,y); r==0 -> copysign(r,y);
 * sign(r) != sign(y) -> r+y; else r. */
static inline OT ON(o_jlmod)(OT x, OT y) {
    OT r = NF(fmod)(x, y);
    if (r == 0) return NF(copysign)(r, y);
    if ((r > 0) != (y > 0)) return r + y;
    return r;
}
static inline OT ON(o_sign)(OT x) { return x > 0 ? (OT)1 : (x < 0 ? (OT)-1 : x); }

/* digamma for the gamma derivative (SpecialFunctions.digamma; unpinned). */
static inline OW ON(o_digamma)(OW x) {
    OW r = 0;
    if (x <= 0) {
        if (x == WF(floor)(x)) return (OW)NAN;
        /* reflection: psi(1-x) - psi(x) = pi*cot(pi*x) */
        const OW pi = (OW)3.141592653589793238462643383279502884L;
        return ON(o_digamma)(1 - x) - pi / WF(tan)(pi * x);
    }
    while (x < 10) { r -= 1 / x; x += 1; }
    OW f = 1 / (x * x);
    OW t = f * ((OW)-1 / 12 + f * ((OW)1 / 120 + f * ((OW)-1 / 252 + f * ((OW)1 / 240 + f * ((OW)-1 / 132)))));
    return r + WF(log)(x) - (OW)0.5 / x + t;
}

static inline OT ON(o_unary)(int op, OT x) {
    switch (op) {
    case DE_U_NEG: return -x;
    case DE_U_ABS: return NF(fabs)(x);
    case DE_U_SQUARE: return x * x;
    case DE_U_CUBE: return (x * x) * x;
    case DE_U_RELU: return x < 0 ? (OT)0 : x;
    case DE_U_SIGN: return ON(o_sign)(x);
    case DE_U_ROUND: return NF(rint)(x);
    case DE_U_FLOOR: return NF(floor)(x);
    case DE_U_CEIL: return NF(ceil)(x);
    case DE_U_INV: return (OT)1 / x;
    case DE_U_SQRT: return NF(sqrt)(x);
    case DE_U_CBRT: return WIDE1(cbrt, x);
    case DE_U_EXP: return WIDE1(exp, x);
    case DE_U_EXP2: return WIDE1(exp2, x);
    case DE_U_LOG: return WIDE1(log, x);
    case DE_U_LOG2: return WIDE1(log2, x);
    case DE_U_LOG10: return WIDE1(log10, x);
    case DE_U_LOG1P: return WIDE1(log1p, x);
    case DE_U_SIN: return WIDE1(sin, x);
    case DE_U_COS: return WIDE1(cos, x);
    case DE_U_TAN: return WIDE1(tan, x);
    case DE_U_SINH: return WIDE1(sinh, x);
    case DE_U_COSH: return WIDE1(cosh, x);
    case DE_U_TANH: return WIDE1(tanh, x);
    case DE_U_ASIN: return WIDE1(asin, x);
    case DE_U_ACOS: return WIDE1(acos, x);
    case DE_U_ATAN: return WIDE1(atan, x);
    case DE_U_ASINH: return WIDE1(asinh, x);
    case DE_U_ACOSH: return WIDE1(acosh, x);
    case DE_U_ATANH: return WIDE1(atanh, x);
    case DE_U_SAFE_LOG: return x <= 0 ? (OT)NAN : WIDE1(log, x);
    case DE_U_SAFE_LOG2: return x <= 0 ? (OT)NAN : WIDE1(log2, x);
    case DE_U_SAFE_LOG10: return x <= 0 ? (OT)NAN : WIDE1(log10, x);
    case DE_U_SAFE_LOG1P: return x <= -1 ? (OT)NAN : WIDE1(log1p, x);
    case DE_U_SAFE_SQRT: return x < 0 ? (OT)NAN : NF(sqrt)(x);
    case DE_U_SAFE_ACOSH: return x < 1 ? (OT)NAN : WIDE1(acosh, x);
    case DE_U_COS2: { OT c = WIDE1(cos, x); return c * c; }
    case DE_U_GAMMA: return WIDE1(tgamma, x);
    default: return (OT)NAN;
    }
}

static inline OT ON(o_binary)(int op, OT x, OT y) {
    switch (op) {
    case DE_B_ADD: return x + y;
    case DE_B_SUB: return x - y;
    case DE_B_MUL: return x * y;
    case DE_B_DIV: return x / y;
    case DE_B_POW: return WIDE2(pow, x, y);
    case DE_B_MAX: return ON(o_jlmax)(x, y);
    case DE_B_MIN: return ON(o_jlmin)(x, y);
    case DE_B_MOD: return ON(o_jlmod)(x, y);
    case DE_B_REM: return NF(fmod)(x, y);
    case DE_B_GREATER: return x > y ? (OT)1 : (OT)0;
    case DE_B_POW_ABS2: {
        /* exp(y * log(abs(x))) evaluated step by step in T, as the Julia closure does */
        OT l = WIDE1(log, NF(fabs)(x));
        OT m = y * l;
        return WIDE1(exp, m);
    }
    default: return (OT)NAN;
    }
}

static inline OT ON(o_ternary)(int op, OT x, OT y, OT z) {
    switch (op) {
    case DE_T_FMA: return NF(fma)(x, y, z);
    case DE_T_CLAMP: return x > z ? z : (x < y ? y : x);
    case DE_T_ADD3: return (x + y) + z;
    case DE_T_MAX3: return ON(o_jlmax)(ON(o_jlmax)(x, y), z);
    default: return (OT)NAN;
    }
}

/* ---- partial derivatives: g[i] = d op / d arg_i, Zygote `nothing` -> 0 ---- */
static inline void ON(o_unary_grad)(int op, OT x, OT *g) {
    switch (op) {
    case DE_U_NEG: g[0] = (OT)-1; break;
    case DE_U_ABS: g[0] = ON(o_sign)(x); break;
    case DE_U_SQUARE: g[0] = x + x; break;
    case DE_U_CUBE: g[0] = ((OT)3 * x) * x; break;
    case DE_U_RELU: g[0] = x < 0 ? (OT)0 : (OT)1; break;
    case DE_U_SIGN: case DE_U_ROUND: case DE_U_FLOOR: case DE_U_CEIL: g[0] = (OT)0; break;
    case DE_U_INV: { OT o = (OT)1 / x; g[0] = -(o * o); break; }
    case DE_U_SQRT: { OT o = NF(sqrt)(x); g[0] = (OT)1 / ((OT)2 * o); break; }
    case DE_U_CBRT: { OT o = WIDE1(cbrt, x); g[0] = (OT)1 / ((OT)3 * (o * o)); break; }
    case DE_U_EXP: g[0] = WIDE1(exp, x); break;
    case DE_U_EXP2: g[0] = WIDE1(exp2, x) * (OT)0.693147180559945309417232121458176568L; break;
    case DE_U_LOG: g[0] = (OT)1 / x; break;
    case DE_U_LOG2: g[0] = ((OT)1 / x) / (OT)0.693147180559945309417232121458176568L; break;
    case DE_U_LOG10: g[0] = ((OT)1 / x) / (OT)2.302585092994045684017991454684364208L; break;
    case DE_U_LOG1P: g[0] = (OT)1 / (x + (OT)1); break;
    case DE_U_SIN: g[0] = WIDE1(cos, x); break;
    case DE_U_COS: g[0] = -WIDE1(sin, x); break;
    case DE_U_TAN: { OT o = WIDE1(tan, x); g[0] = (OT)1 + o * o; break; }
    case DE_U_SINH: g[0] = WIDE1(cosh, x); break;
    case DE_U_COSH: g[0] = WIDE1(sinh, x); break;
    case DE_U_TANH: { OT o = WIDE1(tanh, x); g[0] = (OT)1 - o * o; break; }
    case DE_U_ASIN: g[0] = (OT)1 / NF(sqrt)((OT)1 - x * x); break;
    case DE_U_ACOS: g[0] = -((OT)1 / NF(sqrt)((OT)1 - x * x)); break;
    case DE_U_ATAN: g[0] = (OT)1 / ((OT)1 + x * x); break;
    case DE_U_ASINH: g[0] = (OT)1 / NF(sqrt)(x * x + (OT)1); break;
    case DE_U_ACOSH: g[0] = (OT)1 / (NF(sqrt)(x - (OT)1) * NF(sqrt)(x + (OT)1)); break;
    case DE_U_ATANH: g[0] = (OT)1 / ((OT)1 - x * x); break;
    case DE_U_SAFE_LOG: g[0] = x <= 0 ? (OT)0 : (OT)1 / x; break;
    case DE_U_SAFE_LOG2: g[0] = x <= 0 ? (OT)0 : ((OT)1 / x) / (OT)0.693147180559945309417232121458176568L; break;
    case DE_U_SAFE_LOG10: g[0] = x <= 0 ? (OT)0 : ((OT)1 / x) / (OT)2.302585092994045684017991454684364208L; break;
    case DE_U_SAFE_LOG1P: g[0] = x <= -1 ? (OT)0 : (OT)1 / (x + (OT)1); break;
    case DE_U_SAFE_SQRT: if (x < 0) g[0] = (OT)0; else { OT o = NF(sqrt)(x); g[0] = (OT)1 / ((OT)2 * o); } break;
    case DE_U_SAFE_ACOSH: g[0] = x < 1 ? (OT)0 : (OT)1 / (NF(sqrt)(x - (OT)1) * NF(sqrt)(x + (OT)1)); break;
    case DE_U_COS2: { OT c = WIDE1(cos, x); OT s = WIDE1(sin, x); g[0] = ((OT)2 * c) * (-s); break; }
    case DE_U_GAMMA: { OW o = WF(tgamma)((OW)x); g[0] = (OT)(o * ON(o_digamma)((OW)x)); break; }
    default: g[0] = (OT)NAN; break;
    }
}

static inline void ON(o_binary_grad)(int op, OT x, OT y, OT *g) {
    switch (op) {
    case DE_B_ADD: g[0] = (OT)1; g[1] = (OT)1; break;
    case DE_B_SUB: g[0] = (OT)1; g[1] = (OT)-1; break;
    case DE_B_MUL: g[0] = y; g[1] = x; break;
    case DE_B_DIV: { OT o = x / y; g[0] = (OT)1 / y; g[1] = -(o / y); break; }
    case DE_B_POW: {
        OT o = WIDE2(pow, x, y);
        /* ChainRules _pow_grad_x / _pow_grad_p (real case) */
        if (x != 0 || y < 0) {
            if (isinf(x) && y == 1) g[0] = (OT)1; else g[0] = (o * y) / x;
        } else if (y == 1) g[0] = (OT)1;
        else if (y == 0 || y > 1) g[0] = (OT)0;
        else g[0] = (OT)INFINITY;
        if (x != 0) g[1] = o * WIDE1(log, NF(fabs)(x));
        else if (y > 0) g[1] = (OT)0;
        else g[1] = (OT)NAN;
        break;
    }
    case DE_B_MAX: { int gt = x > y; g[0] = gt ? (OT)1 : (OT)0; g[1] = gt ? (OT)0 : (OT)1; break; }
    case DE_B_MIN: { int gt = x > y; g[0] = gt ? (OT)0 : (OT)1; g[1] = gt ? (OT)1 : (OT)0; break; }
    case DE_B_MOD: {
        OT u = x / y; int isint = (u == NF(floor)(u)) && isfinite(u);
        g[0] = isint ? (OT)NAN : (OT)1; g[1] = isint ? (OT)NAN : -NF(floor)(u); break;
    }
    case DE_B_REM: {
        OT u = x / y; int isint = (u == NF(floor)(u)) && isfinite(u);
        g[0] = isint ? (OT)NAN : (OT)1; g[1] = isint ? (OT)NAN : -NF(trunc)(u); break;
    }
    case DE_B_GREATER: g[0] = (OT)0; g[1] = (OT)0; break;
    case DE_B_POW_ABS2: {
        OT a = NF(fabs)(x); OT l = WIDE1(log, a); OT m = y * l; OT o = WIDE1(exp, m);
        g[0] = ((o * y) * ((OT)1 / a)) * ON(o_sign)(x);
        g[1] = o * l;
        break;
    }
    default: g[0] = g[1] = (OT)NAN; break;
    }
}

static inline void ON(o_ternary_grad)(int op, OT x, OT y, OT z, OT *g) {
    switch (op) {
    case DE_T_FMA: g[0] = y; g[1] = x; g[2] = (OT)1; break;
    case DE_T_CLAMP: /* clamp(x, lo=y, hi=z) */
        g[0] = (x > z || x < y) ? (OT)0 : (OT)1;
        g[1] = (x > z) ? (OT)0 : (x < y ? (OT)1 : (OT)0);
        g[2] = (x > z) ? (OT)1 : (OT)0;
        break;
    case DE_T_ADD3: g[0] = g[1] = g[2] = (OT)1; break;
    case DE_T_MAX3: {
        OT m = ON(o_jlmax)(x, y); int gt1 = x > y; int gt2 = m > z;
        g[0] = (gt2 && gt1) ? (OT)1 : (OT)0; g[1] = (gt2 && !gt1) ? (OT)1 : (OT)0; g[2] = gt2 ? (OT)0 : (OT)1;
        break;
    }
    default: g[0] = g[1] = g[2] = (OT)NAN; break;
    }
}

#undef WIDE1
#undef WIDE2
#undef WF
#undef NF
