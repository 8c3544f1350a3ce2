/*
 * de_opcodes.h — the closed operator set shared by the C-ABI (de_hip.h), the
 * HIP kernels, the host flatteners (Julia shim / Python mirror) and the CPU
 * oracle.  VERSIONED: bump DE_OPCODE_TABLE_VERSION when an id changes.
 *
 * Why a closed set: the reference's OperatorEnum holds *arbitrary Julia
 * functions* indexed [degree][op_idx] (reference src/OperatorEnum.jl:14-49); a
 * GPU kernel cannot call a Julia closure, so the boundary maps each function
 * (by identity/name) onto one of these ids and reports DE_ERR_UNSUPPORTED_OP
 * for anything else (the Julia shim then keeps the reference CPU path).
 *
 * Coverage = every scalar operator the reference itself pre-declares or
 * exercises on the numeric path (SURVEY.md §8b "Opcode set"):
 *   unary  : src/OperatorEnumConstruction.jl:585-591, test/test_params.jl:7-29,
 *            test/test_evaluation.jl:373 (unary minus), test/test_tree_construction.jl:11
 *   binary : src/OperatorEnumConstruction.jl:587, test/test_expressions.jl:453-492,
 *            test/test_params.jl:14-15, test/test_derivatives.jl:12
 *   ternary: test/test_supposition_consistency.jl:20 (fma, clamp, +, max)
 */
#ifndef DE_OPCODES_H
#define DE_OPCODES_H

#define DE_OPCODE_TABLE_VERSION 1

/* Leaf kinds: value of de_tape_node_t.op when degree == 0. */
enum de_leaf_kind {
    DE_LEAF_CONST = 0,   /* arg = constant slot (0-based, depth-first order)      */
    DE_LEAF_FEATURE = 1, /* arg = 0-based feature row of X                         */
    DE_LEAF_PARAM = 2,   /* arg = 0-based parameter row (ParametricExpression)     */
    /* CSE tapes only (de_program_create_cse): the value of shared subtree `arg`, defined earlier in the
     * same tape by a DE_OP_SHARE marker (GraphNode sharing, src/Node.jl:138-166).                        */
    DE_LEAF_SHARED = 3
};
/* CSE tapes only: a degree-1 pseudo node (op = DE_OP_SHARE, arg = share id 0..15) that follows the
 * post-order slice of a shared subtree at its FIRST occurrence: "this value is shared subtree `arg`".
 * It is transparent (the value passes through) and costs no instruction of its own. */
#define DE_OP_SHARE 0xFE

/* Operator ids: value of de_tape_node_t.op when degree >= 1. */
enum de_opcode {
    DE_OP_INVALID = 0,

    /* ---- degree 1 ------------------------------------------------------- */
    DE_U_NEG = 1,     /* -(x)                                                  */
    DE_U_ABS,         /* abs                                                   */
    DE_U_SQUARE,      /* x*x            (test_params.jl:25)                    */
    DE_U_CUBE,        /* (x*x)*x        (test_params.jl:26)                    */
    DE_U_RELU,        /* x<0 ? 0 : x    (test_params.jl:12)                    */
    DE_U_SIGN,        /* sign: ±1, keeps ±0 and NaN                            */
    DE_U_ROUND,       /* round half to even (Julia default RoundNearest)       */
    DE_U_FLOOR,
    DE_U_CEIL,
    DE_U_INV,         /* inv(x) = 1/x                                          */
    DE_U_SQRT,        /* NaN for x<0 (Julia throws DomainError)                */
    DE_U_CBRT,
    DE_U_EXP,
    DE_U_EXP2,
    DE_U_LOG,         /* NaN for x<0 (Julia throws DomainError)                */
    DE_U_LOG2,
    DE_U_LOG10,
    DE_U_LOG1P,
    DE_U_SIN,
    DE_U_COS,
    DE_U_TAN,
    DE_U_SINH,
    DE_U_COSH,
    DE_U_TANH,
    DE_U_ASIN,
    DE_U_ACOS,
    DE_U_ATAN,
    DE_U_ASINH,
    DE_U_ACOSH,
    DE_U_ATANH,
    DE_U_SAFE_LOG,    /* x<=0 ? NaN : log(x)    (test_params.jl:7)             */
    DE_U_SAFE_LOG2,
    DE_U_SAFE_LOG10,
    DE_U_SAFE_LOG1P,  /* x<=-1 ? NaN : log1p(x)                                */
    DE_U_SAFE_SQRT,   /* x<0 ? NaN : sqrt(x)                                   */
    DE_U_SAFE_ACOSH,  /* x<1 ? NaN : acosh(x)                                  */
    DE_U_COS2,        /* custom_cos(x) = cos(x)^2 = cos(x)*cos(x) (test_params.jl:29) */
    DE_U_GAMMA,       /* SpecialFunctions.gamma (test_tree_construction.jl:11) */
    DE_U_LAST_,

    /* ---- degree 2 ------------------------------------------------------- */
    DE_B_ADD = 64,
    DE_B_SUB,         /* also test_params.jl `sub`                             */
    DE_B_MUL,
    DE_B_DIV,
    DE_B_POW,         /* x^y; NaN where Julia throws DomainError               */
    DE_B_MAX,         /* Julia max: NaN-propagating, max(-0,+0)=+0             */
    DE_B_MIN,
    DE_B_MOD,         /* Julia mod: result has sign of y                       */
    DE_B_REM,         /* Julia rem = C fmod                                    */
    DE_B_GREATER,     /* x>y ? 1 : 0 in T   (test_params.jl:15)                */
    DE_B_POW_ABS2,    /* exp(y*log(abs(x))) (test_derivatives.jl:12)           */
    DE_B_LAST_,

    /* ---- degree 3 (n-ary nodes, src/Evaluate.jl:428-487) ----------------- */
    DE_T_FMA = 128,   /* fma(x,y,z), single rounding                           */
    DE_T_CLAMP,       /* clamp(x, lo, hi) = x>hi ? hi : (x<lo ? lo : x)        */
    DE_T_ADD3,        /* (x+y)+z                                               */
    DE_T_MAX3,        /* max(max(x,y),z)                                       */
    DE_T_LAST_
};

#endif /* DE_OPCODES_H */
