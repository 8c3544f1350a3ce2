/*
 * de_oracle.c — CPU ORACLE for the DynamicExpressions.jl hot path.
 *
 * *** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***  Only tests/, the
 * cpu_baseline leg of bench.py and __graft_entry__.smoke() may load this
 * library, and only as the checker / the reported CPU baseline.  Nothing under
 * dynamicexpressions.jl_amd/ imports, links or calls it; the product path fails
 * loudly when the HIP library is missing.
 *
 * What it is: a plain-C restatement of the reference ALGORITHM — the recursive
 * evaluator of src/Evaluate.jl with its fused 2/3-node kernels, constant
 * folding and early-exit validity tests; the forward-mode gradient of
 * src/EvaluateDerivative.jl; the ParametricExpression wrapper of
 * src/ParametricExpression.jl:371-390 — one full-array pass per (fused) node,
 * exactly how the reference spends its time.  Each function cites the
 * reference lines it follows (see de_oracle_impl.h).
 *
 * Pinning status:
 *   - The reference is 100 % Julia and cannot be built or run in this
 *     container (no julia binary, no network) — there is no oracle/_ref.
 *   - The oracle is pinned against every known-answer / golden vector the
 *     reference's own tests and docs hold for this path; they are transcribed
 *     in tests/golden/reference_known_answers.json (each with its file:line)
 *     and checked by tests/test_oracle_golden.py.
 *   - Values of IEEE-exact operators are therefore fully pinned (bit-exact).
 *     Values of transcendental operators come from Julia Base's libm and
 *     derivative rules from Zygote/ChainRules — third-party code that is not
 *     under /root/reference; they are pinned only at the reference tests'
 *     tolerances.  Bit-level (1-ulp) parity of transcendentals and the
 *     tie/edge conventions of the derivative rules are PARITY UNPINNED.
 *
 * Build: see oracle/Makefile (gcc -O3 -march=native -ffp-contract=off).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/de_hip.h"

/* Parsed tree node (children by index) */
typedef struct onode {
    uint8_t degree, op;
    uint16_t arg;
    int child[3];
    int is_const; /* is_constant(subtree): no feature/param leaf below (src/NodeUtils.jl:73) */
} onode;

/* Post-order tape -> indexed tree.  Returns root index or a negative error:
 * -2 malformed tape, -3 unknown opcode, -6 index out of range. */
static int o_parse(const de_tape_node_t *tape, int64_t n, int64_t n_consts, int F, int P,
                   onode **out) {
    if (n <= 0) return -2;
    onode *nodes = (onode *)calloc((size_t)n, sizeof(onode));
    int *stack = (int *)malloc((size_t)n * sizeof(int));
    int sp = 0;
    int err = 0;
    for (int64_t i = 0; i < n && !err; i++) {
        onode *nd = &nodes[i];
        nd->degree = tape[i].degree;
        nd->op = tape[i].op;
        nd->arg = tape[i].arg;
        if (nd->degree == 0) {
            if (nd->op == DE_LEAF_CONST) { nd->is_const = 1; if (nd->arg >= n_consts) err = -6; }
            else if (nd->op == DE_LEAF_FEATURE) { if (nd->arg >= F) err = -6; }
            else if (nd->op == DE_LEAF_PARAM) { if (nd->arg >= P) err = -6; }
            else err = -2;
        } else if (nd->degree <= 3) {
            int lo = nd->degree == 1 ? DE_U_NEG : (nd->degree == 2 ? DE_B_ADD : DE_T_FMA);
            int hi = nd->degree == 1 ? DE_U_LAST_ : (nd->degree == 2 ? DE_B_LAST_ : DE_T_LAST_);
            if (nd->op < lo || nd->op >= hi) { err = -3; break; }
            if (sp < nd->degree) { err = -2; break; }
            nd->is_const = 1;
            for (int k = nd->degree - 1; k >= 0; k--) {
                nd->child[k] = stack[--sp];
                nd->is_const &= nodes[nd->child[k]].is_const;
            }
        } else err = -2;
        stack[sp++] = (int)i;
    }
    if (!err && sp != 1) err = -2;
    int root = err ? err : stack[0];
    free(stack);
    if (err) { free(nodes); return err; }
    *out = nodes;
    return root;
}

/* ---- float instantiation ---- */
#define OT float
#define OW double
#define OSUF f
#define WSUF
#define ONAME _f32
#include "de_oracle_ops.h"
#include "de_oracle_impl.h"
#undef OT
#undef OW
#undef OSUF
#undef WSUF
#undef ONAME
#undef ON
#undef OCAT
#undef OCAT_

/* ---- double instantiation ---- */
#define OT double
#define OW long double
#define OSUF
#define WSUF l
#define ONAME _f64
#include "de_oracle_ops.h"
#include "de_oracle_impl.h"

/* Scalar probes so tests can pin single operators / partials directly. */
float de_oracle_unary_f32(int op, float x) { return o_unary_f32(op, x); }
double de_oracle_unary_f64(int op, double x) { return o_unary_f64(op, x); }
float de_oracle_binary_f32(int op, float x, float y) { return o_binary_f32(op, x, y); }
double de_oracle_binary_f64(int op, double x, double y) { return o_binary_f64(op, x, y); }
double de_oracle_ternary_f64(int op, double x, double y, double z) { return o_ternary_f64(op, x, y, z); }
float de_oracle_ternary_f32(int op, float x, float y, float z) { return o_ternary_f32(op, x, y, z); }
void de_oracle_unary_grad_f64(int op, double x, double *g) { o_unary_grad_f64(op, x, g); }
void de_oracle_binary_grad_f64(int op, double x, double y, double *g) { o_binary_grad_f64(op, x, y, g); }
void de_oracle_unary_grad_f32(int op, float x, float *g) { o_unary_grad_f32(op, x, g); }
void de_oracle_binary_grad_f32(int op, float x, float y, float *g) { o_binary_grad_f32(op, x, y, g); }
int de_oracle_abi(void) { return DE_HIP_ABI_VERSION; }
