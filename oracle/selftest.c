/*
 * selftest.c — the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`).
 * TEST INFRASTRUCTURE.  The oracle is the checker of every parity test; this driver runs its entry points
 * (eval with all option sets, gradient in the three modes, diff, parametric eval) over a seeded stream of random
 * well-formed tapes — every opcode of include/de_opcodes.h, sizes 1..40 nodes, N = 0, 1, 7, 64, ragged leading
 * dimension — plus malformed tapes that must be rejected, so that an out-of-bounds access, a leak of the node
 * arena or signed overflow in the restatement shows up here rather than as a wrong "expected" value in a test.
 * Known answers: README tree x1*cos(x2-3.2) (README.md:30-39) and d/dx of 0.5*x1+cos(x2-0.2) (docs/src/eval.md:166-217).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/de_hip.h"

int de_oracle_eval_f64(const de_tape_node_t *, int64_t, const double *, int64_t, const double *, int32_t, int64_t, int64_t, uint32_t, int32_t, double *, uint8_t *);
int de_oracle_eval_f32(const de_tape_node_t *, int64_t, const float *, int64_t, const float *, int32_t, int64_t, int64_t, uint32_t, int32_t, float *, uint8_t *);
int de_oracle_grad_f64(const de_tape_node_t *, int64_t, const double *, int64_t, const double *, int32_t, int64_t, int64_t, int32_t, int32_t, double *, double *, uint8_t *, int64_t *);
int de_oracle_diff_f64(const de_tape_node_t *, int64_t, const double *, int64_t, const double *, int32_t, int64_t, int64_t, int32_t, double *, double *, uint8_t *);
int de_oracle_eval_param_f64(const de_tape_node_t *, int64_t, const double *, int64_t, const double *, int32_t, int64_t, int64_t, const double *, int32_t,
                             int64_t, int64_t, const int32_t *, int32_t, uint32_t, int32_t, double *, uint8_t *);

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double urand(void) { return (double)(rnd() >> 11) / 9007199254740992.0 * 4.0 - 2.0; }

/* random post-order tape with n nodes over F features, P parameters; returns the number of constants */
static int make_tape(de_tape_node_t *t, int n, int F, int P, double *consts) {
    int depth = 0, nc = 0;
    for (int i = 0; i < n; i++) {
        int remaining = n - i; /* nodes still to emit, this one included */
        int deg;
        /* keep the stack reducible to one root: depth - (deg - 1) must stay >= 1 and reachable */
        if (depth == 0) deg = 0;
        else if (depth >= remaining) deg = depth >= 3 && (rnd() & 3) == 0 && depth - 2 >= remaining - 1 ? 3 : 2;
        else { deg = (int)(rnd() % 4); if (deg > depth) deg = depth; if (depth - (deg ? deg - 1 : -1) > remaining - 1) deg = depth >= 2 ? 2 : 1; }
        if (i == n - 1) deg = depth == 1 ? 1 : (depth == 2 ? 2 : 3);
        if (deg > depth) deg = 0;
        if (deg == 0) {
            int k = (int)(rnd() % (P > 0 ? 3 : 2));
            t[i].degree = 0;
            if (k == 0) { t[i].op = DE_LEAF_CONST; t[i].arg = (uint16_t)nc; consts[nc++] = urand(); }
            else if (k == 1) { t[i].op = DE_LEAF_FEATURE; t[i].arg = (uint16_t)(rnd() % (unsigned)F); }
            else { t[i].op = DE_LEAF_PARAM; t[i].arg = (uint16_t)(rnd() % (unsigned)P); }
            depth++;
        } else {
            int lo = deg == 1 ? DE_U_NEG : (deg == 2 ? DE_B_ADD : DE_T_FMA), hi = deg == 1 ? DE_U_LAST_ : (deg == 2 ? DE_B_LAST_ : DE_T_LAST_);
            t[i].degree = (uint8_t)deg;
            t[i].op = (uint8_t)(lo + (int)(rnd() % (unsigned)(hi - lo)));
            t[i].arg = 0;
            depth -= deg - 1;
        }
    }
    return depth == 1 ? nc : -1;
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest: %s failed at line %d\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
    /* known answers */
    {
        const de_tape_node_t t[] = {{0, DE_LEAF_FEATURE, 0}, {0, DE_LEAF_FEATURE, 1}, {0, DE_LEAF_CONST, 0}, {2, DE_B_SUB, 0}, {1, DE_U_COS, 0}, {2, DE_B_MUL, 0}};
        const double c[] = {3.2}, X[] = {1.5, 0.25, -0.5, 2.0};
        double out[2]; uint8_t ok = 0;
        CHECK(de_oracle_eval_f64(t, 6, c, 1, X, 2, 2, 2, DE_OPT_DEFAULT, 0, out, &ok) == 0 && ok);
        CHECK(fabs(out[0] - 1.5 * cos(0.25 - 3.2)) < 1e-15 && fabs(out[1] - -0.5 * cos(2.0 - 3.2)) < 1e-15);
        const de_tape_node_t g[] = {{0, DE_LEAF_CONST, 0}, {0, DE_LEAF_FEATURE, 0}, {2, DE_B_MUL, 0}, {0, DE_LEAF_FEATURE, 1}, {0, DE_LEAF_CONST, 1}, {2, DE_B_SUB, 0}, {1, DE_U_COS, 0}, {2, DE_B_ADD, 0}};
        const double gc[] = {0.5, 0.2}, GX[] = {1, 4, 2, 5, 3, 6};
        double y[3], d[6]; int64_t ng = 0;
        CHECK(de_oracle_grad_f64(g, 8, gc, 2, GX, 2, 3, 2, DE_GRAD_VARIABLE, 0, y, d, &ok, &ng) == 0 && ok && ng == 2);
        CHECK(fabs(d[0] - 0.5) < 1e-15 && fabs(d[1] - -sin(4 - 0.2)) < 1e-12 && fabs(d[5] - -sin(6 - 0.2)) < 1e-12);
    }
    /* malformed tapes are rejected, not read past */
    {
        const de_tape_node_t bad1[] = {{2, DE_B_ADD, 0}}, bad2[] = {{0, DE_LEAF_FEATURE, 9}}, bad3[] = {{0, DE_LEAF_FEATURE, 0}, {0, DE_LEAF_FEATURE, 0}};
        const de_tape_node_t bad4[] = {{0, DE_LEAF_FEATURE, 0}, {1, 250, 0}}, bad5[] = {{0, DE_LEAF_CONST, 3}};
        double X[4] = {0}, out[2]; uint8_t ok;
        CHECK(de_oracle_eval_f64(bad1, 1, NULL, 0, X, 2, 2, 2, 7, 0, out, &ok) < 0);
        CHECK(de_oracle_eval_f64(bad2, 1, NULL, 0, X, 2, 2, 2, 7, 0, out, &ok) < 0);
        CHECK(de_oracle_eval_f64(bad3, 2, NULL, 0, X, 2, 2, 2, 7, 0, out, &ok) < 0);
        CHECK(de_oracle_eval_f64(bad4, 2, NULL, 0, X, 2, 2, 2, 7, 0, out, &ok) < 0);
        CHECK(de_oracle_eval_f64(bad5, 1, X, 1, X, 2, 2, 2, 7, 0, out, &ok) < 0);
    }
    /* random tapes through every entry point */
    enum { F = 4, P = 3, C = 5, NMAX = 64, LD = 6 };
    static const int Ns[] = {0, 1, 7, 64};
    static const uint32_t opts[] = {7, 6, 1, 0, 15};
    de_tape_node_t tape[48];
    double consts[48], *X = malloc(sizeof(double) * LD * NMAX), *out = malloc(sizeof(double) * NMAX), *dout = malloc(sizeof(double) * NMAX);
    double *grad = malloc(sizeof(double) * (F + 48) * NMAX), params[P * C];
    float *Xf = malloc(sizeof(float) * LD * NMAX), *outf = malloc(sizeof(float) * NMAX), cf[48];
    int32_t classes[NMAX];
    long n_eval = 0, n_complete = 0;
    for (int iter = 0; iter < 4000; iter++) {
        const int n = 1 + (int)(rnd() % 40), with_params = iter & 1;
        const int nc = make_tape(tape, n, F, with_params ? P : 0, consts);
        if (nc < 0) continue;
        for (int i = 0; i < LD * NMAX; i++) { X[i] = urand(); Xf[i] = (float)X[i]; }
        for (int i = 0; i < nc; i++) cf[i] = (float)consts[i];
        for (int i = 0; i < P * C; i++) params[i] = urand();
        for (int i = 0; i < NMAX; i++) classes[i] = 1 + (int)(rnd() % C);
        const int64_t N = Ns[iter % 4];
        uint8_t ok = 2;
        if (with_params) {
            CHECK(de_oracle_eval_param_f64(tape, n, consts, nc, X, F, N, LD, params, P, C, P, classes, 1, opts[iter % 5], iter & 2, out, &ok) == 0);
        } else {
            CHECK(de_oracle_eval_f64(tape, n, consts, nc, X, F, N, LD, opts[iter % 5], iter & 2, out, &ok) == 0);
            CHECK(de_oracle_eval_f32(tape, n, cf, nc, Xf, F, N, LD, opts[(iter + 1) % 5], 1, outf, &ok) == 0);
            int has3 = 0;
            for (int i = 0; i < n; i++) has3 |= tape[i].degree == 3;
            int64_t ng = -1;
            CHECK(de_oracle_grad_f64(tape, n, consts, nc, X, F, N, LD, iter % 3, iter & 4, out, grad, &ok, &ng) == 0);
            CHECK(ng == (iter % 3 == 0 ? F : (iter % 3 == 1 ? nc : F + nc)));
            if (!has3) CHECK(de_oracle_diff_f64(tape, n, consts, nc, X, F, N, LD, (int)(rnd() % F), out, dout, &ok) == 0);
        }
        CHECK(ok == 0 || ok == 1);
        n_eval++;
        n_complete += ok;
    }
    free(X); free(out); free(dout); free(grad); free(Xf); free(outf);
    printf("oracle selftest OK: %ld random tapes (%ld complete) through eval/grad/diff/parametric, known answers, malformed tapes rejected\n", n_eval, n_complete);
    return n_eval > 1000 ? 0 : 1;
}
