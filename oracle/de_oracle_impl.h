/*
 * de_oracle_impl.h — the reference ALGORITHM restated on the CPU, one full
 * length-N array pass per (fused) node, with the reference's early-exit
 * checks.  Included twice by de_oracle.c (T=float, T=double).  TEST
 * INFRASTRUCTURE ONLY — see de_oracle.c header.
 *
 * Every function cites the reference code it follows (paths relative to
 * /root/reference).
 */

#define OCAT_(a, b) a##b
#define OCAT(a, b) OCAT_(a, b)
#define ON(name) OCAT(name, ONAME)

/* ------------------------------------------------------------------------- */
typedef struct ON(octx) {
    const onode *nodes;
    const OT *consts; /* constant pool of this tree */
    const OT *X;      /* [F, N] column-major, ld = ldX */
    int64_t N;
    int64_t ldX;
    int F;
    int early_exit;   /* EvalContext.early_exit            */
    int fuse1, fuse2; /* use_fused && nops<=15, per degree */
    int elementwise;  /* 0: is_valid_array = isfinite(sum(x)) (reference);
                         1: all(isfinite, x)  (what a per-element device test computes) */
    /* bump arena for node arrays (the reference allocates one `similar` per
     * leaf/fused node, src/Evaluate.jl:70-72) */
    OT *arena;
    size_t arena_cap, arena_top;
} ON(octx);

typedef struct ON(ores) { OT *x; int ok; } ON(ores);
typedef struct ON(osca) { OT x; int ok; } ON(osca);

static OT *ON(o_alloc)(ON(octx) * c, size_t n) {
    if (c->arena_top + n > c->arena_cap) return NULL;
    OT *p = c->arena + c->arena_top;
    c->arena_top += n;
    return p;
}

/* is_valid_array(x) = is_valid(sum(x))            src/ValueInterface.jl:9
 * Julia's `sum` is a pairwise/@simd reduction whose association order is not
 * specified; 8 interleaved partial sums model the SIMD reassociation.  The
 * order only matters for the documented overflow quirk (finite elements whose
 * T-precision sum overflows). */
static int ON(o_is_valid_array)(const ON(octx) * c, const OT *x, int64_t n) {
    if (c->elementwise) {
        int bad = 0;
        for (int64_t j = 0; j < n; j++) bad |= !isfinite(x[j]);
        return !bad;
    }
    OT s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t j = 0;
    for (; j + 8 <= n; j += 8)
        for (int k = 0; k < 8; k++) s[k] += x[j + k];
    OT t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    for (; j < n; j++) t += x[j];
    return isfinite(t);
}

#define FEAT(c, f, j) ((c)->X[(int64_t)(f) + (c)->ldX * (j)])
#define LEAFVAL(c, nd) ((c)->consts[(nd)->arg])
#define IS_CONST_LEAF(nd) ((nd)->degree == 0 && (nd)->op == DE_LEAF_CONST)

/* @return_on_nonfinite_val                         src/Evaluate.jl:16-24 */
#define RET_NONFINITE_VAL(c, v)                                         \
    do {                                                                \
        if ((c)->early_exit && !isfinite(v)) {                          \
            ON(ores) r_ = {ON(o_alloc)((c), (size_t)(c)->N), 0};        \
            return r_;                                                  \
        }                                                               \
    } while (0)
/* @return_on_nonfinite_array                       src/Evaluate.jl:26-32 */
#define RET_NONFINITE_ARR(c, r)                                                    \
    do {                                                                           \
        if ((c)->early_exit && !ON(o_is_valid_array)((c), (r).x, (c)->N)) {        \
            ON(ores) r_ = {(r).x, 0};                                              \
            return r_;                                                             \
        }                                                                          \
    } while (0)

static ON(ores) ON(o_eval)(ON(octx) * c, int ni);

/* dispatch_constant_tree / degn_eval_constant      src/Evaluate.jl:1002-1067 */
static ON(osca) ON(o_const_tree)(const ON(octx) * c, int ni) {
    const onode *nd = &c->nodes[ni];
    ON(osca) r;
    if (nd->degree == 0) {
        r.x = LEAFVAL(c, nd);
        r.ok = isfinite(r.x);
        return r;
    }
    OT in[3] = {0, 0, 0};
    for (int i = 0; i < nd->degree; i++) {
        ON(osca) ci = ON(o_const_tree)(c, nd->child[i]);
        if (!ci.ok) return ci;
        in[i] = ci.x;
    }
    if (nd->degree == 1) r.x = ON(o_unary)(nd->op, in[0]);
    else if (nd->degree == 2) r.x = ON(o_binary)(nd->op, in[0], in[1]);
    else r.x = ON(o_ternary)(nd->op, in[0], in[1], in[2]);
    r.ok = isfinite(r.x);
    return r;
}

/* deg0_eval                                        src/Evaluate.jl:394-404 */
static ON(ores) ON(o_deg0)(ON(octx) * c, const onode *nd) {
    ON(ores) r = {ON(o_alloc)(c, (size_t)c->N), 1};
    if (nd->op == DE_LEAF_CONST) {
        OT v = LEAFVAL(c, nd);
        for (int64_t j = 0; j < c->N; j++) r.x[j] = v;
    } else {
        int f = nd->arg;
        for (int64_t j = 0; j < c->N; j++) r.x[j] = FEAT(c, f, j);
    }
    return r;
}

/* Per-op loops: the switch sits OUTSIDE the sample loop, like the reference's
 * @nif over operators selects one specialised kernel (src/Evaluate.jl:488-577). */
#define LOOP1(EXPR_IN, STORE)                                         \
    for (int64_t j = 0; j < n; j++) { OT a = (EXPR_IN); STORE = ON(o_unary)(op, a); }

/* degn_eval (deg1)                                 src/Evaluate.jl:366-392 */
static void ON(o_deg1_inplace)(int op, OT *x, int64_t n) {
    switch (op) {
    case DE_U_NEG: for (int64_t j = 0; j < n; j++) x[j] = -x[j]; break;
    case DE_U_SQUARE: for (int64_t j = 0; j < n; j++) x[j] = x[j] * x[j]; break;
    default: for (int64_t j = 0; j < n; j++) x[j] = ON(o_unary)(op, x[j]); break;
    }
}
/* degn_eval (deg2): cum_l[j] = op(cum_l[j], cum_r[j]) */
static void ON(o_deg2_inplace)(int op, OT *l, const OT *r, int64_t n) {
    switch (op) {
    case DE_B_ADD: for (int64_t j = 0; j < n; j++) l[j] = l[j] + r[j]; break;
    case DE_B_SUB: for (int64_t j = 0; j < n; j++) l[j] = l[j] - r[j]; break;
    case DE_B_MUL: for (int64_t j = 0; j < n; j++) l[j] = l[j] * r[j]; break;
    case DE_B_DIV: for (int64_t j = 0; j < n; j++) l[j] = l[j] / r[j]; break;
    default: for (int64_t j = 0; j < n; j++) l[j] = ON(o_binary)(op, l[j], r[j]); break;
    }
}

/* A leaf operand inside a fused loop: constant value or strided feature read
 * (feature_at / first_feature_index, src/Evaluate.jl:124-128). */
#define LEAF_AT(c, nd, j) (IS_CONST_LEAF(nd) ? LEAFVAL(c, nd) : FEAT(c, (nd)->arg, j))

/* deg2_l0_r0_eval                                  src/Evaluate.jl:874-933 */
static ON(ores) ON(o_deg2_l0_r0)(ON(octx) * c, const onode *nd) {
    const onode *l = &c->nodes[nd->child[0]], *r = &c->nodes[nd->child[1]];
    int op = nd->op;
    int64_t n = c->N;
    if (IS_CONST_LEAF(l) && IS_CONST_LEAF(r)) {
        OT vl = LEAFVAL(c, l); RET_NONFINITE_VAL(c, vl);
        OT vr = LEAFVAL(c, r); RET_NONFINITE_VAL(c, vr);
        OT x = ON(o_binary)(op, vl, vr); RET_NONFINITE_VAL(c, x);
        ON(ores) o = {ON(o_alloc)(c, (size_t)n), 1};
        for (int64_t j = 0; j < n; j++) o.x[j] = x;
        return o;
    }
    if (IS_CONST_LEAF(l)) { OT v = LEAFVAL(c, l); RET_NONFINITE_VAL(c, v); }
    if (IS_CONST_LEAF(r)) { OT v = LEAFVAL(c, r); RET_NONFINITE_VAL(c, v); }
    ON(ores) o = {ON(o_alloc)(c, (size_t)n), 1};
    switch (op) {
#define L0R0(OPX) for (int64_t j = 0; j < n; j++) { OT a = LEAF_AT(c, l, j), b = LEAF_AT(c, r, j); o.x[j] = (OPX); }
    case DE_B_ADD: L0R0(a + b) break;
    case DE_B_SUB: L0R0(a - b) break;
    case DE_B_MUL: L0R0(a * b) break;
    case DE_B_DIV: L0R0(a / b) break;
    default: L0R0(ON(o_binary)(op, a, b)) break;
#undef L0R0
    }
    return o;
}

/* deg2_l0_eval: op(leaf, cum)                      src/Evaluate.jl:936-963
 * deg2_r0_eval: op(cum, leaf)                      src/Evaluate.jl:966-993 */
static ON(ores) ON(o_deg2_leafcum)(ON(octx) * c, const onode *nd, OT *cum, int leaf_is_left) {
    const onode *lf = &c->nodes[nd->child[leaf_is_left ? 0 : 1]];
    int op = nd->op;
    int64_t n = c->N;
    if (IS_CONST_LEAF(lf)) { OT v = LEAFVAL(c, lf); RET_NONFINITE_VAL(c, v); }
    if (leaf_is_left) {
        switch (op) {
#define LC(OPX) for (int64_t j = 0; j < n; j++) { OT a = LEAF_AT(c, lf, j), b = cum[j]; cum[j] = (OPX); }
        case DE_B_ADD: LC(a + b) break;
        case DE_B_SUB: LC(a - b) break;
        case DE_B_MUL: LC(a * b) break;
        case DE_B_DIV: LC(a / b) break;
        default: LC(ON(o_binary)(op, a, b)) break;
#undef LC
        }
    } else {
        switch (op) {
#define CL(OPX) for (int64_t j = 0; j < n; j++) { OT a = cum[j], b = LEAF_AT(c, lf, j); cum[j] = (OPX); }
        case DE_B_ADD: CL(a + b) break;
        case DE_B_SUB: CL(a - b) break;
        case DE_B_MUL: CL(a * b) break;
        case DE_B_DIV: CL(a / b) break;
        default: CL(ON(o_binary)(op, a, b)) break;
#undef CL
        }
    }
    ON(ores) o = {cum, 1};
    return o;
}

/* deg2_branch0_eval + _fused_binary3               src/Evaluate.jl:795-871 */
static ON(ores) ON(o_deg2_branch0)(ON(octx) * c, const onode *nd, int left) {
    const onode *branch = &c->nodes[nd->child[left ? 0 : 1]];
    const onode *leaf1 = left ? &c->nodes[branch->child[0]] : &c->nodes[nd->child[0]];
    const onode *leaf2 = left ? &c->nodes[branch->child[1]] : &c->nodes[branch->child[0]];
    const onode *leaf3 = left ? &c->nodes[nd->child[1]] : &c->nodes[branch->child[1]];
    int64_t n = c->N;
    if (c->early_exit) {
        const onode *lv[3] = {leaf1, leaf2, leaf3};
        for (int i = 0; i < 3; i++)
            if (IS_CONST_LEAF(lv[i]) && !isfinite(LEAFVAL(c, lv[i]))) {
                ON(ores) r_ = {ON(o_alloc)(c, (size_t)n), 0};
                return r_;
            }
    }
    int op = nd->op, bop = branch->op, ee = c->early_exit;
    ON(ores) o = {ON(o_alloc)(c, (size_t)n), 1};
    for (int64_t j = 0; j < n; j++) {
        OT x1 = LEAF_AT(c, leaf1, j), x2 = LEAF_AT(c, leaf2, j), x3 = LEAF_AT(c, leaf3, j);
        if (left) {
            OT bx = ON(o_binary)(bop, x1, x2);
            o.x[j] = (ee && (!isfinite(bx) || !isfinite(x3))) ? (OT)INFINITY : ON(o_binary)(op, bx, x3);
        } else {
            OT bx = ON(o_binary)(bop, x2, x3);
            o.x[j] = (ee && (!isfinite(x1) || !isfinite(bx))) ? (OT)INFINITY : ON(o_binary)(op, x1, bx);
        }
    }
    return o;
}

/* deg1_l2_ll0_lr0_eval: op(op_l(leaf, leaf))       src/Evaluate.jl:693-761
 * (the both-constant case cannot be reached from _eval_tree_array — the
 * is_constant shortcut catches it — but direct callers exist in the reference
 * tests, test/test_evaluation.jl:228-236, so it is kept.) */
static ON(ores) ON(o_deg1_l2)(ON(octx) * c, const onode *nd) {
    const onode *ch = &c->nodes[nd->child[0]];
    const onode *ll = &c->nodes[ch->child[0]], *lr = &c->nodes[ch->child[1]];
    int op = nd->op, opl = ch->op;
    int64_t n = c->N;
    if (IS_CONST_LEAF(ll) && IS_CONST_LEAF(lr)) {
        OT a = LEAFVAL(c, ll), b = LEAFVAL(c, lr);
        RET_NONFINITE_VAL(c, a); RET_NONFINITE_VAL(c, b);
        OT xl = ON(o_binary)(opl, a, b); RET_NONFINITE_VAL(c, xl);
        OT x = ON(o_unary)(op, xl); RET_NONFINITE_VAL(c, x);
        ON(ores) o = {ON(o_alloc)(c, (size_t)n), 1};
        for (int64_t j = 0; j < n; j++) o.x[j] = x;
        return o;
    }
    if (IS_CONST_LEAF(ll)) { OT v = LEAFVAL(c, ll); RET_NONFINITE_VAL(c, v); }
    if (IS_CONST_LEAF(lr)) { OT v = LEAFVAL(c, lr); RET_NONFINITE_VAL(c, v); }
    ON(ores) o = {ON(o_alloc)(c, (size_t)n), 1};
    for (int64_t j = 0; j < n; j++) {
        OT xl = ON(o_binary)(opl, LEAF_AT(c, ll, j), LEAF_AT(c, lr, j));
        o.x[j] = isfinite(xl) ? ON(o_unary)(op, xl) : (OT)INFINITY;
    }
    return o;
}

/* deg1_l1_ll0_eval: op(op_l(leaf))                 src/Evaluate.jl:764-793 */
static ON(ores) ON(o_deg1_l1)(ON(octx) * c, const onode *nd) {
    const onode *ch = &c->nodes[nd->child[0]];
    const onode *ll = &c->nodes[ch->child[0]];
    int op = nd->op, opl = ch->op;
    int64_t n = c->N;
    if (IS_CONST_LEAF(ll)) {
        OT a = LEAFVAL(c, ll); RET_NONFINITE_VAL(c, a);
        OT xl = ON(o_unary)(opl, a); RET_NONFINITE_VAL(c, xl);
        OT x = ON(o_unary)(op, xl); RET_NONFINITE_VAL(c, x);
        ON(ores) o = {ON(o_alloc)(c, (size_t)n), 1};
        for (int64_t j = 0; j < n; j++) o.x[j] = x;
        return o;
    }
    ON(ores) o = {ON(o_alloc)(c, (size_t)n), 1};
    int f = ll->arg;
    for (int64_t j = 0; j < n; j++) {
        OT xl = ON(o_unary)(opl, FEAT(c, f, j));
        o.x[j] = isfinite(xl) ? ON(o_unary)(op, xl) : (OT)INFINITY;
    }
    return o;
}

#define IS_LEAF(ci) (c->nodes[(ci)].degree == 0)
static int ON(o_bin_of_leaves)(const ON(octx) * c, int ci) {
    const onode *x = &c->nodes[ci];
    return x->degree == 2 && IS_LEAF(x->child[0]) && IS_LEAF(x->child[1]);
}

/* _eval_tree_array + dispatch_deg1_eval + dispatch_deg2_eval + dispatch_degn_eval
 *                                                   src/Evaluate.jl:337-364,428-651 */
static ON(ores) ON(o_eval)(ON(octx) * c, int ni) {
    const onode *nd = &c->nodes[ni];
    int64_t n = c->N;
    if (nd->degree == 0) return ON(o_deg0)(c, nd);
    if (nd->is_const) { /* is_constant(tree) shortcut, :347-354 */
        ON(osca) s = ON(o_const_tree)(c, ni);
        ON(ores) o = {ON(o_alloc)(c, (size_t)n), s.ok};
        if (s.ok) for (int64_t j = 0; j < n; j++) o.x[j] = s.x;
        return o;
    }
    if (nd->degree == 1) { /* dispatch_deg1_eval :599-651 */
        int ci = nd->child[0];
        const onode *ch = &c->nodes[ci];
        if (c->fuse1 && ON(o_bin_of_leaves)(c, ci)) return ON(o_deg1_l2)(c, nd);
        if (c->fuse1 && ch->degree == 1 && IS_LEAF(ch->child[0])) return ON(o_deg1_l1)(c, nd);
        ON(ores) r = ON(o_eval)(c, ci);
        if (!r.ok) return r;
        RET_NONFINITE_ARR(c, r);
        ON(o_deg1_inplace)(nd->op, r.x, n);
        return r;
    }
    if (nd->degree == 2) { /* dispatch_deg2_eval :488-577 */
        int li = nd->child[0], ri = nd->child[1];
        if (c->fuse2 && IS_LEAF(li) && IS_LEAF(ri)) return ON(o_deg2_l0_r0)(c, nd);
        if (c->fuse2 && IS_LEAF(ri)) {
            if (ON(o_bin_of_leaves)(c, li)) return ON(o_deg2_branch0)(c, nd, 1);
            ON(ores) rl = ON(o_eval)(c, li);
            if (!rl.ok) return rl;
            RET_NONFINITE_ARR(c, rl);
            return ON(o_deg2_leafcum)(c, nd, rl.x, 0);
        }
        if (c->fuse2 && IS_LEAF(li)) {
            if (ON(o_bin_of_leaves)(c, ri)) return ON(o_deg2_branch0)(c, nd, 0);
            ON(ores) rr = ON(o_eval)(c, ri);
            if (!rr.ok) return rr;
            RET_NONFINITE_ARR(c, rr);
            return ON(o_deg2_leafcum)(c, nd, rr.x, 1);
        }
        ON(ores) rl = ON(o_eval)(c, li);
        if (!rl.ok) return rl;
        RET_NONFINITE_ARR(c, rl);
        ON(ores) rr = ON(o_eval)(c, ri);
        if (!rr.ok) return rr;
        RET_NONFINITE_ARR(c, rr);
        ON(o_deg2_inplace)(nd->op, rl.x, rr.x, n);
        return rl;
    }
    /* degree 3: inner_dispatch_degn_eval :428-467 */
    ON(ores) r[3];
    for (int i = 0; i < 3; i++) {
        r[i] = ON(o_eval)(c, nd->child[i]);
        if (!r[i].ok) return r[i];
        RET_NONFINITE_ARR(c, r[i]);
    }
    for (int64_t j = 0; j < n; j++) r[0].x[j] = ON(o_ternary)(nd->op, r[0].x[j], r[1].x[j], r[2].x[j]);
    return r[0];
}

/* _bumper_eval_tree_array + KernelDispatcher       ext/DynamicExpressionsBumperExt.jl:11-89 */
static ON(ores) ON(o_eval_bumper)(ON(octx) * c, int ni) {
    const onode *nd = &c->nodes[ni];
    int64_t n = c->N;
    if (nd->degree == 0) {
        ON(ores) r = ON(o_deg0)(c, nd);
        if (nd->op == DE_LEAF_CONST) r.ok = c->early_exit ? isfinite(LEAFVAL(c, nd)) : 1;
        return r;
    }
    ON(ores) r[3];
    for (int i = 0; i < nd->degree; i++) r[i] = ON(o_eval_bumper)(c, nd->child[i]);
    for (int i = 0; i < nd->degree; i++) if (!r[i].ok) return r[i];
    if (nd->degree == 1) ON(o_deg1_inplace)(nd->op, r[0].x, n);
    else if (nd->degree == 2) ON(o_deg2_inplace)(nd->op, r[0].x, r[1].x, n);
    else for (int64_t j = 0; j < n; j++) r[0].x[j] = ON(o_ternary)(nd->op, r[0].x[j], r[1].x[j], r[2].x[j]);
    r[0].ok = c->early_exit ? ON(o_is_valid_array)(c, r[0].x, n) : 1;
    return r[0];
}

/* eval_tree_array                                   src/Evaluate.jl:279-309
 * options: DE_OPT_* bits of include/de_hip.h.  Returns 0, or a negative value for a
 * malformed tape.  `out` receives result.x (contents unspecified when ok==0,
 * exactly like the reference, which returns a partially evaluated buffer). */
int ON(de_oracle_eval)(const de_tape_node_t *tape, int64_t n_nodes, const OT *consts,
                       int64_t n_consts, const OT *X, int32_t F, int64_t N, int64_t ldX,
                       uint32_t options, int32_t elementwise, OT *out, uint8_t *ok) {
    onode *nodes = NULL;
    int root = o_parse(tape, n_nodes, n_consts, F, 0, &nodes);
    if (root < 0) return root;
    ON(octx) c;
    memset(&c, 0, sizeof c);
    c.nodes = nodes; c.consts = consts; c.X = X; c.N = N; c.ldX = ldX; c.F = F;
    c.early_exit = (options & DE_OPT_EARLY_EXIT) != 0;
    c.fuse1 = (options & DE_OPT_FUSE_DEG1) != 0;
    c.fuse2 = (options & DE_OPT_FUSE_DEG2) != 0;
    c.elementwise = elementwise;
    c.arena_cap = (size_t)(n_nodes + 2) * (size_t)(N > 0 ? N : 1);
    c.arena = (OT *)malloc(c.arena_cap * sizeof(OT));
    if (!c.arena) { free(nodes); return -100; }
    ON(ores) r;
    if (options & DE_OPT_BUMPER_CHECKS) {
        r = ON(o_eval_bumper)(&c, root);
    } else {
        r = ON(o_eval)(&c, root);
        /* result.ok && (early_exit==false || is_valid_array(result.x))   :305-308 */
        if (r.ok && c.early_exit) r.ok = ON(o_is_valid_array)(&c, r.x, N);
    }
    if (r.x && out) memcpy(out, r.x, (size_t)N * sizeof(OT));
    *ok = (uint8_t)(r.ok ? 1 : 0);
    free(c.arena);
    free(nodes);
    return 0;
}

/* eval_tree_array(ex::ParametricExpression, X, classes, operators)
 *                                                   src/ParametricExpression.jl:371-390
 * Materialises indexed_parameters (P x N), vcat's it above X and re-indexes the
 * leaves (LeafConverter :305-313: parameter p -> feature p, feature f -> f+P),
 * exactly like the reference; classes are 1-based when class_base==1. */
int ON(de_oracle_eval_param)(const de_tape_node_t *tape, int64_t n_nodes, const OT *consts,
                             int64_t n_consts, const OT *X, int32_t F, int64_t N, int64_t ldX,
                             const OT *params, int32_t P, int64_t n_classes, int64_t ld_params,
                             const int32_t *classes, int32_t class_base, uint32_t options,
                             int32_t elementwise, OT *out, uint8_t *ok) {
    for (int64_t j = 0; j < N; j++) {
        int64_t cl = (int64_t)classes[j] - class_base;
        if (cl < 0 || cl >= n_classes) return -6; /* @assert maximum(classes) <= n_classes :379 */
    }
    int32_t F2 = F + P;
    OT *PX = (OT *)malloc((size_t)F2 * (size_t)(N > 0 ? N : 1) * sizeof(OT));
    de_tape_node_t *t2 = (de_tape_node_t *)malloc((size_t)n_nodes * sizeof *t2);
    if (!PX || !t2) { free(PX); free(t2); return -100; }
    for (int64_t j = 0; j < N; j++) {
        int64_t cl = (int64_t)classes[j] - class_base;
        for (int p = 0; p < P; p++) PX[p + (int64_t)F2 * j] = params[p + ld_params * cl];
        for (int f = 0; f < F; f++) PX[P + f + (int64_t)F2 * j] = X[f + ldX * j];
    }
    for (int64_t i = 0; i < n_nodes; i++) {
        t2[i] = tape[i];
        if (tape[i].degree == 0 && tape[i].op == DE_LEAF_PARAM) { t2[i].op = DE_LEAF_FEATURE; }
        else if (tape[i].degree == 0 && tape[i].op == DE_LEAF_FEATURE) { t2[i].arg = (uint16_t)(tape[i].arg + P); }
    }
    int rc = ON(de_oracle_eval)(t2, n_nodes, consts, n_consts, PX, F2, N, F2, options, elementwise, out, ok);
    free(PX);
    free(t2);
    return rc;
}

/* ------------------------------------------------------------------------- *
 * Forward-mode gradient                             src/EvaluateDerivative.jl
 * ------------------------------------------------------------------------- */
typedef struct ON(ogres) { OT *x; OT *dx; int ok; } ON(ogres);

typedef struct ON(ogctx) {
    ON(octx) e;
    int mode;     /* de_grad_mode */
    int64_t G;    /* n_gradients */
    OT *garena; size_t gcap, gtop;
} ON(ogctx);

static OT *ON(o_galloc)(ON(ogctx) * g, size_t n) {
    if (g->gtop + n > g->gcap) return NULL;
    OT *p = g->garena + g->gtop;
    g->gtop += n;
    return p;
}

static ON(ogres) ON(o_grad_inner)(ON(ogctx) * g, int ni);

/* eval_grad_tree_array (inner wrapper): after EVERY node,
 * ok = is_valid_array(x) && is_valid_array(dx)       :230-243 */
static ON(ogres) ON(o_grad)(ON(ogctx) * g, int ni) {
    ON(ogres) r = ON(o_grad_inner)(g, ni);
    if (!r.ok) return r;
    r.ok = ON(o_is_valid_array)(&g->e, r.x, g->e.N) &&
           ON(o_is_valid_array)(&g->e, r.dx, g->e.N * g->G);
    return r;
}

static ON(ogres) ON(o_grad_inner)(ON(ogctx) * g, int ni) {
    ON(octx) *c = &g->e;
    const onode *nd = &c->nodes[ni];
    int64_t n = c->N, G = g->G;
    if (nd->degree == 0) { /* grad_deg0_eval :367-404 */
        ON(ores) v = ON(o_deg0)(c, nd);
        ON(ogres) r = {v.x, ON(o_galloc)(g, (size_t)(G * n)), 1};
        memset(r.dx, 0, (size_t)(G * n) * sizeof(OT));
        int is_const = nd->op == DE_LEAF_CONST;
        int64_t idx = -1;
        if (g->mode == DE_GRAD_VARIABLE) { if (!is_const) idx = nd->arg; }
        else if (g->mode == DE_GRAD_CONSTANT) { if (is_const) idx = nd->arg; }
        else idx = is_const ? (int64_t)nd->arg + c->F : (int64_t)nd->arg; /* features first :220 */
        if (idx >= 0) for (int64_t j = 0; j < n; j++) r.dx[idx + G * j] = (OT)1;
        return r;
    }
    /* dispatch_grad_degn_eval :285-338: children through the WRAPPER */
    ON(ogres) ch[3];
    for (int i = 0; i < nd->degree; i++) {
        ch[i] = ON(o_grad)(g, nd->child[i]);
        if (!ch[i].ok) return ch[i];
    }
    /* grad_degn_eval :340-365: d[k,j] = g1*d1[k,j] + g2*d2[k,j] (+ g3*d3[k,j]) */
    int op = nd->op;
    for (int64_t j = 0; j < n; j++) {
        OT gr[3];
        if (nd->degree == 1) {
            OT a = ch[0].x[j];
            ON(o_unary_grad)(op, a, gr);
            ch[0].x[j] = ON(o_unary)(op, a);
            for (int64_t k = 0; k < G; k++) ch[0].dx[k + G * j] = gr[0] * ch[0].dx[k + G * j];
        } else if (nd->degree == 2) {
            OT a = ch[0].x[j], b = ch[1].x[j];
            ON(o_binary_grad)(op, a, b, gr);
            ch[0].x[j] = ON(o_binary)(op, a, b);
            for (int64_t k = 0; k < G; k++)
                ch[0].dx[k + G * j] = gr[0] * ch[0].dx[k + G * j] + gr[1] * ch[1].dx[k + G * j];
        } else {
            OT a = ch[0].x[j], b = ch[1].x[j], z = ch[2].x[j];
            ON(o_ternary_grad)(op, a, b, z, gr);
            ch[0].x[j] = ON(o_ternary)(op, a, b, z);
            for (int64_t k = 0; k < G; k++)
                ch[0].dx[k + G * j] = (gr[0] * ch[0].dx[k + G * j] + gr[1] * ch[1].dx[k + G * j]) +
                                      gr[2] * ch[2].dx[k + G * j];
        }
    }
    ch[0].ok = 1;
    return ch[0];
}

/* eval_grad_tree_array (public)                     src/EvaluateDerivative.jl:193-228
 * grad: [n_grad, N] column-major.  *n_grad_out receives n_gradients. */
int ON(de_oracle_grad)(const de_tape_node_t *tape, int64_t n_nodes, const OT *consts,
                       int64_t n_consts, const OT *X, int32_t F, int64_t N, int64_t ldX, int32_t mode,
                       int32_t elementwise, OT *out, OT *grad, uint8_t *ok, int64_t *n_grad_out) {
    onode *nodes = NULL;
    int root = o_parse(tape, n_nodes, n_consts, F, 0, &nodes);
    if (root < 0) return root;
    ON(ogctx) g;
    memset(&g, 0, sizeof g);
    ON(octx) *c = &g.e;
    c->nodes = nodes; c->consts = consts; c->X = X; c->N = N; c->ldX = ldX; c->F = F;
    c->early_exit = 1; c->elementwise = elementwise;
    int64_t nc = 0;
    for (int64_t i = 0; i < n_nodes; i++) nc += (tape[i].degree == 0 && tape[i].op == DE_LEAF_CONST);
    g.mode = mode;
    g.G = mode == DE_GRAD_VARIABLE ? F : (mode == DE_GRAD_CONSTANT ? nc : F + nc);
    if (n_grad_out) *n_grad_out = g.G;
    size_t nn = (size_t)(N > 0 ? N : 1);
    c->arena_cap = (size_t)(n_nodes + 2) * nn;
    c->arena = (OT *)malloc(c->arena_cap * sizeof(OT));
    g.gcap = (size_t)(n_nodes + 2) * nn * (size_t)(g.G > 0 ? g.G : 1);
    g.garena = (OT *)malloc(g.gcap * sizeof(OT));
    if (!c->arena || !g.garena) { free(c->arena); free(g.garena); free(nodes); return -100; }
    ON(ogres) r = ON(o_grad)(&g, root);
    if (r.x && out) memcpy(out, r.x, (size_t)N * sizeof(OT));
    if (r.dx && grad) memcpy(grad, r.dx, (size_t)(N * g.G) * sizeof(OT));
    *ok = (uint8_t)(r.ok ? 1 : 0);
    free(c->arena); free(g.garena); free(nodes);
    return 0;
}

/* eval_diff_tree_array                              src/EvaluateDerivative.jl:40-168
 * Single direction (0-based feature); NO validity test on this path: ok is
 * always true (diff_degn_eval returns ResultOk2(..., true), :117). */
typedef struct ON(odres) { OT *x; OT *dx; } ON(odres);
static ON(odres) ON(o_diff)(ON(octx) * c, int ni, int direction) {
    const onode *nd = &c->nodes[ni];
    int64_t n = c->N;
    if (nd->degree == 0) { /* diff_deg0_eval :87-97 */
        ON(ores) v = ON(o_deg0)(c, nd);
        ON(odres) r = {v.x, ON(o_alloc)(c, (size_t)n)};
        OT d = (nd->op != DE_LEAF_CONST && nd->arg == direction) ? (OT)1 : (OT)0;
        for (int64_t j = 0; j < n; j++) r.dx[j] = d;
        return r;
    }
    ON(odres) ch[3];
    for (int i = 0; i < nd->degree; i++) ch[i] = ON(o_diff)(c, nd->child[i], direction);
    int op = nd->op;
    for (int64_t j = 0; j < n; j++) { /* diff_degn_eval :99-119 */
        OT gr[3];
        if (nd->degree == 1) {
            OT a = ch[0].x[j];
            ON(o_unary_grad)(op, a, gr);
            ch[0].x[j] = ON(o_unary)(op, a);
            ch[0].dx[j] = gr[0] * ch[0].dx[j];
        } else if (nd->degree == 2) {
            OT a = ch[0].x[j], b = ch[1].x[j];
            ON(o_binary_grad)(op, a, b, gr);
            ch[0].x[j] = ON(o_binary)(op, a, b);
            ch[0].dx[j] = gr[0] * ch[0].dx[j] + gr[1] * ch[1].dx[j];
        } else {
            OT a = ch[0].x[j], b = ch[1].x[j], z = ch[2].x[j];
            ON(o_ternary_grad)(op, a, b, z, gr);
            ch[0].x[j] = ON(o_ternary)(op, a, b, z);
            ch[0].dx[j] = (gr[0] * ch[0].dx[j] + gr[1] * ch[1].dx[j]) + gr[2] * ch[2].dx[j];
        }
    }
    return ch[0];
}

int ON(de_oracle_diff)(const de_tape_node_t *tape, int64_t n_nodes, const OT *consts,
                       int64_t n_consts, const OT *X, int32_t F, int64_t N, int64_t ldX,
                       int32_t direction, OT *out, OT *dout, uint8_t *ok) {
    onode *nodes = NULL;
    int root = o_parse(tape, n_nodes, n_consts, F, 0, &nodes);
    if (root < 0) return root;
    ON(octx) c;
    memset(&c, 0, sizeof c);
    c.nodes = nodes; c.consts = consts; c.X = X; c.N = N; c.ldX = ldX; c.F = F;
    c.arena_cap = 2 * (size_t)(n_nodes + 2) * (size_t)(N > 0 ? N : 1);
    c.arena = (OT *)malloc(c.arena_cap * sizeof(OT));
    if (!c.arena) { free(nodes); return -100; }
    ON(odres) r = ON(o_diff)(&c, root, direction);
    memcpy(out, r.x, (size_t)N * sizeof(OT));
    memcpy(dout, r.dx, (size_t)N * sizeof(OT));
    *ok = 1;
    free(c.arena); free(nodes);
    return 0;
}

#undef FEAT
#undef LEAFVAL
#undef IS_CONST_LEAF
#undef RET_NONFINITE_VAL
#undef RET_NONFINITE_ARR
#undef LEAF_AT
#undef IS_LEAF
#undef LOOP1
#undef ON
#undef OCAT
#undef OCAT_
