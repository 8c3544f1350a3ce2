// de_api_program.cpp — C ABI (include/de_hip.h): de_program_create / _create_cse / _set_consts / _destroy, constant folding (host, de_fold_kernel,
// auxiliary program), the bound / threaded / chained streams of the eval program, and the program hooks (verify, dump, hash, host-only lowering).
#include "de_api_internal.h"

// (src/Evaluate.jl:1002-1067: every node's output is validity-tested; the arithmetic goes on, IEEE propagates what it must).
template <typename T>
static bool host_fold_eval(const de_tape_node_t *nd, int64_t n, const double *consts, const int64_t *csrc, T *value) {
    T stack_small[32];
    std::vector<T> stack_big;
    T *st = stack_small;
    if (n > 32) { stack_big.resize((size_t)n); st = stack_big.data(); }
    int sp = 0;
    bool ok = true;
    for (int64_t i = 0; i < n; i++) {
        T v;
        if (nd[i].degree == 0) v = (T)consts[csrc[nd[i].arg]];
        else {
            const T b = st[--sp], a = st[--sp];
            switch (nd[i].op) {
            case DE_B_ADD: v = a + b; break;
            case DE_B_SUB: v = a - b; break;
            case DE_B_MUL: v = a * b; break;
            default: v = a / b; break; // DE_B_DIV (host_foldable admits nothing else)
            }
        }
        ok = ok && std::isfinite(v);
        st[sp++] = v;
    }
    *value = st[0];
    return ok;
}
static bool host_foldable(const de_tape_node_t *nd, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        if (nd[i].degree == 0) { if (nd[i].op != DE_LEAF_CONST) return false; }
        else if (nd[i].degree != 2 || nd[i].op < DE_B_ADD || nd[i].op > DE_B_DIV) return false;
    }
    return n > 0;
}

// No exception leaves this file: the gradient entry points build host vectors (and run passes on the host pool, which reports a worker's
extern "C" {
static void write_imm(Instr &ins, int dtype, double v) {
    if (dtype == DE_F32) {
        ins.imm.u32[1] = 0;
        ins.imm.f32 = (float)v;
    } else ins.imm.f64 = v;
}
// Early exit at tree granularity (kernels: a workgroup does not evaluate the trees whose flag is already 0).  DE_NO_TREE_SKIP=1
// restores the evaluate-everything behaviour for A/B measurements (the option bit DE_OPT_FULL_EVAL does the same per program).
bool tree_skip_enabled() {
    static const bool on = [] { const char *v = getenv("DE_NO_TREE_SKIP"); return !(v && *v == '1'); }();
    return on;
}
static bool finite_in(int dtype, double v) { return dtype == DE_F32 ? std::isfinite((float)v) : std::isfinite(v); }

// Parameters as staged rows (round 3): every use of a parameter was a gather of its samples' values through the vector cache (h_param:
// 4 loads per lane and use; per-sample parameters, C = N, ran at 27 % VALU utilisation).  With <= 16 parameters the eval kernels
// instead stage the tile's parameter values once per workgroup, transposed like X, into P more LDS rows behind the spill slots and
// the binder treats a parameter operand as a row operand (every fused form applies).  DE_NO_PARAM_ROWS=1: the gathers.
static bool param_rows_enabled() { // (read at every de_program_create: the tests switch it inside one process)
    const char *v = getenv("DE_NO_PARAM_ROWS");
    return !(v && *v == '1');
}
static int64_t eval_rows(const de_program *p) { return (int64_t)p->n_features + p->n_slots + (p->prows ? p->n_params : 0); }
static void rebind(de_program *p) {
    const bool ee = (p->options & DE_OPT_EARLY_EXIT) != 0;
    p->prows = p->uses_params && p->n_params > 0 && p->n_params <= 16 && param_rows_enabled() &&
               ((size_t)p->n_features + (size_t)p->n_slots + (size_t)p->n_params) * 257 * 16 <= 150 * 1024;
    const int prb = p->prows ? p->n_features + p->n_slots : -1;
    const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
    const std::vector<int32_t> &off = p->folded ? p->fcode_off : p->code_off;
    const int nf = p->n_features;
    build_stream_by_trees<BoundInstr>(p->n_trees, &p->bcode, &p->bcode_off, [&](int64_t t, std::vector<BoundInstr> *out) {
        const int32_t i0 = off[(size_t)t], i1 = off[(size_t)t + 1];
        bind_tree(src.data() + i0, (size_t)(i1 - i0), ee, nf, out, prb);
    });
    match_const_sites(src, off, p->bcode, p->bcode_off, p->n_trees, [](const BoundInstr &b) { return bop_is_const_source(b.bop); }, &p->bsite);
    p->tsite.clear();
    p->site_gen++;
}
// (bind_tree / fuse_tree are ~0.3 us per tree — 3 ms each for 10^4 trees on one thread: build_stream_by_trees)

// Threaded-code form of the bound program (de_kernels.hip, de_eval_threaded_kernel): word 0 =
// handler address - handler_base, word 1 = LDS byte offset of the operand row | aux << 24.
// DE_DEBUG_TIMING: microseconds since the previous lap of this thread, on stderr
static void make_chained(de_program *p);
// Waves per workgroup of the threaded eval kernel (de_api_internal.h `waves`).  A one-wave workgroup keeps F rows of X, S spill-slot rows
// and (parametric) P staged parameter rows in LDS: at F = 5, S = 2 that is 21 resident waves per CU (5.25 per SIMD; the kernel's registers
// allow 28) and more buys nothing (measured: W = 2 +1 %, W = 4 +6 % on the headline) — but 8 parameter rows leave 10 waves, 20 features 7, 30
// features 4.  W waves share X and the parameter rows: W is doubled (1 -> 2 -> 4 -> 8) while the workgroup is below 20 resident waves per CU and
// the step raises them by half or more.  Measured, 10^6 samples x 1000 trees: 8 per-sample parameters 1.35 -> 0.92 ms, F = 12 1.29 -> 0.97,
// F = 20 2.06 -> 1.08, F = 30 3.10 -> 1.22 (profiles/r6_wave_groups.txt).  DE_EVAL_WAVES = 1 | 2 | 4 | 8 overrides (1 = the kernel of rounds 1-5).
static int choose_waves(const de_program *p) {
    if (TBLK != 64 || (p->uses_params && !p->prows)) return 1; // (gathered parameters: the class row is per wave)
    if (const char *e = getenv("DE_EVAL_WAVES")) { const int v = atoi(e); return (v == 2 || v == 4 || v == 8) ? v : 1; }
    const size_t rb = trow_bytes(p->dtype), lds_cu = 160u << 10, shared = (size_t)p->n_features + (p->prows ? (size_t)p->n_params : 0);
    auto resident = [&](int W) -> size_t {
        const size_t lds = (shared + (size_t)W * (size_t)p->n_slots) * rb + (size_t)W * 256;
        // (7 waves per SIMD by the kernel's vector registers; a group of W >= 4 waves puts W / 4 on every SIMD)
        const size_t by_regs = W >= 4 ? 7 / ((size_t)W / 4) : 28 / (size_t)W;
        return lds > 150u * 1024 ? 0 : std::min<size_t>(lds_cu / lds, by_regs) * (size_t)W;
    };
    int best = 1;
    size_t have = resident(1);
    for (int W : {2, 4, 8}) {
        const size_t w = resident(W);
        if (have >= 20) break;
        if (w * 2 < have * 3) { // (many slot rows: two waves gain too little — four may still, when the workgroup is below 3 waves per SIMD)
            if (W == 2 && have < 12) continue;
            break;
        }
        best = W;
        have = w;
    }
    return best;
}
static int make_threaded(de_ctx *c, de_program *p) {
    p->threaded = false;
    p->waves = 1;
    p->var_stride = 0;
    p->ccode_w.clear();
    dbg_lap(nullptr);
    // the LDS-staged kernels need (n_features + n_slots) rows of 4112 B; wider X uses the direct variant
    p->direct = (size_t)eval_rows(p) * 257 * 16 > 150 * 1024; // (the flat-switch kernel's geometry: 256 threads x 16 bytes per row; it gathers the features of a wider X from global memory)
    // the threaded kernel's rows are a quarter of that (one wave's 64 vectors): it stages X up to ~140 rows — and shares them among the
    // waves of a wave group (round 6: F = 36 ... 120 ran the gathering flat-switch kernel, 14 - 20 ms per 10^6 samples x 1000 trees)
    // (a program the flat-switch kernel would gather for takes the threaded kernel while a group of FOUR waves fits: one wave per CU
    // would be no better than the gathers)
    if ((p->direct && ((size_t)eval_rows(p) + 3 * (size_t)p->n_slots) * trow_bytes(p->dtype) + 4 * 256 > 150 * 1024) || !eval_uses_threaded()) return DE_OK;
    if (eval_rows(p) > 4000) return DE_OK; // row offsets must fit 24 bits
    uint64_t table[TOPX_TABLE];
    hipError_t st = eval_handler_table(p->dtype, (p->options & DE_OPT_TURBO) != 0, table);
    if (st != hipSuccess) return fail(c, DE_ERR_HIP, "handler table: %s", hipGetErrorString(st));
    dbg_lap("handler table");
    uint64_t base = table[0];
    for (int i = 0; i < (int)TOPX_TABLE; i++) base = std::min<uint64_t>(base, table[i]);
    for (int i = 0; i < (int)TOPX_TABLE; i++) {
        if (table[i] - base > 0xFFFFFFFFull) return DE_OK; // cannot encode: keep the switch kernel
        // Float64 records carry 32 bits of the next handler's address (the high half is the current pc's)
        if (p->dtype != DE_F32 && (table[i] >> 32) != (table[0] >> 32)) return DE_OK;
    }
    const bool hot_unary = !getenv("DE_NO_CONST_UNARY_HOT"); // unary operators outside the binder's hot set: their own handlers
    const uint32_t row_bytes = (uint32_t)trow_bytes(p->dtype);
    // superinstructions (de_bind.h): fewer dispatches for the same arithmetic
    const char *nf = getenv("DE_NO_FUSE");
    const bool fuse = !(nf && *nf == '1');
    // wave groups (below): the waves per workgroup, and how far the last variant of the stream moves the slot rows — a fusion must fit all
    if (p->waves_choice == 0) p->waves_choice = choose_waves(p);
    const int W = fuse ? p->waves_choice : 1;
    const int wave_prows = p->prows ? p->n_params : 0;
    const FuseRows rows0{(uint32_t)p->n_features, (uint32_t)(p->n_features + p->n_slots), 0, W > 1 ? wave_prows + (W - 1) * p->n_slots : 0};
    build_stream_by_trees<BoundInstr>(p->n_trees, &p->fbcode, &p->tcode_off, [&](int64_t t, std::vector<BoundInstr> *out) {
        const int32_t b0 = p->bcode_off[(size_t)t], b1 = p->bcode_off[(size_t)t + 1];
        if (fuse) fuse_tree(p->bcode.data() + b0, (size_t)(b1 - b0), out, W > 1 && p->n_slots > 0 ? &rows0 : nullptr);
        else out->insert(out->end(), p->bcode.begin() + b0, p->bcode.begin() + b1);
    });
    dbg_lap("fuse_tree");
    p->tcode.resize(p->fbcode.size());
    // fused instruction -> threaded words (also what make_wave_variants re-derives the operand words of the other waves' streams with)
    auto threaded_words = [p, &table, base, hot_unary, row_bytes](const BoundInstr &b) {
        BoundInstr t = b;
        t.bop = (uint32_t)(table[b.bop] - base);
        if (hot_unary && (b.bop == BOP_GEN_ROW || b.bop == BOP_GEN_ACC)) {
            const int k = gun_index((int)(b.arg >> 24), DE_U_COS, DE_U_EXP, DE_U_SIN, DE_U_NEG, DE_U_SQUARE, DE_U_CUBE, DE_U_ABS, DE_U_LOG, DE_U_SAFE_LOG,
                                    DE_U_SQRT, DE_U_SAFE_SQRT, DE_U_TANH, DE_U_RELU);
            if (k >= 3) t.bop = (uint32_t)(table[TOPX_UN_BASE + (uint32_t)(k - 3) * 2 + (b.bop == BOP_GEN_ACC ? 1 : 0)] - base);
        }
        if (hot_unary && (b.bop == BOP_GEN_ROW || b.bop == BOP_GEN_CONST) && ((b.arg >> 24) == (uint32_t)DE_B_MAX || (b.arg >> 24) == (uint32_t)DE_B_MIN))
            t.bop = (uint32_t)(table[TOPX_BIN_BASE + ((b.arg >> 24) == (uint32_t)DE_B_MAX ? 0u : 2u) + (b.bop == BOP_GEN_CONST ? 1u : 0u)] - base);
        if (b.bop < BOP_COUNT && bop_is_const_source(b.bop)) {
            t.arg = b.arg & 0xFF000000u; // constant ordinal is only for the gradient kernel
        } else if (b.bop == BOP_GEN_PARAM) { // immediate = LDS byte offset of the class row (behind X and the spill slots)
            t.lo = (uint32_t)(p->n_features + p->n_slots) * row_bytes;
            t.hi = 0;
        } else {
            const uint32_t row = b.arg & 0xFFFFFFu, aux = b.arg >> 24;
            t.arg = (row * row_bytes) | (aux << 24);
            if (b.bop == BOP_TERN) t.lo = (b.lo - row) * row_bytes; // byte distance row B -> row C (mod 2^32)
            if (b.bop >= TOP_BIN2_BASE && b.bop < TOP_COUNT && !(((b.bop - TOP_BIN2_BASE) >> 2) & 1))
                t.lo = (uint32_t)((int32_t)b.lo * (int32_t)row_bytes); // row-row: byte distance row A -> row B
        }
        return t;
    };
    parallel_tree_ranges(p->n_trees, [&](int, int64_t tb, int64_t te) {
        for (size_t i = (size_t)p->tcode_off[(size_t)tb]; i < (size_t)p->tcode_off[(size_t)te]; i++) p->tcode[i] = threaded_words(p->fbcode[i]);
    });
    dbg_lap("threaded words");
    {
        const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
        const std::vector<int32_t> &off = p->folded ? p->fcode_off : p->code_off;
        match_const_sites(src, off, p->fbcode, p->tcode_off, p->n_trees, [](const BoundInstr &b) { return top_carries_const(b.bop); }, &p->tsite);
        p->site_gen++;
    }
    dbg_lap("constant sites");
    p->handler_base = base;
    p->end_handler = table[TOPX_END];
    for (uint32_t k = 0; k < TOPX_ENDV_COUNT; k++) p->endv_handler[k] = table[TOPX_ENDV_BASE + k];
    make_chained(p);
    dbg_lap("chained records");
    p->threaded = true;
    // ---- wave groups: the stream variants of waves 1 .. W - 1 (de_api_internal.h `waves`)
    p->waves = 1;
    p->var_stride = 0;
    p->ccode_w.clear();
    if (W > 1 && p->n_slots > 0 && fuse) {
        const bool ee = (p->options & DE_OPT_EARLY_EXIT) != 0, f32 = p->dtype == DE_F32;
        const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
        const std::vector<int32_t> &off = p->folded ? p->fcode_off : p->code_off;
        const int prb = p->prows ? p->n_features + p->n_slots : -1, n_prows = wave_prows;
        const size_t n_rec = p->ccode.size();
        p->ccode_w.resize(n_rec * (size_t)(W - 1));
        std::atomic<bool> same{true};
        for (int w = 1; w < W; w++) {
            BoundInstr *cw = p->ccode_w.data() + n_rec * (size_t)(w - 1);
            const int shift = n_prows + w * p->n_slots; // [X | slots of wave 0 | parameter rows | slots of wave 1 | ...]: slot s of wave w = row F + S + P + (w - 1) S + s
            const FuseRows rows_w{(uint32_t)(p->n_features + shift), (uint32_t)(p->n_features + shift + p->n_slots), shift, rows0.headroom};
            parallel_tree_ranges(p->n_trees, [&](int, int64_t tb, int64_t te) {
                std::vector<BoundInstr> b, f;
                // (the records of trees tb .. te - 1 behind the header in front of tree tb, which is tree tb - 1's end record: disjoint ranges)
                const size_t r0 = (size_t)p->ccode_off[(size_t)tb] - 1, r1 = te < p->n_trees ? (size_t)p->ccode_off[(size_t)te] - 1 : n_rec;
                std::copy(p->ccode.begin() + (long)r0, p->ccode.begin() + (long)r1, cw + r0);
                for (int64_t t = tb; t < te; t++) {
                    b.clear();
                    f.clear();
                    bind_tree(src.data() + off[(size_t)t], (size_t)(off[(size_t)t + 1] - off[(size_t)t]), ee, p->n_features, &b, prb, shift);
                    fuse_tree(b.data(), b.size(), &f, &rows_w);
                    const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1];
                    if ((int32_t)f.size() != i1 - i0) { same = false; return; }
                    const size_t h = (size_t)p->ccode_off[(size_t)t];
                    for (int32_t i = i0; i < i1; i++) {
                        if (f[(size_t)(i - i0)].bop != p->fbcode[(size_t)i].bop) { same = false; return; }
                        const BoundInstr tw = threaded_words(f[(size_t)(i - i0)]);
                        BoundInstr &r = cw[h + (size_t)(i - i0)]; // (make_chained `put`: the operand words; the handler words stay)
                        r.bop = tw.arg;
                        if (f32) r.arg = tw.lo;
                        else { r.lo = tw.lo; r.hi = tw.hi; }
                    }
                }
            });
        }
        if (same) {
            p->waves = W;
            // (the device keeps a variant in cbytes = (bcode + trees + 2) records: de_program_create)
            p->var_stride = (int64_t)(p->bcode.size() + (size_t)p->n_trees + 2);
        } else p->ccode_w.clear(); // (a fusion decided differently with shifted rows: never seen — rows are < 128 apart — but then one wave)
        dbg_lap("wave-group stream variants");
    } else if (W > 1 && fuse) p->waves = W; // no slot: one stream for every wave
    return DE_OK;
}

// The device layout of the threaded program (see de_kernels.hip): one head record, then per tree one record per instruction
// and an end record.  A record = {its operand word, its immediate, the address of the NEXT record's handler}: the head record
// names the first handler of tree 0, a tree's last instruction names h_tree_end (whose operand word is the tree's index), an
// end record names the first handler of the next tree — a chunk of consecutive trees is ONE chain and a handler knows where
// to jump before the record it has to fetch arrives.  BoundInstr fields by word: Float32 {bop: operand word, arg: imm, lo/hi:
// next handler}; Float64 {bop: operand word, arg: next handler (low half), lo/hi: imm}.
static void make_chained(de_program *p) {
    const bool f32 = p->dtype == DE_F32;
    p->ccode.assign(p->tcode.size() + (size_t)p->n_trees + 1, BoundInstr{0u, 0u, 0u, 0u});
    p->ccode_off.assign((size_t)p->n_trees + 1, 0);
    auto put = [&](BoundInstr &r, uint32_t la, uint32_t lo, uint32_t hi) { // operand words; the handler word is set by the predecessor
        r.bop = la;
        if (f32) r.arg = lo;
        else { r.lo = lo; r.hi = hi; }
    };
    auto name_next = [&](BoundInstr &r, uint64_t handler) {
        if (f32) { r.lo = (uint32_t)handler; r.hi = (uint32_t)(handler >> 32); }
        else r.arg = (uint32_t)handler;
    };
    // (the DE_NO_END_FUSE switch of round 2 measured <= 1 % and is gone)
    // a tree that finishes in a validity-tested hot operator runs that instruction and its end as ONE dispatch (h_chain_end);
    // a one-instruction tree keeps the plain form (the kernel's first call cannot tell the two apart)
    auto ev_of = [&](int64_t t) -> int {
        const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1];
        return i1 - i0 >= 2 ? topx_endv_of(p->fbcode[(size_t)i1 - 1].bop) : -1;
    };
    auto handler_of = [&](int32_t i, int32_t i1, int ev) -> uint64_t {
        return (i == i1 - 1 && ev >= 0) ? p->endv_handler[ev] : p->handler_base + p->tcode[(size_t)i].bop;
    };
    // Pass A, on the host threads: the records a tree OWNS — its instruction records and its end record (h = one end record per
    // preceding tree + the head record).  Pass B, serial (three writes per tree): what a tree writes into its PREDECESSOR's last two
    // records — the handler of its first instruction, and the header words over the end record's.
    parallel_tree_ranges(p->n_trees, [&](int, int64_t tb, int64_t te) {
        for (int64_t t = tb; t < te; t++) {
            const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1];
            const size_t h = (size_t)i0 + (size_t)t + 1;
            p->ccode_off[(size_t)t] = (int32_t)h;
            const int ev = ev_of(t);
            for (int32_t i = i0; i < i1; i++) {
                const BoundInstr &s = p->tcode[(size_t)i];
                put(p->ccode[h + (size_t)(i - i0)], s.arg, s.lo, s.hi);
                if (i > i0) name_next(p->ccode[h + (size_t)(i - i0) - 1], handler_of(i, i1, ev)); // in the record in front
            }
            put(p->ccode[h + (size_t)(i1 - i0)], (uint32_t)t, 0u, 0u); // end record (operand word: the tree's index, informational)
            if (i1 > i0) name_next(p->ccode[h + (size_t)(i1 - i0) - 1], p->end_handler);
        }
    });
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1];
        const size_t h = (size_t)p->ccode_off[(size_t)t];
        const int ev = ev_of(t);
        if (i1 > i0) {
            const uint64_t first = handler_of(i0, i1, ev);
            name_next(p->ccode[h - 1], first); // in the head record / the previous tree's end record
            // the previous tree finishes in an end-fused handler: ITS last instruction names this tree's first handler, stepping over its end record
            if (t > 0 && ev_of(t - 1) >= 0) name_next(p->ccode[h - 2], first);
        } else name_next(p->ccode[h - 1], p->end_handler);
        // the record in front of the tree (head record / previous tree's end record) is its HEADER: its immediate = the number of
        // instruction records of the tree, which is what h_tree_skip needs to step over a tree that is not evaluated
        // bit 31 = the tree finishes in an end-fused handler (its last instruction record names the next tree's first handler too):
        // what de_compact_live_kernel (de_kernels.hip) needs to re-link a tree behind another one
        put(p->ccode[h - 1], t == 0 ? 0u : (uint32_t)(t - 1), (uint32_t)(i1 - i0) | (ev >= 0 ? DE_HDR_FUSED_END : 0u), 0u);
    }
    if (p->n_trees > 0) name_next(p->ccode.back(), p->end_handler); // never followed: the last tree's end returns (left == 1)
    p->ccode_off[(size_t)p->n_trees] = (int32_t)p->ccode.size();
}
static inline void patch_chained_imm(de_program *p, int32_t c, uint32_t lo, uint32_t hi) {
    if (p->dtype == DE_F32) p->ccode[(size_t)c].arg = lo;
    else { p->ccode[(size_t)c].lo = lo; p->ccode[(size_t)c].hi = hi; }
    for (size_t w = 0; w * p->ccode.size() < p->ccode_w.size(); w++) { // the other waves' variants of the record (wave groups)
        BoundInstr &r = p->ccode_w[w * p->ccode.size() + (size_t)c];
        if (p->dtype == DE_F32) r.arg = lo;
        else { r.lo = lo; r.hi = hi; }
    }
}
// the stream variants of waves 1 .. (wave groups) to their places behind variant 0
static hipError_t upload_wave_variants(de_program *p) {
    for (size_t w = 0; p->var_stride && w * p->ccode.size() < p->ccode_w.size(); w++) {
        const hipError_t st = hipMemcpy(p->d_code + (w + 1) * (size_t)p->var_stride, p->ccode_w.data() + w * p->ccode.size(),
                                        p->ccode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice);
        if (st != hipSuccess) return st;
    }
    return hipSuccess;
}

static void recompute_host_ok(de_program *p) {
    const bool ee = (p->options & DE_OPT_EARLY_EXIT) != 0;
    parallel_for_trees(p->n_trees, [&](int64_t t) {
        bool ok_eval = true, ok_grad = true;
        for (int64_t k = p->const_off[t]; k < p->const_off[t + 1]; k++) {
            const bool fin = finite_in(p->dtype, p->consts[k]);
            ok_grad = ok_grad && fin;
            const uint8_t ch = p->const_checks[k];
            if (!fin && ((ch & CONST_CHECK_ALWAYS) || (ee && (ch & CONST_CHECK_EE)))) ok_eval = false;
        }
        p->host_ok_eval[t] = ok_eval;
        p->host_ok_grad[t] = ok_grad;
    }, 1024);
    // a constant subtree that evaluates to a non-finite value clears the flag — with the flag
    // semantics of the program's own options (dispatch_constant_tree tests unconditionally,
    // the Bumper path only under early_exit): that is exactly what `aux` was lowered with
    // — except for a subtree the reference never hands to dispatch_constant_tree (inner branch of a fused
    // 3-node kernel): its non-finite value is only noticed by the early-exit tests
    for (size_t j = 0; j < p->folds.size() && j < p->fold_ok.size(); j++)
        if (!p->fold_ok[j] && (p->folds[j].tested_always || ee)) p->host_ok_eval[(size_t)p->folds[j].tree] = 0;
}

static int create_impl(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                       int64_t n_trees, const void *consts, const int64_t *const_offsets, int32_t n_features,
                       int32_t n_params, uint32_t options, bool allow_fold, de_program_t **out_program,
                       const de_tape_node_t *cse_nodes = nullptr, const int64_t *cse_offsets = nullptr);

// Device copy of the host part of the eval flag: every de_eval starts from it with one device-to-device copy
// (a pageable host-to-device copy per call costs ~10 us, a fifth of a small-population call).
static int upload_ok_eval(de_ctx *c, de_program *p) {
    p->tab_ok_stale = true; // host_ok_grad moves with the constants too
    if (p->n_trees == 0) return DE_OK;
    if (!p->d_ok_eval) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_ok_eval), (size_t)std::max<int64_t>(p->n_trees, 1))); // (never taken since round 6: the arena holds it)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(p->d_ok_eval, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice));
    return DE_OK;
}

// (Re-)evaluate the folded constant subtrees — the IEEE-exact ones on the host, the others on the device — and patch their
// values into fcode.
// `aux_current`: the auxiliary program was created with the present constants this very moment (de_program_create: setting them
// again cost 0.9 of the 1.1 ms this step took for 10^4 trees).
static int refresh_folds(de_ctx *c, de_program *p, bool aux_current = false) {
    if (!p->folded || p->folds.empty()) return DE_OK; // (a CSE-only eval program has no constant subtrees to evaluate)
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const size_t nf = p->folds.size();
    p->fold_ok.assign(nf, 0);
    parallel_for_trees((int64_t)nf, [&](int64_t j) {
        if (p->fold_host[(size_t)j] != 1) return;
        const de_tape_node_t *nd = p->fold_nodes.data() + p->fold_noff[(size_t)j];
        const int64_t n = p->fold_noff[(size_t)j + 1] - p->fold_noff[(size_t)j];
        const int64_t *csrc = p->aux_const_src.data() + p->fold_coff[(size_t)j];
        double v;
        bool ok;
        if (p->dtype == DE_F32) { float f; ok = host_fold_eval<float>(nd, n, p->consts.data(), csrc, &f); v = (double)f; }
        else ok = host_fold_eval<double>(nd, n, p->consts.data(), csrc, &v);
        p->fold_ok[(size_t)j] = ok ? 1 : 0;
        write_imm(p->fcode[(size_t)p->folds[(size_t)j].instr], p->dtype, v);
    }, 256);
    if (!p->kfold.empty()) {
        // the subtrees with other operators: one thread each on the device (de_fold_kernel), the operators' own device code.
        // `aux_current`: the image uploaded at creation already holds these constants.
        const size_t nk = p->kfold.size();
        HIP_TRY(c, hipSetDevice(c->device));
        if (!aux_current) {
            std::vector<unsigned char> cv(std::max<size_t>(p->kf_csrc.size(), 1) * es);
            for (size_t k = 0; k < p->kf_csrc.size(); k++) {
                const double v = p->consts[(size_t)p->kf_csrc[k]];
                if (p->dtype == DE_F32) reinterpret_cast<float *>(cv.data())[k] = (float)v;
                else reinterpret_cast<double *>(cv.data())[k] = v;
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream)); // (an earlier launch may still read the values)
            if (!p->kf_csrc.empty()) HIP_TRY(c, hipMemcpy(p->d_kf + p->kf_o_cvals, cv.data(), p->kf_csrc.size() * es, hipMemcpyHostToDevice));
        }
        HIP_TRY(c, launch_fold(p->dtype, p->d_kf, reinterpret_cast<const int64_t *>(p->d_kf + p->kf_o_noff), reinterpret_cast<const int64_t *>(p->d_kf + p->kf_o_coff),
                               p->d_kf + p->kf_o_cvals, (int64_t)nk, p->d_kf + p->kf_o_out, reinterpret_cast<uint8_t *>(p->d_kf + p->kf_o_ok), c->stream));
        std::vector<unsigned char> res(p->kf_bytes - p->kf_o_out);
        HIP_TRY(c, hipMemcpyAsync(res.data(), p->d_kf + p->kf_o_out, res.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const unsigned char *okb = res.data() + (p->kf_o_ok - p->kf_o_out);
        for (size_t k = 0; k < nk; k++) {
            const size_t j = (size_t)p->kfold[k];
            const double v = p->dtype == DE_F32 ? (double)reinterpret_cast<const float *>(res.data())[k] : reinterpret_cast<const double *>(res.data())[k];
            p->fold_ok[j] = okb[k];
            write_imm(p->fcode[(size_t)p->folds[j].instr], p->dtype, v);
        }
    }
    if (!p->aux) return DE_OK;
    const size_t na = p->aux_fold.size();
    int rc = DE_OK;
    if (!aux_current) {
        std::vector<unsigned char> ac(std::max<size_t>(p->aux_csrc.size(), 1) * es);
        for (size_t k = 0; k < p->aux_csrc.size(); k++) {
            const double v = p->consts[(size_t)p->aux_csrc[k]];
            if (p->dtype == DE_F32) reinterpret_cast<float *>(ac.data())[k] = (float)v;
            else reinterpret_cast<double *>(ac.data())[k] = v;
        }
        rc = de_program_set_consts(p->aux, ac.data());
        if (rc != DE_OK) return fail(c, rc, "constant folding: %s", p->aux->ctx->err.c_str());
    }
    std::vector<unsigned char> X(std::max<size_t>((size_t)p->n_features, 1) * es, 0), out(na * es);
    std::vector<uint8_t> aok(na, 0);
    rc = de_eval(c, p->aux, X.data(), 1, std::max<int64_t>(p->n_features, 1), nullptr, out.data(), 1, aok.data());
    if (rc != DE_OK) return rc;
    for (size_t a = 0; a < na; a++) {
        const size_t j = (size_t)p->aux_fold[a];
        const double v = p->dtype == DE_F32 ? (double)reinterpret_cast<float *>(out.data())[a]
                                            : reinterpret_cast<double *>(out.data())[a];
        p->fold_ok[j] = aok[a];
        write_imm(p->fcode[(size_t)p->folds[j].instr], p->dtype, v);
    }
    return DE_OK;
}


int de_program_create(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                      int64_t n_trees, const void *consts, const int64_t *const_offsets,
                      int32_t n_features, int32_t n_params, uint32_t options, de_program_t **out_program) {
    const char *nf = getenv("DE_NO_FOLD");
    if (!ctx) return DE_ERR_INVALID_ARG;
    DE_NOTHROW(ctx, create_impl(ctx, dtype, nodes, node_offsets, n_trees, consts, const_offsets, n_features, n_params, options,
                                !(nf && *nf == '1'), out_program));
}

int de_program_create_cse(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                          const de_tape_node_t *cse_nodes, const int64_t *cse_offsets, int64_t n_trees, const void *consts,
                          const int64_t *const_offsets, int32_t n_features, int32_t n_params, uint32_t options,
                          de_program_t **out_program) {
    const char *nf = getenv("DE_NO_FOLD"), *nc = getenv("DE_NO_CSE");
    const bool fold = !(nf && *nf == '1'), cse = !(nc && *nc == '1');
    if (!ctx) return DE_ERR_INVALID_ARG;
    if (n_trees > 0 && cse_nodes && !cse_offsets) return fail(ctx, DE_ERR_INVALID_ARG, "cse_offsets is null");
    DE_NOTHROW(ctx, create_impl(ctx, dtype, nodes, node_offsets, n_trees, consts, const_offsets, n_features, n_params, options, fold, out_program,
                                fold && cse ? cse_nodes : nullptr, cse_offsets));
}

// The eval program of tree t is lowered from its CSE tape when the caller supplied one (a GraphNode tree: shared subtrees
// appear once, de_program_create_cse); everything else — gradients, constant bookkeeping, flags — follows the expanded tape.
static int create_impl(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                       int64_t n_trees, const void *consts, const int64_t *const_offsets, int32_t n_features,
                       int32_t n_params, uint32_t options, bool allow_fold, de_program_t **out_program,
                       const de_tape_node_t *cse_nodes, const int64_t *cse_offsets) {
    if (!ctx) return DE_ERR_INVALID_ARG;
    if (!out_program) return fail(ctx, DE_ERR_INVALID_ARG, "out_program is null");
    *out_program = nullptr;
    if (dtype != DE_F32 && dtype != DE_F64) return fail(ctx, DE_ERR_INVALID_ARG, "dtype must be DE_F32 or DE_F64");
    if (n_trees < 0 || n_features < 0 || n_params < 0 || n_features > 65535 || n_params > 65535)
        return fail(ctx, DE_ERR_INVALID_ARG, "bad sizes");
    if (n_trees > 0 && (!nodes || !node_offsets || !const_offsets))
        return fail(ctx, DE_ERR_INVALID_ARG, "null tape pointers");
    if (n_trees > 0x7fffffff) return fail(ctx, DE_ERR_UNSUPPORTED, "too many trees");
    std::unique_ptr<de_program> p(new (std::nothrow) de_program());
    if (!p) return fail(ctx, DE_ERR_HIP, "out of host memory");
    adopt_parked(ctx, p.get());
    // DE_DEBUG_TIMING: microseconds per phase of the creation on stderr (tools/bench_create.py)
    const bool timing = getenv("DE_DEBUG_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[de_program_create %lld trees%s] %-28s %9.1f us\n", (long long)n_trees, allow_fold ? "" : " (aux)", what,
                std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = std::chrono::steady_clock::now();
    };
    try {
        p->ctx = ctx;
        p->dtype = dtype;
        p->options = options;
        p->n_features = n_features;
        p->n_params = n_params;
        p->n_trees = n_trees;
        LowerOptions lo;
        lo.early_exit = (options & DE_OPT_EARLY_EXIT) != 0;
        lo.fuse1 = (options & DE_OPT_FUSE_DEG1) != 0;
        lo.fuse2 = (options & DE_OPT_FUSE_DEG2) != 0;
        lo.bumper = (options & DE_OPT_BUMPER_CHECKS) != 0;
        lo.n_features = n_features;
        lo.n_params = n_params;
        lo.dtype = dtype;
        p->code_off.assign((size_t)n_trees + 1, 0);
        p->const_off.assign((size_t)n_trees + 1, 0);
        p->n_consts_tree.assign((size_t)n_trees, 0);
        p->host_ok_eval.assign((size_t)n_trees, 1);
        p->host_ok_grad.assign((size_t)n_trees, 1);
        const int64_t total_consts = n_trees ? const_offsets[n_trees] - const_offsets[0] : 0;
        if (total_consts < 0) return fail(ctx, DE_ERR_INVALID_ARG, "const_offsets not monotone");
        if (total_consts > 0 && !consts) return fail(ctx, DE_ERR_INVALID_ARG, "consts is null");
        p->consts.resize((size_t)total_consts);
        p->const_instr.assign((size_t)total_consts, -1);
        p->const_checks.assign((size_t)total_consts, 0);
        for (int64_t t = 0; t < n_trees; t++)
            if (node_offsets[t + 1] < node_offsets[t] || const_offsets[t + 1] < const_offsets[t])
                return fail(ctx, DE_ERR_INVALID_ARG, "offsets not monotone at tree %lld", (long long)t);
        // both lowerings of every tree (plain, and with constant subtrees folded), on host threads
        struct Lowered { TreeProgram plain, folded; int rc = DE_OK, rcf = DE_OK; bool cse = false, cse_plain = false; std::string why; };
        std::vector<Lowered> low((size_t)n_trees);
        {
            LowerOptions lof = lo;
            lof.fold = true;
            std::atomic<bool> oom{false};
            parallel_for_trees(n_trees, [&](int64_t t) {
                Lowered &L = low[(size_t)t];
                const int64_t n0 = node_offsets[t], c0 = const_offsets[t];
                try {
                    L.rc = lower_tree(nodes + n0, node_offsets[t + 1] - n0, const_offsets[t + 1] - c0, lo, &L.plain, &L.why);
                    if (L.rc == DE_OK && cse_nodes && cse_offsets[t + 1] > cse_offsets[t] && !getenv("DE_NO_GRAD_CSE")) {
                        // GraphNode sharing in the GENERIC program too (round 3; the gradient kernels, eval_diff and the unfolded
                        // eval run it): a shared subtree's dual number is computed once into a persistent slot and read by every
                        // consumer — the reference evaluates it once per parent with the same arithmetic, so values, Jacobian rows
                        // of features / parameters and flags are those of the expansion.  A constant inside a shared subtree keeps
                        // the gradient row of its FIRST occurrence, which receives every consumer's contribution; the rows of its
                        // later occurrences stay zero (callers sum the occurrence rows: the reference's shared NodeIndex row).
                        TreeProgram pc;
                        std::string why2;
                        LowerOptions loc = lo;
                        loc.cse = true;
                        if (lower_tree(cse_nodes + cse_offsets[t], cse_offsets[t + 1] - cse_offsets[t], const_offsets[t + 1] - c0, loc, &pc, &why2) == DE_OK) {
                            L.plain = std::move(pc);
                            L.cse_plain = true;
                        }
                    }
                    if (L.rc == DE_OK && allow_fold) {
                        if (cse_nodes && cse_offsets[t + 1] > cse_offsets[t]) {
                            LowerOptions loc = lof;
                            loc.cse = true;
                            L.rcf = lower_tree(cse_nodes + cse_offsets[t], cse_offsets[t + 1] - cse_offsets[t], const_offsets[t + 1] - c0, loc, &L.folded, &L.why);
                            L.cse = L.rcf == DE_OK;
                            if (L.rcf == DE_ERR_UNSUPPORTED) {
                                // the CSE form does not fit (spill slots + shared rows > 16, a share in an unsupported position): the
                                // expanded tape has the same values and flags (the reference evaluates a shared node once per
                                // parent), so this tree alone runs expanded instead of failing the whole population
                                L.folded = TreeProgram();
                                L.why.clear();
                                L.rcf = lower_tree(nodes + n0, node_offsets[t + 1] - n0, const_offsets[t + 1] - c0, lof, &L.folded, &L.why);
                            }
                        } else L.rcf = lower_tree(nodes + n0, node_offsets[t + 1] - n0, const_offsets[t + 1] - c0, lof, &L.folded, &L.why);
                    }
                } catch (const std::bad_alloc &) { oom = true; }
            });
            if (oom) return fail(ctx, DE_ERR_HIP, "out of host memory");
        }
        lap("lower (host threads)");
        // merge: offsets by a serial prefix sum, the copies on the host threads (every tree writes slices of its own)
        {
            uint64_t total = 0;
            for (int64_t t = 0; t < n_trees; t++) {
                if (low[(size_t)t].rc != DE_OK) return fail(ctx, low[(size_t)t].rc, "tree %lld: %s", (long long)t, low[(size_t)t].why.c_str());
                total += low[(size_t)t].plain.code.size();
                if (total > 0x7fff0000u) return fail(ctx, DE_ERR_UNSUPPORTED, "program too large");
                p->code_off[(size_t)t + 1] = (int32_t)total;
            }
            p->code.resize((size_t)total);
            struct Part { int32_t n_slots = 0; bool cse = false, params = false; int64_t nodes = 0; } part[HOST_RANGES_MAX];
            parallel_tree_ranges(n_trees, [&](int wk, int64_t tb, int64_t te) {
                Part &pt = part[wk];
                for (int64_t t = tb; t < te; t++) {
                    const int64_t n0 = node_offsets[t], n1 = node_offsets[t + 1];
                    const int64_t c0 = const_offsets[t], c1 = const_offsets[t + 1];
                    TreeProgram &tp = low[(size_t)t].plain;
                    const int64_t cb = c0 - const_offsets[0];
                    p->const_off[(size_t)t + 1] = cb + (c1 - c0); // (entry t is tree t - 1's, entry 0 stays 0)
                    p->n_consts_tree[(size_t)t] = (int32_t)(c1 - c0);
                    const int32_t ib = p->code_off[(size_t)t];
                    for (int64_t k = 0; k < c1 - c0; k++) {
                        const double v = dtype == DE_F32 ? (double)static_cast<const float *>(consts)[c0 + k]
                                                         : static_cast<const double *>(consts)[c0 + k];
                        p->consts[(size_t)(cb + k)] = v;
                        // (a CSE lowering has no instruction for the later occurrences of a constant inside a shared subtree: -1)
                        p->const_instr[(size_t)(cb + k)] = tp.const_instr[(size_t)k] >= 0 ? ib + tp.const_instr[(size_t)k] : -1;
                        p->const_checks[(size_t)(cb + k)] = tp.const_checks[(size_t)k];
                        if (tp.const_instr[(size_t)k] >= 0) write_imm(tp.code[(size_t)tp.const_instr[(size_t)k]], dtype, v);
                    }
                    std::copy(tp.code.begin(), tp.code.end(), p->code.begin() + ib);
                    pt.cse = pt.cse || low[(size_t)t].cse_plain;
                    pt.n_slots = std::max(pt.n_slots, tp.n_slots);
                    pt.params = pt.params || tp.uses_params;
                    pt.nodes += n1 - n0;
                }
            });
            for (const Part &pt : part) {
                p->cse_generic = p->cse_generic || pt.cse;
                p->n_slots = std::max(p->n_slots, pt.n_slots);
                p->uses_params = p->uses_params || pt.params;
                p->n_nodes += pt.nodes;
            }
        }
        lap("merge plain");
        // ---- folded lowering of the eval program + the auxiliary population of constant subtrees
        if (allow_fold) {
            lo.fold = true;
            std::vector<de_tape_node_t> &anodes = p->fold_nodes; // (retained: the host-folded subtrees are re-evaluated from them)
            std::vector<int64_t> &anoff = p->fold_noff, &acoff = p->fold_coff;
            bool any_cse = false;
            p->fcode_off.assign((size_t)n_trees + 1, 0);
            p->fconst_instr.assign((size_t)total_consts, -1);
            // offsets of every tree's instructions, folds, auxiliary tape nodes and auxiliary constants by a serial prefix sum ...
            std::vector<int64_t> fold0((size_t)n_trees + 1, 0), anode0((size_t)n_trees + 1, 0), acs0((size_t)n_trees + 1, 0);
            {
                uint64_t total = 0;
                for (int64_t t = 0; t < n_trees; t++) {
                    const TreeProgram &tp = low[(size_t)t].folded;
                    if (low[(size_t)t].rcf != DE_OK)
                        return fail(ctx, low[(size_t)t].rcf, "tree %lld (folded): %s", (long long)t, low[(size_t)t].why.c_str());
                    total += tp.code.size();
                    if (total > 0x7fff0000u) return fail(ctx, DE_ERR_UNSUPPORTED, "program too large");
                    p->fcode_off[(size_t)t + 1] = (int32_t)total;
                    int64_t nn = 0, nc = 0;
                    for (const FoldSpan &sp : tp.folds) { nn += sp.node_end - sp.node_begin; nc += sp.const_end - sp.const_begin; }
                    fold0[(size_t)t + 1] = fold0[(size_t)t] + (int64_t)tp.folds.size();
                    anode0[(size_t)t + 1] = anode0[(size_t)t] + nn;
                    acs0[(size_t)t + 1] = acs0[(size_t)t] + nc;
                }
                p->fcode.resize((size_t)total);
            }
            const size_t n_folds = (size_t)fold0[(size_t)n_trees];
            p->folds.resize(n_folds);
            anodes.resize((size_t)anode0[(size_t)n_trees]);
            p->aux_const_src.resize((size_t)acs0[(size_t)n_trees]);
            anoff.assign(n_folds + 1, 0);
            acoff.assign(n_folds + 1, 0);
            // ... the copies on the host threads
            struct PartF { int32_t n_slots = 0; bool cse = false; } partf[HOST_RANGES_MAX];
            parallel_tree_ranges(n_trees, [&](int wk, int64_t tb, int64_t te) {
                PartF &pt = partf[wk];
                for (int64_t t = tb; t < te; t++) {
                    const int64_t n0 = node_offsets[t];
                    const int64_t c0 = const_offsets[t], c1 = const_offsets[t + 1];
                    TreeProgram &tp = low[(size_t)t].folded;
                    const bool is_cse = low[(size_t)t].cse;
                    const de_tape_node_t *src_nodes = is_cse ? cse_nodes + cse_offsets[t] : nodes + n0; // the tape the fold spans index
                    pt.cse = pt.cse || is_cse;
                    pt.n_slots = std::max(pt.n_slots, tp.n_slots); // a CSE program keeps one persistent row per shared subtree
                    const int64_t cb = c0 - const_offsets[0];
                    const int32_t ib = p->fcode_off[(size_t)t];
                    for (int64_t k = 0; k < c1 - c0; k++) {
                        const int32_t ci = tp.const_instr[(size_t)k];
                        if (ci < 0) continue; // constant lives inside a folded subtree
                        p->fconst_instr[(size_t)(cb + k)] = ib + ci;
                        write_imm(tp.code[(size_t)ci], dtype, p->consts[(size_t)(cb + k)]);
                    }
                    size_t an = (size_t)anode0[(size_t)t], ac = (size_t)acs0[(size_t)t];
                    for (size_t f = 0; f < tp.folds.size(); f++) {
                        const FoldSpan &sp = tp.folds[f];
                        const size_t fi = (size_t)fold0[(size_t)t] + f;
                        p->folds[fi] = {(int32_t)t, ib + tp.const_instr[(size_t)(c1 - c0) + f], sp.tested_always};
                        for (int32_t q = sp.node_begin; q < sp.node_end; q++) {
                            de_tape_node_t nd = src_nodes[q];
                            if (nd.degree == 0 && nd.op == DE_LEAF_CONST) nd.arg = (uint16_t)(nd.arg - sp.const_begin);
                            anodes[an++] = nd;
                        }
                        for (int32_t q = sp.const_begin; q < sp.const_end; q++) p->aux_const_src[ac++] = cb + q;
                        anoff[fi + 1] = (int64_t)an;
                        acoff[fi + 1] = (int64_t)ac;
                    }
                    std::copy(tp.code.begin(), tp.code.end(), p->fcode.begin() + ib);
                    { Lowered done; std::swap(done, low[(size_t)t]); } // both lowerings of the tree are merged: released here, by the thread that is at it
                }
            });
            for (const PartF &pt : partf) {
                any_cse = any_cse || pt.cse;
                p->n_slots = std::max(p->n_slots, pt.n_slots);
            }
            lap("merge folded");
            if (!p->folds.empty()) {
                const size_t es = dtype == DE_F32 ? 4 : 8;
                // which folds stay on the host: subtrees of + - * / only (the turbo division is not IEEE: such programs fold everything on
                // the device, with the operators they evaluate with; DE_NO_HOST_FOLD=1: everything on the device, for A/B tests)
                const char *nh = getenv("DE_NO_HOST_FOLD");
                const bool host_fold = !(nh && *nh == '1') && !(options & DE_OPT_TURBO);
                // ... and which go to de_fold_kernel (everything else whose evaluation stack fits; a turbo program evaluates with other
                // operators than that kernel has: its subtrees stay with the auxiliary program; DE_NO_KERNEL_FOLD=1 for A/B tests)
                const char *nk_env = getenv("DE_NO_KERNEL_FOLD");
                const bool kernel_fold = !(nk_env && *nk_env == '1') && !(options & DE_OPT_TURBO);
                p->fold_host.assign(n_folds, 0);
                parallel_for_trees((int64_t)n_folds, [&](int64_t j) {
                    const de_tape_node_t *nd = anodes.data() + anoff[(size_t)j];
                    const int64_t n = anoff[(size_t)j + 1] - anoff[(size_t)j];
                    if (host_fold && host_foldable(nd, n)) { p->fold_host[(size_t)j] = 1; return; }
                    if (!kernel_fold) return;
                    int depth = 0, worst = 0;
                    for (int64_t i = 0; i < n; i++) { depth += 1 - (int)nd[i].degree; worst = std::max(worst, depth); }
                    if (worst <= DE_FOLD_STACK) p->fold_host[(size_t)j] = 2;
                }, 512);
                // the kernel's image and the auxiliary population: tape slices, offsets and constant sources of their subtrees, in fold order —
                // positions by one serial prefix sum over the folds, the copies on the host threads
                std::vector<de_tape_node_t> knodes, xnodes;
                std::vector<int64_t> knoff{0}, kcoff{0}, xnoff{0}, xcoff{0};
                {
                    std::vector<int64_t> pos_n(n_folds), pos_c(n_folds);
                    std::vector<int32_t> ord(n_folds);
                    int64_t kn = 0, kc = 0, xn = 0, xc = 0;
                    p->kfold.clear();
                    p->aux_fold.clear();
                    for (size_t j = 0; j < n_folds; j++) {
                        const int64_t nn = anoff[j + 1] - anoff[j], nc = acoff[j + 1] - acoff[j];
                        if (p->fold_host[j] == 2) {
                            ord[j] = (int32_t)p->kfold.size(); p->kfold.push_back((int32_t)j);
                            pos_n[j] = kn; pos_c[j] = kc; kn += nn; kc += nc;
                            knoff.push_back(kn); kcoff.push_back(kc);
                        } else if (p->fold_host[j] == 0) {
                            ord[j] = (int32_t)p->aux_fold.size(); p->aux_fold.push_back((int32_t)j);
                            pos_n[j] = xn; pos_c[j] = xc; xn += nn; xc += nc;
                            xnoff.push_back(xn); xcoff.push_back(xc);
                        }
                    }
                    knodes.resize((size_t)kn);
                    xnodes.resize((size_t)xn);
                    p->kf_csrc.resize((size_t)kc);
                    p->aux_csrc.resize((size_t)xc);
                    parallel_for_trees((int64_t)n_folds, [&](int64_t jj) {
                        const size_t j = (size_t)jj;
                        if (p->fold_host[j] == 1) return;
                        const bool k = p->fold_host[j] == 2;
                        std::copy(anodes.begin() + anoff[j], anodes.begin() + anoff[j + 1], (k ? knodes : xnodes).begin() + pos_n[j]);
                        std::copy(p->aux_const_src.begin() + acoff[j], p->aux_const_src.begin() + acoff[j + 1], (k ? p->kf_csrc : p->aux_csrc).begin() + pos_c[j]);
                    }, 512);
                }
                if (!p->kfold.empty()) {
                    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
                    const size_t nk = p->kfold.size();
                    p->kf_o_noff = al(knodes.size() * sizeof(de_tape_node_t));
                    p->kf_o_coff = p->kf_o_noff + al(knoff.size() * sizeof(int64_t));
                    p->kf_o_cvals = p->kf_o_coff + al(kcoff.size() * sizeof(int64_t));
                    p->kf_o_out = p->kf_o_cvals + al(std::max<size_t>(p->kf_csrc.size(), 1) * es);
                    p->kf_o_ok = p->kf_o_out + al(nk * es);
                    p->kf_bytes = p->kf_o_ok + al(nk);
                    std::vector<unsigned char> img(p->kf_o_out, 0);
                    std::memcpy(img.data(), knodes.data(), knodes.size() * sizeof(de_tape_node_t));
                    std::memcpy(img.data() + p->kf_o_noff, knoff.data(), knoff.size() * sizeof(int64_t));
                    std::memcpy(img.data() + p->kf_o_coff, kcoff.data(), kcoff.size() * sizeof(int64_t));
                    for (size_t k = 0; k < p->kf_csrc.size(); k++) {
                        const double v = p->consts[(size_t)p->kf_csrc[k]];
                        if (dtype == DE_F32) reinterpret_cast<float *>(img.data() + p->kf_o_cvals)[k] = (float)v;
                        else reinterpret_cast<double *>(img.data() + p->kf_o_cvals)[k] = v;
                    }
                    HIP_TRY(ctx, hipSetDevice(ctx->device));
                    const hipError_t kst = prog_malloc(ctx, reinterpret_cast<void **>(&p->d_kf), p->kf_bytes);
                    if (kst != hipSuccess) return fail(ctx, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(kst));
                    HIP_TRY(ctx, hipMemcpy(p->d_kf, img.data(), img.size(), hipMemcpyHostToDevice));
                }
                p->folded = true;
                lap("folds: classify, kernel image, auxiliary tapes");
                if (!p->aux_fold.empty()) {
                    std::vector<unsigned char> ac(std::max<size_t>(p->aux_csrc.size(), 1) * es, 0);
                    for (size_t k = 0; k < p->aux_csrc.size(); k++) {
                        const double v = p->consts[(size_t)p->aux_csrc[k]];
                        if (dtype == DE_F32) reinterpret_cast<float *>(ac.data())[k] = (float)v;
                        else reinterpret_cast<double *>(ac.data())[k] = v;
                    }
                    int rc = create_impl(ctx, dtype, xnodes.data(), xnoff.data(), (int64_t)p->aux_fold.size(), ac.data(),
                                         xcoff.data(), n_features, 0, options, false, &p->aux);
                    if (rc != DE_OK) return rc;
                    lap("aux program (create)");
                }
                int rc = refresh_folds(ctx, p.get(), true);
                if (rc != DE_OK) return rc;
                lap("folds (evaluate: host + kernel + aux)");
            } else if (any_cse) {
                p->folded = true; // the eval program is the CSE lowering even without a constant subtree to fold
            } else {
                p->fcode.clear();
                p->fcode_off.clear();
            }
        }
        // the per-tree lowerings are ~16 small vectors each: released on the threads that allocated them (one thread took 5 ms for 10^4 trees)
        if (!allow_fold) parallel_for_trees(n_trees, [&](int64_t t) { Lowered done; std::swap(done, low[(size_t)t]); }); // (with folding: released in the merge above)
        lap("release lowerings");
        recompute_host_ok(p.get());
        rebind(p.get());
        lap("bind");
    } catch (const std::bad_alloc &) {
        return fail(ctx, DE_ERR_HIP, "out of host memory");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        int rc = DE_OK;
        try { rc = make_threaded(ctx, p.get()); } catch (const std::bad_alloc &) { rc = fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (rc != DE_OK) return rc;
    }
    lap("threaded + chained records");
    // one trailing pad instruction: the flat-switch interpreter prefetches code[pc + 1]; the chained form of the
    // threaded kernel has one end record per tree and a head record (and the fused form is never longer than the bound one)
    const size_t cbytes = (p->bcode.size() + (size_t)p->n_trees + 2) * sizeof(BoundInstr); // + head record + one of padding
    {
        // the early-exit walk (h_tree_skip) rebuilds record addresses from their low 32 bits: the stream must lie inside one
        // 4 GiB window.  An allocation that straddles a boundary (once in ~10^4 for a 400 KB stream) is set aside and redone.
        // (a threaded program allocates the stream twice: the second half receives the re-linked stream of the live trees, de_compact_live_kernel)
        // ONE device arena per eval program (round 6): [record stream | its second half for the compacted live trees | tree offsets |
        // compaction control ints | initial flags], one allocation from the context's pool; a small program (the one-tree call of
        // de_eval_tree_array) goes up in ONE copy from a zero-filled host image, a large one in one memset + three copies.
        // (wave groups: `waves` variants of the stream var_stride = cbytes / 16 records apart, and as much room again for their compacted forms)
        const size_t nvar = p->threaded && p->var_stride ? (size_t)p->waves : 1;
        if (nvar > 1 && (size_t)p->var_stride * sizeof(BoundInstr) != cbytes) return fail(ctx, DE_ERR_HIP, "wave-group stream variants: stride and arena disagree");
        const size_t abytes = p->threaded ? 2 * nvar * cbytes : cbytes;
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t off_bytes = p->bcode_off.size() * sizeof(int32_t);
        const size_t ints_bytes = p->threaded ? ((size_t)2 * (size_t)p->n_trees + 5) * sizeof(int32_t) : 0;
        const size_t o_off = al(abytes), o_ints = o_off + al(off_bytes), o_ok = o_ints + al(ints_bytes);
        const size_t total = o_ok + al((size_t)std::max<int64_t>(p->n_trees, 1));
        const hipError_t ast = prog_malloc(ctx, reinterpret_cast<void **>(&p->d_code), total); // (one 4 GiB window: prog_malloc's contract)
        if (ast != hipSuccess) return fail(ctx, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(ast));
        if (!in_one_window(p->d_code, abytes)) return fail(ctx, DE_ERR_HIP, "instruction stream straddles a 4 GiB boundary");
        char *base = reinterpret_cast<char *>(p->d_code);
        p->eval_arena = true; // (d_code_off, d_compact_ints, d_ok_eval live inside d_code's allocation: never freed on their own)
        p->d_code_off = reinterpret_cast<int32_t *>(base + o_off);
        p->d_ok_eval = reinterpret_cast<uint8_t *>(base + o_ok);
        if (p->threaded) {
            p->d_compact_code = p->d_code + nvar * cbytes / sizeof(BoundInstr);
            p->d_compact_ints = reinterpret_cast<int32_t *>(base + o_ints);
        }
        lap("hipMalloc (arena)");
        const std::vector<BoundInstr> &stream = p->threaded ? p->ccode : p->bcode;
        const std::vector<int32_t> &offs = p->threaded ? p->ccode_off : p->bcode_off;
        hipError_t st = hipSuccess;
        if (total <= (size_t)(128u << 10)) {
            // (the second half of a threaded stream needs no initial content: de_compact_live_kernel writes what the launch proper reads)
            std::vector<unsigned char> img(total, 0);
            if (!stream.empty()) std::memcpy(img.data(), stream.data(), stream.size() * sizeof(BoundInstr));
            for (size_t w = 1; w < nvar; w++)
                std::memcpy(img.data() + w * cbytes, p->ccode_w.data() + (w - 1) * p->ccode.size(), p->ccode.size() * sizeof(BoundInstr));
            std::memcpy(img.data() + o_off, offs.data(), off_bytes);
            if (p->n_trees > 0) std::memcpy(img.data() + o_ok, p->host_ok_eval.data(), (size_t)p->n_trees);
            st = hipMemcpy(base, img.data(), total, hipMemcpyHostToDevice);
        } else {
            st = hipMemset(p->d_code, 0, nvar * cbytes);
            if (st == hipSuccess && !stream.empty()) st = hipMemcpy(p->d_code, stream.data(), stream.size() * sizeof(BoundInstr), hipMemcpyHostToDevice);
            if (st == hipSuccess) st = upload_wave_variants(p.get());
            if (st == hipSuccess) st = hipMemcpy(p->d_code_off, offs.data(), off_bytes, hipMemcpyHostToDevice);
            if (st == hipSuccess && p->n_trees > 0) st = hipMemcpy(p->d_ok_eval, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice);
        }
        if (st != hipSuccess) {
            prog_free(ctx, p->d_code);
            p->d_code = nullptr;
            return fail(ctx, DE_ERR_HIP, "program upload failed: %s", hipGetErrorString(st));
        }
    }
    lap("memset + upload (stream, offsets, flags)");
    if (getenv("DE_VERIFY") && *getenv("DE_VERIFY") == '1') {
        const int rc = de_program_verify(p.get());
        if (rc != DE_OK) return rc;
    }
    *out_program = p.release();
    return DE_OK;
}

static int set_consts_impl(de_program_t *p, const void *consts);
static int set_consts_nothrow(de_program_t *p, const void *consts) {
    if (!p) return DE_ERR_INVALID_ARG;
    DE_NOTHROW(p->ctx, set_consts_impl(p, consts));
}
int de_program_set_consts(de_program_t *p, const void *consts) {
    const int rc = set_consts_nothrow(p, consts);
    if (rc == DE_OK && p && getenv("DE_VERIFY") && *getenv("DE_VERIFY") == '1') return de_program_verify(p);
    return rc;
}
static int set_consts_impl(de_program_t *p, const void *consts) {
    if (!p) return DE_ERR_INVALID_ARG;
    de_ctx *ctx = p->ctx;
    if (!consts && !p->consts.empty()) return fail(ctx, DE_ERR_INVALID_ARG, "consts is null");
    p->consts_gen++; // (the cached certificate program belongs to the old constants)
    const bool timing = getenv("DE_DEBUG_TIMING") != nullptr; // stderr: microseconds per phase
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    const auto t0 = now();
    // (every loop of this function over constants, trees or sites writes slots of its own: on the host pool; 10^4 trees: 0.85 -> see DESIGN 9.1)
    try {
        parallel_tree_ranges((int64_t)p->consts.size(), [&](int, int64_t kb, int64_t ke) {
            for (size_t k = (size_t)kb; k < (size_t)ke; k++) {
                const double v = p->dtype == DE_F32 ? (double)static_cast<const float *>(consts)[k]
                                                    : static_cast<const double *>(consts)[k];
                p->consts[k] = v;
                if (p->const_instr[k] >= 0) write_imm(p->code[(size_t)p->const_instr[k]], p->dtype, v);
                if (p->folded && p->fconst_instr[k] >= 0) write_imm(p->fcode[(size_t)p->fconst_instr[k]], p->dtype, v);
            }
        }, 4096);
    } catch (const std::bad_alloc &) { return fail(ctx, DE_ERR_HIP, "out of host memory"); }
    const auto t1 = now();
    if (p->folded) {
        int rc = DE_OK;
        try { rc = refresh_folds(ctx, p); } catch (const std::bad_alloc &) { rc = fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (rc != DE_OK) return rc;
    }
    const auto t2 = now();
    recompute_host_ok(p);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const int rc = upload_ok_eval(ctx, p);
        if (rc != DE_OK) return rc;
    }
    const auto t3 = now();
    // Same tree shapes, new immediates: patch the bits where they live (the optimiser calls this once per
    // step — re-binding 10^4 trees costs milliseconds, the kernel it feeds a few hundred microseconds).
    const char *nopatch = getenv("DE_NO_CONST_PATCH");
    if (!(nopatch && *nopatch == '1') && p->threaded && !p->tsite.empty() && !p->bsite.empty()) {
        const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
        const bool gpatch = p->d_gcode && !p->gcode_stale && !p->gbsite.empty();
        const bool tpatch = gpatch && p->gt_valid && !p->gtsite_of_gb.empty();
        const bool rpatch = gpatch && p->rt_valid && !p->rtsite_of_gb.empty();
        if (p->lists_gen != p->site_gen) { // one pass over all instructions, then only the immediates are visited
            p->eval_sites.clear();
            p->grad_sites.clear();
            for (size_t i = 0; i < src.size(); i++)
                if (p->bsite[i] >= 0) {
                    // record of tcode[j] in the chained stream: one end record per preceding tree, behind the head record
                    const int32_t j = p->tsite[i];
                    const int64_t tree = (std::upper_bound(p->tcode_off.begin(), p->tcode_off.end(), j) - p->tcode_off.begin()) - 1;
                    p->eval_sites.push_back({(int32_t)i, p->bsite[i], j, (int32_t)(j + tree + 1)}); // + the head record
                }
            if (!p->gbsite.empty())
                for (size_t i = 0; i < p->code.size(); i++) {
                    const int32_t gj = p->gbsite[i];
                    if (gj < 0) continue;
                    p->grad_sites.push_back({(int32_t)i, gj, p->gtsite_of_gb.empty() ? -1 : p->gtsite_of_gb[(size_t)gj],
                                             p->rtsite_of_gb.empty() ? -1 : p->rtsite_of_gb[(size_t)gj]});
                }
            p->lists_gen = p->site_gen;
        }
        try {
        parallel_tree_ranges((int64_t)p->eval_sites.size(), [&](int, int64_t sb, int64_t se) {
            for (size_t q = (size_t)sb; q < (size_t)se; q++) {
                const de_program::EvalSite &e = p->eval_sites[q];
                const uint32_t lo = src[(size_t)e.src].imm.u32[0], hi = src[(size_t)e.src].imm.u32[1];
                p->bcode[(size_t)e.b].lo = lo;
                p->bcode[(size_t)e.b].hi = hi;
                p->tcode[(size_t)e.t].lo = lo;
                p->tcode[(size_t)e.t].hi = hi;
                patch_chained_imm(p, e.c, lo, hi);
            }
        }, 4096);
        if (gpatch)
            parallel_tree_ranges((int64_t)p->grad_sites.size(), [&](int, int64_t sb, int64_t se) {
                for (size_t q = (size_t)sb; q < (size_t)se; q++) {
                    const de_program::GradSite &g = p->grad_sites[q];
                    const uint32_t lo = p->code[(size_t)g.src].imm.u32[0], hi = p->code[(size_t)g.src].imm.u32[1];
                    p->gbcode[(size_t)g.gb].lo = lo;
                    p->gbcode[(size_t)g.gb].hi = hi;
                    if (tpatch && g.gt >= 0) {
                        for (int64_t w = 0; w < (p->gt_share ? 4 : 1); w++) { // (every stream variant of a shared-leaf-row program)
                            p->gtcode[(size_t)(g.gt + w * p->gt_stride)].lo = lo;
                            p->gtcode[(size_t)(g.gt + w * p->gt_stride)].hi = hi;
                        }
                    }
                    if (rpatch && g.rt >= 0) {
                        p->rtcode[(size_t)g.rt].lo = lo;
                        p->rtcode[(size_t)g.rt].hi = hi;
                    }
                }
            }, 4096);
        } catch (const std::bad_alloc &) { return fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (gpatch) {
            if (!tpatch) p->gt_valid = false;
            if (!rpatch) p->rt_valid = false;
        } else {
            p->gcode_stale = true;
        }
        const auto t4 = now();
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // the program may be in use by work already queued
        if (!p->ccode.empty())
            HIP_TRY(ctx, hipMemcpy(p->d_code, p->ccode.data(), p->ccode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        HIP_TRY(ctx, upload_wave_variants(p));
        if (gpatch) {
            if (!p->gbcode.empty())
                HIP_TRY(ctx, hipMemcpy(p->d_gcode, p->gbcode.data(), p->gbcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
            if (p->gt_valid && !p->gtcode.empty())
                HIP_TRY(ctx, hipMemcpy(p->d_gtcode, p->gtcode.data(), p->gtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
            if (p->rt_valid && !p->rtcode.empty())
                HIP_TRY(ctx, hipMemcpy(p->d_rtcode, p->rtcode.data(), p->rtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        }
        if (timing)
            fprintf(stderr, "set_consts us: write %ld, refresh_folds %ld, flags %ld, patch %ld, upload %ld\n", us(t0, t1), us(t1, t2), us(t2, t3),
                    us(t3, t4), us(t4, now()));
        return DE_OK;
    }
    p->gcode_stale = true;
    try {
        rebind(p); // same shape: only immediates change
    } catch (const std::bad_alloc &) {
        return fail(ctx, DE_ERR_HIP, "out of host memory");
    }
    {
        int rc = DE_OK;
        try { rc = make_threaded(ctx, p); } catch (const std::bad_alloc &) { rc = fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (rc != DE_OK) return rc;
    }
    // the program may be in use by work already queued on the stream
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!p->bcode.empty())
        HIP_TRY(ctx, hipMemcpy(p->d_code, (p->threaded ? p->ccode : p->bcode).data(),
                               (p->threaded ? p->ccode : p->bcode).size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
    if (p->threaded) HIP_TRY(ctx, upload_wave_variants(p)); // (same shapes: the variants the creation made room for)
    return DE_OK;
}

int de_program_destroy(de_program_t *p) {
    if (!p) return DE_OK;
    (void)hipSetDevice(p->ctx->device);
    dbg_lap(nullptr);
    (void)hipStreamSynchronize(p->ctx->stream);
    de_ctx *c = p->ctx;
    dbg_lap("destroy: stream sync");
    prog_free(c, p->d_code);
    if (!p->eval_arena) {
        if (p->d_code_off) (void)hipFree(p->d_code_off);
        if (p->d_compact_ints) (void)hipFree(p->d_compact_ints);
    }
    if (p->d_cert_code) (void)hipFree(p->d_cert_code);
    if (p->d_cert_off) (void)hipFree(p->d_cert_off);
    dbg_lap("destroy: eval streams");
    if (p->aux) de_program_destroy(p->aux);
    prog_free(c, p->d_kf);
    p->d_kf = nullptr;
    dbg_lap(nullptr);
    prog_free(c, p->d_gcode);
    if (p->d_gcode_off) (void)hipFree(p->d_gcode_off);
    prog_free(c, p->d_gtcode);
    if (p->d_gtcode_off) (void)hipFree(p->d_gtcode_off);
    if (p->d_gt_ids) (void)hipFree(p->d_gt_ids);
    prog_free(c, p->d_rtcode);
    for (void *q : {(void *)p->d_rtcode_off, (void *)p->d_rtcode_mid, (void *)p->d_rt_ids})
        if (q) (void)hipFree(q);
    if (p->d_ok_eval && !p->eval_arena) (void)hipFree(p->d_ok_eval);
    for (void *q : {(void *)p->d_ok_grad, (void *)p->d_ng, (void *)p->d_goff})
        if (q) (void)hipFree(q);
    dbg_lap("destroy: gradient streams, flags");
    p->aux = nullptr;
    park_program(c, p);
    dbg_lap("destroy: host vectors");
    return DE_OK;
}

int64_t de_program_n_trees(const de_program_t *p) { return p ? p->n_trees : -1; }
int64_t de_program_n_nodes(const de_program_t *p) { return p ? p->n_nodes : -1; }
int64_t de_program_n_grad(const de_program_t *p, int64_t tree, int mode) {
    if (!p || tree < 0 || tree >= p->n_trees) return -1;
    const int64_t nc = p->n_consts_tree[(size_t)tree];
    const int64_t nv = (int64_t)p->n_features + p->n_params;
    switch (mode) {
    case DE_GRAD_VARIABLE: return nv;
    case DE_GRAD_CONSTANT: return nc;
    case DE_GRAD_BOTH: return nv + nc;
    default: return -1;
    }
}

int de_program_last_live_trees(de_program_t *p, int64_t *n_live) {
    if (!p || !n_live) return DE_ERR_INVALID_ARG;
    *n_live = -1;
    if (!p->last_compacted || !p->d_compact_ints) return DE_OK;
    de_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int32_t v = -1;
    HIP_TRY(c, hipMemcpy(&v, p->d_compact_ints + (2 * (size_t)p->n_trees + 1), sizeof v, hipMemcpyDeviceToHost));
    *n_live = v;
    return DE_OK;
}

int de_prio_tiles_wanted(int64_t N, int32_t n_features, int64_t n_trees) { return prio_tiles_wanted(N, n_features, n_trees) ? 1 : 0; }

int de_eval_plan(const de_program_t *p, int64_t N, int32_t *plan) {
    if (!p || !plan || N < 0) return DE_ERR_INVALID_ARG;
    eval_plan(p->dtype, p->n_trees, N, &plan[0], &plan[1], &plan[2], p->threaded ? p->waves : 1);
    return DE_OK;
}

// Host-only test hook (no HIP call): n items over the pool of host threads that de_program_create's per-tree passes run on; returns how
// many items were visited exactly once (n when all is well), *n_ranges = the ranges the items were split into (1 = ran inline).
uint64_t de_program_stream_hash(const de_program_t *p) {
    if (!p) return 0;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *data, size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(data);
        for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ull; }
        const uint64_t n = bytes; // (the length too: an empty vector and a missing one differ from a shifted boundary)
        for (int i = 0; i < 8; i++) { h ^= (n >> (8 * i)) & 0xFF; h *= 1099511628211ull; }
    };
    auto vec = [&](const auto &v) { mix(v.data(), v.size() * sizeof(v[0])); };
    vec(p->code); vec(p->code_off); vec(p->fcode); vec(p->fcode_off); vec(p->bcode); vec(p->bcode_off);
    vec(p->fbcode); vec(p->tcode); vec(p->tcode_off); vec(p->ccode); vec(p->ccode_off); vec(p->bsite); vec(p->tsite);
    vec(p->consts); vec(p->const_off); vec(p->const_instr); vec(p->fconst_instr); vec(p->const_checks); vec(p->n_consts_tree);
    vec(p->aux_const_src); vec(p->host_ok_eval); vec(p->host_ok_grad); vec(p->fold_ok);
    vec(p->fold_host); vec(p->fold_noff); vec(p->fold_coff); vec(p->aux_fold); vec(p->aux_csrc); vec(p->kfold); vec(p->kf_csrc);
    mix(p->fold_nodes.data(), p->fold_nodes.size() * sizeof(de_tape_node_t));
    for (const auto &f : p->folds) { const int32_t w[3] = {f.tree, f.instr, f.tested_always ? 1 : 0}; mix(w, sizeof w); }
    const int64_t scal[6] = {p->n_trees, p->n_nodes, p->n_slots, p->uses_params ? 1 : 0, p->folded ? 1 : 0, p->threaded ? 1 : 0};
    mix(scal, sizeof scal);
    if (!p->ccode_w.empty()) mix(p->ccode_w.data(), p->ccode_w.size() * sizeof(BoundInstr)); // the stream variants of a wave group
    if (p->aux) { const uint64_t a = de_program_stream_hash(p->aux); mix(&a, sizeof a); }
    return h;
}

// Program sanitizer (SURVEY.md §5 "sanitizer / bounds-checked debug"): the kernels trust the instruction streams —
// an LDS offset, a spill slot, a handler address are used as they come.  This walks every stream of the program on the
// host and checks each field against the bounds the launch will allocate: generic code (opcodes, operand rows <
// n_features + n_slots, push / pop slots, constant slots), bound and fused code (handler ids, rows, the int8 push
// distance of the superinstructions), and the chained stream the threaded kernel executes (every handler address is an
// entry of the device handler table, LDS byte offsets lie inside the launch's allocation, every tree ends in the end
// record).  DE_VERIFY=1 runs it after every de_program_create / de_program_set_consts.
int de_program_verify(const de_program_t *p) {
    if (!p) return DE_ERR_INVALID_ARG;
    de_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device)); // the handler tables are cached per device: the program's addresses are its OWN device's
    const int64_t rows = eval_rows(p), spill_end = (int64_t)p->n_features + p->n_slots;
    auto bad = [&](const char *what, int64_t tree, int64_t i, uint64_t v) {
        return fail(c, DE_ERR_BAD_TAPE, "program verify: %s (tree %lld, instruction %lld, value 0x%llx)", what, (long long)tree, (long long)i,
                    (unsigned long long)v);
    };
    const std::vector<Instr> *gens[2] = {&p->code, p->folded ? &p->fcode : nullptr};
    const std::vector<int32_t> *goffs[2] = {&p->code_off, p->folded ? &p->fcode_off : nullptr};
    for (int g = 0; g < 2; g++) {
        if (!gens[g]) continue;
        const auto &code = *gens[g];
        const auto &off = *goffs[g];
        if ((int64_t)off.size() != p->n_trees + 1 || off[0] != 0 || off.back() != (int32_t)code.size()) return bad("generic offsets", -1, g, off.size());
        for (int64_t t = 0; t < p->n_trees; t++) {
            if (off[(size_t)t + 1] <= off[(size_t)t]) return bad("empty tree", t, 0, 0);
            for (int32_t i = off[(size_t)t]; i < off[(size_t)t + 1]; i++) {
                const Instr &ins = code[(size_t)i];
                const uint32_t op = ins.hdr & H_OP_MASK, src = (ins.hdr >> H_SRC_SHIFT) & H_SRC_MASK;
                const bool known = op == DOP_LOAD || (op >= DE_U_NEG && op < DE_U_LAST_) || (op >= DE_B_ADD && op < DE_B_LAST_) ||
                                   (op >= DE_T_FMA && op < DE_T_LAST_) || (op >= DOP_RSUB && op <= DOP_RPOW_ABS2);
                if (!known) return bad("unknown opcode", t, i, op);
                if (src == SRC_ROW && (int64_t)(ins.feat & 0xFFFFu) >= rows) return bad("operand row outside X + spill slots", t, i, ins.feat);
                if (src == SRC_PARAM && (int64_t)(ins.feat & 0xFFFFu) >= p->n_params) return bad("parameter row out of range", t, i, ins.feat);
                if (src != SRC_ACC && src != SRC_ROW && src != SRC_CONST && src != SRC_PARAM) return bad("operand kind", t, i, src);
                if ((ins.hdr & H_PUSH) && (int)((ins.hdr >> H_PUSH_SHIFT) & H_SLOT_MASK) >= p->n_slots) return bad("push slot", t, i, ins.hdr);
                if (op >= DE_T_FMA && op < DE_T_LAST_ && (int)((ins.hdr >> H_POPC_SHIFT) & H_SLOT_MASK) >= p->n_slots) return bad("ternary slot", t, i, ins.hdr);
            }
        }
    }
    for (size_t i = 0; i < p->bcode.size(); i++) {
        const BoundInstr &b = p->bcode[i];
        if (b.bop >= BOP_COUNT) return bad("bound handler id", -1, (int64_t)i, b.bop);
        if (!bop_is_const_source(b.bop) && b.bop != BOP_GEN_PARAM && b.bop != BOP_LOAD_CONST && b.bop != BOP_CHECK_ACC && b.bop != BOP_GEN_ACC &&
            b.bop != BOP_INJ_ACC && !(b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END && !((b.bop - BOP_UN_BASE) & 2)) &&
            (int64_t)(b.arg & 0xFFFFFFu) >= rows)
            return bad("bound operand row", -1, (int64_t)i, b.arg);
    }
    if (p->threaded) {
        uint64_t table[TOPX_TABLE];
        if (eval_handler_table(p->dtype, (p->options & DE_OPT_TURBO) != 0, table) != hipSuccess) return fail(c, DE_ERR_HIP, "handler table");
        std::vector<uint64_t> valid(table, table + TOPX_TABLE); // (the end-fused variants included)
        std::sort(valid.begin(), valid.end());
        const uint64_t lds_bytes = (uint64_t)(rows + (p->uses_params ? 2 : 0)) * trow_bytes(p->dtype);
        if ((int64_t)p->ccode_off.size() != p->n_trees + 1 || p->ccode.size() != p->tcode.size() + (size_t)p->n_trees + 1) return bad("chained layout", -1, 0, p->ccode.size());
        const bool f32 = p->dtype == DE_F32;
        for (int64_t t = 0; t < p->n_trees; t++) {
            const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1], h = p->ccode_off[(size_t)t];
            if (h != i0 + (int32_t)t + 1) return bad("chained offset", t, h, (uint64_t)i0);
            if (i1 > i0) { // a tree never starts by reading the accumulator: the end of the previous tree leaves it as it is
                const BoundInstr &f0 = p->fbcode[(size_t)i0];
                const uint32_t aux0 = f0.arg >> 24;
                const int deg0 = aux0 == (uint32_t)DOP_LOAD ? 0 : de_opcode_degree((int)aux0);
                if (top_reads_acc(f0.bop, deg0)) return bad("first instruction of a tree reads the accumulator", t, 0, f0.bop);
            }
            { // the header record (in front of the tree) carries the tree's record count: h_tree_skip steps over the tree with it
                const BoundInstr &hd = p->ccode[(size_t)h - 1];
                if (((f32 ? hd.arg : hd.lo) & ~DE_HDR_FUSED_END) != (uint32_t)(i1 - i0)) return bad("tree header does not carry the tree's length", t, 0, f32 ? hd.arg : hd.lo);
            }
            for (int32_t i = i0; i <= i1; i++) {
                const BoundInstr &r = p->ccode[(size_t)(h + (i - i0))];
                const BoundInstr &q = p->ccode[(size_t)(h + (i - i0) - 1)]; // a record's handler is named by the record in front of it
                const uint64_t addr = f32 ? (((uint64_t)q.hi << 32) | q.lo) : ((table[0] & 0xFFFFFFFF00000000ull) | q.arg);
                if (!std::binary_search(valid.begin(), valid.end(), addr)) return bad("handler address not in the device table", t, i - i0, addr);
                // an end-fused last instruction (h_chain_end) steps over the end record: its own record names what the end record names
                const int ev = i1 - i0 >= 2 ? topx_endv_of(p->fbcode[(size_t)i1 - 1].bop) : -1;
                const uint64_t last_plain = p->handler_base + p->tcode[(size_t)i1 - 1].bop;
                const BoundInstr &lastq = p->ccode[(size_t)(h + (i1 - 1 - i0) - 1)];
                const uint64_t last_addr = f32 ? (((uint64_t)lastq.hi << 32) | lastq.lo) : ((table[0] & 0xFFFFFFFF00000000ull) | lastq.arg);
                const bool fused_end = ev >= 0 && last_addr == p->endv_handler[ev] && last_addr != last_plain;
                if (i == i0 && (((f32 ? p->ccode[(size_t)h - 1].arg : p->ccode[(size_t)h - 1].lo) & DE_HDR_FUSED_END) != 0) != fused_end)
                    return bad("tree header's end-fused bit", t, 0, (uint64_t)fused_end);
                if (i == i1) {
                    const BoundInstr &e = p->ccode[(size_t)(h + (i1 - i0))]; // the end record itself names the next tree's first handler
                    const uint64_t after = f32 ? (((uint64_t)e.hi << 32) | e.lo) : ((table[0] & 0xFFFFFFFF00000000ull) | e.arg);
                    if (fused_end ? addr != after : addr != p->end_handler) return bad("tree does not end in the end record", t, i - i0, addr);
                    if (r.bop != (uint32_t)t) return bad("end record does not name its tree", t, i - i0, r.bop);
                    continue;
                }
                const BoundInstr &fb = p->fbcode[(size_t)i];
                if (i == i1 - 1 ? (addr != last_plain && !fused_end) : addr != p->handler_base + p->tcode[(size_t)i].bop)
                    return bad("record / threaded code mismatch", t, i - i0, addr);
                if (fb.bop >= TOPX_COUNT) return bad("fused handler id", t, i - i0, fb.bop);
                const bool no_row = top_is_const_source(fb.bop) || fb.bop == BOP_CHECK_ACC || fb.bop == BOP_GEN_ACC || fb.bop == BOP_INJ_ACC ||
                                    fb.bop == BOP_GEN_PARAM || (fb.bop >= BOP_UN_BASE && fb.bop < BOP_UN_END && !((fb.bop - BOP_UN_BASE) & 2)) ||
                                    (fb.bop >= TOPX_UN_BASE && fb.bop < TOPX_BIN_BASE && ((fb.bop - TOPX_UN_BASE) & 1)) ||
                                    (fb.bop >= TOPX_BIN_BASE && ((fb.bop - TOPX_BIN_BASE) & 1));
                if (!no_row) {
                    const uint64_t off = r.bop & 0xFFFFFFu;
                    if (off % trow_bytes(p->dtype) != 0 || off + trow_bytes(p->dtype) > lds_bytes) return bad("LDS operand offset outside the launch's allocation", t, i - i0, r.bop);
                    const bool pushes = (fb.bop >= TOP_LOADROW_BASE && fb.bop < TOP_LOADCONST_PUSH && ((fb.bop - TOP_LOADROW_BASE) & 2)) ||
                                        (fb.bop >= TOP_UNROW_BASE && fb.bop < TOP_BINROWC_BASE && ((fb.bop - TOP_UNROW_BASE) & 2)) ||
                                        (fb.bop >= TOP_BIN2_BASE && fb.bop < TOP_COUNT && ((fb.bop - TOP_BIN2_BASE) & 1));
                    if (pushes) {
                        const int64_t prow = (int64_t)(off / trow_bytes(p->dtype)) + (int8_t)(r.bop >> 24);
                        if (prow < p->n_features || prow >= spill_end) return bad("push row of a superinstruction outside the spill slots", t, i - i0, r.bop);
                    }
                    if (fb.bop >= TOP_BIN2_BASE && fb.bop < TOP_COUNT && !(((fb.bop - TOP_BIN2_BASE) >> 2) & 1)) { // row-row: second row by distance
                        const int64_t second = (int64_t)off + (int32_t)(f32 ? r.arg : r.lo);
                        if (second < 0 || second % (int64_t)trow_bytes(p->dtype) != 0 || (uint64_t)second + trow_bytes(p->dtype) > lds_bytes) return bad("second operand row of a two-operand form", t, i - i0, (uint64_t)second);
                    }
                }
                if (fb.bop == BOP_GEN_PARAM && (f32 ? r.arg : r.lo) != (uint32_t)rows * (uint32_t)trow_bytes(p->dtype)) return bad("class-row offset of a parameter operand", t, i - i0, r.arg);
            }
        }
        // wave groups: variant w of the stream = variant 0 but for the slot rows, which lie `shift` rows further on (behind the staged
        // parameter rows and the slots of the waves before): same handler words, every operand row inside the group's allocation, a row
        // of variant 0's slot area moved by exactly the shift and every other row left where it is
        if (p->waves > 1 && p->ccode_w.size() != (p->var_stride ? p->ccode.size() * (size_t)(p->waves - 1) : 0)) return bad("stream variants of the wave group", -1, p->waves, p->ccode_w.size());
        const uint64_t rb = trow_bytes(p->dtype);
        for (size_t w = 1; p->var_stride && w < (size_t)p->waves; w++) {
            const BoundInstr *cw = p->ccode_w.data() + (w - 1) * p->ccode.size();
            const int64_t shift = (p->prows ? p->n_params : 0) + (int64_t)w * p->n_slots, rows_all = rows + (int64_t)(p->waves - 1) * p->n_slots;
            auto moved = [&](int64_t row0, int64_t roww) { return roww == (row0 >= p->n_features && row0 < spill_end ? row0 + shift : row0) && roww < rows_all; };
            for (int64_t t = 0; t < p->n_trees; t++) {
                const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1], h = p->ccode_off[(size_t)t];
                for (int32_t i = i0 - 1; i <= i1; i++) { // the header in front, the instructions, the end record
                    const BoundInstr &r0 = p->ccode[(size_t)(h + (i - i0))], &r = cw[(size_t)(h + (i - i0))];
                    const bool same_next = f32 ? (r.lo == r0.lo && r.hi == r0.hi) : r.arg == r0.arg;
                    if (!same_next) return bad("stream variant names another handler", t, i - i0, w);
                    if (i < i0 || i == i1) { if (std::memcmp(&r, &r0, sizeof r)) return bad("stream variant: header / end record differs", t, i - i0, w); continue; }
                    const BoundInstr &fb = p->fbcode[(size_t)i];
                    if (fb.bop == BOP_GEN_PARAM) { if (std::memcmp(&r, &r0, sizeof r)) return bad("stream variant: a gathered-parameter record differs", t, i - i0, w); continue; }
                    // the operand word: a row offset (a slot row moves by the shift, any other row stays; a record without an LDS operand
                    // carries row 0 or no row offset at all: unchanged) | aux << 24
                    const uint64_t off0 = r0.bop & 0xFFFFFFu, off = r.bop & 0xFFFFFFu;
                    if (off0 % rb != 0 || off % rb != 0) { if (off != off0) return bad("stream variant: operand word", t, i - i0, r.bop); }
                    else if (!moved((int64_t)(off0 / rb), (int64_t)(off / rb))) return bad("stream variant: operand row", t, i - i0, r.bop);
                    const bool pushes = (fb.bop >= TOP_LOADROW_BASE && fb.bop < TOP_LOADCONST_PUSH && ((fb.bop - TOP_LOADROW_BASE) & 2)) ||
                                        (fb.bop >= TOP_UNROW_BASE && fb.bop < TOP_BINROWC_BASE && ((fb.bop - TOP_UNROW_BASE) & 2)) ||
                                        (fb.bop >= TOP_BIN2_BASE && fb.bop < TOP_COUNT && ((fb.bop - TOP_BIN2_BASE) & 1));
                    if (pushes ? !moved((int64_t)(off0 / rb) + (int8_t)(r0.bop >> 24), (int64_t)(off / rb) + (int8_t)(r.bop >> 24)) : (r.bop >> 24) != (r0.bop >> 24))
                        return bad("stream variant: push row / aux byte", t, i - i0, r.bop);
                    const bool row_row = fb.bop >= TOP_BIN2_BASE && fb.bop < TOP_COUNT && !(((fb.bop - TOP_BIN2_BASE) >> 2) & 1);
                    if (row_row || fb.bop == BOP_TERN) { // the immediate = byte distance to the second / third operand row
                        const int64_t s0 = (int64_t)off0 + (int32_t)(f32 ? r0.arg : r0.lo), s1 = (int64_t)off + (int32_t)(f32 ? r.arg : r.lo);
                        if (s1 < 0 || s1 % (int64_t)rb != 0 || !moved(s0 / (int64_t)rb, s1 / (int64_t)rb)) return bad("stream variant: second operand row", t, i - i0, (uint64_t)s1);
                    } else if (f32 ? r.arg != r0.arg : (r.lo != r0.lo || r.hi != r0.hi))
                        return bad("stream variant: immediate differs", t, i - i0, w);
                }
            }
        }
    }
    return DE_OK;
}

int64_t de_program_dump(const de_program_t *p, int64_t tree, uint32_t *words, int64_t cap, int which) {
    if (!p || tree < 0 || tree >= p->n_trees) return -DE_ERR_INVALID_ARG;
    if (which == 1) { // metadata: n_slots, host_ok_eval, host_ok_grad, uses_params
        if (cap < 4) return -DE_ERR_INVALID_ARG;
        words[0] = (uint32_t)p->n_slots;
        words[1] = p->host_ok_eval[(size_t)tree];
        words[2] = p->host_ok_grad[(size_t)tree];
        words[3] = p->uses_params;
        if (cap < 5) return 4;
        words[4] = (uint32_t)(p->threaded ? p->waves : 1); // waves per workgroup of the eval kernel (wave groups: 2 / 4 / 8)
        return 5;
    }
    if (which == 2) { // bound instructions (de_bind.h)
        const int32_t b0 = p->bcode_off[(size_t)tree], b1 = p->bcode_off[(size_t)tree + 1];
        const int64_t nb = (int64_t)(b1 - b0) * 4;
        if (!words || cap < nb) return nb;
        std::memcpy(words, p->bcode.data() + b0, (size_t)nb * 4);
        return nb;
    }
    if (which == 3) { // fused (superinstruction) form of the threaded kernel; empty when that kernel is not in use
        if (!p->threaded) return 0;
        const int32_t b0 = p->tcode_off[(size_t)tree], b1 = p->tcode_off[(size_t)tree + 1];
        const int64_t nb = (int64_t)(b1 - b0) * 4;
        if (!words || cap < nb) return nb;
        std::memcpy(words, p->fbcode.data() + b0, (size_t)nb * 4);
        return nb;
    }
    const int32_t i0 = p->code_off[(size_t)tree], i1 = p->code_off[(size_t)tree + 1];
    const int64_t nw = (int64_t)(i1 - i0) * 4;
    if (!words || cap < nw) return nw;
    std::memcpy(words, p->code.data() + i0, (size_t)nw * 4);
    return nw;
}

int64_t de_lower_tape(int dtype, const de_tape_node_t *nodes, int64_t n_nodes, const void *consts,
                      int64_t n_consts, int32_t n_features, int32_t n_params, uint32_t options, uint32_t *words,
                      int64_t cap, int32_t *meta) {
    if (!nodes || (dtype != DE_F32 && dtype != DE_F64) || (n_consts > 0 && !consts)) return -DE_ERR_INVALID_ARG;
    try {
        LowerOptions lo;
        lo.early_exit = (options & DE_OPT_EARLY_EXIT) != 0;
        lo.fuse1 = (options & DE_OPT_FUSE_DEG1) != 0;
        lo.fuse2 = (options & DE_OPT_FUSE_DEG2) != 0;
        lo.bumper = (options & DE_OPT_BUMPER_CHECKS) != 0;
        lo.n_features = n_features;
        lo.n_params = n_params;
        lo.dtype = dtype;
        for (int64_t i = 0; i < n_nodes; i++) // a CSE tape (GraphNode sharing) announces itself by its markers
            if ((nodes[i].degree == 1 && nodes[i].op == DE_OP_SHARE) || (nodes[i].degree == 0 && nodes[i].op == DE_LEAF_SHARED)) lo.cse = true;
        TreeProgram tp;
        std::string why;
        int rc = lower_tree(nodes, n_nodes, n_consts, lo, &tp, &why);
        if (rc != DE_OK) return -rc;
        bool ok_eval = true, ok_grad = true;
        for (int64_t k = 0; k < n_consts; k++) {
            const double v = dtype == DE_F32 ? (double)static_cast<const float *>(consts)[k]
                                             : static_cast<const double *>(consts)[k];
            if (tp.const_instr[(size_t)k] >= 0) write_imm(tp.code[(size_t)tp.const_instr[(size_t)k]], dtype, v);
            const bool fin = finite_in(dtype, v);
            ok_grad = ok_grad && fin;
            const uint8_t ch = tp.const_checks[(size_t)k];
            if (!fin && ((ch & CONST_CHECK_ALWAYS) || (lo.early_exit && (ch & CONST_CHECK_EE)))) ok_eval = false;
        }
        if (meta) {
            meta[0] = tp.n_slots;
            meta[1] = ok_eval;
            meta[2] = ok_grad;
            meta[3] = tp.uses_params;
        }
        const int64_t nw = (int64_t)tp.code.size() * 4;
        if (!words || cap < nw) return nw;
        std::memcpy(words, tp.code.data(), (size_t)nw * 4);
        return nw;
    } catch (...) {
        return -DE_ERR_HIP;
    }
}

int64_t de_lower_tape_stage(int dtype, const de_tape_node_t *nodes, int64_t n_nodes, const void *consts,
                            int64_t n_consts, int32_t n_features, int32_t n_params, uint32_t options, int stage,
                            uint32_t *words, int64_t cap) {
    if (stage != 2 && stage != 3) return -DE_ERR_INVALID_ARG;
    std::vector<uint32_t> g;
    int64_t nw = de_lower_tape(dtype, nodes, n_nodes, consts, n_consts, n_features, n_params, options, nullptr, 0, nullptr);
    if (nw < 0) return nw;
    try {
        g.resize((size_t)nw);
        nw = de_lower_tape(dtype, nodes, n_nodes, consts, n_consts, n_features, n_params, options, g.data(), nw, nullptr);
        if (nw < 0) return nw;
        std::vector<BoundInstr> b, f;
        bind_tree(reinterpret_cast<const Instr *>(g.data()), (size_t)nw / 4, (options & DE_OPT_EARLY_EXIT) != 0, n_features, &b);
        if (stage == 3) fuse_tree(b.data(), b.size(), &f);
        const std::vector<BoundInstr> &o = stage == 3 ? f : b;
        const int64_t n = (int64_t)o.size() * 4;
        if (!words || cap < n) return n;
        std::memcpy(words, o.data(), (size_t)n * 4);
        return n;
    } catch (...) {
        return -DE_ERR_HIP;
    }
}

// ---------------------------------------------------------------------------
// Stage a caller buffer: device pointers are used in place; host pointers are
// copied into context scratch (and copied back by the caller of this helper).
} // extern "C"
