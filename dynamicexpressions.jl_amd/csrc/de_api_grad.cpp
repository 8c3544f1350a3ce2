// de_api_grad.cpp — C ABI (include/de_hip.h): de_eval_grad / de_eval_diff / de_eval_pullback_dX, de_eval_loss_grad, de_eval_loss_grad_by_class: the
// generic gradient program, its direct-threaded and reverse-accumulation forms, launch planning of the bucketed gradient kernels.
#include "de_api_internal.h"

extern "C" {
static int ensure_generic_code(de_ctx *c, de_program *p) {
    if (p->gcode_stale || !p->d_gcode) {
        // gradients flow through constant subtrees, so this is the UNFOLDED program; every value the
        // reference tests is tested (ee binding) whatever the eval options were
        p->gt_valid = false;
        p->rt_valid = false;
        // bound per worker into a vector of its own, then concatenated (10^4 trees: 3 ms on one thread)
        build_stream_by_trees<BoundInstr>(p->n_trees, &p->gbcode, &p->gbcode_off, [&](int64_t t, std::vector<BoundInstr> *out) {
            const int32_t i0 = p->code_off[(size_t)t], i1 = p->code_off[(size_t)t + 1];
            bind_tree(p->code.data() + i0, (size_t)(i1 - i0), true, p->n_features, out);
        });
        match_const_sites(p->code, p->code_off, p->gbcode, p->gbcode_off, p->n_trees,
                          [](const BoundInstr &b) { return bop_is_const_source(b.bop); }, &p->gbsite);
        p->gtsite_of_gb.clear();
        p->site_gen++;
    }
    if (!p->d_gcode) {
        HIP_TRY(c, prog_malloc(c, reinterpret_cast<void **>(&p->d_gcode), (p->gbcode.size() + 1) * sizeof(BoundInstr)));
        HIP_TRY(c, hipMemset(p->d_gcode, 0, (p->gbcode.size() + 1) * sizeof(BoundInstr)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_gcode_off), p->gbcode_off.size() * sizeof(int32_t)));
        HIP_TRY(c, hipMemcpy(p->d_gcode_off, p->gbcode_off.data(), p->gbcode_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        p->gcode_stale = true;
    }
    if (p->gcode_stale) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!p->gbcode.empty())
            HIP_TRY(c, hipMemcpy(p->d_gcode, p->gbcode.data(), p->gbcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        p->gcode_stale = false;
    }
    return DE_OK;
}

// Threaded form of the gradient program (de_grad_threaded.hip) for `mode`.  Trees are grouped into
// buckets by gradient width n_grad(t, mode): widths 1..6 and 7-8 each run in the module built for that
// window (every seed is known here and compiled into the handler choice), wider trees in windows of 8
// with run-time seeds.  Fills g->threaded_code & co. when the program can be expressed this way; otherwise
// leaves them null and the flat-switch kernel runs.  Call after ensure_generic_code().
static int ensure_grad_threaded(de_ctx *c, de_program *p, int mode, const std::vector<int32_t> &ng, int64_t N, GradArgs *g) {
    g->threaded_code = nullptr;
    g->n_buckets = 0;
    g->gt_share = false;
    g->gt_var_stride = 0;
    const char *env = getenv("DE_GRAD_THREADED");
    if (env && *env == '0') return DE_OK;
    const int F = p->n_features, P = p->n_params;
    // Parameter leaves are LDS rows of their own: the kernel gathers params[:, class] into P rows behind the X rows when it
    // stages a tile (the reference's formulation, src/ParametricExpression.jl:381-389), so every hot handler serves them.
    const int FE = F + (p->uses_params ? P : 0);
    const bool hot_const_unary = !getenv("DE_NO_CONST_UNARY_HOT"); // unary operators outside the binder's hot set through hot handlers
    const bool fuse_push = true;                                   // PUSH + LOAD pairs as one instruction
    auto gun_of = [&](uint32_t op) { // hot unary index of a de_opcode (de_bind.h), or -1
        return hot_const_unary ? gun_index((int)op, DE_U_COS, DE_U_EXP, DE_U_SIN, DE_U_NEG, DE_U_SQUARE, DE_U_CUBE, DE_U_ABS, DE_U_LOG, DE_U_SAFE_LOG,
                                           DE_U_SQRT, DE_U_SAFE_SQRT, DE_U_TANH, DE_U_RELU) : -1;
    };
    // Two samples per lane double the buckets (launches) and the tile: they pay from ~10^5 samples on (10^4 trees x
    // 10^3 rows: 0.55 ms with them, 0.35 ms without; 10^3 trees x 10^6 rows: 12.1 against 13.4 ms)
    const char *envn = getenv("DE_GRAD_VS2_MIN_N");
    const bool wide = N >= (envn ? atoll(envn) : 65536);
    if (!(p->gt_valid && p->gt_mode == mode && p->gt_wide == wide)) {
        dbg_lap(nullptr);
        // bucket of a tree: (width index 0..6 = single window of width 1,2,3,4,5,6,8; 7 = several windows of 8)
        // x (samples per lane - 1).  The two-sample modules exist for Float32 windows <= 6; their rows are
        // twice as long, so they only pay while a workgroup's LDS stays small: at most DE_GRAD_VS2_ROWS (15) rows per wave.
        // width index 0..6 = single window of width 1,2,3,4,5,6,8; 7,8,9 = several windows of 8,5,6 (the
        // narrowest module that covers the gradient in ceil(G/8) windows: 9-10 rows -> 2x5, 11-12 -> 2x6, 17-18 -> 3x6)
        static const int WIDTH[10] = {1, 2, 3, 4, 5, 6, 8, 8, 5, 6};
        constexpr int NW = 10, NB = 2 * NW;
        const char *env2 = getenv("DE_GRAD_VS2_ROWS"); // most LDS rows per wave (X + parameters + slots) that still run two samples per lane
        // 15 rows x 512 B x 4 waves = 30.7 KB: 5 workgroups per CU.  Parametric populations (their P parameter rows come on top of X) take 18:
        // measured in round 6 (tools/experiments/sweep_vs2_rows.sh, same box): C5 7.98 / 7.82 / 7.81 / 8.06 / 8.42 ms and C5Ng 7.30 / 7.06 / 7.14 /
        // 7.40 / 7.66 ms at 15 / 18 / 20 / 22 / 24 rows — two samples per lane pay a little further out when most rows are shared inputs
        const int vs2_rows = env2 ? atoi(env2) : (p->uses_params ? 18 : 15);
        // SHARED LEAF ROWS (GradArgs::gt_share): with many leaf rows most of a wave's LDS is a copy of inputs the other three waves could
        // read as well — the four waves then take different trees on the same samples.  Measured (round 6, constant-mode Jacobians of 1000
        // trees x 10^5 samples, tools/experiments/wide_x_grad.py): F = 5 0.80 -> 0.86 ms (worse: the staging is spread over fewer samples),
        // F = 20 1.14 -> 1.03, F = 40 1.82 -> 1.05, F = 60 2.58 -> 1.18, F = 120 5.82 -> 1.91; 5 features + 8 parameter rows (C5): 7.42 -> 7.35 ms,
        // nothing — those kernels are not short of resident waves.  From 16 leaf rows on; DE_GRAD_SHARE = 0 | 1 overrides.
        const char *envs = getenv("DE_GRAD_SHARE"), *envf = getenv("DE_GRAD_SHARE_MIN_ROWS");
        const bool share = envs ? *envs == '1' : FE >= (envf ? atoi(envf) : 16);
        std::vector<int32_t> tslots((size_t)p->n_trees, 0); // spill slots of each tree (rows >= F the code names)
        parallel_for_trees(p->n_trees, [&](int64_t t) {
            int32_t need = 0;
            for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1]; i++) {
                const BoundInstr &b = p->gbcode[(size_t)i];
                const uint32_t row = b.arg & 0xFFFFFFu;
                const bool names_row = b.bop == BOP_PUSH || b.bop == BOP_LOAD_ROW || b.bop == BOP_GEN_ROW || b.bop == BOP_TERN ||
                                       (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END && !((b.bop - BOP_BIN_BASE) & 2)) ||
                                       (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END && ((b.bop - BOP_UN_BASE) & 2));
                if (names_row && row >= (uint32_t)F) need = std::max(need, (int32_t)(row - (uint32_t)F) + 1);
                if (b.bop == BOP_TERN && b.lo >= (uint32_t)F) need = std::max(need, (int32_t)(b.lo - (uint32_t)F) + 1);
            }
            tslots[(size_t)t] = need;
        });
        dbg_lap("grad threaded: spill slots per tree");
        auto bucket_of = [&](int64_t t) {
            const int32_t G = ng[(size_t)t];
            int w;
            if (G <= 6) w = G < 1 ? 0 : G - 1;
            else if (G <= 8) w = 6;
            else {
                const int windows = (G + 7) / 8, per = (G + windows - 1) / windows;
                w = per <= 5 ? 8 : (per <= 6 ? 9 : 7);
                if (!grad_threaded_has(p->dtype, WIDTH[w], 1)) w = 7;
            }
            const int rows2 = FE + std::max(tslots[(size_t)t] * (1 + WIDTH[w]), WIDTH[w]);
            const bool two = wide && p->dtype == DE_F32 && WIDTH[w] <= 6 && rows2 <= vs2_rows && grad_threaded_has(p->dtype, WIDTH[w], 2);
            return w + (two ? NW : 0);
        };
        int32_t count[NB] = {0}, maxg[NB] = {0}, slots[NB] = {0};
        for (int64_t t = 0; t < p->n_trees; t++) {
            const int32_t G = ng[(size_t)t];
            if (G > 240) return DE_OK; // gradient rows travel in 8 bits
            const int b = bucket_of(t);
            // Float64 states wider than 16 dwords are passed through scratch memory by the calling convention
            if (!grad_threaded_has(p->dtype, WIDTH[b % NW], 1 + b / NW)) return DE_OK;
            count[b]++;
            maxg[b] = std::max(maxg[b], G);
            slots[b] = std::max(slots[b], tslots[(size_t)t]);
        }
        const uint32_t es32 = p->dtype == DE_F32 ? 4u : 8u;
        std::vector<std::array<uint64_t, GOP_MAX>> tables(NB);
        uint64_t bases[NB] = {0};
        for (int b = 0; b < NB; b++) {
            if (!count[b]) continue;
            const int GC = WIDTH[b % NW], VS = 1 + b / NW;
            const uint64_t RBb = 64ull * VS * es32; // one wave's row
            const uint64_t rows = (uint64_t)FE + std::max<uint64_t>((uint64_t)slots[b] * (1 + GC), (uint64_t)GC);
            const uint64_t srows = std::max<uint64_t>((uint64_t)slots[b] * (1 + GC), (uint64_t)GC);
            if ((share ? ((uint64_t)FE + 4 * srows) : 4 * rows) * RBb > 160 * 1024) return DE_OK; // four waves' rows must fit the CU's LDS
            hipError_t st = grad_handler_table(p->dtype, GC, VS, tables[b].data());
            if (st != hipSuccess) return fail(c, DE_ERR_HIP, "gradient handler table: %s", hipGetErrorString(st));
            uint64_t base = tables[b][0];
            for (int i = 0; i < (int)gop_count(GC); i++) base = std::min<uint64_t>(base, tables[b][i]);
            for (int i = 0; i < (int)gop_count(GC); i++)
                if (tables[b][i] - base > 0xFFFFFFFFull) return DE_OK;
            bases[b] = base;
        }
        dbg_lap("grad threaded: buckets, handler tables");
        auto leaf_seed = [&](uint32_t f) -> uint32_t { return mode != DE_GRAD_CONSTANT ? (uint32_t)P + f : 0xFFu; };
        auto const_seed = [&](uint32_t ord) -> uint32_t {
            return mode == DE_GRAD_CONSTANT ? ord : (mode == DE_GRAD_BOTH ? (uint32_t)(P + F) + ord : 0xFFu);
        };
        p->gtcode.clear();
        p->gtcode_off.assign((size_t)p->n_trees + 1, 0);
        p->gtsite_of_gb.assign(p->gbcode.size(), -1);
        p->site_gen++;
        std::atomic<bool> ok{true};
        // encoded per worker into a vector of its own (sites = positions in that vector), concatenated afterwards
        std::vector<BoundInstr> parts[HOST_RANGES_MAX];
        // ... and per record what a stream variant of the shared-leaf-row launch adds the wave's slot bytes to: 0 nothing, 1 the operand
        // word (a slot operand, a push, the spilled operands of a ternary operator), 2 the immediate (push + load of a leaf: row -> slot distance)
        std::vector<uint8_t> kparts[HOST_RANGES_MAX];
        std::vector<int32_t> tree_cnt((size_t)p->n_trees, 0);
        int64_t part_first[HOST_RANGES_MAX], part_last[HOST_RANGES_MAX];
        for (int k = 0; k < HOST_RANGES_MAX; k++) part_first[k] = part_last[k] = 0;
        parallel_tree_ranges(p->n_trees, [&](int wk, int64_t tb, int64_t te) {
        std::vector<BoundInstr> &out = parts[wk];
        std::vector<uint8_t> &kout = kparts[wk];
        part_first[wk] = tb;
        part_last[wk] = te;
        for (int64_t t = tb; t < te && ok; t++) {
            const size_t out_before = out.size();
            const int bkt = bucket_of(t);
            const int GC = WIDTH[bkt % NW];
            const uint32_t RB = 64u * (uint32_t)(1 + bkt / NW) * es32; // bytes of one wave's row
            const bool one_window = bkt % NW < 7; // then g0 = 0 and every seed is known here
            const uint64_t *table = tables[bkt].data();
            const uint64_t base = bases[bkt];
            auto slot_off = [&](uint32_t row) { return (uint32_t)((FE + (row - (uint32_t)F) * (1 + GC)) * RB); };
            // seed variant of a handler (de_bind.h): 0 run-time, 1 none, 2 + k
            auto seed_variant = [&](uint32_t sd) -> int { return !one_window ? 0 : (sd == 0xFFu ? 1 : (sd < (uint32_t)GC ? 2 + (int)sd : 0)); };
            for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1] && ok; i++) {
                const BoundInstr &b = p->gbcode[(size_t)i];
                const uint32_t row = b.arg & 0xFFFFFFu, aux = b.arg >> 24;
                BoundInstr o = b;
                int src = GSRC_ACC, sv = 0;
                auto row_operand = [&](bool rt = false) { // sets src, sv and o.arg for a row operand (rt: handler reads the seed at run time)
                    if (row < (uint32_t)F) {
                        const uint32_t sd = leaf_seed(row);
                        if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                        src = GSRC_LEAF;
                        sv = rt ? 0 : seed_variant(sd);
                        o.arg = (row * RB) | (sv == 0 ? sd << 24 : 0u); // known seeds are compiled into the handler
                    } else {
                        src = GSRC_SLOT;
                        o.arg = slot_off(row);
                    }
                };
                auto const_operand = [&](uint32_t ord, uint32_t low, bool rt = false) {
                    const uint32_t sd = const_seed(ord);
                    if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                    src = GSRC_CONST;
                    sv = rt ? 0 : seed_variant(sd);
                    o.arg = low | (sv == 0 ? sd << 24 : 0u);
                };
                auto param_operand = [&](uint32_t prm, bool rt = false) { // parameter row prm = LDS row F + prm, seed = its gradient row
                    const uint32_t sd = mode != DE_GRAD_CONSTANT ? prm : 0xFFu;
                    if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                    src = GSRC_LEAF;
                    sv = rt ? 0 : seed_variant(sd);
                    o.arg = (((uint32_t)F + prm) * RB) | (sv == 0 ? sd << 24 : 0u);
                };
                uint32_t gop = 0;
                if (b.bop == BOP_CHECK_ROW) continue; // leaf operands are tested where they are read
                if (b.bop == BOP_LOAD_ROW) { row_operand(); gop = gop_load(GC, src, sv); }
                else if (b.bop == BOP_LOAD_CONST) { const_operand(b.arg & 0xFFFFu, 0); gop = gop_load(GC, src, sv); }
                else if (b.bop == BOP_PUSH && fuse_push && i + 1 < p->gbcode_off[(size_t)t + 1] &&
                         (p->gbcode[(size_t)i + 1].bop == BOP_LOAD_CONST ||
                          (p->gbcode[(size_t)i + 1].bop == BOP_LOAD_ROW && (p->gbcode[(size_t)i + 1].arg & 0xFFFFFFu) < (uint32_t)F))) {
                    // PUSH followed by the LOAD that starts the next subtree: one dispatch (g_pushload)
                    const BoundInstr &b2 = p->gbcode[(size_t)i + 1];
                    const uint32_t slot = slot_off(row);
                    if (b2.bop == BOP_LOAD_CONST) {
                        const_operand(b2.arg & 0xFFFFu, slot);
                        o.lo = b2.lo;
                        o.hi = b2.hi;
                        p->gtsite_of_gb[(size_t)i + 1] = (int32_t)out.size(); // the constant lives in the fused instruction
                    } else {
                        const uint32_t row2 = b2.arg & 0xFFFFFFu, sd = leaf_seed(row2);
                        if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                        src = GSRC_LEAF;
                        sv = seed_variant(sd);
                        o.arg = (row2 * RB) | (sv == 0 ? sd << 24 : 0u);
                        o.lo = slot - row2 * RB; // byte distance row -> slot
                        o.hi = 0;
                    }
                    if (!ok) break;
                    o.bop = (uint32_t)(table[gop_pushload(GC, src, sv)] - base);
                    out.push_back(o);
                    kout.push_back(b2.bop == BOP_LOAD_CONST ? 1 : 2);
                    i++; // the LOAD is part of this instruction
                    continue;
                }
                else if (b.bop == BOP_PUSH) { gop = gop_push(GC); o.arg = slot_off(row); }
                else if (b.bop == BOP_CHECK_ACC) { gop = gop_check_acc(GC); o.arg = 0; }
                else if (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END) {
                    const uint32_t v = b.bop - BOP_BIN_BASE;
                    if (v & 2) const_operand(b.arg & 0xFFFFu, 0);
                    else row_operand();
                    gop = gop_bin(GC, (int)(v >> 2), src, sv, (v & 1) != 0);
                } else if (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END) {
                    const uint32_t v = b.bop - BOP_UN_BASE;
                    if (v & 2) row_operand();
                    else o.arg = 0;
                    gop = gop_un(GC, (int)(v >> 2), src, sv, (v & 1) != 0);
                } else if (b.bop == BOP_GEN_ROW && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    row_operand(); gop = gop_bin(GC, aux == (uint32_t)DE_B_MAX ? 6 : 7, src, sv, false); o.lo = o.hi = 0;
                } else if (b.bop == BOP_GEN_ROW && gun_of(aux) >= 0) { row_operand(); gop = gop_un(GC, gun_of(aux), src, sv, false); o.lo = o.hi = 0; }
                else if (b.bop == BOP_GEN_ROW) { row_operand(true); gop = gop_gen(GC, src); o.lo = aux; o.hi = 0; }
                else if (b.bop == BOP_GEN_CONST && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    const_operand(b.arg & 0xFFFFu, 0); gop = gop_bin(GC, aux == (uint32_t)DE_B_MAX ? 6 : 7, src, sv, false);
                }
                else if (b.bop == BOP_GEN_CONST && gun_of(aux) >= 0) {
                    // cos / exp / sin of a constant leaf (common: half the leaves are constants and the gradient program
                    // is not folded): load the constant, then the hot unary handler on the accumulator — not the generic
                    // handler (out-of-line operator switch, OCML functions, scratch traffic of its spills)
                    const_operand(b.arg & 0xFFFFu, 0);
                    o.bop = (uint32_t)(table[gop_load(GC, src, sv)] - base);
                    p->gtsite_of_gb[(size_t)i] = (int32_t)out.size();
                    out.push_back(o);
                    kout.push_back(0);
                    BoundInstr u = b;
                    u.arg = 0;
                    u.lo = u.hi = 0;
                    u.bop = (uint32_t)(table[gop_un(GC, gun_of(aux), GSRC_ACC, 0, false)] - base);
                    out.push_back(u);
                    kout.push_back(0);
                    continue;
                }
                else if (b.bop == BOP_GEN_CONST) { const_operand(b.arg & 0xFFFFu, aux << 16, true); gop = gop_gen(GC, GSRC_CONST); }
                else if (b.bop == BOP_GEN_ACC && gun_of(aux) >= 0) { gop = gop_un(GC, gun_of(aux), GSRC_ACC, 0, false); o.arg = 0; o.lo = o.hi = 0; }
                else if (b.bop == BOP_GEN_ACC) { gop = gop_gen(GC, GSRC_ACC); o.arg = 0; o.lo = aux; o.hi = 0; }
                else if (b.bop == BOP_GEN_PARAM) { // operand = parameter row (b.arg & 0xFFFF), operator aux: the leaf-operand handlers
                    const uint32_t prm = b.arg & 0xFFFFu;
                    int k = -1, ku = -1;
                    switch (aux) {
                    case DE_B_ADD: k = 0; break;
                    case DE_B_SUB: k = 1; break;
                    case DOP_RSUB: k = 2; break;
                    case DE_B_MUL: k = 3; break;
                    case DE_B_DIV: k = 4; break;
                    case DOP_RDIV: k = 5; break;
                    case DE_B_MAX: k = hot_const_unary ? 6 : -1; break;
                    case DE_B_MIN: k = hot_const_unary ? 7 : -1; break;
                    default: ku = gun_of(aux); break;
                    }
                    o.lo = o.hi = 0;
                    if (aux == (uint32_t)DOP_LOAD) { param_operand(prm); gop = gop_load(GC, src, sv); }
                    else if (k >= 0) { param_operand(prm); gop = gop_bin(GC, k, src, sv, false); }
                    else if (ku >= 0) { param_operand(prm); gop = gop_un(GC, ku, src, sv, false); }
                    else { param_operand(prm, true); gop = gop_gen(GC, GSRC_LEAF); o.lo = aux; }
                }
                else if (b.bop == BOP_TERN) {
                    if (row < (uint32_t)F || b.lo < (uint32_t)F) ok = false; // both operands are spilled duals
                    else { gop = gop_tern(GC); o.arg = slot_off(row) | (aux << 24); o.lo = slot_off(b.lo) - slot_off(row); o.hi = 0; }
                } else ok = false; // INJ_*: only bound with early_exit=false, never for gradients
                if (!ok) break;
                o.bop = (uint32_t)(table[gop] - base);
                p->gtsite_of_gb[(size_t)i] = (int32_t)out.size();
                out.push_back(o);
                kout.push_back((src == GSRC_SLOT || b.bop == BOP_PUSH || b.bop == BOP_TERN) ? 1 : 0);
            }
            // the end record: every tree's chain finishes in g_end (the table slot of round 1's parameter handler)
            out.push_back(BoundInstr{(uint32_t)(table[gop_param(GC)] - base), 0u, 0u, 0u});
            kout.push_back(0);
            tree_cnt[(size_t)t] = (int32_t)(out.size() - out_before);
        }
        });
        dbg_lap("grad threaded: encode (host threads)");
        if (!ok) { p->gtsite_of_gb.clear(); p->site_gen++; return DE_OK; }
        for (int64_t t = 0; t < p->n_trees; t++) p->gtcode_off[(size_t)t + 1] = p->gtcode_off[(size_t)t] + tree_cnt[(size_t)t];
        p->gtcode.resize((size_t)p->gtcode_off[(size_t)p->n_trees]);
        // ranges are in tree order; sites move from worker-local to global positions (the same partition as the encoding pass: worker k
        // copies the piece it encoded)
        parallel_tree_ranges(p->n_trees, [&](int k, int64_t tb, int64_t te) {
            if (part_first[k] != tb || part_last[k] != te) return; // (never: the partition depends on n_trees only)
            const int32_t base_k = p->gtcode_off[(size_t)tb];
            if (!parts[k].empty()) std::memcpy(static_cast<void *>(p->gtcode.data() + base_k), parts[k].data(), parts[k].size() * sizeof(BoundInstr));
            if (base_k != 0)
                for (int32_t i = p->gbcode_off[(size_t)tb]; i < p->gbcode_off[(size_t)te]; i++)
                    if (p->gtsite_of_gb[(size_t)i] >= 0) p->gtsite_of_gb[(size_t)i] += base_k;
        });
        {
            size_t copied = 0;
            for (int k = 0; k < HOST_RANGES_MAX; k++) copied += parts[k].size();
            if (copied != p->gtcode.size()) { p->gtsite_of_gb.clear(); p->site_gen++; return fail(c, DE_ERR_HIP, "gradient program: the host threads' partitions disagree"); }
        }
        dbg_lap("grad threaded: concatenate + sites");
        // the handler word of a record names the handler of the record BEHIND it, the end record names the tree's first handler
        // (de_grad_threaded.hip: a handler knows its successor at entry and jumps without waiting for the record it loads)
        parallel_for_trees(p->n_trees, [&](int64_t t) {
            const int32_t a0 = p->gtcode_off[(size_t)t], b0 = p->gtcode_off[(size_t)t + 1];
            if (b0 - a0 < 2) return;
            const uint32_t first = p->gtcode[(size_t)a0].bop;
            for (int32_t i = a0; i < b0 - 1; i++) p->gtcode[(size_t)i].bop = p->gtcode[(size_t)i + 1].bop;
            p->gtcode[(size_t)b0 - 1].bop = first;
        });
        dbg_lap("grad threaded: successor words");
        p->gt_share = false;
        p->gt_stride = 0;
        if (share) { // the stream variants of waves 1 .. 3: the same records with the wave's slot bytes added where a record names a slot
            const size_t n0 = p->gtcode.size();
            std::vector<uint8_t> kinds(n0, 0);
            parallel_tree_ranges(p->n_trees, [&](int k, int64_t tb, int64_t) {
                if (!kparts[k].empty()) std::memcpy(kinds.data() + p->gtcode_off[(size_t)tb], kparts[k].data(), kparts[k].size());
            });
            p->gtcode.resize(4 * n0);
            parallel_for_trees(p->n_trees, [&](int64_t t) {
                const int bkt = bucket_of(t), GC = WIDTH[bkt % NW];
                const uint32_t RB = 64u * (uint32_t)(1 + bkt / NW) * es32;
                const uint32_t sbytes = (uint32_t)std::max<int64_t>((int64_t)slots[bkt] * (1 + GC), (int64_t)GC) * RB; // one wave's slot area
                for (int32_t i = p->gtcode_off[(size_t)t]; i < p->gtcode_off[(size_t)t + 1]; i++)
                    for (uint32_t w = 1; w < 4; w++) {
                        BoundInstr r = p->gtcode[(size_t)i];
                        if (kinds[(size_t)i] == 1) r.arg += w * sbytes; // (the low 24 bits: an LDS offset < 2^18)
                        else if (kinds[(size_t)i] == 2) r.lo += w * sbytes;
                        p->gtcode[(size_t)w * n0 + (size_t)i] = r;
                    }
            });
            p->gt_share = true;
            p->gt_stride = (int64_t)n0;
            dbg_lap("grad threaded: stream variants (shared leaf rows)");
        }
        std::vector<int32_t> ids((size_t)p->n_trees);
        int32_t start[NB], run = 0;
        for (int b = 0; b < NB; b++) { start[b] = run; run += count[b]; }
        {
            int32_t fill[NB];
            for (int b = 0; b < NB; b++) fill[b] = start[b];
            for (int64_t t = 0; t < p->n_trees; t++) ids[(size_t)fill[bucket_of(t)]++] = (int32_t)t;
        }
        // a unary operator on a constant leaf becomes two instructions: at most twice the bound program
        // ... plus one end record per tree and one of padding (every handler reads the record behind its own)
        const size_t gt_cap = (2 * p->gbcode.size() + (size_t)p->n_trees + 1) * (share ? 4 : 1);
        if (p->d_gtcode && p->gt_cap < gt_cap) { // (DE_GRAD_SHARE switched between two encodings of one program: tests)
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            prog_free(c, p->d_gtcode);
            p->d_gtcode = nullptr;
            if (p->d_gtcode_off) { (void)hipFree(p->d_gtcode_off); p->d_gtcode_off = nullptr; }
            if (p->d_gt_ids) { (void)hipFree(p->d_gt_ids); p->d_gt_ids = nullptr; }
        }
        if (!p->d_gtcode) {
            p->gt_cap = gt_cap;
            // (inside one 4 GiB window: the handlers bump the record pointer without a carry; a straddling allocation is set aside and redone)
            {
                const hipError_t ast = prog_malloc(c, reinterpret_cast<void **>(&p->d_gtcode), gt_cap * sizeof(BoundInstr));
                if (ast != hipSuccess) return fail(c, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(ast));
                if (!in_one_window(p->d_gtcode, gt_cap * sizeof(BoundInstr))) return fail(c, DE_ERR_HIP, "gradient instruction stream straddles a 4 GiB boundary");
            }
            HIP_TRY(c, hipMemset(p->d_gtcode, 0, gt_cap * sizeof(BoundInstr)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_gtcode_off), p->gtcode_off.size() * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_gt_ids), std::max<size_t>(ids.size(), 1) * sizeof(int32_t)));
        }
        dbg_lap("grad threaded: ids, hipMalloc, memset");
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // the previous form may be in use by queued work
        if (!p->gtcode.empty())
            HIP_TRY(c, hipMemcpy(p->d_gtcode, p->gtcode.data(), p->gtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(p->d_gtcode_off, p->gtcode_off.data(), p->gtcode_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!ids.empty()) HIP_TRY(c, hipMemcpy(p->d_gt_ids, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        p->gt_n_buckets = 0;
        for (int b = 0; b < NB; b++) {
            if (!count[b]) continue;
            GradArgs::Bucket &bk = p->gt_buckets[p->gt_n_buckets++];
            bk.GC = WIDTH[b % NW];
            bk.VS = 1 + b / NW;
            bk.windows = b % NW >= 7 ? (maxg[b] + bk.GC - 1) / bk.GC : 1;
            bk.max_grad = maxg[b];
            bk.n_slots = slots[b];
            bk.ids = p->d_gt_ids + start[b];
            bk.n = count[b];
            bk.handler_base = bases[b];
            bk.param_handler_off = (uint32_t)(tables[b][gop_param(WIDTH[b % NW])] - bases[b]);
        }
        dbg_lap("grad threaded: upload");
        p->gt_mode = mode;
        p->gt_wide = wide;
        p->gt_valid = true;
    }
    g->threaded_code = p->d_gtcode;
    g->gt_share = p->gt_share;
    g->gt_var_stride = p->gt_stride;
    g->e.code_off = p->d_gtcode_off;
    g->n_buckets = p->gt_n_buckets;
    for (int b = 0; b < p->gt_n_buckets; b++) g->buckets[b] = p->gt_buckets[b];
    return DE_OK;
}

// Reverse-accumulation form of the gradient program (de_rev_threaded.hip) for `mode`: per tree the forward
// instructions (every operator also stores its partials in LDS rows of its own), then the backward instructions
// in execution order.  Fills g->rev_* when the program can be expressed this way (otherwise leaves rev_code
// null and the forward-dual kernels run).  Call after ensure_generic_code().
static int ensure_rev_threaded(de_ctx *c, de_program *p, int mode, GradArgs *g) {
    g->rev_code = nullptr;
    // Reverse accumulation costs two sweeps whatever the number of gradient rows; forward duals cost one sweep
    // of (1 + rows) values (and one sweep per window of 8 rows).  Measured break-even on MI355X: ~8 rows per tree
    // (20-node trees: 3.5 rows 9.6 ms forward / 17.2 ms reverse; 17 rows 44.6 ms / 30.2 ms).  DE_LOSS_GRAD_REVERSE=1|0 forces.
    const char *env = getenv("DE_LOSS_GRAD_REVERSE");
    if (env && *env == '0') return DE_OK;
    // DE_OPT_FORWARD_GRAD: the caller wants the reference's forward-mode flag semantics exactly (a product chain that overflows in one
    // association only flips `ok` in ~0.03 % of Float32 fuzz cases under reverse accumulation): forward duals whatever the width
    if (p->options & DE_OPT_FORWARD_GRAD) return DE_OK;
    // ABI 3 (round 6): reverse accumulation is an OPT-IN (DE_OPT_REVERSE_GRAD, or DE_LOSS_GRAD_REVERSE=1 for the tests / experiments): the
    // default keeps the reference's forward-mode flag semantics
    if (!(p->options & DE_OPT_REVERSE_GRAD) && !(env && *env == '1')) return DE_OK;
    // a CSE program (GraphNode trees, §3.1) reads a persistent row from several consumers: the backward sweep ACCUMULATES their adjoints
    // into that row (round 4: `acc_use` below); DE_REV_NO_SHARED=1 restores round 3's fall-back to forward duals for such populations
    if (p->cse_generic && getenv("DE_REV_NO_SHARED")) return DE_OK;
    if (!(env && *env == '1')) {
        int64_t total = 0;
        for (int64_t t = 0; t < p->n_trees; t++) total += de_program_n_grad(p, t, mode);
        if (total < 8 * p->n_trees) return DE_OK;
    }
    const int F = p->n_features, P = p->n_params;
    if (!(p->rt_valid && p->rt_mode == mode)) {
        const uint32_t es32 = p->dtype == DE_F32 ? 4u : 8u, RB = 64u * es32;
        // parameter leaves are LDS rows F .. F+P (gathered by class when the kernel stages a tile), slots follow
        const uint32_t FE = (uint32_t)F + (p->uses_params ? (uint32_t)P : 0u);
        const bool hot_const_unary = !getenv("DE_NO_CONST_UNARY_HOT");
        auto gun_of = [&](uint32_t op) { // hot unary index of a de_opcode (de_bind.h), or -1
            return hot_const_unary ? gun_index((int)op, DE_U_COS, DE_U_EXP, DE_U_SIN, DE_U_NEG, DE_U_SQUARE, DE_U_CUBE, DE_U_ABS, DE_U_LOG, DE_U_SAFE_LOG,
                                               DE_U_SQRT, DE_U_SAFE_SQRT, DE_U_TANH, DE_U_RELU) : -1;
        };
        const uint32_t PR0 = FE + (uint32_t)p->n_slots; // first partial row
        uint64_t table[ROP_COUNT];
        hipError_t hst = rev_handler_table(p->dtype, table);
        if (hst != hipSuccess) return fail(c, DE_ERR_HIP, "reverse handler table: %s", hipGetErrorString(hst));
        uint64_t base = table[0];
        for (int i = 0; i < (int)ROP_COUNT; i++) base = std::min<uint64_t>(base, table[i]);
        for (int i = 0; i < (int)ROP_COUNT; i++)
            if (table[i] - base > 0xFFFFFFFFull) return DE_OK;
        constexpr uint32_t NONE = 0xFFFFFFFFu, ACC = 0x80000000u;
        auto leaf_col = [&](uint32_t f) -> uint32_t { return mode != DE_GRAD_CONSTANT ? (1u + (uint32_t)P + f) | ACC : NONE; };
        auto param_col = [&](uint32_t r) -> uint32_t { return mode != DE_GRAD_CONSTANT ? (1u + r) | ACC : NONE; };
        auto const_col = [&](uint32_t ord) -> uint32_t {
            return mode == DE_GRAD_CONSTANT ? 1u + ord : (mode == DE_GRAD_BOTH ? 1u + (uint32_t)(P + F) + ord : NONE);
        };
        auto rowb = [&](uint32_t row) { return (row < (uint32_t)F ? row : row + (FE - (uint32_t)F)) * RB; }; // LDS byte offset of a bound row
        p->rtcode.clear();
        p->rtcode_off.assign((size_t)p->n_trees + 1, 0);
        p->rtcode_mid.assign((size_t)p->n_trees, 0);
        p->rtsite_of_gb.assign(p->gbcode.size(), -1);
        p->site_gen++;
        uint32_t max_prows = 0;
        bool ok = true;
        std::vector<uint32_t> need((size_t)p->n_trees, 0);
        std::vector<BoundInstr> rv;
        std::vector<uint8_t> rv_col; // rv[k] carries a gradient column word in .lo
        std::vector<uint32_t> rv_rop; // rop of rv[k]
        std::vector<BoundInstr> bw;   // a tree's backward records in execution order
        std::vector<uint32_t> bw_rop;
        std::vector<uint8_t> acc_use; // per instruction of the tree: reads a shared row and is not its last reader (adds its adjoint)
        std::map<uint32_t, std::pair<uint32_t, std::pair<uint32_t, uint32_t>>> occ; // column -> (leaves, (seen, row))
        std::map<uint32_t, uint32_t> rop_of_off; // handler offset -> rop id (DE_REV_STATS)
        uint32_t mk_rop = 0;                     // rop of the record `mk` made last (the emitters below read it)
        const bool rfuse = !getenv("DE_REV_NO_FUSE"); // fused pairs / triples (de_rev_threaded.hip rh_pushload ...): same bits, fewer dispatches
        auto mk = [&](uint32_t rop, uint32_t y, uint32_t z, uint32_t w) {
            BoundInstr o;
            o.bop = (uint32_t)(table[rop] - base);
            rop_of_off[o.bop] = rop;
            mk_rop = rop;
            o.arg = y;
            o.lo = z;
            o.hi = w;
            return o;
        };
        for (int64_t t = 0; t < p->n_trees && ok; t++) {
            uint32_t n_prows = 0;
            rv.clear();
            rv_col.clear();
            rv_rop.clear();
            auto alloc = [&](uint32_t n) { const uint32_t r = (PR0 + n_prows) * RB; n_prows += n; return r; };
            uint32_t last_f_rop = 0xFFFFFFFFu; // rop of this tree's last forward record
            auto F_ = [&](const BoundInstr &o) -> int32_t {
                const uint32_t rop = mk_rop;
                if (rfuse && last_f_rop == ROP_PUSH) { // PUSH + the load / unary function of a leaf that starts the next subtree: one record
                    const uint32_t push_off = p->rtcode.back().arg;
                    const bool un_leaf = rop >= ROP_UN_BASE && rop < ROP_GEN_BASE && (((rop - ROP_UN_BASE) >> 1) & 1u);
                    BoundInstr f{0u, 0u, 0u, 0u};
                    bool fused = false;
                    if (rop == rop_load(RSRC_LEAF) && push_off < 65536u && o.arg < 65536u) { f = mk(ROP_F_PUSHLOAD_BASE + 0, push_off | (o.arg << 16), 0, 0); fused = true; }
                    else if (rop == rop_load(RSRC_CONST) && push_off < 65536u) { f = mk(ROP_F_PUSHLOAD_BASE + 1, push_off, o.lo, o.hi); fused = true; }
                    else if (un_leaf && push_off < 65536u && o.arg < 65536u) {
                        const uint32_t v = rop - ROP_UN_BASE;
                        f = mk(rop_pushun((int)(v >> 2), (v & 1u) != 0), push_off | (o.arg << 16), o.lo, 0);
                        fused = true;
                    }
                    if (fused) {
                        p->rtcode.back() = f;
                        last_f_rop = 0xFFFFFFFEu;
                        return (int32_t)p->rtcode.size() - 1;
                    }
                }
                p->rtcode.push_back(o);
                last_f_rop = rop;
                return (int32_t)p->rtcode.size() - 1; // (the record that carries o's immediate: de_program_set_consts patches it there)
            };
            auto R_ = [&](const BoundInstr &o, bool has_col = false) { rv.push_back(o); rv_col.push_back(has_col ? 1 : 0); rv_rop.push_back(mk_rop); }; // pushed in forward order, reversed below
            // SHARED ROWS.  A slot row is written by a PUSH and normally read once; a GraphNode program reads a persistent row from several
            // consumers.  Backwards the consumers run in reverse order and the definition's r_pop last: the consumer that runs FIRST in the
            // backward sweep (the last reader in program order) stores its adjoint contribution into the row, every other one adds to it.
            // acc_use[i] = instruction i reads a slot row and is NOT that row's last reader before its next PUSH.
            acc_use.assign((size_t)(p->gbcode_off[(size_t)t + 1] - p->gbcode_off[(size_t)t]), 0);
            {
                std::map<uint32_t, int32_t> last_reader; // slot row -> the last instruction seen reading it since its PUSH
                for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1]; i++) {
                    const BoundInstr &b = p->gbcode[(size_t)i];
                    const uint32_t row = b.arg & 0xFFFFFFu;
                    if (b.bop == BOP_PUSH) { last_reader.erase(row); continue; }
                    const bool reads_row = b.bop == BOP_LOAD_ROW || b.bop == BOP_GEN_ROW || (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END && !((b.bop - BOP_BIN_BASE) & 2)) ||
                                           (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END && ((b.bop - BOP_UN_BASE) & 2));
                    if (!reads_row || row < (uint32_t)F) continue;
                    auto it = last_reader.find(row);
                    if (it != last_reader.end()) acc_use[(size_t)(it->second - p->gbcode_off[(size_t)t])] = 1; // no longer the last reader: it adds
                    last_reader[row] = i;
                }
            }
            auto accumulates = [&](int32_t i) { return acc_use[(size_t)(i - p->gbcode_off[(size_t)t])] != 0; };
            // backward of "acc' = op(acc, operand)" whose partial rows (d/d acc, d/d operand) start at pr
            auto back_binary = [&](int pk, uint32_t pr, bool slot, uint32_t slot_byte, uint32_t col, bool add = false) {
                if (slot && add) R_(mk(ROP_R_BINACC_BASE + (uint32_t)pk, pk == 0 ? pr : 0, slot_byte, 0));
                else if (slot) R_(mk(rop_rbin(pk, 0), pk == 0 ? pr : 0, slot_byte, 0));
                else if (col != NONE) R_(mk(rop_rbin(pk, 1), pk == 0 ? pr : 0, col, 0), true);
                else if (pk == 0) R_(mk(ROP_R_UN, pr, 0, 0));
                else if (pk == 3) R_(mk(ROP_R_NEG, 0, 0, 0));
            };
            // backward of "acc' = f(leaf)": first the unary partial, then the leaf's row — pushed in reverse
            auto back_unary_leaf = [&](uint32_t pr, uint32_t col) {
                if (col != NONE) R_(mk(ROP_R_LEAF, 0, col, 0), true);
                R_(mk(ROP_R_UN, pr, 0, 0));
            };
            // backward of "acc' = f(shared row)": the unary partial, then the row's adjoint receives the result
            auto back_unary_slot = [&](uint32_t pr, uint32_t slot_byte, bool add) {
                R_(mk(ROP_R_SLOTACC_BASE + (add ? 1u : 0u), slot_byte, 0, 0));
                R_(mk(ROP_R_UN, pr, 0, 0));
            };
            for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1] && ok; i++) {
                const BoundInstr &b = p->gbcode[(size_t)i];
                const uint32_t row = b.arg & 0xFFFFFFu, aux = b.arg >> 24, ord = b.arg & 0xFFFFu;
                const bool is_leaf = row < (uint32_t)F;
                if (b.bop == BOP_CHECK_ROW) continue; // leaf operands are tested where they are read
                if (b.bop == BOP_LOAD_ROW && !is_leaf) { // acc = a shared (persistent) row
                    F_(mk(rop_load(RSRC_SLOT), rowb(row), 0, 0));
                    R_(mk(ROP_R_SLOTACC_BASE + (accumulates(i) ? 1u : 0u), rowb(row), 0, 0));
                } else if (b.bop == BOP_LOAD_ROW) {
                    F_(mk(rop_load(RSRC_LEAF), rowb(row), 0, 0));
                    if (leaf_col(row) != NONE) R_(mk(ROP_R_LEAF, 0, leaf_col(row), 0), true);
                } else if (b.bop == BOP_LOAD_CONST) {
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_load(RSRC_CONST), 0, b.lo, b.hi));
                    if (const_col(ord) != NONE) R_(mk(ROP_R_LEAF, 0, const_col(ord), 0), true);
                } else if (b.bop == BOP_PUSH) {
                    // A spill is followed by the load that starts the next subtree (the accumulator's value is dead: the backward sweep
                    // continues with the slot's adjoint).  A SHARED definition that is used at once stays in the accumulator: the next
                    // instruction reads it, and backwards BOTH adjoints — the accumulator's and the row's — flow into the definition.
                    bool acc_live = false;
                    for (int32_t q = i + 1; q < p->gbcode_off[(size_t)t + 1]; q++) {
                        const BoundInstr &nx = p->gbcode[(size_t)q];
                        if (nx.bop == BOP_CHECK_ROW || nx.bop == BOP_CHECK_ACC || nx.bop == BOP_PUSH) continue;
                        const uint32_t nau = nx.arg >> 24;
                        acc_live = top_reads_acc(nx.bop, nau == (uint32_t)DOP_LOAD ? 0 : de_opcode_degree((int)nau));
                        break;
                    }
                    F_(mk(ROP_PUSH, rowb(row), 0, 0));
                    R_(mk(acc_live ? (uint32_t)ROP_R_POPADD : (uint32_t)ROP_R_POP, rowb(row), 0, 0));
                } else if (b.bop == BOP_CHECK_ACC) {
                    F_(mk(ROP_CHECK, 0, 0, 0));
                } else if (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END) {
                    const uint32_t v = b.bop - BOP_BIN_BASE;
                    const int k = (int)(v >> 2);
                    const bool cst = (v & 2) != 0, chk = (v & 1) != 0;
                    const uint32_t pr = k >= 3 ? alloc(2) : 0;
                    const int pk = k == 0 ? 1 : (k == 1 ? 2 : (k == 2 ? 3 : 0));
                    if (cst) {
                                                p->rtsite_of_gb[(size_t)i] = F_(mk(rop_bin(k, RSRC_CONST, chk), pr, b.lo, b.hi));
                        back_binary(pk, pr, false, 0, const_col(ord));
                    } else {
                        F_(mk(rop_bin(k, is_leaf ? RSRC_LEAF : RSRC_SLOT, chk), rowb(row), pr, 0));
                        back_binary(pk, pr, !is_leaf, rowb(row), is_leaf ? leaf_col(row) : NONE, !is_leaf && accumulates(i));
                    }
                } else if (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END) {
                    const uint32_t v = b.bop - BOP_UN_BASE;
                    const int k = (int)(v >> 2);
                    const bool from_row = (v & 2) != 0, chk = (v & 1) != 0;
                    const uint32_t pr = alloc(1);
                    if (from_row && !is_leaf) { // unary function of a shared row
                        F_(mk(rop_un_slot(k, chk), rowb(row), pr, 0));
                        back_unary_slot(pr, rowb(row), accumulates(i));
                    } else if (from_row) {
                        F_(mk(rop_un(k, RSRC_LEAF, chk), rowb(row), pr, 0));
                        back_unary_leaf(pr, leaf_col(row));
                    } else {
                        F_(mk(rop_un(k, RSRC_ACC, chk), pr, 0, 0));
                        R_(mk(ROP_R_UN, pr, 0, 0));
                    }
                } else if (b.bop == BOP_GEN_ROW && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    const uint32_t pr = alloc(2);
                    F_(mk(rop_bin(aux == (uint32_t)DE_B_MAX ? 6 : 7, is_leaf ? RSRC_LEAF : RSRC_SLOT, false), rowb(row), pr, 0));
                    back_binary(0, pr, !is_leaf, rowb(row), is_leaf ? leaf_col(row) : NONE, !is_leaf && accumulates(i));
                } else if (b.bop == BOP_GEN_ROW && gun_of(aux) >= 0 && is_leaf) {
                    const uint32_t pr = alloc(1);
                    F_(mk(rop_un(gun_of(aux), RSRC_LEAF, false), rowb(row), pr, 0));
                    back_unary_leaf(pr, leaf_col(row));
                } else if (b.bop == BOP_GEN_ROW) {
                    const bool unary = aux < (uint32_t)DE_B_ADD;
                    const uint32_t pr = alloc(unary ? 1 : 2);
                    F_(mk(rop_gen(is_leaf ? RSRC_LEAF : RSRC_SLOT), rowb(row), pr | (aux << 24), 0));
                    if (unary && !is_leaf) back_unary_slot(pr, rowb(row), accumulates(i));
                    else if (unary) back_unary_leaf(pr, leaf_col(row));
                    else back_binary(0, pr, !is_leaf, rowb(row), is_leaf ? leaf_col(row) : NONE, !is_leaf && accumulates(i));
                } else if (b.bop == BOP_GEN_CONST && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    const uint32_t pr = alloc(2);
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_bin(aux == (uint32_t)DE_B_MAX ? 6 : 7, RSRC_CONST, false), pr, b.lo, b.hi));
                    back_binary(0, pr, false, 0, const_col(ord));
                } else if (b.bop == BOP_GEN_CONST && gun_of(aux) >= 0) {
                    // cos / exp / sin of a constant leaf: load + hot unary handler instead of the generic one
                    const uint32_t pr = alloc(1);
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_load(RSRC_CONST), 0, b.lo, b.hi));
                    F_(mk(rop_un(gun_of(aux), RSRC_ACC, false), pr, 0, 0));
                    back_unary_leaf(pr, const_col(ord));
                } else if (b.bop == BOP_GEN_CONST) {
                    const bool unary = aux < (uint32_t)DE_B_ADD;
                    const uint32_t pr = alloc(unary ? 1 : 2);
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_gen(RSRC_CONST), pr | (aux << 24), b.lo, b.hi));
                    if (unary) back_unary_leaf(pr, const_col(ord));
                    else back_binary(0, pr, false, 0, const_col(ord));
                } else if (b.bop == BOP_GEN_ACC && gun_of(aux) >= 0) {
                    const uint32_t pr = alloc(1);
                    F_(mk(rop_un(gun_of(aux), RSRC_ACC, false), pr, 0, 0));
                    R_(mk(ROP_R_UN, pr, 0, 0));
                } else if (b.bop == BOP_GEN_ACC) {
                    const uint32_t pr = alloc(1);
                    F_(mk(rop_gen(RSRC_ACC), pr | (aux << 24), 0, 0));
                    R_(mk(ROP_R_UN, pr, 0, 0));
                } else if (b.bop == BOP_GEN_PARAM) { // operand = parameter row prm = LDS leaf row F + prm
                    const uint32_t prm = b.arg & 0xFFFFu, prow = ((uint32_t)F + prm) * RB;
                    int k = -1, ku = -1;
                    switch (aux) {
                    case DE_B_ADD: k = 0; break;
                    case DE_B_SUB: k = 1; break;
                    case DOP_RSUB: k = 2; break;
                    case DE_B_MUL: k = 3; break;
                    case DE_B_DIV: k = 4; break;
                    case DOP_RDIV: k = 5; break;
                    case DE_B_MAX: k = hot_const_unary ? 6 : -1; break;
                    case DE_B_MIN: k = hot_const_unary ? 7 : -1; break;
                    default: ku = gun_of(aux); break;
                    }
                    if (aux == (uint32_t)DOP_LOAD) {
                        F_(mk(rop_load(RSRC_LEAF), prow, 0, 0));
                        if (param_col(prm) != NONE) R_(mk(ROP_R_LEAF, 0, param_col(prm), 0), true);
                    } else if (k >= 0) {
                        const uint32_t pr = k >= 3 ? alloc(2) : 0;
                        F_(mk(rop_bin(k, RSRC_LEAF, false), prow, pr, 0));
                        back_binary(k == 0 ? 1 : (k == 1 ? 2 : (k == 2 ? 3 : 0)), pr, false, 0, param_col(prm));
                    } else if (ku >= 0) {
                        const uint32_t pr = alloc(1);
                        F_(mk(rop_un(ku, RSRC_LEAF, false), prow, pr, 0));
                        back_unary_leaf(pr, param_col(prm));
                    } else {
                        const bool unary = aux < (uint32_t)DE_B_ADD;
                        const uint32_t pr = alloc(unary ? 1 : 2);
                        F_(mk(rop_gen(RSRC_LEAF), prow, pr | (aux << 24), 0));
                        if (unary) back_unary_leaf(pr, param_col(prm));
                        else back_binary(0, pr, false, 0, param_col(prm));
                    }
                } else if (b.bop == BOP_TERN) {
                    if (is_leaf || b.lo < (uint32_t)F || row > 0xFFFFu || b.lo > 0xFFFFu) { ok = false; break; }
                    if (p->cse_generic) { ok = false; break; } // (a ternary operator's slot operands may be shared rows: r_tern stores; such populations keep forward duals)
                    const uint32_t pr = alloc(3);
                    const uint32_t rb_ = row + (FE - (uint32_t)F), rc_ = b.lo + (FE - (uint32_t)F);
                    if (rb_ > 0xFFFFu || rc_ > 0xFFFFu) { ok = false; break; }
                    F_(mk(ROP_TERN, pr | (aux << 24), rb_ | (rc_ << 16), 0));
                    R_(mk(ROP_R_TERN, pr, rb_ | (rc_ << 16), 0));
                } else ok = false; // INJ_*: only bound with early_exit=false, never for gradients
            }
            if (!ok) break;
            // end record of the forward sweep (r_end: the table slot of round 1's parameter handler); the backward sweep's
            // first record follows it
            p->rtcode.push_back(mk(ROP_PARAM, 0, 0, 0));
            p->rtcode_mid[(size_t)t] = (int32_t)p->rtcode.size();
            // Gradient rows several leaves share (features, parameters): the leaves' contributions are added per
            // SAMPLE in an LDS row and reduced once, at the last of them — paths that cancel within a sample then
            // cancel before the reduction, as they do in the forward Jacobian.
            // column word: [15:0] column, [29:16] accumulation row, [31:30] 0 reduce now, 1 first, 2 middle, 3 last
            occ.clear();
            for (size_t k = 0; k < rv.size(); k++)
                if (rv_col[k] && (rv[k].lo & ACC)) occ[rv[k].lo & 0xFFFFu].first++;
            uint32_t n_acc = 0;
            bw.clear();
            bw_rop.clear();
            for (size_t k = rv.size(); k-- > 0;) { // execution order
                BoundInstr o = rv[k];
                if (rv_col[k]) {
                    const uint32_t col = o.lo & 0xFFFFu;
                    if ((o.lo & 0x7FFFFFFFu) > 0xFFFFu) { ok = false; break; }
                    uint32_t word = col;
                    if (o.lo & ACC) {
                        auto &oc = occ[col];
                        if (oc.first > 1) {
                            if (oc.second.first == 0) oc.second.second = n_acc++;
                            const uint32_t nth = ++oc.second.first;
                            const uint32_t md = nth == 1 ? 1u : (nth == oc.first ? 3u : 2u);
                            word = col | ((PR0 + n_prows + oc.second.second) << 16) | (md << 30);
                        }
                    }
                    o.lo = word;
                }
                bw.push_back(o);
                bw_rop.push_back(rv_rop[k]);
            }
            if (!ok) break;
            for (size_t k = 0; k < bw.size();) { // fused backward sequences: [r_un] r_leaf [r_pop]  and  r_bin<PK, column> r_leaf [r_pop]
                auto is = [&](size_t q, uint32_t rop) { return q < bw.size() && bw_rop[q] == rop; };
                auto small = [&](size_t q) { return q >= bw.size() || bw[q].arg < 65536u; };
                if (rfuse && is(k, ROP_R_UN) && is(k + 1, ROP_R_LEAF) && small(k) && (!is(k + 2, ROP_R_POP) || small(k + 2))) {
                    const bool pop = is(k + 2, ROP_R_POP);
                    p->rtcode.push_back(mk(rop_leafx(true, pop), bw[k].arg | (pop ? bw[k + 2].arg << 16 : 0u), bw[k + 1].lo, 0));
                    k += pop ? 3 : 2;
                } else if (rfuse && is(k, ROP_R_LEAF) && is(k + 1, ROP_R_POP) && small(k + 1)) {
                    p->rtcode.push_back(mk(rop_leafx(false, true), bw[k + 1].arg << 16, bw[k].lo, 0));
                    k += 2;
                } else if (rfuse && k < bw.size() && bw_rop[k] >= ROP_R_BIN_BASE && bw_rop[k] < ROP_R_TERN && ((bw_rop[k] - ROP_R_BIN_BASE) & 1u) && is(k + 1, ROP_R_LEAF) &&
                           small(k) && (!is(k + 2, ROP_R_POP) || small(k + 2))) {
                    const bool pop = is(k + 2, ROP_R_POP);
                    p->rtcode.push_back(mk(rop_bincolx((int)((bw_rop[k] - ROP_R_BIN_BASE) >> 1), pop), bw[k].arg | (pop ? bw[k + 2].arg << 16 : 0u), bw[k].lo, bw[k + 1].lo));
                    k += pop ? 3 : 2;
                } else {
                    p->rtcode.push_back(bw[k]);
                    k += 1;
                }
            }
            if (PR0 + n_prows + n_acc > 0x3FFFu) { ok = false; break; }
            p->rtcode.push_back(mk(ROP_PARAM, 0, 0, 0)); // end record of the backward sweep
            p->rtcode_off[(size_t)t + 1] = (int32_t)p->rtcode.size();
            max_prows = std::max(max_prows, n_prows + n_acc);
            need[(size_t)t] = n_prows + n_acc;
        }
        if (getenv("DE_REV_STATS") && ok) { // dispatch classes and adjacent pairs of the two sweeps (what a fusion would save)
            auto cls = [&](uint32_t off) -> std::string {
                const uint32_t r = rop_of_off.count(off) ? rop_of_off[off] : 9999u;
                char buf[48];
                if (r < 3) snprintf(buf, sizeof buf, "LOAD%c", "LSC"[r]);
                else if (r == ROP_PUSH) return "PUSH";
                else if (r == ROP_CHECK) return "CHECK";
                else if (r >= ROP_BIN_BASE && r < ROP_UN_BASE) snprintf(buf, sizeof buf, "BIN%c", "LSC"[((r - ROP_BIN_BASE) / 2) % 3]);
                else if (r >= ROP_UN_BASE && r < ROP_GEN_BASE) snprintf(buf, sizeof buf, "UN%c", ((r - ROP_UN_BASE) / 2) % 2 ? 'L' : 'A');
                else if (r >= ROP_GEN_BASE && r < ROP_TERN) return "GEN";
                else if (r == ROP_PARAM) return "END";
                else if (r == ROP_R_UN) return "r_un";
                else if (r == ROP_R_NEG) return "r_neg";
                else if (r == ROP_R_POP) return "r_pop";
                else if (r == ROP_R_LEAF) return "r_leaf";
                else if (r >= ROP_R_BIN_BASE && r < ROP_R_TERN) snprintf(buf, sizeof buf, "r_bin%s", (r - ROP_R_BIN_BASE) % 2 ? "col" : "slot");
                else if (r >= ROP_F_PUSHLOAD_BASE && r < ROP_R_LEAFX_BASE) return "PUSH+";
                else if (r >= ROP_R_LEAFX_BASE && r < ROP_R_BINCOLX_BASE) return "r_leafx";
                else if (r >= ROP_R_BINCOLX_BASE && r < ROP_COUNT) return "r_bincolx";
                else return "other";
                return buf;
            };
            std::map<std::string, int64_t> one, two;
            for (size_t i = 0; i < p->rtcode.size(); i++) {
                const std::string a = cls(p->rtcode[i].bop);
                one[a]++;
                if (i + 1 < p->rtcode.size() && a != "END") two[a + " " + cls(p->rtcode[i + 1].bop)]++;
            }
            fprintf(stderr, "DE_REV_STATS: %zu records, %lld trees: %.2f dispatches per tree\n", p->rtcode.size(), (long long)p->n_trees, (double)p->rtcode.size() / (double)p->n_trees);
            for (auto &kv : one) fprintf(stderr, "  %-10s %8.3f per tree\n", kv.first.c_str(), (double)kv.second / (double)p->n_trees);
            std::vector<std::pair<int64_t, std::string>> v;
            for (auto &kv : two) v.push_back({kv.second, kv.first});
            std::sort(v.rbegin(), v.rend());
            for (size_t i = 0; i < v.size() && i < 24; i++) fprintf(stderr, "  pair %-22s %8.3f per tree\n", v[i].second.c_str(), (double)v[i].first / (double)p->n_trees);
        }
        // per-wave staging of the column sums: one LDS row, or the widest tree's columns
        int64_t stage_cols = 64;
        for (int64_t t = 0; t < p->n_trees; t++) stage_cols = std::max<int64_t>(stage_cols, 1 + de_program_n_grad(p, t, mode));
        const uint64_t stage_rows = ((uint64_t)stage_cols * es32 + RB - 1) / RB;
        const uint64_t rows = (uint64_t)PR0 + max_prows + stage_rows;
        if (!ok || 4 * rows * RB > 160 * 1024 || rows * RB >= (1u << 24)) { p->rtsite_of_gb.clear(); p->site_gen++; return DE_OK; }
        // The kernel is latency-bound and its occupancy is set by the LDS rows of the neediest tree of a launch
        // (5 -> 4 workgroups per CU: +17 % time): trees are grouped by the number of workgroups per CU their own
        // need allows and every group is a launch of its own (small groups join the next needier one).
        auto wgs_of = [&](uint32_t nd) { return (int)std::min<uint64_t>(8, (160 * 1024) / (4 * ((uint64_t)PR0 + nd + stage_rows) * RB)); };
        std::vector<int32_t> ids((size_t)p->n_trees);
        for (int64_t t = 0; t < p->n_trees; t++) ids[(size_t)t] = (int32_t)t;
        std::stable_sort(ids.begin(), ids.end(), [&](int32_t x, int32_t y) { return need[(size_t)x] < need[(size_t)y]; });
        p->rt_n_groups = 0;
        const bool grouping = true;
        for (int64_t k = 0; k < p->n_trees;) {
            int64_t e = k;
            const int w = wgs_of(need[(size_t)ids[(size_t)k]]);
            while (e < p->n_trees && grouping && wgs_of(need[(size_t)ids[(size_t)e]]) == w) e++;
            if (!grouping) e = p->n_trees;
            // a group too small to fill the chip, or the last slot: extend to the end / absorb into the next group
            if (p->rt_n_groups == 7) e = p->n_trees;
            while (e < p->n_trees && e - k < std::max<int64_t>(64, p->n_trees / 16)) e++;
            if (p->n_trees - e < std::max<int64_t>(64, p->n_trees / 16)) e = p->n_trees;
            GradArgs::RevGroup &gr = p->rt_groups[p->rt_n_groups++];
            gr.first = (int32_t)k;
            gr.n = (int32_t)(e - k);
            gr.rows = (int32_t)(PR0 + need[(size_t)ids[(size_t)e - 1]] + stage_rows);
            std::sort(ids.begin() + k, ids.begin() + e); // tree order inside a group: adjacent trees share staging batches
            k = e;
        }
        // the handler word of a record names the handler of the record BEHIND it; the end record of a sweep names the sweep's first
        // handler (de_rev_threaded.hip: a handler knows its successor at entry)
        for (int64_t t = 0; t < p->n_trees; t++) {
            const int32_t lim[3] = {p->rtcode_off[(size_t)t], p->rtcode_mid[(size_t)t], p->rtcode_off[(size_t)t + 1]};
            for (int sw = 0; sw < 2; sw++) {
                const int32_t a0 = lim[sw], b0 = lim[sw + 1];
                if (b0 - a0 < 2) continue;
                const uint32_t first = p->rtcode[(size_t)a0].bop;
                for (int32_t i = a0; i < b0 - 1; i++) p->rtcode[(size_t)i].bop = p->rtcode[(size_t)i + 1].bop;
                p->rtcode[(size_t)b0 - 1].bop = first;
            }
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // the previous form may be in use by queued work
        if (p->d_rtcode) { // sizes depend on the mode
            prog_free(c, p->d_rtcode);
            p->d_rtcode = nullptr;
        }
        { // (inside one 4 GiB window: the handlers bump the record pointer without a carry)
            const size_t rbytes = (p->rtcode.size() + 1) * sizeof(BoundInstr);
            const hipError_t ast = prog_malloc(c, reinterpret_cast<void **>(&p->d_rtcode), rbytes);
            if (ast != hipSuccess) return fail(c, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(ast));
            if (!in_one_window(p->d_rtcode, rbytes)) return fail(c, DE_ERR_HIP, "reverse instruction stream straddles a 4 GiB boundary");
        }
        HIP_TRY(c, hipMemset(p->d_rtcode, 0, (p->rtcode.size() + 1) * sizeof(BoundInstr)));
        if (!p->d_rtcode_off) {
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_rtcode_off), p->rtcode_off.size() * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_rtcode_mid), std::max<size_t>(p->rtcode_mid.size(), 1) * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_rt_ids), std::max<size_t>(ids.size(), 1) * sizeof(int32_t)));
        }
        if (!p->rtcode.empty())
            HIP_TRY(c, hipMemcpy(p->d_rtcode, p->rtcode.data(), p->rtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(p->d_rtcode_off, p->rtcode_off.data(), p->rtcode_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!ids.empty()) {
            HIP_TRY(c, hipMemcpy(p->d_rtcode_mid, p->rtcode_mid.data(), p->rtcode_mid.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(p->d_rt_ids, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        p->rt_stage_cols = (int)stage_cols;
        p->rt_handler_base = base;
        p->rt_param_off = (uint32_t)(table[ROP_PARAM] - base);
        p->rt_mode = mode;
        p->rt_valid = true;
    }
    g->rev_code = p->d_rtcode;
    g->rev_code_off = p->d_rtcode_off;
    g->rev_code_mid = p->d_rtcode_mid;
    g->rev_ids = p->d_rt_ids;
    g->rev_n_groups = p->rt_n_groups;
    for (int k = 0; k < p->rt_n_groups; k++) g->rev_groups[k] = p->rt_groups[k];
    g->rev_stage_cols = p->rt_stage_cols;
    g->rev_handler_base = p->rt_handler_base;
    g->rev_param_off = p->rt_param_off;
    return DE_OK;
}

// Shared body of de_eval_grad / de_eval_diff.
static int grad_impl(de_ctx *c, de_program *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                     int mode, int diff_direction, void *out, int64_t ld_out, void *grad,
                     const int64_t *grad_offsets, uint8_t *ok, const void *dY = nullptr) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    const bool diff = diff_direction >= 0;
    if (N < 0 || !ok || (p->n_trees > 0 && N > 0 && (!X || !grad))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (ldX < p->n_features || ((out || diff) && ld_out < N)) return fail(c, DE_ERR_INVALID_ARG, "ldX < n_features or ld_out < N");
    if (!diff && mode != DE_GRAD_VARIABLE && mode != DE_GRAD_CONSTANT && mode != DE_GRAD_BOTH)
        return fail(c, DE_ERR_INVALID_ARG, "bad gradient mode");
    if (diff && diff_direction >= p->n_features) return fail(c, DE_ERR_OUT_OF_RANGE, "direction >= n_features");
    int rc = check_param_args(c, p, pa, N);
    if (rc != DE_OK) return rc;
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const bool ok_dev = is_device_ptr(ok);
    std::vector<uint8_t> ones;
    const uint8_t *ok_init = p->host_ok_grad.data();
    if (diff) { // no validity test on this path: always complete (src/EvaluateDerivative.jl:117)
        ones.assign((size_t)p->n_trees, 1);
        ok_init = ones.data();
    }
    if (N == 0) {
        if (ok_dev) {
            HIP_TRY(c, hipMemcpyAsync(ok, ok_init, (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        } else std::memcpy(ok, ok_init, (size_t)p->n_trees);
        return DE_OK;
    }
    // per-tree gradient geometry
    std::vector<int32_t> ng((size_t)p->n_trees);
    std::vector<int64_t> goff((size_t)p->n_trees);
    int64_t span = 0, run = 0;
    int32_t maxg = 0;
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t g = diff ? 1 : (int32_t)de_program_n_grad(p, t, mode);
        ng[(size_t)t] = g;
        maxg = std::max(maxg, g);
        const int64_t off = diff ? t * ld_out : (grad_offsets ? grad_offsets[t] : run);
        if (off < 0) return fail(c, DE_ERR_INVALID_ARG, "negative gradient offset");
        goff[(size_t)t] = off;
        run += (int64_t)g * N;
        span = std::max(span, off + (int64_t)g * N);
    }
    const size_t lds_need = ((size_t)p->n_features + (size_t)p->n_slots * (1 + (size_t)std::min(maxg, 8))) * 260 * es;
    if (lds_need > 160 * 1024) return fail(c, DE_ERR_UNSUPPORTED, "gradient kernel: LDS footprint too large for this tree shape");
    rc = ensure_generic_code(c, p);
    if (rc) return rc;

    Staged sX, sOut, sGrad, sOk, sPar, sCls;
    rc = stage_in(c, c->sX, X, (size_t)ldX * (size_t)N * es, &sX);
    if (rc) return rc;
    if (out) {
        rc = stage_out(c, c->sOut, out, ((size_t)(p->n_trees - 1) * (size_t)ld_out + (size_t)N) * es, &sOut);
        if (rc) return rc;
    }
    rc = stage_out(c, diff ? c->sOut2 : c->sGrad, grad, (size_t)span * es, &sGrad);
    if (rc) return rc;
    if (ok_dev) sOk.dev = ok;
    else {
        HIP_TRY(c, c->sOk.reserve((size_t)p->n_trees));
        sOk.dev = c->sOk.p;
        sOk.staged = true;
    }
    // The initial flags, the gradient widths and (packed layout) the offsets depend on the program, the mode and N only:
    // they live on the device and are refreshed when one of those changes — the usual call copies nothing from pageable
    // host memory and does not block.  Caller-supplied offsets and eval_diff take the staged path.
    const bool cached = !diff && !grad_offsets;
    const int64_t *d_goff_use = nullptr;
    const int32_t *d_ng_use = nullptr;
    if (cached) {
        const size_t nt = (size_t)p->n_trees;
        if (!p->d_ok_grad) {
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_ok_grad), nt));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_ng), nt * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_goff), nt * sizeof(int64_t)));
            p->tab_ok_stale = true;
            p->tab_mode = -1;
        }
        if (p->tab_ok_stale || p->tab_mode != mode || p->tab_N != N) {
            HIP_TRY(c, hipStreamSynchronize(c->stream)); // earlier calls may still read the tables
            HIP_TRY(c, hipMemcpy(p->d_ok_grad, p->host_ok_grad.data(), nt, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(p->d_ng, ng.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(p->d_goff, goff.data(), nt * sizeof(int64_t), hipMemcpyHostToDevice));
            p->tab_ok_stale = false;
            p->tab_mode = mode;
            p->tab_N = N;
        }
        HIP_TRY(c, hipMemcpyAsync(sOk.dev, p->d_ok_grad, nt, hipMemcpyDeviceToDevice, c->stream));
        d_goff_use = p->d_goff;
        d_ng_use = p->d_ng;
    } else {
        HIP_TRY(c, hipMemcpyAsync(sOk.dev, ok_init, (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, c->sGoff.reserve(goff.size() * sizeof(int64_t)));
        HIP_TRY(c, c->sNg.reserve(ng.size() * sizeof(int32_t)));
        HIP_TRY(c, hipMemcpyAsync(c->sGoff.p, goff.data(), goff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->sNg.p, ng.data(), ng.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        d_goff_use = static_cast<const int64_t *>(c->sGoff.p);
        d_ng_use = static_cast<const int32_t *>(c->sNg.p);
    }
    if (p->uses_params) {
        rc = stage_in(c, c->sParams, pa->params, (size_t)pa->ld_params * (size_t)pa->n_classes * es, &sPar);
        if (rc) return rc;
        rc = stage_in(c, c->sClasses, pa->classes, (size_t)N * (pa->classes_is_i64 ? 8 : 4), &sCls);
        if (rc) return rc;
    }
    // the pageable host vectors of the staged path must outlive their async copies
    if (!cached) HIP_TRY(c, hipStreamSynchronize(c->stream));

    GradArgs g;
    std::memset(&g, 0, sizeof g);
    g.generic_code = p->d_gcode;
    g.e.code_off = nullptr;
    g.e.n_trees = (int32_t)p->n_trees;
    g.e.skip_flagged = !(p->options & DE_OPT_FULL_EVAL) && tree_skip_enabled(); // (the gradient entry points always test validity)
    if (c->sPrio.reserve((size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long)) == hipSuccess) g.e.prio_keys = c->sPrio.p; // priority tiles (de_kernels.hip)
    g.prio_ready = false;
    g.e.prio_keys_ready = !sX.staged && g.e.prio_keys && dataset_keys(c, p->dtype, X, N, ldX, p->n_features, &g.e.prio_keys);
    g.e.n_slots = p->n_slots;
    g.e.uses_params = p->uses_params;
    g.e.X = sX.dev;
    g.e.N = N;
    g.e.ldX = ldX;
    g.e.F = p->n_features;
    g.e.out = out ? sOut.dev : nullptr;
    g.e.ld_out = ld_out;
    g.e.ok = static_cast<uint8_t *>(sOk.dev);
    if (p->uses_params) {
        g.e.params = sPar.dev;
        g.e.ld_params = pa->ld_params;
        g.e.n_classes = pa->n_classes;
        g.e.classes = sCls.dev;
        g.e.classes_is_i64 = pa->classes_is_i64;
        g.e.class_base = pa->class_base;
    }
    g.mode = diff ? DE_GRAD_VARIABLE : mode;
    g.P = p->n_params;
    g.grad = sGrad.dev;
    g.grad_off = d_goff_use;
    g.n_grad = d_ng_use;
    g.max_grad = maxg;
    g.diff_direction = diff ? diff_direction : -1;
    g.e.code_off = p->d_gcode_off;
    if (!diff) {
        rc = ensure_grad_threaded(c, p, mode, ng, N, &g);
        if (rc) return rc;
    }
    Staged sDY;
    if (dY) {
        rc = stage_in(c, c->sY, dY, (size_t)N * es, &sDY);
        if (rc) return rc;
    }
    HIP_TRY(c, time_begin(c));
    HIP_TRY(c, launch_grad(p->dtype, g, c->stream, &c->last_kernel));
    if (dY) // the pullback's dX .* dY' (and its NaN fill) on the Jacobians just written
        HIP_TRY(c, launch_pullback_scale(p->dtype, sGrad.dev, g.grad_off, g.n_grad, g.e.ok, sDY.dev, N, p->n_trees, maxg, c->stream));
    HIP_TRY(c, time_end(c));
    if (out && sOut.staged)
        for (int64_t t = 0; t < p->n_trees; t++)
            HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(out) + (size_t)t * (size_t)ld_out * es,
                                      static_cast<char *>(sOut.dev) + (size_t)t * (size_t)ld_out * es, (size_t)N * es,
                                      hipMemcpyDeviceToHost, c->stream));
    if (sGrad.staged) {
        if (diff) {
            for (int64_t t = 0; t < p->n_trees; t++)
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(grad) + (size_t)t * (size_t)ld_out * es,
                                          static_cast<char *>(sGrad.dev) + (size_t)t * (size_t)ld_out * es, (size_t)N * es,
                                          hipMemcpyDeviceToHost, c->stream));
        } else {
            for (int64_t t = 0; t < p->n_trees; t++)
                if (ng[(size_t)t] > 0)
                    HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(grad) + (size_t)goff[(size_t)t] * es,
                                              static_cast<char *>(sGrad.dev) + (size_t)goff[(size_t)t] * es,
                                              (size_t)ng[(size_t)t] * (size_t)N * es, hipMemcpyDeviceToHost, c->stream));
        }
    }
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    if (sX.staged || sOut.staged || sGrad.staged || sOk.staged || sPar.staged || sCls.staged) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (g.e.skip_flagged && ((out && sOut.staged) || sGrad.staged)) {
        // host buffers: rows / Jacobians of incomplete trees were only partly written into staging buffers every program of the context
        // shares — NaN-fill them (as eval_impl does; src/EvaluationHelpers.jl:56-62 does the same one level up)
        std::vector<uint8_t> okh;
        const uint8_t *okp = ok;
        if (ok_dev) {
            okh.resize((size_t)p->n_trees);
            HIP_TRY(c, hipMemcpy(okh.data(), ok, (size_t)p->n_trees, hipMemcpyDeviceToHost));
            okp = okh.data();
        }
        auto fill = [&](void *base, size_t off, size_t n) {
            if (p->dtype == DE_F32) std::fill_n(static_cast<float *>(base) + off, n, std::nanf(""));
            else std::fill_n(static_cast<double *>(base) + off, n, std::nan(""));
        };
        for (int64_t t = 0; t < p->n_trees; t++) {
            if (okp[t]) continue;
            if (out && sOut.staged) fill(out, (size_t)t * (size_t)ld_out, (size_t)N);
            if (sGrad.staged && diff) fill(grad, (size_t)t * (size_t)ld_out, (size_t)N);
            else if (sGrad.staged && ng[(size_t)t] > 0) fill(grad, (size_t)goff[(size_t)t], (size_t)ng[(size_t)t] * (size_t)N);
        }
    }
    return DE_OK;
}

// By-class reduction in ONE pass (de_eval_loss_grad_by_class): class-aligned tiles, then one pair of finish passes per
// class into loss_c / dloss_c ([C][n_trees] and [C][span], device).  Only the reverse kernel takes a tile table:
// `done` stays false when the population runs forward duals and the caller falls back to one call per class.
struct ByClassPlan {
    const int64_t *class_starts;
    int64_t C, span;
    void *loss_c, *dloss_c;
    bool done;
};
static int loss_grad_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                          int mode, const void *y, const void *w, int32_t loss_kind, void *loss, void *dloss,
                          const int64_t *dloss_offsets, uint8_t *ok, ByClassPlan *plan);
int de_eval_loss_grad(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                      int mode, const void *y, const void *w, int32_t loss_kind, void *loss, void *dloss,
                      const int64_t *dloss_offsets, uint8_t *ok) {
    DE_NOTHROW(c, loss_grad_impl(c, p, X, N, ldX, pa, mode, y, w, loss_kind, loss, dloss, dloss_offsets, ok, nullptr));
}
static int loss_grad_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                          int mode, const void *y, const void *w, int32_t loss_kind, void *loss, void *dloss,
                          const int64_t *dloss_offsets, uint8_t *ok, ByClassPlan *plan) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    if (N < 0 || !ok || (p->n_trees > 0 && (!dloss || (N > 0 && (!X || !y))))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (ldX < p->n_features) return fail(c, DE_ERR_INVALID_ARG, "ldX < n_features");
    if (mode != DE_GRAD_VARIABLE && mode != DE_GRAD_CONSTANT && mode != DE_GRAD_BOTH) return fail(c, DE_ERR_INVALID_ARG, "bad gradient mode");
    if (loss_kind != DE_LOSS_L2 && loss_kind != DE_LOSS_L1 && loss_kind != DE_LOSS_PULLBACK)
        return fail(c, DE_ERR_INVALID_ARG, "unknown loss_kind %d", loss_kind);
    int rc = check_param_args(c, p, pa, N);
    if (rc != DE_OK) return rc;
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    // per-tree geometry: tree t owns reduction columns col_off[t] (loss) .. col_off[t] + n_grad[t]
    std::vector<int32_t> ng((size_t)p->n_trees);
    std::vector<int64_t> coloff((size_t)p->n_trees + 1, 0), doff((size_t)p->n_trees);
    int64_t span = 0, run = 0;
    int32_t maxg = 0;
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t g = (int32_t)de_program_n_grad(p, t, mode);
        ng[(size_t)t] = g;
        maxg = std::max(maxg, g);
        const int64_t off = dloss_offsets ? dloss_offsets[t] : run;
        if (off < 0) return fail(c, DE_ERR_INVALID_ARG, "negative dloss offset");
        doff[(size_t)t] = off;
        run += g;
        span = std::max(span, off + g);
        coloff[(size_t)t + 1] = coloff[(size_t)t] + 1 + g;
    }
    const int64_t n_cols = coloff[(size_t)p->n_trees];
    const bool ok_dev = is_device_ptr(ok);
    if (N == 0) { // empty sums: 0, or NaN where a constant already fails the flag
        std::vector<unsigned char> zl((size_t)p->n_trees * es), zd((size_t)std::max<int64_t>(span, 1) * es);
        auto put = [&](unsigned char *b, int64_t i, double v) {
            if (p->dtype == DE_F32) reinterpret_cast<float *>(b)[i] = (float)v;
            else reinterpret_cast<double *>(b)[i] = v;
        };
        for (int64_t t = 0; t < p->n_trees; t++) {
            const double v = p->host_ok_grad[(size_t)t] ? 0.0 : std::nan("");
            put(zl.data(), t, v);
            for (int32_t k = 0; k < ng[(size_t)t]; k++) put(zd.data(), doff[(size_t)t] + k, v);
        }
        for (int64_t t = 0; t < p->n_trees; t++) // only the entries each tree owns are written
            if (ng[(size_t)t] > 0)
                HIP_TRY(c, hipMemcpy(static_cast<char *>(dloss) + (size_t)doff[(size_t)t] * es, zd.data() + (size_t)doff[(size_t)t] * es,
                                     (size_t)ng[(size_t)t] * es, hipMemcpyDefault));
        if (loss) HIP_TRY(c, hipMemcpy(loss, zl.data(), zl.size(), hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(ok, p->host_ok_grad.data(), (size_t)p->n_trees, hipMemcpyDefault));
        return DE_OK;
    }
    const bool timing = getenv("DE_DEBUG_TIMING") != nullptr;
    const auto tg0 = std::chrono::steady_clock::now();
    rc = ensure_generic_code(c, p);
    if (rc) return rc;
    const auto tg1 = std::chrono::steady_clock::now();

    Staged sX, sY, sW, sLoss, sDl, sOk, sPar, sCls;
    rc = stage_in(c, c->sX, X, (size_t)ldX * (size_t)N * es, &sX);
    if (rc) return rc;
    rc = stage_in(c, c->sY, y, (size_t)N * es, &sY);
    if (rc) return rc;
    if (w) {
        rc = stage_in(c, c->sW, w, (size_t)N * es, &sW);
        if (rc) return rc;
    }
    if (loss) {
        rc = stage_out(c, c->sLoss, loss, (size_t)p->n_trees * es, &sLoss);
        if (rc) return rc;
    }
    rc = stage_out(c, c->sDloss, dloss, (size_t)std::max<int64_t>(span, 1) * es, &sDl);
    if (rc) return rc;
    if (ok_dev) sOk.dev = ok;
    else {
        HIP_TRY(c, c->sOk.reserve((size_t)p->n_trees));
        sOk.dev = c->sOk.p;
        sOk.staged = true;
    }
    int64_t n_tiles = (N + 255) / 256;
    std::vector<int64_t> tile_range, class_tile0; // by-class: (first, last) sample of every class-aligned tile; first tile of every class
    if (plan) {
        class_tile0.assign((size_t)plan->C + 1, 0);
        for (int64_t k = 0; k < plan->C; k++) {
            const int64_t j0 = plan->class_starts[k], j1 = plan->class_starts[k + 1];
            for (int64_t b = j0; b < j1; b += 256) {
                tile_range.push_back(b);
                tile_range.push_back(j1 - 1);
            }
            class_tile0[(size_t)k + 1] = (int64_t)(tile_range.size() / 2);
        }
        n_tiles = (int64_t)(tile_range.size() / 2);
    }
    HIP_TRY(c, c->sPartial.reserve((size_t)n_tiles * (size_t)n_cols * 4 * es));
    // (by class: three regions — the finish passes of the classes run on the caller's stream and two side streams, launch_loss_grad_finish_ranges)
    const size_t seg_region = (size_t)loss_segments(n_tiles) * (size_t)n_cols * 4 * sizeof(double);
    const int seg_regions = plan ? 3 : 1;
    HIP_TRY(c, c->sSeg.reserve(seg_region * (size_t)seg_regions));
    HIP_TRY(c, c->sNg.reserve(ng.size() * sizeof(int32_t)));
    HIP_TRY(c, c->sColOff.reserve(coloff.size() * sizeof(int64_t)));
    HIP_TRY(c, c->sDoff.reserve(doff.size() * sizeof(int64_t)));
    HIP_TRY(c, hipMemcpyAsync(sOk.dev, p->host_ok_grad.data(), (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sNg.p, ng.data(), ng.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sColOff.p, coloff.data(), coloff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sDoff.p, doff.data(), doff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (p->uses_params) {
        rc = stage_in(c, c->sParams, pa->params, (size_t)pa->ld_params * (size_t)pa->n_classes * es, &sPar);
        if (rc) return rc;
        rc = stage_in(c, c->sClasses, pa->classes, (size_t)N * (pa->classes_is_i64 ? 8 : 4), &sCls);
        if (rc) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // the pageable host vectors above must outlive their async copies

    LossArgs la;
    std::memset(&la, 0, sizeof la);
    la.y = sY.dev;
    la.w = w ? sW.dev : nullptr;
    la.kind = loss_kind;
    la.partial = c->sPartial.p;
    la.seg_sum = c->sSeg.p;
    la.loss = loss ? sLoss.dev : nullptr;
    GradArgs g;
    std::memset(&g, 0, sizeof g);
    g.generic_code = p->d_gcode;
    g.e.code_off = p->d_gcode_off;
    g.e.n_trees = (int32_t)p->n_trees;
    g.e.skip_flagged = !(p->options & DE_OPT_FULL_EVAL) && tree_skip_enabled(); // (the gradient entry points always test validity)
    if (c->sPrio.reserve((size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long)) == hipSuccess) g.e.prio_keys = c->sPrio.p; // priority tiles (de_kernels.hip)
    g.prio_ready = false;
    g.e.prio_keys_ready = !sX.staged && g.e.prio_keys && dataset_keys(c, p->dtype, X, N, ldX, p->n_features, &g.e.prio_keys);
    g.e.n_slots = p->n_slots;
    g.e.uses_params = p->uses_params;
    g.e.X = sX.dev;
    g.e.N = N;
    g.e.ldX = ldX;
    g.e.F = p->n_features;
    g.e.out = nullptr;
    g.e.ld_out = N;
    g.e.ok = static_cast<uint8_t *>(sOk.dev);
    if (p->uses_params) {
        g.e.params = sPar.dev;
        g.e.ld_params = pa->ld_params;
        g.e.n_classes = pa->n_classes;
        g.e.classes = sCls.dev;
        g.e.classes_is_i64 = pa->classes_is_i64;
        g.e.class_base = pa->class_base;
    }
    g.mode = mode;
    g.P = p->n_params;
    g.grad = nullptr;
    g.grad_off = nullptr;
    g.n_grad = static_cast<const int32_t *>(c->sNg.p);
    g.max_grad = maxg;
    g.diff_direction = -1;
    g.loss = &la;
    g.col_off = static_cast<const int64_t *>(c->sColOff.p);
    g.n_cols = n_cols;
    g.dloss = sDl.dev;
    g.dloss_off = static_cast<const int64_t *>(c->sDoff.p);
    rc = ensure_rev_threaded(c, p, mode, &g);
    if (rc) return rc;
    if (plan && !g.rev_code) return DE_OK; // forward duals: the caller runs one call per class (plan->done stays false)
    if (plan) {
        HIP_TRY(c, c->sBcTiles.reserve(std::max<size_t>(tile_range.size(), 2) * sizeof(int64_t)));
        if (!tile_range.empty())
            HIP_TRY(c, hipMemcpyAsync(c->sBcTiles.p, tile_range.data(), tile_range.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // tile_range is pageable
        g.rev_tile_range = static_cast<const int64_t *>(c->sBcTiles.p);
        g.rev_n_tiles = n_tiles;
    }
    if (!g.rev_code) {
        const size_t lds_need = ((size_t)p->n_features + (size_t)p->n_slots * (1 + (size_t)std::min(maxg, 8))) * 260 * es;
        if (lds_need > 160 * 1024) return fail(c, DE_ERR_UNSUPPORTED, "gradient kernel: LDS footprint too large for this tree shape");
        rc = ensure_grad_threaded(c, p, mode, ng, N, &g);
        if (rc) return rc;
    }
    if (timing) {
        const auto tg2 = std::chrono::steady_clock::now();
        fprintf(stderr, "loss_grad host us: generic code %ld, staging + threaded/reverse code %ld\n",
                (long)std::chrono::duration_cast<std::chrono::microseconds>(tg1 - tg0).count(),
                (long)std::chrono::duration_cast<std::chrono::microseconds>(tg2 - tg1).count());
    }
    if (!c->nested) HIP_TRY(c, time_begin(c));
    if (g.rev_code) HIP_TRY(c, launch_rev_threaded(p->dtype, g, c->stream, &c->last_kernel));
    else HIP_TRY(c, launch_grad(p->dtype, g, c->stream, &c->last_kernel));
    if (plan) { // one pair of finish passes per class over its own tiles
        HIP_TRY(c, launch_loss_grad_finish_ranges(p->dtype, g, plan->C, class_tile0.data(), plan->loss_c, (size_t)p->n_trees * es, plan->dloss_c,
                                                  (size_t)plan->span * es, seg_region, seg_regions, c->stream));
        plan->done = true;
    }
    if (!c->nested) HIP_TRY(c, time_end(c));
    if (sLoss.staged) HIP_TRY(c, hipMemcpyAsync(loss, sLoss.dev, (size_t)p->n_trees * es, hipMemcpyDeviceToHost, c->stream));
    if (sDl.staged)
        for (int64_t t = 0; t < p->n_trees; t++)
            if (ng[(size_t)t] > 0)
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(dloss) + (size_t)doff[(size_t)t] * es,
                                          static_cast<char *>(sDl.dev) + (size_t)doff[(size_t)t] * es, (size_t)ng[(size_t)t] * es,
                                          hipMemcpyDeviceToHost, c->stream));
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    if (sX.staged || sY.staged || sW.staged || sLoss.staged || sDl.staged || sOk.staged || sPar.staged || sCls.staged)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DE_OK;
}

static int by_class_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX,
                               const de_param_args_t *pa, int mode, const void *y, const void *w, int32_t loss_kind,
                               const int64_t *class_starts, void *loss, void *dloss, const int64_t *dloss_offsets,
                               void *dparams, uint8_t *ok);
int de_eval_loss_grad_by_class(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX,
                               const de_param_args_t *pa, int mode, const void *y, const void *w, int32_t loss_kind,
                               const int64_t *class_starts, void *loss, void *dloss, const int64_t *dloss_offsets,
                               void *dparams, uint8_t *ok) {
    DE_NOTHROW(c, by_class_impl(c, p, X, N, ldX, pa, mode, y, w, loss_kind, class_starts, loss, dloss, dloss_offsets, dparams, ok));
}
static int by_class_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX,
                               const de_param_args_t *pa, int mode, const void *y, const void *w, int32_t loss_kind,
                               const int64_t *class_starts, void *loss, void *dloss, const int64_t *dloss_offsets,
                               void *dparams, uint8_t *ok) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    if (p->n_params <= 0 || !pa) return fail(c, DE_ERR_INVALID_ARG, "not a parametric population (n_params = 0 or no parameter arguments)");
    if (mode != DE_GRAD_VARIABLE && mode != DE_GRAD_BOTH)
        return fail(c, DE_ERR_INVALID_ARG, "by-class reduction needs a mode with parameter rows (DE_GRAD_VARIABLE / DE_GRAD_BOTH)");
    if (!pa->params || !pa->classes || pa->ld_params < p->n_params || pa->n_classes <= 0)
        return fail(c, DE_ERR_INVALID_ARG, "bad parameter arguments");
    if (N < 0 || !ok || !class_starts || (p->n_trees > 0 && (!dloss || !dparams))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    const int64_t C = pa->n_classes;
    if (class_starts[0] != 0 || class_starts[C] != N) return fail(c, DE_ERR_INVALID_ARG, "class_starts must run from 0 to N");
    for (int64_t k = 0; k < C; k++)
        if (class_starts[k + 1] < class_starts[k]) return fail(c, DE_ERR_INVALID_ARG, "class_starts must be non-decreasing");
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const int P = p->n_params;
    std::vector<int32_t> ng((size_t)p->n_trees);
    std::vector<int64_t> doff((size_t)p->n_trees);
    int64_t span = 0, run = 0;
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t g = (int32_t)de_program_n_grad(p, t, mode);
        ng[(size_t)t] = g;
        const int64_t off = dloss_offsets ? dloss_offsets[t] : run;
        if (off < 0) return fail(c, DE_ERR_INVALID_ARG, "negative dloss offset");
        doff[(size_t)t] = off;
        run += g;
        span = std::max(span, off + g);
    }
    span = std::max<int64_t>(span, 1);
    HIP_TRY(c, c->sBcLoss.reserve((size_t)C * (size_t)p->n_trees * es));
    HIP_TRY(c, c->sBcDloss.reserve((size_t)C * (size_t)span * es));
    HIP_TRY(c, c->sBcOk.reserve((size_t)C * (size_t)p->n_trees));
    HIP_TRY(c, c->sBcNg.reserve(ng.size() * sizeof(int32_t)));
    HIP_TRY(c, c->sBcDoff.reserve(doff.size() * sizeof(int64_t)));
    HIP_TRY(c, hipMemcpyAsync(c->sBcNg.p, ng.data(), ng.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sBcDoff.p, doff.data(), doff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    // dloss entries no tree owns (caller-chosen offsets) are never read by the combine pass
    HIP_TRY(c, hipMemsetAsync(c->sBcDloss.p, 0, (size_t)C * (size_t)span * es, c->stream));
    HIP_TRY(c, time_begin(c));
    ByClassPlan plan{class_starts, C, span, c->sBcLoss.p, c->sBcDloss.p, false};
    bool shared_ok = false; // one pass: a single flag array instead of one per class
    {
        const char *env1 = getenv("DE_BY_CLASS_ONE_PASS");
        if (!(env1 && *env1 == '0') && N > 0) {
            c->nested++;
            const int rc1 = loss_grad_impl(c, p, X, N, ldX, pa, mode, y, w, loss_kind, c->sBcLoss.p, c->sBcDloss.p, dloss_offsets,
                                           static_cast<uint8_t *>(c->sBcOk.p), &plan);
            c->nested--;
            if (rc1 != DE_OK) return rc1;
            shared_ok = plan.done;
        }
    }
    struct Nest { // inner calls leave the timing events alone; restored on every exit path
        de_ctx *c;
        explicit Nest(de_ctx *c_) : c(c_) { c->nested++; }
        ~Nest() { c->nested--; }
    };
    int rc = DE_OK;
    const size_t cls_es = pa->classes_is_i64 ? 8 : 4;
    {
    Nest nest(c);
    for (int64_t k = 0; k < C && rc == DE_OK && !plan.done; k++) {
        const int64_t j0 = class_starts[k], n = class_starts[k + 1] - j0;
        de_param_args_t sub = *pa;
        sub.classes = static_cast<const char *>(pa->classes) + (size_t)j0 * cls_es;
        rc = de_eval_loss_grad(c, p, static_cast<const char *>(X) + (size_t)j0 * (size_t)ldX * es, n, ldX, &sub, mode,
                               y ? static_cast<const char *>(y) + (size_t)j0 * es : nullptr,
                               w ? static_cast<const char *>(w) + (size_t)j0 * es : nullptr, loss_kind,
                               static_cast<char *>(c->sBcLoss.p) + (size_t)k * (size_t)p->n_trees * es,
                               static_cast<char *>(c->sBcDloss.p) + (size_t)k * (size_t)span * es, dloss_offsets,
                               static_cast<uint8_t *>(c->sBcOk.p) + (size_t)k * (size_t)p->n_trees);
    }
    }
    if (rc != DE_OK) return rc;
    Staged sLoss, sDl, sDp, sOk;
    if (loss) {
        rc = stage_out(c, c->sLoss, loss, (size_t)p->n_trees * es, &sLoss);
        if (rc) return rc;
    }
    rc = stage_out(c, c->sDloss, dloss, (size_t)span * es, &sDl);
    if (rc) return rc;
    const size_t dp_bytes = (size_t)p->n_trees * (size_t)C * (size_t)P * es;
    rc = stage_out(c, c->sBcOut, dparams, dp_bytes, &sDp);
    if (rc) return rc;
    rc = stage_out(c, c->sOk, ok, (size_t)p->n_trees, &sOk);
    if (rc) return rc;
    ByClassArgs a;
    a.loss_c = c->sBcLoss.p;
    a.dloss_c = c->sBcDloss.p;
    a.ok_c = static_cast<const uint8_t *>(c->sBcOk.p);
    a.n_classes = (int32_t)C;
    a.ok_stride = shared_ok ? 0 : p->n_trees;
    a.n_params = P;
    a.n_trees = p->n_trees;
    a.span = span;
    a.n_grad = static_cast<const int32_t *>(c->sBcNg.p);
    a.dloss_off = static_cast<const int64_t *>(c->sBcDoff.p);
    a.loss = loss ? sLoss.dev : nullptr;
    a.dloss = sDl.dev;
    a.dparams = sDp.dev;
    a.ok = static_cast<uint8_t *>(sOk.dev);
    HIP_TRY(c, launch_by_class_combine(p->dtype, a, c->stream));
    HIP_TRY(c, time_end(c));
    if (sLoss.staged) HIP_TRY(c, hipMemcpyAsync(loss, sLoss.dev, (size_t)p->n_trees * es, hipMemcpyDeviceToHost, c->stream));
    if (sDl.staged)
        for (int64_t t = 0; t < p->n_trees; t++)
            if (ng[(size_t)t] > 0)
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(dloss) + (size_t)doff[(size_t)t] * es,
                                          static_cast<char *>(sDl.dev) + (size_t)doff[(size_t)t] * es, (size_t)ng[(size_t)t] * es,
                                          hipMemcpyDeviceToHost, c->stream));
    if (sDp.staged) HIP_TRY(c, hipMemcpyAsync(dparams, sDp.dev, dp_bytes, hipMemcpyDeviceToHost, c->stream));
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // ng/doff (pageable) were copied asynchronously
    return DE_OK;
}

int de_eval_grad(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                 int mode, void *out, int64_t ld_out, void *grad, const int64_t *grad_offsets, uint8_t *ok) {
    DE_NOTHROW(c, grad_impl(c, p, X, N, ldX, pa, mode, -1, out, ld_out, grad, grad_offsets, ok));
}

int de_eval_pullback_dX(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                        const void *dY, void *dX, const int64_t *dX_offsets, uint8_t *ok) {
    if (c && N > 0 && !dY) return fail(c, DE_ERR_INVALID_ARG, "null cotangent dY");
    DE_NOTHROW(c, grad_impl(c, p, X, N, ldX, pa, DE_GRAD_VARIABLE, -1, nullptr, N, dX, dX_offsets, ok, dY));
}

int de_eval_diff(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, int32_t direction, void *out,
                 void *dout, int64_t ld_out, uint8_t *ok) {
    if (direction < 0) return fail(c, DE_ERR_INVALID_ARG, "direction < 0");
    if (p && p->uses_params) return fail(c, DE_ERR_UNSUPPORTED, "eval_diff on parametric trees");
    DE_NOTHROW(c, grad_impl(c, p, X, N, ldX, nullptr, DE_GRAD_VARIABLE, direction, out, ld_out, dout, nullptr, ok));
}

} // extern "C"
