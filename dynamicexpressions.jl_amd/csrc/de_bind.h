// de_bind.h — second lowering stage: generic accumulator instructions (de_program.h) ->
// BOUND instructions, one dense handler id per (operator, operand source, check) combination.
//
// Why: the eval kernel is scalar-issue bound (1 scalar instruction / 4 cycles / SIMD on gfx950,
// measured — DESIGN.md §Tuning).  Decoding {opcode, source kind, PUSH, CHECK_B, CHECK_OUT,
// INJECT} with nested uniform branches costs ~60 scalar instructions per interpreted
// instruction under LLVM's structurizer; a single flat switch over a pre-bound handler id
// costs a third of that.  Flags become explicit micro-instructions (PUSH, CHECK_ROW,
// CHECK_ACC) or handler variants, so handlers are straight-line code.
#pragma once
#include <stdint.h>

#include <vector>

#include "de_program.h"

namespace de {

// Hot operators get dedicated handlers:   op in {ADD,SUB,RSUB,MUL,DIV,RDIV} x {ROW,CONST} x {plain,checked}
//                                         op in {COS,EXP,SIN} x {ACC,ROW} x {plain,checked}
enum BoundOp : uint32_t {
    BOP_LOAD_ROW = 0,
    BOP_LOAD_CONST,
    BOP_PUSH,       // spill acc to row `arg`
    BOP_CHECK_ROW,  // validity-test LDS row `arg` (a leaf the reference tests)
    BOP_CHECK_ACC,  // validity-test acc
    // binary hot block: base + 4*k + 2*(src==CONST) + checked,  k = ADD,SUB,RSUB,MUL,DIV,RDIV
    BOP_BIN_BASE,
    BOP_BIN_END = BOP_BIN_BASE + 24,
    // unary hot block: base + 4*k + 2*(src==ROW) + checked,     k = COS,EXP,SIN
    BOP_UN_BASE = BOP_BIN_END,
    BOP_UN_END = BOP_UN_BASE + 12,
    // generic handlers: the de_opcode travels in arg[31:24]
    BOP_GEN_ROW = BOP_UN_END, // acc = op(acc, row) / op(row)
    BOP_GEN_CONST,
    BOP_GEN_ACC,
    BOP_GEN_PARAM,  // operand = params[arg & 0xFFFF, class]; checked when arg bit 23 set
    BOP_TERN,       // acc = op3(row arg, row lo, acc)
    BOP_INJ_ACC,    // early_exit=false fused deg1: acc = finite(acc) ? op(acc) : Inf
    BOP_INJ_ROW,
    BOP_COUNT
};

// Handler ids of the reverse-accumulation kernel (de_rev_threaded.hip).  Operand kinds RSRC_*; hot binary
// K = 0 ADD, 1 SUB, 2 RSUB, 3 MUL, 4 DIV, 5 RDIV; hot unary K = 0 cos, 1 exp, 2 sin.
enum { RSRC_LEAF = 0, RSRC_SLOT = 1, RSRC_CONST = 2, RSRC_ACC = 3 };
enum RevOp : uint32_t {
    ROP_LOAD_BASE = 0,                 // + src (LEAF, SLOT, CONST)
    ROP_PUSH = 3,
    ROP_CHECK = 4,
    ROP_BIN_BASE = 5,                  // + ((K*3 + src)*2 + checked),  K < GBIN_K (8)
    ROP_UN_BASE = ROP_BIN_BASE + 48,   // + ((K*2 + (src == LEAF))*2 + checked),  K < GUN_K (13)
    ROP_GEN_BASE = ROP_UN_BASE + 52,   // + src (LEAF, SLOT, CONST, ACC)
    ROP_TERN = ROP_GEN_BASE + 4,
    ROP_PARAM,
    ROP_R_UN,                          // backward: adjoint *= partial
    ROP_R_NEG,
    ROP_R_POP,
    ROP_R_LEAF,
    ROP_R_BIN_BASE,                    // + PK*2 + OK   (PK: 0 partial rows, 1 ADD, 2 SUB, 3 RSUB; OK: 0 slot, 1 column)
    ROP_R_TERN = ROP_R_BIN_BASE + 8,
    // fused pairs / triples (round 4: 37.8 -> ~28 dispatches per tree of the C5 pullback; de_rev_threaded.hip):
    ROP_F_PUSHLOAD_BASE,                                // forward  PUSH + LOAD: + 0 leaf, + 1 constant
    ROP_F_PUSHUN_BASE = ROP_F_PUSHLOAD_BASE + 2,        // forward  PUSH + unary(leaf): + K*2 + checked, K < GUN_K
    ROP_R_LEAFX_BASE = ROP_F_PUSHUN_BASE + 26,          // backward [r_un] r_leaf [r_pop]: + PRE*2 + POP  (0: unused)
    ROP_R_BINCOLX_BASE = ROP_R_LEAFX_BASE + 4,          // backward r_bin<PK, column> r_leaf [r_pop]: + PK*2 + POP
    // shared (GraphNode) rows, round 4: a persistent slot row read by SEVERAL consumers — the backward sweep accumulates their adjoints
    ROP_UN_SLOT_BASE = ROP_R_BINCOLX_BASE + 8,          // forward  unary(slot row): + K*2 + checked, K < GUN_K
    ROP_R_SLOTACC_BASE = ROP_UN_SLOT_BASE + 26,         // backward of "acc = slot" / of unary(slot): + 0 slot = adjoint, + 1 slot += adjoint
    ROP_R_BINACC_BASE = ROP_R_SLOTACC_BASE + 2,         // backward r_bin<PK, slot> that ADDS to the slot's adjoint: + PK
    ROP_R_POPADD = ROP_R_BINACC_BASE + 4,               // backward of a PUSH behind which the accumulator stays live (a shared definition used at once): adjoint += slot
    ROP_COUNT
};
constexpr uint32_t rop_load(int src) { return ROP_LOAD_BASE + (uint32_t)src; }
constexpr uint32_t rop_bin(int k, int src, bool chk) { return ROP_BIN_BASE + (uint32_t)((k * 3 + src) * 2 + (chk ? 1 : 0)); }
constexpr uint32_t rop_un(int k, int src, bool chk) { return ROP_UN_BASE + (uint32_t)((k * 2 + (src == RSRC_LEAF ? 1 : 0)) * 2 + (chk ? 1 : 0)); }
constexpr uint32_t rop_gen(int src) { return ROP_GEN_BASE + (uint32_t)src; }
constexpr uint32_t rop_rbin(int pk, int ok) { return ROP_R_BIN_BASE + (uint32_t)(pk * 2 + ok); }
constexpr uint32_t rop_pushun(int k, bool chk) { return ROP_F_PUSHUN_BASE + (uint32_t)(k * 2 + (chk ? 1 : 0)); }
constexpr uint32_t rop_leafx(bool pre, bool pop) { return ROP_R_LEAFX_BASE + (pre ? 2u : 0u) + (pop ? 1u : 0u); }
constexpr uint32_t rop_un_slot(int k, bool chk) { return ROP_UN_SLOT_BASE + (uint32_t)(k * 2 + (chk ? 1 : 0)); }
constexpr uint32_t rop_bincolx(int pk, bool pop) { return ROP_R_BINCOLX_BASE + (uint32_t)(pk * 2 + (pop ? 1 : 0)); }

struct alignas(16) BoundInstr {
    uint32_t bop;
    uint32_t arg; // [23:0] LDS row index / parameter row / constant ordinal (constant operands); [31:24] de_opcode for generic handlers
    uint32_t lo, hi; // immediate bits (f32: lo; f64: lo,hi); BOP_TERN: lo = second row
};
static_assert(sizeof(BoundInstr) == 16, "BoundInstr must be 16 bytes");

// True for handlers whose operand is an inline constant (arg = constant ordinal, not an LDS row).
inline bool bop_is_const_source(uint32_t bop) {
    if (bop == BOP_LOAD_CONST || bop == BOP_GEN_CONST) return true;
    return bop >= BOP_BIN_BASE && bop < BOP_BIN_END && ((bop - BOP_BIN_BASE) & 2);
}

// ---- third stage (threaded kernel only): superinstructions -----------------------------------
// On the bench population 12 % of the bound dispatches are PUSH, 8 % CHECK_ROW and 13 % LOAD_*;
// a dispatch costs ~24 scalar + 3 vector instructions whatever it does, and the kernel is
// co-limited by scalar issue and VALU.  fuse_tree() folds
//   [PUSH s] [CHECK_ROW r] LOAD_ROW r                    -> TOP_LOADROW   (push, chk)
//   [PUSH s] LOAD_CONST c                                -> TOP_LOADCONST_PUSH
//   [PUSH s] [CHECK_ROW r] UN(row r)                     -> TOP_UNROW     (k, out-check, push, chk)
//   CHECK_ROW r; BIN(row r)                              -> TOP_BINROWC   (k, out-check)
//   [PUSH s] LOAD_ROW a; BIN(row b | const c)            -> TOP_BIN2      (k, const?, out-check, push)
//   [PUSH s] LOAD_CONST c; BIN(row a)                    -> TOP_BIN2 with the operator mirrored
// into one dispatch each; everything else keeps its BoundOp id (ids < BOP_COUNT are shared).
// Encoding of a fused instruction (BoundInstr fields):
//   arg[23:0]  = row A (the LDS row operand)             arg[31:24] = int8 (push row - row A), 0 if no push
//   lo, hi     = the constant's bits, or for TOP_BIN2 row-row: lo = (row B - row A) as int32
//   TOP_LOADCONST_PUSH: arg = the push row itself.
enum ThreadedOp : uint32_t {
    TOP_LOADROW_BASE = BOP_COUNT,                 // + 2*push + chk                         (4)
    TOP_LOADCONST_PUSH = TOP_LOADROW_BASE + 4,    //                                        (1)
    TOP_UNROW_BASE = TOP_LOADCONST_PUSH + 1,      // + ((k*2 + out)*2 + push)*2 + chk       (24)
    TOP_BINROWC_BASE = TOP_UNROW_BASE + 24,       // + k*2 + out                            (12)
    TOP_BIN2_BASE = TOP_BINROWC_BASE + 12,        // + ((k*2 + const)*2 + out)*2 + push     (48)
    TOP_COUNT = TOP_BIN2_BASE + 48
};
constexpr uint32_t top_loadrow(bool push, bool chk) { return TOP_LOADROW_BASE + 2 * push + chk; }
constexpr uint32_t top_unrow(int k, bool out, bool push, bool chk) { return TOP_UNROW_BASE + ((k * 2 + out) * 2 + push) * 2 + chk; }
constexpr uint32_t top_binrowc(int k, bool out) { return TOP_BINROWC_BASE + k * 2 + out; }
constexpr uint32_t top_bin2(int k, bool cst, bool out, bool push) { return TOP_BIN2_BASE + ((k * 2 + cst) * 2 + out) * 2 + push; }

// ---- handler ids of the threaded GRADIENT kernel (de_grad_threaded.hip) ----------------------------
// Operand kinds are resolved on the host: LEAF = feature row, SLOT = spilled dual number, CONST = inline
// constant, ACC.  LEAF and CONST operands seed ONE gradient component (or none in this mode); with a
// single window that component is known on the host, so those handlers exist once per seed
// ("seed variant" sv = 0: read the row at run time (several windows), 1: no gradient, 2 + k: row k):
// the dense  g1*d1[k] + g2*d2[k]  then needs no one-hot materialisation.  Window width GC gives
// NS = GC + 2 seed variants; ids depend on GC (one module per GC anyway).
constexpr int GOP_MAX = 760; // >= gop_count(8)
// hot binary operators of the gradient kernels: ADD SUB RSUB MUL DIV RDIV | MAX MIN  (the lowering treats max/min as commutative)
constexpr int GBIN_K = 8;
// hot unary operators of the gradient kernels: cos exp sin | neg square cube abs log safe_log sqrt safe_sqrt tanh relu
constexpr int GUN_K = 13;
static_assert(ROP_GEN_BASE - ROP_UN_BASE == 4 * GUN_K, "RevOp layout and GUN_K disagree");
static_assert(ROP_UN_BASE - ROP_BIN_BASE == 6 * GBIN_K, "RevOp layout and GBIN_K disagree");
// de_opcode -> hot unary index, or -1 (de_opcodes.h values are passed in: this header stays free of that include)
constexpr int gun_index(int op, int u_cos, int u_exp, int u_sin, int u_neg, int u_square, int u_cube, int u_abs, int u_log,
                        int u_safe_log, int u_sqrt, int u_safe_sqrt, int u_tanh, int u_relu) {
    return op == u_cos ? 0 : op == u_exp ? 1 : op == u_sin ? 2 : op == u_neg ? 3 : op == u_square ? 4 : op == u_cube ? 5 : op == u_abs ? 6 :
           op == u_log ? 7 : op == u_safe_log ? 8 : op == u_sqrt ? 9 : op == u_safe_sqrt ? 10 : op == u_tanh ? 11 : op == u_relu ? 12 : -1;
}
enum { GSRC_LEAF = 0, GSRC_SLOT = 1, GSRC_CONST = 2, GSRC_ACC = 3 };
constexpr uint32_t gop_ns(int GC) { return (uint32_t)GC + 2; }
constexpr uint32_t gop_load(int GC, int src, int sv) { // LEAF: [0,NS)  SLOT: NS  CONST: NS+1+sv
    return src == GSRC_LEAF ? (uint32_t)sv : (src == GSRC_SLOT ? gop_ns(GC) : gop_ns(GC) + 1 + (uint32_t)sv);
}
constexpr uint32_t gop_push(int GC) { return 2 * gop_ns(GC) + 1; }
constexpr uint32_t gop_check_acc(int GC) { return gop_push(GC) + 1; }
constexpr uint32_t gop_bin_base(int GC) { return gop_check_acc(GC) + 1; }
constexpr uint32_t gop_bin(int GC, int k, int src, int sv, bool chk) { // GBIN_K x 2 chk x (LEAF NS + SLOT + CONST NS)
    return gop_bin_base(GC) + (uint32_t)(k * 2 + chk) * (2 * gop_ns(GC) + 1) + gop_load(GC, src, sv);
}
constexpr uint32_t gop_un_base(int GC) { return gop_bin_base(GC) + 2 * GBIN_K * (2 * gop_ns(GC) + 1); }
constexpr uint32_t gop_un(int GC, int k, int src, int sv, bool chk) { // GUN_K x 2 chk x (LEAF NS + SLOT + ACC)
    return gop_un_base(GC) + (uint32_t)(k * 2 + chk) * (gop_ns(GC) + 2) +
           (src == GSRC_LEAF ? (uint32_t)sv : (src == GSRC_SLOT ? gop_ns(GC) : gop_ns(GC) + 1));
}
constexpr uint32_t gop_gen_base(int GC) { return gop_un_base(GC) + 2 * GUN_K * (gop_ns(GC) + 2); }
constexpr uint32_t gop_gen(int GC, int src) { return gop_gen_base(GC) + (uint32_t)src; } // LEAF, SLOT, CONST, ACC (run-time seeds)
constexpr uint32_t gop_param(int GC) { return gop_gen_base(GC) + 4; }
constexpr uint32_t gop_tern(int GC) { return gop_gen_base(GC) + 5; }
// spill the accumulator to a slot, then load a leaf row / a constant: one dispatch for the PUSH + LOAD pair that starts every
// right-hand subtree (LEAF: NS seed variants, then CONST: NS)
constexpr uint32_t gop_pushload(int GC, int src, int sv) { return gop_gen_base(GC) + 6 + (src == GSRC_LEAF ? 0u : gop_ns(GC)) + (uint32_t)sv; }
constexpr uint32_t gop_count(int GC) { return gop_gen_base(GC) + 6 + 2 * gop_ns(GC); }
static_assert(gop_count(8) <= GOP_MAX, "GOP_MAX too small");

// Handlers the threaded EVAL kernel has beyond the fused set: unary operators outside the binder's hot set
// (GUN_K - 3 of them: neg .. relu, the list of gun_index) on a row or on the accumulator.  Chosen when the
// threaded code is made (de_api_program.cpp make_threaded) instead of the generic handler; ids follow TOP_COUNT.
constexpr uint32_t TOPX_UN_BASE = TOP_COUNT;                 // + (k - 3) * 2 + (src == ACC)
constexpr uint32_t TOPX_BIN_BASE = TOPX_UN_BASE + 2 * (GUN_K - 3); // max / min: + (k - 6) * 2 + (src == CONST)
constexpr uint32_t TOPX_COUNT = TOPX_BIN_BASE + 4;
// handler table of the threaded eval kernel: the ids above + the end-of-tree handler every chain finishes in
constexpr uint32_t TOPX_END = TOPX_COUNT;
// ... and "last instruction of a tree + its end" variants of the handlers most trees finish in (a validity-tested hot binary or
// unary operator): the stream's end record is then skipped, one dispatch less per tree.  Chosen by make_chained (de_api_program.cpp).
constexpr uint32_t TOPX_ENDV_BASE = TOPX_END + 1;                 // + k * 2 + (operand is a constant), k < 6   | 12 + k, k < 3 (unary on acc)
constexpr uint32_t TOPX_ENDV_COUNT = 15;
// ... and two handlers no record names but every chain can reach (the out-of-line end of a tree, the early-exit walk over skipped
// trees): in the table so that the host's address-window checks (32-bit offsets; Float64: one 4 GiB window) cover them too
constexpr uint32_t TOPX_AUX_BASE = TOPX_ENDV_BASE + TOPX_ENDV_COUNT; // + 0: h_tree_end_slow, + 1: h_tree_skip
constexpr uint32_t TOPX_TABLE = TOPX_AUX_BASE + 2;
// chained stream (de_api_program.cpp make_chained): bit 31 of a tree header's length word = the tree finishes in an end-fused handler
constexpr uint32_t DE_HDR_FUSED_END = 0x80000000u;
// end variant of a fused / bound handler id, or -1
constexpr int topx_endv_of(uint32_t id) {
    return (id >= BOP_BIN_BASE && id < BOP_BIN_END && ((id - BOP_BIN_BASE) & 1)) ? (int)(((id - BOP_BIN_BASE) >> 2) * 2 + (((id - BOP_BIN_BASE) >> 1) & 1))
         : (id >= BOP_UN_BASE && id < BOP_UN_END && ((id - BOP_UN_BASE) & 3) == 1) ? 12 + (int)((id - BOP_UN_BASE) >> 2) : -1;
}

// True when a (bound or fused) instruction READS the accumulator (or spills it).  A tree never starts with one — the accumulator
// is undefined at the start of a tree — which is what lets the end of a tree leave it uncleared (de_kernels.hip HTREE_END_TAIL);
// de_program_verify checks it.  (aux = the de_opcode's degree for the generic row / parameter handlers: unary = operand only.)
inline bool top_reads_acc(uint32_t top, int generic_degree) {
    if (top == BOP_LOAD_ROW || top == BOP_LOAD_CONST || top == BOP_CHECK_ROW || top == BOP_INJ_ROW) return false;
    if (top == BOP_GEN_ROW || top == BOP_GEN_CONST || top == BOP_GEN_PARAM) return generic_degree != 1 && generic_degree != 0; // degree 0: plain load of a parameter
    if (top >= BOP_UN_BASE && top < BOP_UN_END) return ((top - BOP_UN_BASE) & 2u) == 0; // bit 1 = the operand is a row
    if (top < BOP_COUNT) return true; // PUSH, CHECK_ACC, the binary block, GEN_ACC, TERN, INJ_ACC
    if (top >= TOP_LOADROW_BASE && top < TOP_LOADCONST_PUSH) return ((top - TOP_LOADROW_BASE) & 2u) != 0; // push
    if (top == TOP_LOADCONST_PUSH) return true;
    if (top >= TOP_UNROW_BASE && top < TOP_BINROWC_BASE) return ((top - TOP_UNROW_BASE) & 2u) != 0; // push
    if (top >= TOP_BINROWC_BASE && top < TOP_BIN2_BASE) return true;
    if (top >= TOP_BIN2_BASE && top < TOP_COUNT) return ((top - TOP_BIN2_BASE) & 1u) != 0; // push
    if (top >= TOPX_UN_BASE && top < TOPX_BIN_BASE) return ((top - TOPX_UN_BASE) & 1u) != 0; // src == ACC
    return true; // max / min handlers, anything unknown
}

// True when a (bound or fused) instruction carries a constant's bits in lo/hi.  Every generic
// instruction with a constant operand becomes exactly one such instruction, in program order, in the
// bound and in the fused stream — which is how de_program_set_consts patches immediates in place.
inline bool top_carries_const(uint32_t top) {
    if (top < BOP_COUNT) return bop_is_const_source(top);
    if (top == TOP_LOADCONST_PUSH) return true;
    if (top >= TOP_BIN2_BASE && top < TOP_COUNT) return (((top - TOP_BIN2_BASE) >> 2) & 1u) != 0;
    return false;
}

// Fused form of one tree's bound instructions (appended to `out`).
// `rows` (wave groups, de_api_program.cpp make_threaded): the tree's spill-slot rows are [slot_lo, slot_hi) in THIS stream variant, `shift`
// rows behind their place in variant 0, and no variant moves them by more than `headroom` rows: a fusion whose int8 row distance would not
// fit in EVERY variant is not made in any — the variants of a stream differ in operand words only.  Null: one stream, the plain +-127 rule.
struct FuseRows { uint32_t slot_lo, slot_hi; int32_t shift, headroom; };
void fuse_tree(const BoundInstr *b, size_t n, std::vector<BoundInstr> *out, const FuseRows *rows = nullptr);
// True when the fused instruction's operand is an inline constant (arg carries no operand row).
bool top_is_const_source(uint32_t top);

// Append the bound form of `code` (one tree) to `out`.
// param_row_base >= 0: parameter operands are LDS rows param_row_base + p (the eval kernels stage the tile's parameter values like
// features: de_api_program.cpp `prows`); < 0: BOP_GEN_PARAM, a gather per use
// slot_shift: added to every SPILL-SLOT row (rows >= n_features of the lowering: pushes, popped operands, ternary operands, shared rows) —
// the stream variant of wave w > 0 of a wave group keeps its slots behind the staged parameter rows (de_api_program.cpp make_wave_variants)
void bind_tree(const Instr *code, size_t n, bool early_exit, int n_features, std::vector<BoundInstr> *out, int param_row_base = -1, int slot_shift = 0);

} // namespace de
