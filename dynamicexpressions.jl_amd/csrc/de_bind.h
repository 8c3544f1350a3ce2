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
// Operand kinds are resolved on the host: LEAF = feature row (one-hot seed), SLOT = spilled dual
// number, CONST = inline constant (seed by ordinal), ACC.
enum GradOp : uint32_t {
    GOP_LOAD_LEAF = 0,
    GOP_LOAD_SLOT,
    GOP_LOAD_CONST,
    GOP_PUSH,
    GOP_CHECK_ACC,
    GOP_BIN_BASE,                     // + (k*3 + src)*2 + chk,  src: 0 LEAF 1 SLOT 2 CONST   (36)
    GOP_UN_BASE = GOP_BIN_BASE + 36,  // + (k*3 + src)*2 + chk,  src: 0 LEAF 1 SLOT 2 ACC     (18)
    GOP_GEN_LEAF = GOP_UN_BASE + 18,
    GOP_GEN_SLOT,
    GOP_GEN_CONST,
    GOP_GEN_ACC,
    GOP_PARAM,
    GOP_TERN,
    GOP_COUNT
};
constexpr uint32_t gop_bin(int k, int src, bool chk) { return GOP_BIN_BASE + (k * 3 + src) * 2 + chk; }
constexpr uint32_t gop_un(int k, int src, bool chk) { return GOP_UN_BASE + (k * 3 + (src == 3 ? 2 : src)) * 2 + chk; } // src 3 = ACC

// Fused form of one tree's bound instructions (appended to `out`).
void fuse_tree(const BoundInstr *b, size_t n, std::vector<BoundInstr> *out);
// True when the fused instruction's operand is an inline constant (arg carries no operand row).
bool top_is_const_source(uint32_t top);

// Append the bound form of `code` (one tree) to `out`.
void bind_tree(const Instr *code, size_t n, bool early_exit, int n_features, std::vector<BoundInstr> *out);

} // namespace de
