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

// Append the bound form of `code` (one tree) to `out`.
void bind_tree(const Instr *code, size_t n, bool early_exit, int n_features, std::vector<BoundInstr> *out);

} // namespace de
