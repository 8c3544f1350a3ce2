// de_program.h — the DEVICE program: what a post-order tape (include/de_hip.h) is lowered
// to by de_lower.cpp and what the gfx950 kernels in de_kernels.hip interpret.
//
// Execution model: an accumulator machine.  Each sample owns one accumulator
// `acc` (registers) and a small spill stack (LDS).  Every instruction fetches
// at most ONE memory operand B (a feature row from the LDS-staged X tile, an
// inline constant, a spill slot, or a per-class parameter) and applies
//     LOAD   : acc = B
//     unary  : acc = op(B)            (B may be ACC itself)
//     binary : acc = op(acc, B)       (SWAP: op(B, acc))
//     ternary: acc = op(B, C, acc)    (B, C spill slots)
// so a 20-node tree is ~11-14 instructions, leaves never occupy a stack slot,
// and the instruction word is wave-uniform: it is fetched with scalar loads
// and decoded on the scalar unit.
#pragma once
#include <stdint.h>

namespace de {

// Internal opcodes beyond include/de_opcodes.h (which are used verbatim).
enum : uint32_t {
    DOP_LOAD = 0xF0, // acc = B
    // reversed forms of the non-commutative binary operators: acc = op(B, acc)
    DOP_RSUB = 0xF1,
    DOP_RDIV = 0xF2,
    DOP_RPOW = 0xF3,
    DOP_RMOD = 0xF4,
    DOP_RREM = 0xF5,
    DOP_RGREATER = 0xF6,
    DOP_RPOW_ABS2 = 0xF7,
};

// Operand-B kinds (hdr bits 8..10).  SRC_ROW reads an LDS row: rows 0..F-1 are the
// staged features of X, row F+s is spill slot s (so a leaf and a popped value are the
// same kind of access to the kernel).
enum : uint32_t { SRC_ACC = 0, SRC_ROW = 1, SRC_CONST = 2, SRC_PARAM = 4 };

// hdr layout
constexpr uint32_t H_OP_MASK = 0xFFu;
constexpr uint32_t H_SRC_SHIFT = 8, H_SRC_MASK = 0x7u;
constexpr uint32_t H_PUSH = 1u << 11;         // spill acc to slot PUSH_SLOT before executing
constexpr uint32_t H_CHECK_B = 1u << 12;      // validity-test operand B (a leaf the reference tests)
constexpr uint32_t H_CHECK_ALWAYS = 1u << 13; // result of a constant-folded subtree: tested even without early_exit
constexpr uint32_t H_INJECT = 1u << 14;       // reference fused deg1 kernels: non-finite input => Inf
// early-exit mode: validity-test the RESULT of this instruction.  Cleared by the lowering when
// the (single) consumer of the value provably turns a non-finite input into a non-finite,
// tested output (x+y, x-y, x*y, numerator of x/y, cos, ...): the flag stays exact while most
// tests disappear from the instruction stream.
constexpr uint32_t H_CHECK_OUT = 1u << 15;
constexpr uint32_t H_PUSH_SHIFT = 20;         // bits 20..23: slot written by H_PUSH
constexpr uint32_t H_POPC_SHIFT = 24;         // bits 24..27: second slot of a ternary op
constexpr uint32_t H_SLOT_MASK = 0xFu;
constexpr int MAX_SLOTS = 16;

// One instruction = 16 bytes = one s_load_dwordx4.
struct alignas(16) Instr {
    uint32_t hdr;
    uint32_t feat; // [15:0] LDS row (SRC_ROW) / parameter row (SRC_PARAM) of B; [31:16] gradient row of a CONST operand
    union {
        float f32;
        double f64;
        uint32_t u32[2];
    } imm; // value of a CONST operand
};
static_assert(sizeof(Instr) == 16, "Instr must be 16 bytes");

} // namespace de
