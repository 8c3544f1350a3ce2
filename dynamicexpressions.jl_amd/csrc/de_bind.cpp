// de_bind.cpp — see de_bind.h.
#include "de_bind.h"

#include "../../include/de_opcodes.h"

namespace de {
namespace {

int hot_binary_index(uint32_t op) {
    switch (op) {
    case DE_B_ADD: return 0;
    case DE_B_SUB: return 1;
    case DOP_RSUB: return 2;
    case DE_B_MUL: return 3;
    case DE_B_DIV: return 4;
    case DOP_RDIV: return 5;
    default: return -1;
    }
}
int hot_unary_index(uint32_t op) {
    switch (op) {
    case DE_U_COS: return 0;
    case DE_U_EXP: return 1;
    case DE_U_SIN: return 2;
    default: return -1;
    }
}
BoundInstr mk(uint32_t bop, uint32_t arg, uint32_t lo = 0, uint32_t hi = 0) {
    BoundInstr b;
    b.bop = bop;
    b.arg = arg;
    b.lo = lo;
    b.hi = hi;
    return b;
}

} // namespace

void bind_tree(const Instr *code, size_t n, bool ee, int n_features, std::vector<BoundInstr> *out) {
    for (size_t i = 0; i < n; i++) {
        const Instr &ins = code[i];
        const uint32_t hdr = ins.hdr;
        const uint32_t op = hdr & H_OP_MASK;
        const uint32_t src = (hdr >> H_SRC_SHIFT) & H_SRC_MASK;
        const uint32_t row = ins.feat & 0xFFFFu;
        const uint32_t lo = ins.imm.u32[0], hi = ins.imm.u32[1];
        if (hdr & H_PUSH) out->push_back(mk(BOP_PUSH, (uint32_t)n_features + ((hdr >> H_PUSH_SHIFT) & H_SLOT_MASK)));
        const bool check_b = ee && (hdr & H_CHECK_B);
        if (check_b && src == SRC_ROW) out->push_back(mk(BOP_CHECK_ROW, row));
        const bool check_out = op != DOP_LOAD && (hdr & (ee ? H_CHECK_OUT : H_CHECK_ALWAYS));
        const bool inject = !ee && (hdr & H_INJECT);
        bool check_done = false;
        if (op == DOP_LOAD) {
            if (src == SRC_ROW) out->push_back(mk(BOP_LOAD_ROW, row));
            else if (src == SRC_CONST) out->push_back(mk(BOP_LOAD_CONST, ins.feat >> 16, lo, hi)); // arg = constant ordinal (gradient seed)
            else out->push_back(mk(BOP_GEN_PARAM, row | (check_b ? 1u << 23 : 0u) | (DOP_LOAD << 24)));
        } else if (op >= DE_T_FMA && op < DOP_LOAD) {
            out->push_back(mk(BOP_TERN, row | (op << 24), (uint32_t)n_features + ((hdr >> H_POPC_SHIFT) & H_SLOT_MASK)));
        } else if (inject && (src == SRC_ACC || src == SRC_ROW)) {
            out->push_back(mk(src == SRC_ACC ? BOP_INJ_ACC : BOP_INJ_ROW, row | (op << 24)));
        } else if (src == SRC_PARAM) {
            out->push_back(mk(BOP_GEN_PARAM, row | (check_b ? 1u << 23 : 0u) | (op << 24)));
        } else {
            const int kb = hot_binary_index(op), ku = hot_unary_index(op);
            if (kb >= 0 && (src == SRC_ROW || src == SRC_CONST)) {
                out->push_back(mk(BOP_BIN_BASE + 4 * kb + (src == SRC_CONST ? 2 : 0) + (check_out ? 1 : 0),
                                  src == SRC_CONST ? (ins.feat >> 16) : row, lo, hi));
                check_done = true;
            } else if (ku >= 0 && (src == SRC_ACC || src == SRC_ROW)) {
                out->push_back(mk(BOP_UN_BASE + 4 * ku + (src == SRC_ROW ? 2 : 0) + (check_out ? 1 : 0), row));
                check_done = true;
            } else if (src == SRC_ROW) {
                out->push_back(mk(BOP_GEN_ROW, row | (op << 24)));
            } else if (src == SRC_CONST) {
                out->push_back(mk(BOP_GEN_CONST, ((ins.feat >> 16) & 0xFFFFu) | (op << 24), lo, hi));
            } else {
                out->push_back(mk(BOP_GEN_ACC, op << 24));
            }
        }
        if (check_out && !check_done) out->push_back(mk(BOP_CHECK_ACC, 0));
    }
}

} // namespace de
