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

void bind_tree(const Instr *code, size_t n, bool ee, int n_features, std::vector<BoundInstr> *out, int param_row_base, int slot_shift) {
    const uint32_t slot0 = (uint32_t)n_features + (uint32_t)slot_shift; // row of spill slot 0
    for (size_t i = 0; i < n; i++) {
        const Instr &ins = code[i];
        const uint32_t hdr = ins.hdr;
        const uint32_t op = hdr & H_OP_MASK;
        uint32_t src = (hdr >> H_SRC_SHIFT) & H_SRC_MASK;
        uint32_t row = ins.feat & 0xFFFFu;
        if (src == SRC_PARAM && param_row_base >= 0) { // a parameter = one more staged row
            src = SRC_ROW;
            row += (uint32_t)param_row_base;
        } else if (src == SRC_ROW && row >= (uint32_t)n_features) row += (uint32_t)slot_shift; // a popped / shared slot row
        const uint32_t lo = ins.imm.u32[0], hi = ins.imm.u32[1];
        if (hdr & H_PUSH) out->push_back(mk(BOP_PUSH, slot0 + ((hdr >> H_PUSH_SHIFT) & H_SLOT_MASK)));
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
            out->push_back(mk(BOP_TERN, row | (op << 24), slot0 + ((hdr >> H_POPC_SHIFT) & H_SLOT_MASK)));
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

// ---- superinstruction pass ---------------------------------------------------------------------
namespace {
struct HotBin { int k; bool cst, out; };
bool hot_bin_of(const BoundInstr &b, HotBin *h) {
    if (b.bop < BOP_BIN_BASE || b.bop >= BOP_BIN_END) return false;
    const uint32_t v = b.bop - BOP_BIN_BASE;
    h->k = (int)(v >> 2);
    h->cst = (v & 2) != 0;
    h->out = (v & 1) != 0;
    return true;
}
// acc = x op b  ->  the same value written as  b' op' x'  with the operands swapped
int mirrored(int k) {
    switch (k) {
    case 1: return 2; // SUB  -> RSUB
    case 2: return 1;
    case 4: return 5; // DIV  -> RDIV
    case 5: return 4;
    default: return k; // ADD, MUL
    }
}
bool delta_ok(uint32_t a, uint32_t b, const FuseRows *fr = nullptr) {
    int64_t d = (int64_t)b - (int64_t)a, lo = -127, hi = 127;
    if (fr) { // the distance in variant 0, and the range every variant's distance stays in
        const bool sa = a >= fr->slot_lo && a < fr->slot_hi, sb = b >= fr->slot_lo && b < fr->slot_hi;
        d -= (int64_t)fr->shift * ((sb ? 1 : 0) - (sa ? 1 : 0));
        if (sb && !sa) hi -= fr->headroom;
        if (sa && !sb) lo += fr->headroom;
    }
    return d >= lo && d <= hi;
}
uint32_t with_delta(uint32_t rowA, bool push, uint32_t push_row) {
    const int32_t d = push ? (int32_t)push_row - (int32_t)rowA : 0;
    return (rowA & 0xFFFFFFu) | ((uint32_t)(d & 0xFF) << 24);
}
} // namespace

bool top_is_const_source(uint32_t top) {
    if (top < BOP_COUNT) return bop_is_const_source(top);
    return top == TOP_LOADCONST_PUSH; // TOP_BIN2 row-const keeps row A in arg
}

void fuse_tree(const BoundInstr *b, size_t n, std::vector<BoundInstr> *out, const FuseRows *fr) {
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        bool push = false, chk = false;
        uint32_t push_row = 0, chk_row = 0;
        if (b[j].bop == BOP_PUSH) { push = true; push_row = b[j].arg & 0xFFFFFFu; j++; }
        if (j < n && b[j].bop == BOP_CHECK_ROW) { chk = true; chk_row = b[j].arg & 0xFFFFFFu; j++; }
        if (j < n && (push || chk)) {
            const BoundInstr &m = b[j];
            const uint32_t row = m.arg & 0xFFFFFFu;
            HotBin hb;
            if (m.bop == BOP_LOAD_ROW && (!chk || chk_row == row) && (!push || delta_ok(row, push_row, fr))) {
                if (!chk && j + 1 < n && hot_bin_of(b[j + 1], &hb) && (hb.cst || delta_ok(row, b[j + 1].arg & 0xFFFFFFu, fr))) {
                    const BoundInstr &nb = b[j + 1];
                    BoundInstr f = nb;
                    f.bop = top_bin2(hb.k, hb.cst, hb.out, push);
                    f.arg = with_delta(row, push, push_row);
                    if (!hb.cst) { f.lo = (uint32_t)((int32_t)(nb.arg & 0xFFFFFFu) - (int32_t)row); f.hi = 0; }
                    out->push_back(f);
                    i = j + 2;
                    continue;
                }
                BoundInstr f = m;
                f.bop = top_loadrow(push, chk);
                f.arg = with_delta(row, push, push_row);
                out->push_back(f);
                i = j + 1;
                continue;
            }
            if (m.bop == BOP_LOAD_CONST && push && !chk) {
                if (j + 1 < n && hot_bin_of(b[j + 1], &hb) && !hb.cst && delta_ok(b[j + 1].arg & 0xFFFFFFu, push_row, fr)) {
                    const uint32_t rowA = b[j + 1].arg & 0xFFFFFFu;
                    BoundInstr f = m; // keeps the constant's bits
                    f.bop = top_bin2(mirrored(hb.k), true, hb.out, true);
                    f.arg = with_delta(rowA, true, push_row);
                    out->push_back(f);
                    i = j + 2;
                    continue;
                }
                BoundInstr f = m;
                f.bop = TOP_LOADCONST_PUSH;
                f.arg = push_row;
                out->push_back(f);
                i = j + 1;
                continue;
            }
            if (m.bop >= BOP_UN_BASE && m.bop < BOP_UN_END && ((m.bop - BOP_UN_BASE) & 2) && (!chk || chk_row == row) &&
                (!push || delta_ok(row, push_row, fr))) {
                const uint32_t v = m.bop - BOP_UN_BASE;
                BoundInstr f = m;
                f.bop = top_unrow((int)(v >> 2), (v & 1) != 0, push, chk);
                f.arg = with_delta(row, push, push_row);
                out->push_back(f);
                i = j + 1;
                continue;
            }
            if (!push && chk && hot_bin_of(m, &hb) && !hb.cst && chk_row == row) {
                BoundInstr f = m;
                f.bop = top_binrowc(hb.k, hb.out);
                f.arg = row;
                out->push_back(f);
                i = j + 1;
                continue;
            }
        } else if (j < n) { // no PUSH / CHECK_ROW prefix: LOAD + BIN pairs
            const BoundInstr &m = b[j];
            HotBin hb;
            if (j + 1 < n && hot_bin_of(b[j + 1], &hb)) {
                const BoundInstr &nb = b[j + 1];
                if (m.bop == BOP_LOAD_ROW && (hb.cst || delta_ok(m.arg & 0xFFFFFFu, nb.arg & 0xFFFFFFu, fr))) {
                    const uint32_t row = m.arg & 0xFFFFFFu;
                    BoundInstr f = nb;
                    f.bop = top_bin2(hb.k, hb.cst, hb.out, false);
                    f.arg = row;
                    if (!hb.cst) { f.lo = (uint32_t)((int32_t)(nb.arg & 0xFFFFFFu) - (int32_t)row); f.hi = 0; }
                    out->push_back(f);
                    i = j + 2;
                    continue;
                }
                if (m.bop == BOP_LOAD_CONST && !hb.cst) {
                    BoundInstr f = m;
                    f.bop = top_bin2(mirrored(hb.k), true, hb.out, false);
                    f.arg = nb.arg & 0xFFFFFFu;
                    out->push_back(f);
                    i = j + 2;
                    continue;
                }
            }
        }
        out->push_back(b[i]); // unfused: same id as the bound form
        i++;
    }
}

} // namespace de
