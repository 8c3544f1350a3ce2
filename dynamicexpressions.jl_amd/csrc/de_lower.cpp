// de_lower.cpp — host lowering of one post-order tape into the accumulator-machine
// program of de_program.h.
//
// Two jobs:
//  1. ANNOTATE: walk the tree exactly like the reference's recursive dispatch
//     (reference src/Evaluate.jl:337-364 `_eval_tree_array`, :599-651
//     `dispatch_deg1_eval`, :488-577 `dispatch_deg2_eval`, :428-487 degree>2) to
//     decide WHICH values the reference validity-tests, so that the device `ok`
//     flag is identical to the reference's `complete` flag for the same inputs —
//     including which feature leaves are (not) tested by the fused 2/3-node
//     kernels, which constants are value-tested, and where the fused deg1
//     kernels substitute Inf (:722,737,754,787).
//  2. CODEGEN: emit accumulator code with Sethi-Ullman child ordering so the
//     spill stack stays minimal (a leaf never needs a slot).
#include "de_lower.h"

#include <algorithm>
#include <cstring>

namespace de {
namespace {

struct LNode {
    uint8_t degree = 0, op = 0;
    uint16_t arg = 0;
    int child[3] = {-1, -1, -1};
    bool is_const = false;   // is_constant(subtree), src/NodeUtils.jl:73
    bool chk_leaf = false;   // leaf whose array the reference validity-tests (early_exit)
    bool inject = false;     // outer op of a fused deg1 kernel
    bool constfold = false;  // inside a subtree evaluated by dispatch_constant_tree
    int fold_slot = -1;      // >= 0: maximal constant subtree replaced by extended constant slot
    int first = 0;           // index of the first tape node of this subtree (post-order slice [first, i])
    int cfirst = 0, ccount = 0; // constant slots owned by the subtree
    int regs = 0;            // spill slots needed to evaluate into acc
    int share_def = -1;      // >= 0: this node is the first occurrence of shared subtree share_def (CSE tapes)
    uint32_t defs = 0, uses = 0; // share ids defined / referenced inside this subtree (bit masks)
};

struct Lowerer {
    const LowerOptions &opt;
    std::vector<LNode> nodes;
    TreeProgram *out;
    int pending_push = -1;
    int max_slots = 0;
    std::string *err;

    Lowerer(const LowerOptions &o, TreeProgram *p, std::string *e) : opt(o), out(p), err(e) {}

    int share_base = 0;      // first persistent slot of the shared subtrees (= spill slots the tree needs)
    // A DE_LEAF_SHARED operand is an operand for codegen (one LDS row read) but a SUBTREE for the reference's dispatch: the
    // reference evaluates the shared node again at every occurrence (src/Evaluate.jl:337-364 has no cache), so the parent
    // never takes a fused leaf kernel with it and its value is one the parent's @return_on_nonfinite_array tests — which
    // the definition's unconditional result test (assign_check_out) already did on the same values.
    bool is_shared_leaf(int i) const { return nodes[i].degree == 0 && nodes[i].op == DE_LEAF_SHARED; }
    bool is_leaf(int i) const { return nodes[i].degree == 0 && nodes[i].op != DE_LEAF_SHARED; }
    // codegen view: a folded constant subtree is an operand, like a leaf
    bool is_opnd(int i) const { return (nodes[i].degree == 0 || nodes[i].fold_slot >= 0) && nodes[i].share_def < 0; }
    // a definition must be evaluated before its references: the tape has it first (depth-first, left to right), the
    // Sethi-Ullman order may not — the left child goes first whenever the right one reads a row the left one writes
    // the node whose SHAPE the reference's dispatch sees at position i: a shared reference stands for the subtree it names
    std::vector<int> share_node;
    int shp(int i) const { return is_shared_leaf(i) ? share_node[nodes[i].arg] : i; }
    bool left_first(int l, int r) const { return (nodes[r].uses & nodes[l].defs) != 0 || nodes[l].regs > nodes[r].regs; }
    bool is_const_leaf(int i) const { return nodes[i].degree == 0 && nodes[i].op == DE_LEAF_CONST; }
    bool bin_of_leaves(int i) const {
        return nodes[i].degree == 2 && is_leaf(nodes[i].child[0]) && is_leaf(nodes[i].child[1]);
    }

    // ---- annotate ---------------------------------------------------------
    void host_check_ee(int leaf) { // @return_on_nonfinite_val on a constant leaf
        if (is_const_leaf(leaf)) out->const_checks[nodes[leaf].arg] |= CONST_CHECK_EE;
    }
    // `@return_on_nonfinite_array` applied to the result array of child c.
    void checked_child(int c) {
        if (!is_leaf(c)) return; // operator results are always tested in early-exit mode
        if (is_const_leaf(c)) host_check_ee(c);
        else nodes[c].chk_leaf = true;
    }
    void mark_constfold(int i) { // dispatch_constant_tree, src/Evaluate.jl:1002-1067
        LNode &n = nodes[i];
        n.constfold = true;
        if (n.degree == 0) out->const_checks[n.arg] |= CONST_CHECK_ALWAYS;
        for (int k = 0; k < n.degree; k++) mark_constfold(n.child[k]);
    }
    void annotate_bumper(int i) { // ext/DynamicExpressionsBumperExt.jl:25-36,63
        LNode &n = nodes[i];
        if (n.degree == 0) { host_check_ee(i); return; }
        for (int k = 0; k < n.degree; k++) annotate_bumper(n.child[k]);
    }
    void annotate(int i) { // == _eval_tree_array(node)
        LNode &n = nodes[i];
        if (n.degree == 0) return;
        if (n.is_const) { mark_constfold(i); return; }
        if (n.degree == 1) {
            int c = n.child[0];
            const int cs = shp(c); // a later occurrence of a shared subtree takes the same fused kernel as the first
            if (opt.fuse1 && bin_of_leaves(cs)) { // deg1_l2_ll0_lr0_eval
                host_check_ee(nodes[cs].child[0]);
                host_check_ee(nodes[cs].child[1]);
                n.inject = true;
                return;
            }
            if (opt.fuse1 && nodes[cs].degree == 1 && is_leaf(nodes[cs].child[0])) { // deg1_l1_ll0_eval
                n.inject = true;
                return;
            }
            annotate(c);
            checked_child(c);
            return;
        }
        if (n.degree == 2) {
            int l = n.child[0], r = n.child[1];
            if (opt.fuse2 && is_leaf(l) && is_leaf(r)) { // deg2_l0_r0_eval
                host_check_ee(l);
                host_check_ee(r);
                return;
            }
            if (opt.fuse2 && is_leaf(r)) {
                if (bin_of_leaves(shp(l))) { // deg2_branch0_eval :left
                    host_check_ee(nodes[shp(l)].child[0]);
                    host_check_ee(nodes[shp(l)].child[1]);
                    host_check_ee(r);
                    if (!is_const_leaf(r)) nodes[r].chk_leaf = true; // _fused_binary3 tests x3
                    return;
                }
                annotate(l);
                checked_child(l);
                host_check_ee(r); // deg2_r0_eval
                return;
            }
            if (opt.fuse2 && is_leaf(l)) {
                if (bin_of_leaves(shp(r))) { // deg2_branch0_eval :right
                    host_check_ee(l);
                    host_check_ee(nodes[shp(r)].child[0]);
                    host_check_ee(nodes[shp(r)].child[1]);
                    if (!is_const_leaf(l)) nodes[l].chk_leaf = true; // _fused_binary3 tests x1
                    return;
                }
                annotate(r);
                checked_child(r);
                host_check_ee(l); // deg2_l0_eval
                return;
            }
            annotate(l);
            checked_child(l);
            annotate(r);
            checked_child(r);
            return;
        }
        for (int k = 0; k < n.degree; k++) { // inner_dispatch_degn_eval
            annotate(n.child[k]);
            checked_child(n.child[k]);
        }
    }

    // ---- codegen ----------------------------------------------------------
    void compute_regs(int i) {
        LNode &n = nodes[i];
        if (n.fold_slot >= 0) { n.regs = 0; return; }
        for (int k = 0; k < n.degree; k++) compute_regs(n.child[k]);
        if (n.degree == 0) n.regs = 0;
        else if (n.degree == 1) n.regs = nodes[n.child[0]].regs;
        else if (n.degree == 2) {
            int l = n.child[0], r = n.child[1];
            if (is_opnd(l) && is_opnd(r)) n.regs = 0;
            else if (is_opnd(l)) n.regs = nodes[r].regs;
            else if (is_opnd(r)) n.regs = nodes[l].regs;
            else {
                int a = nodes[l].regs, b = nodes[r].regs;
                n.regs = (a == b) ? a + 1 : std::max(a, b);
                if ((nodes[r].uses & nodes[l].defs) != 0) n.regs = std::max(a, b + 1); // forced order: l is held while r runs
            }
        } else {
            n.regs = std::max({nodes[n.child[0]].regs, 1 + nodes[n.child[1]].regs,
                               2 + nodes[n.child[2]].regs});
        }
    }

    Instr &emit(uint32_t op) {
        Instr ins;
        std::memset(&ins, 0, sizeof ins);
        ins.hdr = op & H_OP_MASK;
        if (pending_push >= 0) {
            ins.hdr |= H_PUSH | ((uint32_t)pending_push << H_PUSH_SHIFT);
            pending_push = -1;
        }
        out->code.push_back(ins);
        return out->code.back();
    }
    // Make leaf `li` the B operand of `ins`.
    void set_leaf_operand(Instr &ins, int li) {
        const LNode &lf = nodes[li];
        if (lf.fold_slot >= 0) { // folded constant subtree: extended constant slot, value patched by the caller
            ins.hdr |= SRC_CONST << H_SRC_SHIFT;
            ins.feat = (uint32_t)lf.fold_slot << 16;
            return;
        }
        if (lf.op == DE_LEAF_CONST) {
            ins.hdr |= SRC_CONST << H_SRC_SHIFT;
            ins.feat = (uint32_t)lf.arg << 16;
            out->const_instr[lf.arg] = (int32_t)(&ins - out->code.data());
        } else if (lf.op == DE_LEAF_SHARED) { // persistent row of shared subtree lf.arg, written at its definition
            ins.hdr |= SRC_ROW << H_SRC_SHIFT;
            ins.feat = (uint32_t)(opt.n_features + share_base + lf.arg);
        } else if (lf.op == DE_LEAF_FEATURE) {
            ins.hdr |= SRC_ROW << H_SRC_SHIFT;
            ins.feat = lf.arg;
            if (lf.chk_leaf) ins.hdr |= H_CHECK_B;
        } else {
            ins.hdr |= SRC_PARAM << H_SRC_SHIFT;
            ins.feat = lf.arg;
            if (lf.chk_leaf) ins.hdr |= H_CHECK_B;
            out->uses_params = true;
        }
    }
    void set_pop_operand(Instr &ins, int slot) { // spill slot s lives in LDS row F + s
        ins.hdr |= SRC_ROW << H_SRC_SHIFT;
        ins.feat = (uint32_t)(opt.n_features + slot);
    }
    static uint32_t swapped(uint32_t op, bool *unsupported) {
        *unsupported = false;
        switch (op) {
        case DE_B_ADD: case DE_B_MUL: case DE_B_MAX: case DE_B_MIN: return op; // commutative
        case DE_B_SUB: return DOP_RSUB;
        case DE_B_DIV: return DOP_RDIV;
        case DE_B_POW: return DOP_RPOW;
        case DE_B_MOD: return DOP_RMOD;
        case DE_B_REM: return DOP_RREM;
        case DE_B_GREATER: return DOP_RGREATER;
        case DE_B_POW_ABS2: return DOP_RPOW_ABS2;
        default: *unsupported = true; return op;
        }
    }
    void op_flags(Instr &ins, const LNode &n) {
        if (n.constfold) ins.hdr |= H_CHECK_ALWAYS;
        if (n.inject) ins.hdr |= H_INJECT;
    }

    // Emit code leaving the value of node i in acc; `depth` = occupied spill slots.  A shared subtree's first occurrence
    // additionally leaves its value in its persistent row: the push rides on the NEXT emitted instruction (H_PUSH runs
    // before that instruction executes; the accumulator is unchanged).
    void gen(int i, int depth) {
        gen_value(i, depth);
        if (nodes[i].share_def >= 0) {
            if (pending_push >= 0) { bad_share = true; return; }
            pending_push = share_base + nodes[i].share_def;
        }
    }
    bool bad_share = false;
    void gen_value(int i, int depth) {
        const LNode n = nodes[i];
        if (n.degree == 0 || n.fold_slot >= 0) {
            Instr &ins = emit(DOP_LOAD);
            set_leaf_operand(ins, i);
            return;
        }
        if (n.degree == 1) {
            int c = n.child[0];
            if (is_opnd(c)) {
                Instr &ins = emit(n.op);
                set_leaf_operand(ins, c);
                op_flags(ins, n);
            } else {
                gen(c, depth);
                Instr &ins = emit(n.op);
                ins.hdr |= SRC_ACC << H_SRC_SHIFT;
                op_flags(ins, n);
            }
            return;
        }
        if (n.degree == 2) {
            int l = n.child[0], r = n.child[1];
            if (is_opnd(r)) {
                gen(l, depth); // l leaf -> LOAD, else its code
                Instr &ins = emit(n.op);
                set_leaf_operand(ins, r);
                op_flags(ins, n);
                return;
            }
            if (is_opnd(l)) { // acc = r ; result = op(leaf, acc)
                gen(r, depth);
                bool flag;
                Instr &ins = emit(swapped(n.op, &flag));
                set_leaf_operand(ins, l);
                op_flags(ins, n);
                return;
            }
            const bool lf = left_first(l, r); // (on ties right first: natural operand order, no swap)
            const int first = lf ? l : r, second = lf ? r : l;
            int pop_slot = depth, next_depth = depth + 1;
            gen(first, depth);
            if (nodes[first].share_def >= 0) { // its persistent row doubles as the spill: no slot, the share push is pending
                pop_slot = share_base + nodes[first].share_def;
                next_depth = depth;
            } else {
                max_slots = std::max(max_slots, depth + 1);
                pending_push = depth;
            }
            gen(second, next_depth);
            bool flag;
            Instr &ins = emit(lf ? swapped(n.op, &flag) : n.op); // left first: result = op(pop = l, acc = r)
            set_pop_operand(ins, pop_slot);
            op_flags(ins, n);
            return;
        }
        // degree 3: x -> slot depth, y -> slot depth+1, z -> acc ; acc = op(x, y, z)
        max_slots = std::max(max_slots, depth + 2);
        gen(n.child[0], depth);
        if (pending_push >= 0) bad_share = true; // a shared subtree directly under a ternary operator: not served (the flattener avoids it)
        pending_push = depth;
        gen(n.child[1], depth + 1);
        if (pending_push >= 0) bad_share = true;
        pending_push = depth + 1;
        gen(n.child[2], depth + 2);
        Instr &ins = emit(n.op);
        set_pop_operand(ins, depth);
        ins.hdr |= (uint32_t)(depth + 1) << H_POPC_SHIFT;
        op_flags(ins, n);
    }
};

// Does `op` map a non-finite value arriving at the given operand position onto a
// non-finite result?  (acc_pos: the value is the accumulator / unary input; otherwise it is
// operand B.)  Conservative: false when unsure.
bool propagates_nonfinite(uint32_t op, bool acc_pos) {
    switch (op) {
    case DE_B_ADD: case DE_B_SUB: case DOP_RSUB: case DE_B_MUL: return true;
    case DE_B_DIV: return acc_pos;   // acc / B : numerator
    case DOP_RDIV: return !acc_pos;  // B / acc : numerator is B
    case DE_U_NEG: case DE_U_ABS: case DE_U_SQUARE: case DE_U_CUBE: case DE_U_ROUND: case DE_U_FLOOR:
    case DE_U_CEIL: case DE_U_SQRT: case DE_U_CBRT: case DE_U_LOG: case DE_U_LOG2: case DE_U_LOG10:
    case DE_U_LOG1P: case DE_U_SIN: case DE_U_COS: case DE_U_TAN: case DE_U_SINH: case DE_U_COSH:
    case DE_U_ASIN: case DE_U_ACOS: case DE_U_ASINH: case DE_U_ACOSH: case DE_U_ATANH:
    case DE_U_SAFE_LOG: case DE_U_SAFE_LOG2: case DE_U_SAFE_LOG10: case DE_U_SAFE_LOG1P:
    case DE_U_SAFE_SQRT: case DE_U_SAFE_ACOSH: case DE_U_COS2:
        return true;
    default: return false; // exp(-Inf)=0, 1/Inf=0, tanh, atan, max/min, pow, mod, rem, greater, n-ary ...
    }
}

// Decide H_CHECK_OUT for every instruction (early-exit flag semantics, see de_program.h).
void assign_check_out(std::vector<Instr> &code, int n_features, int share_base = 1 << 20) {
    const size_t n = code.size();
    for (size_t i = 0; i < n; i++) {
        Instr &ins = code[i];
        const uint32_t op = ins.hdr & H_OP_MASK;
        if (op == DOP_LOAD) continue; // a leaf value, tested only through H_CHECK_B
        bool need = true;
        if (i + 1 < n) {
            const Instr &nx = code[i + 1];
            if ((nx.hdr & H_PUSH) && (int)((nx.hdr >> H_PUSH_SHIFT) & H_SLOT_MASK) >= share_base) {
                // definition of a shared subtree: read by several consumers — keep the result test (each occurrence's
                // value is tested in the reference; they are the same values)
            } else if (nx.hdr & H_PUSH) { // value is spilled: consumed later as an LDS row operand
                const uint32_t row = (uint32_t)n_features + ((nx.hdr >> H_PUSH_SHIFT) & H_SLOT_MASK);
                for (size_t j = i + 2; j < n; j++) {
                    const Instr &c = code[j];
                    const uint32_t cop = c.hdr & H_OP_MASK;
                    const bool is_row = ((c.hdr >> H_SRC_SHIFT) & H_SRC_MASK) == SRC_ROW;
                    if (is_row && (c.feat & 0xFFFFu) == row) { // the consumer reads it as operand B
                        const bool binary = (cop >= DE_B_ADD && cop < DE_T_FMA) || cop >= DOP_RSUB;
                        need = !(binary && propagates_nonfinite(cop, false));
                        break;
                    }
                    if (cop >= DE_T_FMA && cop < DOP_LOAD &&
                        (uint32_t)n_features + ((c.hdr >> H_POPC_SHIFT) & H_SLOT_MASK) == row)
                        break; // second operand of a ternary: keep the test
                }
            } else { // consumed by the next instruction through the accumulator
                const uint32_t nop = nx.hdr & H_OP_MASK;
                const uint32_t nsrc = (nx.hdr >> H_SRC_SHIFT) & H_SRC_MASK;
                const bool ternary = nop >= DE_T_FMA && nop < DOP_LOAD;
                const bool reads_acc = (nop >= DE_B_ADD && nop != DOP_LOAD) || nsrc == SRC_ACC;
                if (reads_acc && !ternary) need = !propagates_nonfinite(nop, true);
            }
        }
        if (need) ins.hdr |= H_CHECK_OUT;
    }
}

} // namespace

int lower_tree(const de_tape_node_t *tape, int64_t n, int64_t n_consts, const LowerOptions &opt,
               TreeProgram *out, std::string *err) {
    auto fail = [&](int code, const std::string &msg) {
        if (err) *err = msg;
        return code;
    };
    if (n <= 0) return fail(DE_ERR_BAD_TAPE, "empty tape");
    if (n > 65535) return fail(DE_ERR_UNSUPPORTED, "tape longer than 65535 nodes");
    *out = TreeProgram();
    out->n_nodes = (int)n;
    out->n_consts = (int)n_consts;
    out->const_instr.assign((size_t)n_consts, -1);
    out->const_checks.assign((size_t)n_consts, 0);

    Lowerer L(opt, out, err);
    L.nodes.resize((size_t)n);
    std::vector<int> stack;
    int n_shared_defs = 0;
    std::vector<int> depth((size_t)n, 1);
    stack.reserve((size_t)n);
    int64_t seen_consts = 0;
    for (int64_t i = 0; i < n; i++) {
        LNode &nd = L.nodes[(size_t)i];
        nd.degree = tape[i].degree;
        nd.op = tape[i].op;
        nd.arg = tape[i].arg;
        nd.first = (int)i;
        if (nd.degree == 0) {
            if (nd.op == DE_LEAF_CONST) {
                nd.is_const = true;
                nd.cfirst = nd.arg;
                nd.ccount = 1;
                if (nd.arg >= n_consts) return fail(DE_ERR_OUT_OF_RANGE, "constant slot out of range");
                if (out->const_instr[nd.arg] == -2)
                    return fail(DE_ERR_BAD_TAPE, "constant slot referenced twice");
                out->const_instr[nd.arg] = -2; // seen
                seen_consts++;
            } else if (nd.op == DE_LEAF_FEATURE) {
                if ((int)nd.arg >= opt.n_features) return fail(DE_ERR_OUT_OF_RANGE, "feature index out of range");
            } else if (nd.op == DE_LEAF_PARAM) {
                if ((int)nd.arg >= opt.n_params) return fail(DE_ERR_OUT_OF_RANGE, "parameter index out of range");
            } else if (nd.op == DE_LEAF_SHARED && opt.cse) {
                if ((int)nd.arg >= n_shared_defs) return fail(DE_ERR_BAD_TAPE, "DE_LEAF_SHARED before its DE_OP_SHARE definition");
                nd.uses = 1u << nd.arg;
            } else return fail(DE_ERR_BAD_TAPE, "unknown leaf kind");
        } else if (nd.degree == 1 && nd.op == DE_OP_SHARE && opt.cse) {
            // transparent marker: the node on top of the stack is shared subtree nd.arg (ids in order of definition)
            if (stack.empty()) return fail(DE_ERR_BAD_TAPE, "DE_OP_SHARE without a subtree");
            LNode &def = L.nodes[(size_t)stack.back()];
            if ((int)nd.arg != n_shared_defs || nd.arg >= 16) return fail(DE_ERR_BAD_TAPE, "share ids must be 0, 1, ... in order of definition (at most 16)");
            if (def.degree == 0 || def.is_const || def.share_def >= 0) return fail(DE_ERR_BAD_TAPE, "only non-constant operator subtrees can be shared");
            def.share_def = n_shared_defs++;
            def.defs |= 1u << def.share_def;
            L.share_node.push_back(stack.back());
            nd.degree = 0xFF; // not a node of the tree
            continue;
        } else if (nd.degree <= 3) {
            int lo = nd.degree == 1 ? DE_U_NEG : (nd.degree == 2 ? DE_B_ADD : DE_T_FMA);
            int hi = nd.degree == 1 ? DE_U_LAST_ : (nd.degree == 2 ? DE_B_LAST_ : DE_T_LAST_);
            if (nd.op < lo || nd.op >= hi) return fail(DE_ERR_UNSUPPORTED_OP, "opcode not in table for this degree");
            if ((int)stack.size() < nd.degree) return fail(DE_ERR_BAD_TAPE, "operator without enough operands");
            nd.is_const = true;
            int d = 0;
            for (int k = nd.degree - 1; k >= 0; k--) {
                nd.child[k] = stack.back();
                stack.pop_back();
                const LNode &ch = L.nodes[(size_t)nd.child[k]];
                nd.is_const = nd.is_const && ch.is_const;
                nd.defs |= ch.defs;
                nd.uses |= ch.uses;
                d = std::max(d, depth[(size_t)nd.child[k]]);
                nd.first = std::min(nd.first, ch.first);
                if (ch.ccount) {
                    nd.cfirst = nd.ccount ? std::min(nd.cfirst, ch.cfirst) : ch.cfirst;
                    nd.ccount += ch.ccount;
                }
            }
            depth[(size_t)i] = d + 1;
            if (d + 1 > 2048) return fail(DE_ERR_UNSUPPORTED, "tree deeper than 2048");
        } else return fail(DE_ERR_BAD_TAPE, "degree > 3");
        stack.push_back((int)i);
    }
    if (stack.size() != 1) return fail(DE_ERR_BAD_TAPE, "tape does not reduce to a single root");
    if (seen_consts != n_consts && !(opt.cse && seen_consts < n_consts)) return fail(DE_ERR_BAD_TAPE, "constant pool size does not match tape");
    for (int64_t k = 0; k < n_consts; k++) // (CSE tapes reference a subset of the slots)
        if (out->const_instr[(size_t)k] == -2) out->const_instr[(size_t)k] = -1;
    int root = stack[0];
    // every shared definition must be READ somewhere: a DE_OP_SHARE whose subtree no DE_LEAF_SHARED names would give a
    // persistent row without a reader, and the reverse sweep's `acc += adjoint of the row` (ROP_R_POPADD) would add a row nothing wrote
    if (opt.cse && n_shared_defs > 0 && (L.nodes[(size_t)root].uses & ((1u << n_shared_defs) - 1u)) != ((1u << n_shared_defs) - 1u))
        return fail(DE_ERR_BAD_TAPE, "DE_OP_SHARE definition that no DE_LEAF_SHARED references");

    if (opt.bumper) {
        L.annotate_bumper(root);
    } else {
        L.annotate(root);
        L.checked_child(root); // final is_valid_array(result.x), src/Evaluate.jl:305-308
    }
    if (opt.fold) { // maximal constant subtrees with at least one operator
        std::vector<int> todo{root};
        while (!todo.empty()) {
            const int i = todo.back();
            todo.pop_back();
            LNode &nd = L.nodes[(size_t)i];
            if (nd.degree == 0) continue;
            if (nd.is_const) {
                nd.fold_slot = (int)n_consts + (int)out->folds.size();
                if (nd.fold_slot > 65535) return fail(DE_ERR_UNSUPPORTED, "too many constants");
                out->folds.push_back(FoldSpan{nd.first, (int32_t)i + 1, nd.cfirst, nd.cfirst + nd.ccount, nd.constfold});
            } else {
                for (int k = 0; k < nd.degree; k++) todo.push_back(nd.child[k]);
            }
        }
    }
    // slots n_consts .. n_consts+folds-1 are the folded subtrees' values
    out->const_instr.resize((size_t)n_consts + out->folds.size(), -1);
    L.compute_regs(root);
    if (L.nodes[(size_t)root].regs + n_shared_defs > MAX_SLOTS)
        return fail(DE_ERR_UNSUPPORTED, "tree needs more than 16 spill slots");
    L.share_base = L.nodes[(size_t)root].regs;
    out->code.reserve((size_t)n);
    L.gen(root, 0);
    if (L.bad_share || L.pending_push >= 0) return fail(DE_ERR_UNSUPPORTED, "shared subtree in a position the lowering cannot serve (root, or child of a ternary operator)");
    // set_leaf_operand stored indices while the vector could still grow: recompute them.
    for (size_t k = 0; k < out->code.size(); k++) {
        const Instr &ins = out->code[k];
        if (((ins.hdr >> H_SRC_SHIFT) & H_SRC_MASK) == SRC_CONST) out->const_instr[ins.feat >> 16] = (int32_t)k;
    }
    assign_check_out(out->code, opt.n_features, n_shared_defs ? L.share_base : 1 << 20);
    out->n_shared = n_shared_defs;
    out->n_slots = n_shared_defs ? L.share_base + n_shared_defs : L.max_slots;
    return DE_OK;
}

} // namespace de
