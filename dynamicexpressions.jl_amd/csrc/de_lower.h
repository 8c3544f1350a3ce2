// de_lower.h — host lowering: post-order tape -> device program (de_program.h).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/de_hip.h"
#include "de_program.h"

namespace de {

struct LowerOptions {
    bool early_exit = true;
    bool fuse1 = true, fuse2 = true;
    bool bumper = false;
    int n_features = 0;
    int n_params = 0;
    int dtype = DE_F32;
    // Replace every maximal constant subtree (>= 1 operator) by a constant operand whose value
    // the caller computes ON THE DEVICE (same operator code as everywhere else) before the first
    // launch — the reference folds the same subtrees on the host, dispatch_constant_tree,
    // src/Evaluate.jl:347-354,1002-1067.  Flag annotation still follows the ORIGINAL tree.
    bool fold = false;
    // The tape is the CSE form of a GraphNode tree (DE_OP_SHARE / DE_LEAF_SHARED, include/de_opcodes.h): its constant
    // leaves use the slot numbering of the EXPANDED tape but only the first occurrence of every shared subtree is
    // present, so fewer than n_consts slots may be referenced.
    bool cse = false;
};

// How a constant slot takes part in the host-side part of the `ok` flag.
enum : uint8_t {
    CONST_CHECK_EE = 1,     // value-tested by the reference when early_exit=true
    CONST_CHECK_ALWAYS = 2, // leaf of a constant-folded subtree: tested unconditionally
};

// A folded constant subtree: the post-order tape slice [node_begin, node_end) and the constant
// slots [const_begin, const_end) it owns; its value lives in extended constant slot n_consts + k.
struct FoldSpan {
    int32_t node_begin, node_end, const_begin, const_end;
    // true: the reference evaluates this subtree with dispatch_constant_tree (reached through
    // _eval_tree_array), whose validity test is unconditional; false: the subtree is the inner branch of a
    // fused 3-node kernel (deg2_branch0_eval, src/Evaluate.jl:795-871) or on the Bumper path — a non-finite
    // value only clears the flag under early_exit
    bool tested_always;
};

struct TreeProgram {
    std::vector<FoldSpan> folds;       // only with LowerOptions.fold
    std::vector<Instr> code;
    std::vector<int32_t> const_instr;  // const slot -> index into `code` holding its immediate
    std::vector<uint8_t> const_checks; // const slot -> CONST_CHECK_* bits
    int n_slots = 0;                   // spill slots needed (CSE tapes: + one persistent row per shared subtree)
    int n_shared = 0;                  // shared subtrees of a CSE tape
    int n_nodes = 0;
    int n_consts = 0;
    bool uses_params = false;
};

// Returns DE_OK or a de_status_t; `err` receives a human-readable reason.
int lower_tree(const de_tape_node_t *tape, int64_t n_nodes, int64_t n_consts, const LowerOptions &opt,
               TreeProgram *out, std::string *err);

} // namespace de
